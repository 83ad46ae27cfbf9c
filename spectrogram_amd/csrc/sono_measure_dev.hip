// sono_measure_dev.hip -- measureFrequencyParameters (batCallDetector.js:2994-3894) for a batch of calls on the GPU,
// one 64-lane wave per call, so that the calls' dB matrices never have to cross to the host.
//
// The searches are the shared source of sono_measure_core.h.  Every lane of a call's wave runs the search logic on the
// same values (it is serial: a hundred thresholds, each depending on the last), and the lanes share out the work
// wherever the logic asks a question that has parts:
//   * about a whole row -- "the first bin above this threshold", "the maximum and where it is": a row is read 64
//     entries per instruction, coalesced, and answered with one ballot;
//   * about many frames at one threshold -- the walks of the two searches, 64 frames at a time from a table built once
//     per search (below: WaveOps::build_table, low_walk, high_walk);
//   * about every point of the contour (its smoothing, derivatives, curvature; the time axis): a lane per point.
// What one lane writes for the others to read (the tables, the per-point arrays, the per-bin maxima of the safety
// re-scan) ends in a fence; everything else the logic writes (its per-threshold log in LDS, the contour) is written by
// every lane with the same value, and a lane reads back what it wrote itself.  What the kernel's time goes into is
// the serial logic itself, now some 150 000 wave instructions per call (400 000 before the walks took 64 frames at a
// time), not the rows' way from memory.
//
// Built with -ffp-contract=off like the host stage: same expressions, same roundings, same numbers.
#include <hip/hip_runtime.h>

#include "sono_measure_core.h"
#include "sono_measure_dev.h"

namespace sono {
namespace {

constexpr int kWavesPerBlock = 4;
#ifndef SONO_MEASURE_WAVES_PER_SIMD
#define SONO_MEASURE_WAVES_PER_SIMD 4      // 128 registers a lane: a few spill, and twice the calls in flight pays for them
#endif

// The scans load UNCONDITIONALLY (the index is clamped into the range and the entry discarded afterwards where it
// lay outside): a load under a branch of its own is waited for before the next one is issued, and four chunks of a row
// are fetched together so that their trips to memory overlap.  Indices are 32-bit inside (a row has at most 65 535
// entries): 64-bit integer arithmetic costs the vector unit several instructions per operation.
//
// The two searches ask, a hundred thresholds in a row, "which is this frame's topmost (lowest) bin above the
// threshold" of the same frames.  begin_high / begin_low answer that once for every frame and all hundred
// thresholds: an entry x of a row counts from threshold number T(x) on (the first one it lies above; the thresholds
// fall), so the topmost bin above threshold t is the largest bin whose T is at most t -- every lane files its bins
// under their T with a maximum (minimum) in LDS, and a running maximum (minimum) over the hundred slots is the frame's
// answer per threshold, kept as 16-bit bin numbers, a hundred per frame, in the wave's scratch.  T comes from
// comparisons with the very floats the frame-by-frame scans compare with, so the table holds what they would return.
// A walk then takes 64 frames at a time, a lane each: one look-up per frame, the searches' own expressions for the
// frame's edge and its tests evaluated by every lane for its frame, and the loop's order restored with ballots (where
// the second silent frame in a row ends it) and one maximum over the wave (the first of equal candidates wins).
struct WaveOps {
    unsigned int lane;
    uint16_t *table;                         // the wave's scratch: kThresholds entries per frame
    float *thr_f;                            // LDS, kThresholds: the thresholds as the scans compare with them
    int *slot;                               // LDS, kThresholds
    int table_first;                         // the frame the table's first row belongs to
    static constexpr unsigned int kNone = 0xffffu;

    // independent entries of an array, a lane each; the lanes read each other's entries afterwards
    template <class F> __device__ void for_each(int64_t n, const F &f) const
    {
        for (int i = (int)lane; i < (int)n; i += 64) f((int64_t)i);
        __threadfence_block();
    }

    template <bool TOP> __device__ void build_table(const CallMatrix &m, double peak, int f0, int f1)
    {
        const int l = (int)lane, nb = (int)m.bins, last = nb - 1;
        for (int i = l; i < mcore::kThresholds; i += 64) thr_f[i] = mcore::float_below_thr(peak + (double)(-(i + 1)));
        __threadfence_block();
        const float lowest = thr_f[mcore::kThresholds - 1], peak_f = (float)peak;
        const int none = TOP ? -1 : 0x7fffffff;
        table_first = f0;
        auto file = [&](int i, float v) {
            if (!(i <= last && v > lowest)) return;                              // not above any threshold (NaN: never)
            int g = (int)(peak_f - v) + 1;                                       // a guess at T, then the comparisons themselves
            g = g < 1 ? 1 : g > mcore::kThresholds ? mcore::kThresholds : g;
            while (g > 1 && v > thr_f[g - 2]) --g;
            while (!(v > thr_f[g - 1])) ++g;                                     // ends at the lowest threshold at the latest
            if (TOP) atomicMax(&slot[g - 1], i); else atomicMin(&slot[g - 1], i);
        };
        for (int f = f0; f <= f1; ++f) {
            slot[l] = none;
            if (l + 64 < mcore::kThresholds) slot[l + 64] = none;
            __threadfence_block();
            const float *p = m.db + (size_t)f * m.bins;
            for (int base = 0; base < nb; base += 256) {
                const int i0 = base + l, i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192;
                const float v0 = p[i0 < last ? i0 : last], v1 = p[i1 < last ? i1 : last], v2 = p[i2 < last ? i2 : last],
                            v3 = p[i3 < last ? i3 : last];
                file(i0, v0); file(i1, v1); file(i2, v2); file(i3, v3);
            }
            __threadfence_block();
            // running maximum (minimum) over the slots: lanes 0..63 hold thresholds 1..64, lanes 0..35 also 65..100
            int a = slot[l], b = l + 64 < mcore::kThresholds ? slot[l + 64] : none;
            for (int o = 1; o < 64; o <<= 1) {
                const int va = __shfl_up(a, o), vb = __shfl_up(b, o);
                if (l >= o) { a = TOP ? max(a, va) : min(a, va); b = TOP ? max(b, vb) : min(b, vb); }
            }
            const int first_half = __shfl(a, 63);
            b = TOP ? max(b, first_half) : min(b, first_half);
            uint16_t *out = table + (size_t)(f - f0) * mcore::kThresholds;
            out[l] = (uint16_t)(a == none ? kNone : (unsigned int)a);
            if (l + 64 < mcore::kThresholds) out[l + 64] = (uint16_t)(b == none ? kNone : (unsigned int)b);
        }
        __threadfence_block();                                                   // the lanes read each other's entries next
    }
    __device__ void begin_low(const CallMatrix &m, double peak, int64_t first, int64_t last) { build_table<false>(m, peak, (int)first, (int)last); }
    __device__ void begin_high(const CallMatrix &m, double peak, int64_t last) { build_table<true>(m, peak, 0, (int)last); }

    // One threshold of the low-frequency search (t = 1..100): forwards from start_frame while frames hold a bin above
    // the threshold at or below max_bin; ends after a frame whose lowest such bin lies below the reference, or at the
    // second frame in a row without one.  active_end: the last frame that had one.
    template <class BelowRef>
    __device__ bool low_walk(int t, int64_t start_frame, int64_t search_end, int64_t max_bin, const BelowRef &below_ref, int64_t &active_end) const
    {
        const int l = (int)lane, end = (int)search_end, top = (int)max_bin;
        unsigned long long silent_before = 0;                                    // bit 0: the frame before this chunk had none
        for (int base = (int)start_frame; base <= end; base += 64) {
            const int f = base + l;
            const bool valid = f <= end;
            const unsigned int b = table[(size_t)((valid ? f : end) - table_first) * mcore::kThresholds + (t - 1)];
            const bool has = valid && b != kNone && (int)b <= top;
            const bool below = has && below_ref((int64_t)b);
            const unsigned long long vmask = __ballot(valid), hmask = __ballot(has), silent = vmask & ~hmask;
            const unsigned long long ends = __ballot(below) | (silent & ((silent << 1) | silent_before));
            const int stop = ends ? __builtin_ctzll(ends) : 64;
            const unsigned long long seen = stop < 63 ? hmask & ((2ull << stop) - 1) : hmask;         // frames up to the one that ends it
            if (seen) active_end = base + 63 - __builtin_clzll(seen);
            if (ends) break;
            silent_before = silent >> 63;
        }
        return true;
    }

    // One threshold of the high-frequency search: backwards from frame `limit`, every frame offers the upper edge of its
    // topmost bin above the threshold (at or above min_bin).  Declines when such a candidate is a harmonic (the loop
    // then looks further down that frame) or not a number.
    template <class Edge, class Harmonic>
    __device__ bool high_walk(int t, int64_t limit, int64_t min_bin, const Edge &edge, const Harmonic &harmonic, double &high_hz,
                              int64_t &hbin, int64_t &hframe, bool &found) const
    {
        const int l = (int)lane, low = (int)min_bin;
        unsigned long long silent_before = 0;
        bool any = false;
        double best = 0;
        int best_bin = 0, best_frame = 0;
        for (int top = (int)limit; top >= 0; top -= 64) {
            const int f = top - l;
            const bool valid = f >= 0;
            const unsigned int b = table[(size_t)((valid ? f : 0) - table_first) * mcore::kThresholds + (t - 1)];
            const bool in = valid && b != kNone && (int)b >= low;
            const unsigned long long vmask = __ballot(valid), imask = __ballot(in), silent = vmask & ~imask;
            const unsigned long long ends = silent & ((silent << 1) | silent_before);
            const int stop = ends ? __builtin_ctzll(ends) : 64;
            const bool offers = in && l < stop;
            double cand = -INFINITY;
            bool odd = false;
            if (offers) {
                cand = edge((int64_t)f, (int64_t)b);
                odd = cand != cand || harmonic(cand);
            }
            if (__ballot(odd)) return false;
            double most = cand;
            for (int o = 32; o >= 1; o >>= 1) { const double v = __shfl_xor(most, o); if (v > most) most = v; }
            const unsigned long long holders = __ballot(offers && cand == most);
            if (holders) {
                const int w = __builtin_ctzll(holders);                          // the first one met: the highest frame
                if (!any || most > best) { best = most; best_bin = __shfl((int)b, w); best_frame = top - w; any = true; }
            }
            if (ends) break;
            silent_before = silent >> 63;
        }
        if (any) { high_hz = best; hbin = best_bin; hframe = best_frame; found = true; }
        return true;
    }

    __device__ int64_t first_gt(const float *p, int64_t lo64, int64_t hi64, float t) const
    {
        const int lo = (int)lo64, hi = (int)hi64, l = (int)lane;
        for (int base = lo; base <= hi; base += 256) {
            const int i0 = base + l, i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192;
            const float v0 = p[i0 < hi ? i0 : hi], v1 = p[i1 < hi ? i1 : hi], v2 = p[i2 < hi ? i2 : hi], v3 = p[i3 < hi ? i3 : hi];
            unsigned long long mask = __ballot(i0 <= hi && v0 > t);
            if (mask) return base + __builtin_ctzll(mask);
            mask = __ballot(i1 <= hi && v1 > t);
            if (mask) return base + 64 + __builtin_ctzll(mask);
            mask = __ballot(i2 <= hi && v2 > t);
            if (mask) return base + 128 + __builtin_ctzll(mask);
            mask = __ballot(i3 <= hi && v3 > t);
            if (mask) return base + 192 + __builtin_ctzll(mask);
        }
        return -1;
    }
    __device__ int64_t last_gt(const float *p, int64_t lo64, int64_t hi64, float t) const
    {
        const int lo = (int)lo64, hi = (int)hi64, l = (int)lane;
        for (int base = hi; base >= lo; base -= 256) {
            const int i0 = base - l, i1 = i0 - 64, i2 = i0 - 128, i3 = i0 - 192;
            const float v0 = p[i0 > lo ? i0 : lo], v1 = p[i1 > lo ? i1 : lo], v2 = p[i2 > lo ? i2 : lo], v3 = p[i3 > lo ? i3 : lo];
            unsigned long long mask = __ballot(i0 >= lo && v0 > t);
            if (mask) return base - __builtin_ctzll(mask);
            mask = __ballot(i1 >= lo && v1 > t);
            if (mask) return base - 64 - __builtin_ctzll(mask);
            mask = __ballot(i2 >= lo && v2 > t);
            if (mask) return base - 128 - __builtin_ctzll(mask);
            mask = __ballot(i3 >= lo && v3 > t);
            if (mask) return base - 192 - __builtin_ctzll(mask);
        }
        return -1;
    }
    __device__ float row_max(const float *p, size_t n64) const
    {
        const int n = (int)n64, l = (int)lane;
        float m = -INFINITY;
        for (int base = 0; base < n; base += 256) {
            const int i0 = base + l, i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192, last = n - 1;
            const float v0 = p[i0 < last ? i0 : last], v1 = p[i1 < last ? i1 : last], v2 = p[i2 < last ? i2 : last],
                        v3 = p[i3 < last ? i3 : last];                 // a repeated last entry changes no maximum
            if (v0 > m) m = v0;                                        // a NaN is never above
            if (v1 > m) m = v1;
            if (v2 > m) m = v2;
            if (v3 > m) m = v3;
        }
        for (int o = 32; o >= 1; o >>= 1) { const float v = __shfl_xor(m, o); if (v > m) m = v; }
        return m;
    }
    __device__ int64_t first_max(const float *p, size_t n64, float *value) const
    {
        const float m = row_max(p, n64);
        *value = -INFINITY;
        if (!(m > -INFINITY)) return -1;
        const int n = (int)n64, l = (int)lane, last = n - 1;
        for (int base = 0; base < n; base += 256) {
            const int i0 = base + l, i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192;
            const float v0 = p[i0 < last ? i0 : last], v1 = p[i1 < last ? i1 : last], v2 = p[i2 < last ? i2 : last],
                        v3 = p[i3 < last ? i3 : last];
            int at = -1;
            unsigned long long mask = __ballot(i0 <= last && v0 == m);
            if (mask) at = base + __builtin_ctzll(mask);
            else if ((mask = __ballot(i1 <= last && v1 == m))) at = base + 64 + __builtin_ctzll(mask);
            else if ((mask = __ballot(i2 <= last && v2 == m))) at = base + 128 + __builtin_ctzll(mask);
            else if ((mask = __ballot(i3 <= last && v3 == m))) at = base + 192 + __builtin_ctzll(mask);
            if (at >= 0) {
                *value = p[at];                              // the entry met first (a -0 before a +0 stays a -0)
                return at;
            }
        }
        return -1;                                           // not reached: m is one of the entries
    }
    // every call measured here comes with its region's zonal map
    __device__ const double *zonal_floors(const CallMatrix &, int64_t, int64_t, size_t *zones) const { *zones = 0; return nullptr; }
    // per bin: the largest entry of frames 0..last and the first frame that holds it (js:3160-3171); the lanes take a
    // bin each and walk down the frames
    __device__ void max_spectrum(const CallMatrix &m, int64_t last, float *max_spec, uint16_t *frame_of) const
    {
        for (uint32_t b = lane; b < m.bins; b += 64) {
            float best = -INFINITY;
            uint16_t at = 0;
            for (int64_t f = 0; f <= last; ++f) {
                const float v = m.db[(size_t)f * m.bins + b];
                if (v > best) { best = v; at = (uint16_t)f; }
            }
            max_spec[b] = best;
            frame_of[b] = at;
        }
        __threadfence_block();                               // the lanes read each other's bins next
    }
};

__global__ __launch_bounds__(64 * kWavesPerBlock, SONO_MEASURE_WAVES_PER_SIMD) void measure_calls_kernel(MeasureArgs a)
{
    __shared__ mcore::LowM s_low[kWavesPerBlock][mcore::kThresholds];
    __shared__ mcore::HighM s_high[kWavesPerBlock][mcore::kThresholds];
    __shared__ int s_valid[kWavesPerBlock][mcore::kThresholds];
    const unsigned int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const unsigned long long wave_global = (unsigned long long)blockIdx.x * kWavesPerBlock + wave;
    const unsigned long long waves = (unsigned long long)gridDim.x * kWavesPerBlock;
    // the wave's scratch: time axis, five per-frame double arrays, one per-frame index array, two per-bin arrays
    char *scr = a.scratch + wave_global * a.scratch_per_wave;
    const size_t Fm = a.max_frames;
    double *tf = (double *)scr;
    double *per_frame = tf + (Fm + 1);
    int64_t *frames = (int64_t *)(per_frame + 5 * Fm);
    float *max_spec = (float *)(frames + Fm);
    uint16_t *frame_of = (uint16_t *)(max_spec + a.bins);
    uint16_t *table = frame_of + (a.bins + 1) / 2 * 2;
    mcore::Work w{s_low[wave], s_high[wave], s_valid[wave], per_frame, per_frame + Fm, per_frame + 2 * Fm, per_frame + 3 * Fm,
                  per_frame + 4 * Fm, frames, max_spec, frame_of};
    // the table's two LDS areas are borrowed from logs that are not in use while a table is built: the high-frequency
    // search's log (written from its first threshold on, after begin_high) and the list of valid entries (filled after
    // a search's last threshold)
    WaveOps ops{lane, table, (float *)s_high[wave], s_valid[wave], 0};
    for (unsigned long long j = wave_global; j < a.n_jobs; j += waves) {
        const MeasureJob k = a.jobs[j];
        ops.for_each(k.times, [&](int64_t f) {                                                       // js:1350-1351
            tf[f] = ((double)(k.start_frame + (uint64_t)f) * a.hop + a.half_n) / a.sample_rate;
        });
        CallMatrix m{a.db + (size_t)k.db_row * a.bins, (size_t)k.frames, a.bins, a.freq_bins,
                     a.zonal + (size_t)k.roi * a.zone_stride, a.zones, a.debug_mode != 0, a.time_expansion != 0};
        CallMeasure cm;
        cm.start_time_s = tf[0];                                                                     // js:791-794
        cm.end_time_s = tf[k.times - 1];
        cm.duration_ms = k.duration_ms;
        mcore::measure(m, ops, w, tf, (size_t)k.times, a.fhigh_khz, a.freq_res, cm);
        if (lane == 0) {
            sono_call_measurement &o = a.out[j];
            o.start_time_s = cm.start_time_s; o.end_time_s = cm.end_time_s; o.duration_ms = cm.duration_ms;
            o.peak_freq_khz = cm.peak_freq_khz; o.peak_freq_ms = cm.peak_freq_ms; o.peak_power_db = cm.peak_power_db;
            o.high_freq_khz = cm.high_freq_khz; o.high_freq_ms = cm.high_freq_ms;
            o.start_freq_khz = cm.start_freq_khz; o.start_freq_ms = cm.start_freq_ms; o.start_freq_time_s = cm.start_freq_time_s;
            o.end_freq_khz = cm.end_freq_khz; o.end_freq_ms = cm.end_freq_ms; o.end_freq_time_s = cm.end_freq_time_s;
            o.low_freq_khz = cm.low_freq_khz; o.low_freq_ms = cm.low_freq_ms;
            o.characteristic_freq_khz = cm.characteristic_freq_khz; o.characteristic_freq_ms = cm.characteristic_freq_ms;
            o.knee_freq_khz = cm.knee_freq_khz; o.knee_freq_ms = cm.knee_freq_ms;
            o.heel_freq_khz = cm.heel_freq_khz; o.heel_freq_ms = cm.heel_freq_ms;
            o.bandwidth_khz = cm.bandwidth_khz; o.high_threshold_db_used = cm.high_threshold_db_used;
            o.low_threshold_db_used = cm.low_threshold_db_used;
            o.peak_frame = cm.peak_frame; o.high_freq_frame = cm.high_freq_frame; o.start_freq_frame = cm.start_freq_frame;
            o.start_freq_bin = cm.start_freq_bin; o.end_frame_for_low_freq = cm.end_frame_for_low_freq;
            o.knee_frame = cm.knee_frame; o.heel_frame = cm.heel_frame;
            o.discarded = cm.discarded ? 1 : 0; o.reference_throws = cm.reference_throws ? 1 : 0;
        }
    }
}

}  // namespace

size_t measure_scratch_per_wave(unsigned int max_frames, unsigned int bins)
{
    const size_t bytes = ((size_t)max_frames + 1) * sizeof(double) + (size_t)max_frames * mcore::work_bytes_per_frame() +
                         (size_t)(bins + 1) / 2 * 2 * mcore::work_bytes_per_bin() +
                         (size_t)max_frames * mcore::kThresholds * sizeof(uint16_t);         // the searches' table
    return (bytes + 255) / 256 * 256;
}

unsigned int measure_waves(unsigned long long n_jobs, unsigned int cus)
{
    // four blocks of four waves fit a CU (the per-threshold logs take 9 KB of LDS per wave)
    const unsigned long long most = (unsigned long long)(cus ? cus : 256) * 4 * kWavesPerBlock;
    unsigned long long waves = n_jobs < most ? n_jobs : most;
    waves = (waves + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;
    return (unsigned int)(waves ? waves : kWavesPerBlock);
}

hipError_t launch_measure_calls(const MeasureArgs &a, unsigned int waves, hipStream_t s)
{
    if (a.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(measure_calls_kernel, dim3(waves / kWavesPerBlock), dim3(64 * kWavesPerBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace sono
