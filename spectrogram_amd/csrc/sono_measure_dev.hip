// sono_measure_dev.hip -- measureFrequencyParameters (batCallDetector.js:2994-3894) for a batch of calls on the GPU,
// one 64-lane wave per call, so that the calls' dB matrices never have to cross to the host.
//
// The searches are the shared source of sono_measure_core.h.  Every lane of a call's wave runs the search logic on the
// same values (it is serial: a hundred thresholds, each depending on the last), and the lanes share out the bins of a
// row whenever the logic asks a question about a whole row -- "the first bin above this threshold", "the maximum and
// where it is" -- which is where the time goes: a row is read 64 entries per instruction, coalesced, and answered with
// one ballot.  Everything the logic writes (its per-threshold log in LDS, the contour arrays in the wave's scratch) is
// written by every lane with the same value, and a lane reads back what it wrote itself, so no ordering between lanes
// is needed except after the one cooperative write (the per-bin maxima of the safety re-scan), which ends in a fence.
//
// Built with -ffp-contract=off like the host stage: same expressions, same roundings, same numbers.
#include <hip/hip_runtime.h>

#include "sono_measure_core.h"
#include "sono_measure_dev.h"

namespace sono {
namespace {

constexpr int kWavesPerBlock = 4;

struct WaveOps {
    unsigned int lane;
    // Four chunks of 64 entries are loaded before the first of them is looked at: the loads overlap, and a 316-bin row
    // is answered after at most two round trips to memory instead of five.
    __device__ int64_t first_gt(const float *p, int64_t lo, int64_t hi, float t) const
    {
        for (int64_t base = lo; base <= hi; base += 256) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int64_t i = base + 64 * c + lane; v[c] = i <= hi ? p[i] : -INFINITY; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned long long mask = __ballot(v[c] > t);
                if (mask) return base + 64 * c + (int64_t)__builtin_ctzll(mask);
            }
        }
        return -1;
    }
    __device__ int64_t last_gt(const float *p, int64_t lo, int64_t hi, float t) const
    {
        for (int64_t base = hi; base >= lo; base -= 256) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int64_t i = base - 64 * c - lane; v[c] = i >= lo ? p[i] : -INFINITY; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned long long mask = __ballot(v[c] > t);
                if (mask) return base - 64 * c - (int64_t)__builtin_ctzll(mask);
            }
        }
        return -1;
    }
    __device__ float row_max(const float *p, size_t n) const
    {
        float m = -INFINITY;
        for (size_t i = lane; i < n; i += 64) { const float v = p[i]; if (v > m) m = v; }      // a NaN is never above
        for (int o = 32; o >= 1; o >>= 1) { const float v = __shfl_xor(m, o); if (v > m) m = v; }
        return m;
    }
    __device__ int64_t first_max(const float *p, size_t n, float *value) const
    {
        const float m = row_max(p, n);
        *value = -INFINITY;
        if (!(m > -INFINITY)) return -1;
        for (size_t base = 0; base < n; base += 256) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const size_t i = base + 64 * c + lane; v[c] = i < n ? p[i] : -INFINITY; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned long long mask = __ballot(v[c] == m);         // m is above -INFINITY here
                if (mask) {
                    const int64_t at = (int64_t)base + 64 * c + (int64_t)__builtin_ctzll(mask);
                    *value = p[at];                          // the entry met first (a -0 before a +0 stays a -0)
                    return at;
                }
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

__global__ __launch_bounds__(64 * kWavesPerBlock, 4) void measure_calls_kernel(MeasureArgs a)
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
    mcore::Work w{s_low[wave], s_high[wave], s_valid[wave], per_frame, per_frame + Fm, per_frame + 2 * Fm, per_frame + 3 * Fm,
                  per_frame + 4 * Fm, frames, max_spec, frame_of};
    WaveOps ops{lane};
    for (unsigned long long j = wave_global; j < a.n_jobs; j += waves) {
        const MeasureJob k = a.jobs[j];
        for (unsigned int f = 0; f < k.times; ++f)                                                   // js:1350-1351
            tf[f] = ((double)(k.start_frame + f) * a.hop + a.half_n) / a.sample_rate;
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
                         (size_t)bins * mcore::work_bytes_per_bin();
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
