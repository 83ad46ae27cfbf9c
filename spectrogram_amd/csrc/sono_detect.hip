// sono_detect.hip -- detector front-end kernels (see sono_detect.h).  All of them
// are single passes over frames x bins arrays in HBM: one wave64 per frame row,
// lanes striding over the bins, so every load is a coalesced 256-byte row piece.
#include "sono_detect.h"

#include <hipcub/hipcub.hpp>

#include <cstring>
#include <type_traits>

namespace sono {

namespace {

constexpr int kThreads = 256;                       // 4 rows per workgroup
constexpr int kRowsPerBlock = kThreads / 64;

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One workgroup per kRowsPerBlock rows (band_active_kernel), or -- for the kernels that loop over rows -- at most 8192.
unsigned int rows_grid(size_t frames) { return (unsigned int)((frames + kRowsPerBlock - 1) / kRowsPerBlock); }
unsigned int rows_grid_capped(size_t frames) { const unsigned int g = rows_grid(frames); return g < 8192u ? g : 8192u; }

// The row kernels below let a wave take row after row (a grid-stride loop over rows instead of one short-lived
// workgroup per four rows) and, for rows of up to 64 * kRowRegs bins, fetch a lane's whole share of the row with
// kRowRegs loads issued together before anything looks at the values; wider rows take the strided loop.
constexpr int kRowRegs = 8;

template <typename F>
__device__ __forceinline__ void for_each_row(size_t frames, F &&body)
{
    const size_t waves = (size_t)gridDim.x * kRowsPerBlock;
    for (size_t f = (size_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6); f < frames; f += waves) body(f);
}

// v[k] = row[lane + 64 k] (fill where the row has ended)
__device__ __forceinline__ void load_row(const float *row, uint32_t nb, int lane, float fill, float (&v)[kRowRegs])
{
#pragma unroll
    for (int k = 0; k < kRowRegs; ++k) {
        const uint32_t b = (uint32_t)lane + 64u * k;
        v[k] = b < nb ? row[b] : fill;
    }
}

__global__ __launch_bounds__(kThreads) void band_active_kernel(const float *mags, size_t frames, uint32_t bins,
                                                               uint32_t min_bin, uint32_t max_bin, double target,
                                                               uint8_t *active, double *band_energy)
{
    const size_t f = (size_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= frames) return;
    const size_t total = frames * (size_t)bins;
    const size_t lo = f * bins + min_bin;
    size_t hi = f * bins + max_bin;                 // inclusive; the flat index may run into the next frame
    if (hi >= total) hi = total - 1;
    double acc = 0.0;
    for (size_t i = lo + lane; i <= hi && lo <= hi; i += 64) {
        const double m = (double)mags[i];
        acc += m * m;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        active[f] = acc > target ? 1 : 0;
        if (band_energy) band_energy[f] = acc;
    }
}

// 10 log10(max(m^2 / N, 1e-16)) of the band's bins in double, as the JavaScript computes it (batCallDetector.js:1353-1360),
// rounded to f32 where it stores into a Float32Array.  A wave per row: no division finds the row of an element.
//
// The library logarithm costs over a hundred instructions per entry, and the f32 it is rounded to needs a fraction of
// its digits.  So the entry first goes a short way: log10 of the mantissa from a 64-entry table of reciprocals
// (c = 1 + (i + 1/2) / 64, r = mant / c - 1, |r| < 2^-7) and seven terms of log1p, good to 4e-14 dB against the exact
// value over the whole range (measured against long double arithmetic on 1e8 entries).  If every number within
// 1e-12 dB of that rounds to the same f32, the library's value -- a few 1e-14 from the exact one as well -- rounds
// to it too, and it is stored; otherwise (two entries in a million: the value sits on a rounding boundary of f32, or
// next to 0 dB) the entry goes the long way.  Either way the stored f32 is the one the library logarithm gives.
struct DbTable { double inv_c[64], log10_c[64]; };
const DbTable &db_table()
{
    static const DbTable t = [] {
        DbTable q;
        for (int i = 0; i < 64; ++i) {
            const long double c = 1.0L + ((long double)i + 0.5L) / 64;
            q.inv_c[i] = (double)(1.0L / c);
            q.log10_c[i] = (double)(-log10l((long double)q.inv_c[i]));          // of the c that inv_c[i] stands for exactly
        }
        return q;
    }();
    return t;
}

// How a magnitude becomes a dB value: the transform's size, 1 / size where that is exact (else 0: divide), the table.
struct DbLaw { double fft_size, scale; DbTable table; };
DbLaw db_law(uint32_t fft_size)
{
    return DbLaw{(double)fft_size, (fft_size & (fft_size - 1)) == 0 ? 1.0 / (double)fft_size : 0.0, db_table()};
}
// The table in LDS, for every thread of the workgroup (call once, by all threads, before db_of_mag).
__device__ __forceinline__ void stage_db_table(const DbTable &table, double *s_inv_c, double *s_log10_c)
{
    if (threadIdx.x < 64) { s_inv_c[threadIdx.x] = table.inv_c[threadIdx.x]; s_log10_c[threadIdx.x] = table.log10_c[threadIdx.x]; }
    __syncthreads();
}
__device__ __forceinline__ float db_of_mag(float mag, double fft_size, double scale, const double *s_inv_c, const double *s_log10_c)
{
    const double m = (double)mag;
    double psd = scale != 0.0 ? (m * m) * scale : (m * m) / fft_size;
    psd = psd > 1e-16 ? psd : 1e-16;                                             // a NaN ends here as well
    const unsigned long long bits = (unsigned long long)__double_as_longlong(psd);
    const int e = (int)((bits >> 52) & 0x7ffu);
    if (e != 0x7ff) {                                                            // finite (and normal: at least 1e-16)
        const int i = (int)((bits >> 46) & 63u);
        const double mant = __longlong_as_double((long long)((bits & 0xFFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
        const double r = fma(mant, s_inv_c[i], -1.0);
        double p = 1.0 / 7;
        p = fma(p, r, -1.0 / 6); p = fma(p, r, 1.0 / 5); p = fma(p, r, -1.0 / 4); p = fma(p, r, 1.0 / 3);
        p = fma(p, r, -0.5); p = fma(p, r, 1.0);
        p *= r;                                                                  // log1p(r)
        const double y = 10.0 * ((double)(e - 1023) * 0.30102999566398119521 + (s_log10_c[i] + p * 0.43429448190325182765));
        const float lo = (float)(y - 1e-12), hi = (float)(y + 1e-12);
        if (lo == hi) return lo;
    }
    return (float)(10.0 * log10(psd));
}

__global__ __launch_bounds__(kThreads) void power_db_kernel(const float *mags, size_t frames, uint32_t bins,
                                                            uint32_t min_bin, uint32_t nb, DbLaw law, float *db)
{
    __shared__ double s_inv_c[64], s_log10_c[64];
    stage_db_table(law.table, s_inv_c, s_log10_c);
    const int lane = threadIdx.x & 63;
    auto to_db = [&](float mag) { return db_of_mag(mag, law.fft_size, law.scale, s_inv_c, s_log10_c); };
    for_each_row(frames, [&](size_t f) {
        const float *src = mags + f * bins + min_bin;
        float *dst = db + f * nb;
        if (nb <= 64u * kRowRegs) {
            float v[kRowRegs];
            load_row(src, nb, lane, 0.0f, v);
#pragma unroll
            for (int k = 0; k < kRowRegs; ++k) {
                const uint32_t b = (uint32_t)lane + 64u * k;
                if (b < nb) dst[b] = to_db(v[k]);
            }
        } else {
            for (uint32_t b = lane; b < nb; b += 64) dst[b] = to_db(src[b]);
        }
    });
}

__global__ __launch_bounds__(kThreads) void db_max_kernel(const float *db, size_t count, uint32_t *max_key)
{
    __shared__ uint32_t s_key[kThreads / 64];
    uint32_t best = 0;                              // below every key
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (size_t)gridDim.x * kThreads) {
        const float v = db[i];
        if (v == v) { const uint32_t k = detect_key(v); best = k > best ? k : best; }    // Math.max over finite dB values
    }
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(best, o, 64); best = other > best ? other : best; }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) best = s_key[w] > best ? s_key[w] : best;
        if (best) atomicMax(max_key, best);
    }
}

__global__ __launch_bounds__(kThreads) void db_active_kernel(const float *db, size_t frames, uint32_t nb,
                                                             double threshold, uint8_t *active)
{
    const int lane = threadIdx.x & 63;
    for_each_row(frames, [&](size_t f) {
        const float *row = db + f * nb;
        bool any = false;
        if (nb <= 64u * kRowRegs) {
            float v[kRowRegs];
            load_row(row, nb, lane, -INFINITY, v);
#pragma unroll
            for (int k = 0; k < kRowRegs; ++k) any = any || ((double)v[k] > threshold);
        } else {
            for (uint32_t b = lane; b < nb; b += 64) any = any || ((double)row[b] > threshold);
        }
        const unsigned long long m = __ballot(any);
        if (lane == 0) active[f] = m ? 1 : 0;
    });
}

__global__ __launch_bounds__(kThreads) void frame_features_kernel(const float *db, size_t frames, uint32_t nb,
                                                                  uint32_t peak_b0, uint32_t peak_b1,
                                                                  uint32_t low_max_bin, double low_thr,
                                                                  uint32_t high_min_bin, double high_thr,
                                                                  sono_frame_features *out)
{
    const int lane = threadIdx.x & 63;
    for_each_row(frames, [&](size_t f) {
        const float *row = db + f * nb;
        // peak: first bin holding the largest value (strict > from -infinity, NaN never wins)
        float best = -INFINITY;
        int best_bin = -1;
        int low = 0x7FFFFFFF, high = -1;
        auto look = [&](uint32_t b, float v) {
            if (b >= peak_b0 && b <= peak_b1 && v > best) { best = v; best_bin = (int)b; }
            if (b <= low_max_bin && (double)v > low_thr && (int)b < low) low = (int)b;
            if (b >= high_min_bin && (double)v > high_thr) high = (int)b;       // b ascends: keeps the last
        };
        if (nb <= 64u * kRowRegs) {
            float v[kRowRegs];
            load_row(row, nb, lane, NAN, v);
#pragma unroll
            for (int k = 0; k < kRowRegs; ++k) {
                const uint32_t b = (uint32_t)lane + 64u * k;
                if (b < nb) look(b, v[k]);
            }
        } else {
            for (uint32_t b = lane; b < nb; b += 64) look(b, row[b]);
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int ob = __shfl_xor(best_bin, o, 64);
            const bool take = ob >= 0 && (best_bin < 0 || ov > best || (ov == best && ob < best_bin));
            if (take) { best = ov; best_bin = ob; }
            const int ol = __shfl_xor(low, o, 64), oh = __shfl_xor(high, o, 64);
            low = ol < low ? ol : low;
            high = oh > high ? oh : high;
        }
        if (lane == 0) {
            sono_frame_features r;
            r.peak_bin = best_bin;
            r.peak_db = best;
            r.low_bin = low == 0x7FFFFFFF ? -1 : low;
            r.high_bin = high;
            out[f] = r;
        }
    });
}

__global__ __launch_bounds__(kThreads) void frame_peaks_kernel(const float *db, size_t frames, uint32_t nb, FramePeak *out)
{
    const int lane = threadIdx.x & 63;
    for_each_row(frames, [&](size_t f) {
        const float *row = db + f * nb;
        const bool in_regs = nb <= 64u * kRowRegs;
        float v[kRowRegs];
        if (in_regs) load_row(row, nb, lane, NAN, v);
        float best = -INFINITY;
        int best_bin = -1;
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < kRowRegs; ++k)
                if (v[k] > best) { best = v[k]; best_bin = lane + 64 * k; }     // strict >, b ascends: the first maximum of the lane; NaN (also the fill) never wins
        } else {
            for (uint32_t b = lane; b < nb; b += 64) {
                const float x = row[b];
                if (x > best) { best = x; best_bin = (int)b; }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int ob = __shfl_xor(best_bin, o, 64);
            const bool take = ob >= 0 && (best_bin < 0 || ov > best || (ov == best && ob < best_bin));
            if (take) { best = ov; best_bin = ob; }
        }
        // second sweep: extent of the bins above peak - 12 dB (js:3061-3073)
        const double thr = (double)best - 12;
        int lo = 0x7FFFFFFF, hi = -1;
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < kRowRegs; ++k)
                if ((double)v[k] > thr) { const int b = lane + 64 * k; if (b < lo) lo = b; hi = b; }
        } else {
            for (uint32_t b = lane; b < nb; b += 64)
                if ((double)row[b] > thr) { if ((int)b < lo) lo = (int)b; hi = (int)b; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int ol = __shfl_xor(lo, o, 64), oh = __shfl_xor(hi, o, 64);
            lo = ol < lo ? ol : lo;
            hi = oh > hi ? oh : hi;
        }
        if (lane == 0) {
            FramePeak r;
            r.lo12 = lo == 0x7FFFFFFF ? -1 : lo;
            r.hi12 = hi;
            r.bin = best_bin;
            r.db = best;
            r.left = best_bin > 0 ? row[best_bin - 1] : NAN;
            r.right = best_bin >= 0 && (uint32_t)best_bin + 1 < nb ? row[best_bin + 1] : NAN;
            out[f] = r;
        }
    });
}

// frame_peaks_kernel on the dB values of a magnitude row that never leave the registers: power_db_kernel and
// frame_peaks_kernel in one pass for rows of up to 64 * kRowRegs bins, when nobody else needs the dB matrix (the region
// analysis: 2.5 KB of traffic per frame less).  Same values, same record.
//
// db_peaks_plain: the record from the dB values v[k] of the row's bins lane + 64 k (NaN beyond the row); every lane of
// the wave returns the same record.
__device__ __forceinline__ FramePeak db_peaks_plain(const float (&v)[kRowRegs], uint32_t nb)
{
    // The row's maximum by one reduction over the lanes, then every question about positions by ballots: which
    // lanes hold the maximum (the first of them is the peak bin), which lie above peak - 12 dB (the first and the
    // last of them are the extent).  A ballot answers for 64 bins at once and its answer is the same in every lane,
    // so no bin numbers travel between lanes.  Same bins as frame_peaks_kernel: strict >, ascending -- the first
    // maximum; a NaN (also the fill) is never a maximum and never above a threshold.
    float best = -INFINITY;
#pragma unroll
    for (int k = 0; k < kRowRegs; ++k)
        if (v[k] > best) best = v[k];
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        if (ov > best) best = ov;
    }
    int best_bin = -1, lo = 0x7FFFFFFF, hi = -1;
    const double thr = (double)best - 12;                                    // js:3061-3073
    const bool any = best > -INFINITY;                                       // else no entry was ever above the start value
#pragma unroll
    for (int k = 0; k < kRowRegs; ++k) {
        if (64u * k >= nb) continue;
        const unsigned long long at_peak = __ballot(any && v[k] == best);
        if (at_peak && best_bin < 0) best_bin = 64 * k + (int)__builtin_ctzll(at_peak);
        const unsigned long long above = __ballot((double)v[k] > thr);
        if (above) {
            if (lo == 0x7FFFFFFF) lo = 64 * k + (int)__builtin_ctzll(above);
            hi = 64 * k + 63 - (int)__builtin_clzll(above);
        }
    }
    // the peak's neighbours, from the lanes that hold them
    auto value_at = [&](int bin) {
        float r = NAN;
#pragma unroll
        for (int k = 0; k < kRowRegs; ++k)
            if ((bin >> 6) == k) r = __shfl(v[k], bin & 63, 64);
        return r;
    };
    if (best_bin >= 0) best = value_at(best_bin);                            // the entry met first (a -0 before a +0 stays a -0)
    FramePeak r;
    r.lo12 = lo == 0x7FFFFFFF ? -1 : lo;
    r.hi12 = hi;
    r.bin = best_bin;
    r.db = best;
    r.left = best_bin > 0 ? value_at(best_bin - 1) : NAN;
    r.right = best_bin >= 0 && (uint32_t)best_bin + 1 < nb ? value_at(best_bin + 1) : NAN;
    return r;
}

// One row's record from its magnitudes at src: every bin of the band converted.
template <typename ToDb>
__device__ __forceinline__ FramePeak db_peaks_row(const float *src, uint32_t nb, int lane, ToDb &&to_db)
{
    float v[kRowRegs];
    load_row(src, nb, lane, 0.0f, v);
#pragma unroll
    for (int k = 0; k < kRowRegs; ++k) {
        if (64u * k < nb) v[k] = to_db(v[k]);                                // a chunk of 64 bins that the band reaches: all lanes or none
        if ((uint32_t)lane + 64u * k >= nb) v[k] = NAN;
    }
    return db_peaks_plain(v, nb);
}

__global__ __launch_bounds__(kThreads) void db_peaks_kernel(const float *mags, size_t frames, uint32_t bins, uint32_t min_bin,
                                                            uint32_t nb, DbLaw law, FramePeak *out)
{
    __shared__ double s_inv_c[64], s_log10_c[64];
    stage_db_table(law.table, s_inv_c, s_log10_c);
    const int lane = threadIdx.x & 63;
    auto to_db = [&](float mag) { return db_of_mag(mag, law.fft_size, law.scale, s_inv_c, s_log10_c); };
    for_each_row(frames, [&](size_t f) {
        const FramePeak r = db_peaks_row(mags + f * bins + min_bin, nb, lane, to_db);
        if (lane == 0) out[f] = r;
    });
}

// db_peaks_kernel for rows that belong to regions: a wave takes a span of consecutive rows of one region (PeakSpan),
// files every row's record straight into the six per-frame arrays of the regions' table (PeakTable, in HBM: the
// caller copies the finished arrays to the host as they are, nothing is re-sorted there) and notes the span's first
// largest peak in row order -- what the automatic high-pass asks of a region (batCallDetector.js:635-644); the caller
// joins a region's spans in order with the same strict comparison.
__global__ __launch_bounds__(kThreads) void db_peaks_spans_kernel(const float *mags, const PeakSpan *spans, size_t n_spans, uint32_t bins,
                                                                  uint32_t min_bin, uint32_t nb, DbLaw law, PeakTable t, SpanBest *span_best)
{
    __shared__ double s_inv_c[64], s_log10_c[64];
    stage_db_table(law.table, s_inv_c, s_log10_c);
    const int lane = threadIdx.x & 63;
    auto to_db = [&](float mag) { return db_of_mag(mag, law.fft_size, law.scale, s_inv_c, s_log10_c); };
    for_each_row(n_spans, [&](size_t i) {
        const PeakSpan sp = spans[i];
        float best = -INFINITY;
        int32_t bin = 0;
        for (uint32_t k = 0; k < sp.rows; ++k) {
            const FramePeak r = db_peaks_row(mags + (size_t)(sp.first_row + k) * bins + min_bin, nb, lane, to_db);
            if (r.db > best) { best = r.db; bin = r.bin; }
            const size_t row = (size_t)sp.table_row + k;
            // six lanes, one array each
            if (lane == 0) t.db[row] = r.db;
            if (lane == 1) t.bin[row] = r.bin;
            if (lane == 2) t.left[row] = r.left;
            if (lane == 3) t.right[row] = r.right;
            if (lane == 4) t.lo12[row] = r.lo12;
            if (lane == 5) t.hi12[row] = r.hi12;
        }
        if (lane == 0 && span_best) span_best[i] = SpanBest{best, bin};
    });
}

__global__ __launch_bounds__(kThreads) void run_edges_kernel(const uint8_t *active, size_t frames, uint8_t *is_start,
                                                             uint8_t *is_end)
{
    const size_t f = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (f >= frames) return;
    const bool a = active[f] != 0;
    const bool prev = f > 0 && active[f - 1] != 0;
    const bool next = f + 1 < frames && active[f + 1] != 0;
    is_start[f] = a && !prev;
    is_end[f] = a && !next;
}

// One sample of the WAV data chunk -> f32 (integer PCM divided by 2^(bits-1), 8-bit PCM offset binary, float as it is).
// ALIGNED: the chunk starts on a multiple of the sample size, so a sample is one load; otherwise byte by byte.
template <int BPS, bool IS_FLOAT, bool ALIGNED>
__device__ __forceinline__ float pcm_sample(const uint8_t *p)
{
    if constexpr (BPS == 1) return ((float)p[0] - 128.0f) * (1.0f / 128.0f);
    else if constexpr (BPS == 2) {
        const int16_t v = ALIGNED ? *(const int16_t *)p : (int16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8));
        return (float)v * (1.0f / 32768.0f);
    } else if constexpr (BPS == 3) {
        const int32_t v = (int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24)) >> 8;
        return (float)v * (1.0f / 8388608.0f);
    } else {
        const uint32_t b = ALIGNED ? *(const uint32_t *)p
                                   : (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        return IS_FLOAT ? __uint_as_float(b) : (float)(int32_t)b * (1.0f / 2147483648.0f);
    }
}

// Eight consecutive sample frames per thread: eight loads in flight, two 16-byte stores (d_out comes from hipMalloc or
// is checked to be 16-byte aligned by the launcher; the tail and unaligned outputs take the scalar stores).
constexpr int kPcmPerThread = 8;
template <int BPS, bool IS_FLOAT, bool ALIGNED, bool WIDE_STORES>
__global__ __launch_bounds__(kThreads) void pcm_to_f32_kernel(const uint8_t *pcm, size_t frames, uint32_t channels,
                                                              uint32_t channel, float *out)
{
    const size_t stride = (size_t)channels * BPS;
    for (size_t i0 = ((size_t)blockIdx.x * kThreads + threadIdx.x) * kPcmPerThread; i0 < frames;
         i0 += (size_t)gridDim.x * kThreads * kPcmPerThread) {
        const uint8_t *p = pcm + (i0 * channels + channel) * (size_t)BPS;
        if (i0 + kPcmPerThread <= frames) {
            float v[kPcmPerThread];
#pragma unroll
            for (int k = 0; k < kPcmPerThread; ++k) v[k] = pcm_sample<BPS, IS_FLOAT, ALIGNED>(p + k * stride);
            if (WIDE_STORES) {
                *(float4 *)(out + i0) = make_float4(v[0], v[1], v[2], v[3]);
                *(float4 *)(out + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
#pragma unroll
                for (int k = 0; k < kPcmPerThread; ++k) out[i0 + k] = v[k];
            }
        } else {
            for (size_t i = i0; i < frames; ++i) out[i] = pcm_sample<BPS, IS_FLOAT, ALIGNED>(pcm + (i * channels + channel) * (size_t)BPS);
        }
    }
}


// One LDS histogram per workgroup, merged into the global one at the end.
__global__ __launch_bounds__(kThreads) void zone_hist_kernel(const float *db, size_t count, uint32_t nb,
                                                             const uint8_t *zone_of_bin, uint32_t num_zones,
                                                             unsigned int *hist)
{
    extern __shared__ unsigned int s_hist[];
    const uint32_t cells = num_zones * (uint32_t)kZoneHistBins;
    for (uint32_t i = threadIdx.x; i < cells; i += kThreads) s_hist[i] = 0u;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (size_t)gridDim.x * kThreads) {
        const uint32_t zone = zone_of_bin[i % nb];
        float v = db[i];
        if (zone >= num_zones || v != v) continue;                  // a NaN lands in no bin of the reference's array either
        v = v < -100.0f ? -100.0f : v;                              // js:1857
        float fl = floorf(v);                                       // js:1867
        fl = fl < -120.0f ? -120.0f : fl > 20.0f ? 20.0f : fl;      // js:1870-1871
        atomicAdd(&s_hist[zone * kZoneHistBins + (uint32_t)((int)fl + 120)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cells; i += kThreads)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

__device__ __forceinline__ double block_sum(double v, double *s_part)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < kThreads / 64; ++w) t += s_part[w];
    __syncthreads();
    return t;                                   // valid in thread 0
}

__global__ __launch_bounds__(kThreads) void box_minmax_kernel(const float *db, uint32_t nb, size_t t0, size_t rows,
                                                              const uint8_t *bin_in_box, uint32_t *keys)
{
    uint32_t hi = 0u, lo = 0xFFFFFFFFu;
    const size_t count = rows * nb;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (size_t)gridDim.x * kThreads) {
        if (!bin_in_box[i % nb]) continue;
        const float v = db[t0 * nb + i];
        if (v != v) continue;                    // a NaN loses both comparisons of the reference (js:444-445)
        const uint32_t k = detect_key(v);
        hi = k > hi ? k : hi;
        lo = k < lo ? k : lo;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t oh = __shfl_xor(hi, o, 64), ol = __shfl_xor(lo, o, 64);
        hi = oh > hi ? oh : hi;
        lo = ol < lo ? ol : lo;
    }
    if ((threadIdx.x & 63) == 0) {
        if (hi != 0u) atomicMax(&keys[0], hi);
        if (lo != 0xFFFFFFFFu) atomicMin(&keys[1], lo);
    }
}

__global__ __launch_bounds__(kThreads) void box_sums_kernel(const float *db, size_t frames, uint32_t nb, size_t t0, size_t t1,
                                                            const uint8_t *bin_in_box, double threshold, int outside_is_noise,
                                                            double *sums, unsigned long long *counts)
{
    __shared__ double s_part[kThreads / 64];
    double sig = 0.0, noise = 0.0;
    unsigned long long n_sig = 0, n_noise = 0;
    const size_t count = frames * nb;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (size_t)gridDim.x * kThreads) {
        const size_t t = i / nb;
        const bool in_box = t >= t0 && t <= t1 && bin_in_box[i - t * nb];
        if (!in_box && !outside_is_noise) continue;
        const double v = (double)db[i];
        if (in_box) {
            if (v > threshold) { sig += pow(10.0, v / 10); ++n_sig; }             // js:478-481
        } else {
            noise += pow(10.0, v / 10);                                          // js:532-536
            ++n_noise;
        }
    }
    const double a = block_sum(sig, s_part), b = block_sum(noise, s_part);
    for (int o = 32; o > 0; o >>= 1) { n_sig += __shfl_xor(n_sig, o, 64); n_noise += __shfl_xor(n_noise, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (n_sig) atomicAdd(&counts[0], n_sig);
        if (n_noise) atomicAdd(&counts[1], n_noise);
    }
    if (threadIdx.x == 0) {
        if (a != 0.0) atomicAdd(&sums[0], a);
        if (b != 0.0) atomicAdd(&sums[1], b);
    }
}

__global__ __launch_bounds__(kThreads) void band_linear_sum_kernel(const float *db, size_t frames, uint32_t nb,
                                                                   const uint8_t *bin_sel, double *sums,
                                                                   unsigned long long *counts)
{
    __shared__ double s_part[kThreads / 64];
    double acc = 0.0;
    unsigned long long n = 0;
    const size_t count = frames * nb;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (size_t)gridDim.x * kThreads) {
        if (!bin_sel[i % nb]) continue;
        acc += pow(10.0, (double)db[i] / 10);                                    // js:512
        ++n;
    }
    const double a = block_sum(acc, s_part);
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(&counts[1], n);
    if (threadIdx.x == 0 && a != 0.0) atomicAdd(&sums[1], a);
}

// FROM_MAGS: `db` is the magnitude matrix (rows of `bins` entries, the band starting at min_bin) and every value is
// converted on the way in; else the dB matrix itself (rows of nb entries).
template <bool FROM_MAGS>
__global__ __launch_bounds__(kThreads) void zone_floors_regions_kernel(const float *db, const unsigned long long *first_row,
                                                                       const unsigned int *rows_used, uint32_t nb,
                                                                       const uint8_t *zone_of_bin, uint32_t num_zones,
                                                                       uint32_t zone_stride, double *floors, uint32_t bins,
                                                                       uint32_t min_bin, DbLaw law)
{
    extern __shared__ unsigned int s_hist[];
    __shared__ double s_inv_c[64], s_log10_c[64];
    if constexpr (FROM_MAGS) stage_db_table(law.table, s_inv_c, s_log10_c);
    const size_t r = blockIdx.x;
    const uint32_t cells = num_zones * (uint32_t)kZoneHistBins;
    for (uint32_t i = threadIdx.x; i < cells; i += kThreads) s_hist[i] = 0u;
    __syncthreads();
    const float *base = db + first_row[r] * (FROM_MAGS ? bins : nb);
    const size_t count = (size_t)rows_used[r] * nb;
    for (size_t i = threadIdx.x; i < count; i += kThreads) {
        const uint32_t zone = zone_of_bin[i % nb];
        float v;
        if constexpr (FROM_MAGS) v = db_of_mag(base[(i / nb) * bins + min_bin + i % nb], law.fft_size, law.scale, s_inv_c, s_log10_c);
        else v = base[i];
        if (zone >= num_zones || v != v) continue;
        v = v < -100.0f ? -100.0f : v;                              // js:1857
        float fl = floorf(v);                                       // js:1867
        fl = fl < -120.0f ? -120.0f : fl > 20.0f ? 20.0f : fl;      // js:1870-1871
        atomicAdd(&s_hist[zone * kZoneHistBins + (uint32_t)((int)fl + 120)], 1u);
    }
    __syncthreads();
    for (uint32_t z = threadIdx.x; z < num_zones; z += kThreads) {
        unsigned int best = 0;
        int mode = -100;
        bool any = false;
        for (int i = 0; i < kZoneHistBins; ++i) {
            const unsigned int c = s_hist[z * kZoneHistBins + i];
            any = any || c > 0;
            if (c > best) { best = c; mode = i - 120; }             // strict >: lowest dB wins a tie, js:1907
        }
        floors[r * zone_stride + z] = any ? (double)mode - 2.0 : -102.0;          // js:1916-1920
    }
}

// in_place != 0: the filtered samples also replace the region's own samples in src (regions are disjoint, and a
// sample is read before it is written).
//
// A lane per region, and a region is a serial recurrence: what the kernel can do about its time is not to wait for
// memory once per sample.  Samples are fetched sixteen at a time, the next sixteen before the current ones are worked
// on (the loads do not depend on the recurrence; written out by hand because the in-place form makes the output alias
// the input, and a load may then not be moved above a store by the compiler): 2.2 -> 0.6 ms for regions of 6 000 samples.
constexpr int kHighpassBlock = 16;
__global__ __launch_bounds__(64) void highpass_regions_kernel(const float *src, float *dst, const HighpassJob *jobs,
                                                              size_t n_jobs, const HighpassCoef *coefs, int in_place)
{
    const size_t k = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (k >= n_jobs) return;
    const HighpassJob job = jobs[k];
    const HighpassCoef c = coefs[job.coef];
    const float *x = src + job.src;
    float *y = dst + job.dst;
    float *z = in_place ? const_cast<float *>(src) + job.src : nullptr;
    double ax1 = 0, ax2 = 0, ay1 = 0, ay2 = 0;      // first section
    double bx1 = 0, bx2 = 0, by1 = 0, by2 = 0;      // second section
    auto filter = [&](float sample) {
        // y0 = b0*x0 + b1*x1 + b2*x2 - a1*y1 - a2*y2, left to right, nothing fused (js:4131)
        const double x0 = (double)sample;
        double s1 = __dadd_rn(__dmul_rn(c.b0, x0), __dmul_rn(c.b1, ax1));
        s1 = __dadd_rn(s1, __dmul_rn(c.b2, ax2));
        s1 = __dsub_rn(s1, __dmul_rn(c.a1, ay1));
        s1 = __dsub_rn(s1, __dmul_rn(c.a2, ay2));
        ax2 = ax1; ax1 = x0; ay2 = ay1; ay1 = s1;
        const double m = (double)(float)s1;          // the first section's Float32Array
        double s2 = __dadd_rn(__dmul_rn(c.b0, m), __dmul_rn(c.b1, bx1));
        s2 = __dadd_rn(s2, __dmul_rn(c.b2, bx2));
        s2 = __dsub_rn(s2, __dmul_rn(c.a1, by1));
        s2 = __dsub_rn(s2, __dmul_rn(c.a2, by2));
        bx2 = bx1; bx1 = m; by2 = by1; by1 = s2;
        return (float)s2;
    };
    const unsigned long long len = job.len, blocks = len / kHighpassBlock;
    float cur[kHighpassBlock], nxt[kHighpassBlock];
    if (blocks) {
#pragma unroll
        for (int j = 0; j < kHighpassBlock; ++j) cur[j] = x[j];
    }
    for (unsigned long long b = 0; b < blocks; ++b) {
        const unsigned long long i = b * kHighpassBlock;
        if (b + 1 < blocks) {
#pragma unroll
            for (int j = 0; j < kHighpassBlock; ++j) nxt[j] = x[i + kHighpassBlock + j];
        }
        float out[kHighpassBlock];
#pragma unroll
        for (int j = 0; j < kHighpassBlock; ++j) out[j] = filter(cur[j]);
#pragma unroll
        for (int j = 0; j < kHighpassBlock; ++j) {
            y[i + j] = out[j];
            if (z) z[i + j] = out[j];
        }
#pragma unroll
        for (int j = 0; j < kHighpassBlock; ++j) cur[j] = nxt[j];
    }
    for (unsigned long long i = blocks * kHighpassBlock; i < len; ++i) {
        const float v = filter(x[i]);
        y[i] = v;
        if (z) z[i] = v;
    }
}

}  // namespace

hipError_t launch_zone_floors_regions(const float *d_db, const unsigned long long *d_first_row,
                                      const unsigned int *d_rows_used, size_t regions, uint32_t nb,
                                      const uint8_t *d_zone_of_bin, uint32_t num_zones, uint32_t zone_stride,
                                      double *d_floors, hipStream_t s)
{
    if (regions == 0 || num_zones == 0) return hipSuccess;
    const size_t lds = (size_t)num_zones * kZoneHistBins * sizeof(unsigned int);
    hipLaunchKernelGGL(zone_floors_regions_kernel<false>, dim3((unsigned int)regions), dim3(kThreads), lds, s, d_db, d_first_row,
                       d_rows_used, nb, d_zone_of_bin, num_zones, zone_stride, d_floors, nb, 0u, DbLaw{});
    return hipGetLastError();
}

hipError_t launch_zone_floors_regions_mags(const float *d_mags, uint32_t bins, uint32_t min_bin, uint32_t fft_size,
                                           const unsigned long long *d_first_row, const unsigned int *d_rows_used, size_t regions,
                                           uint32_t nb, const uint8_t *d_zone_of_bin, uint32_t num_zones, uint32_t zone_stride,
                                           double *d_floors, hipStream_t s)
{
    if (regions == 0 || num_zones == 0) return hipSuccess;
    const size_t lds = (size_t)num_zones * kZoneHistBins * sizeof(unsigned int);
    hipLaunchKernelGGL(zone_floors_regions_kernel<true>, dim3((unsigned int)regions), dim3(kThreads), lds, s, d_mags, d_first_row,
                       d_rows_used, nb, d_zone_of_bin, num_zones, zone_stride, d_floors, bins, min_bin, db_law(fft_size));
    return hipGetLastError();
}

static unsigned int reduce_grid(size_t count)
{
    size_t blocks = (count + (size_t)kThreads * 8 - 1) / ((size_t)kThreads * 8);
    return (unsigned int)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks);
}

hipError_t launch_box_minmax(const float *d_db, size_t frames, uint32_t nb, size_t t0, size_t t1, const uint8_t *d_bin_in_box,
                             uint32_t *d_keys, hipStream_t s)
{
    if (frames == 0 || nb == 0 || t0 > t1 || t0 >= frames) return hipSuccess;
    const size_t rows = (t1 < frames - 1 ? t1 : frames - 1) - t0 + 1;
    hipLaunchKernelGGL(box_minmax_kernel, dim3(reduce_grid(rows * nb)), dim3(kThreads), 0, s, d_db, nb, t0, rows, d_bin_in_box,
                       d_keys);
    return hipGetLastError();
}

hipError_t launch_box_sums(const float *d_db, size_t frames, uint32_t nb, size_t t0, size_t t1, const uint8_t *d_bin_in_box,
                           double threshold_db, int outside_is_noise, double *d_sums, unsigned long long *d_counts, hipStream_t s)
{
    if (frames == 0 || nb == 0) return hipSuccess;
    hipLaunchKernelGGL(box_sums_kernel, dim3(reduce_grid(frames * nb)), dim3(kThreads), 0, s, d_db, frames, nb, t0, t1,
                       d_bin_in_box, threshold_db, outside_is_noise, d_sums, d_counts);
    return hipGetLastError();
}

hipError_t launch_band_linear_sum(const float *d_db, size_t frames, uint32_t nb, const uint8_t *d_bin_sel, double *d_sums,
                                  unsigned long long *d_counts, hipStream_t s)
{
    if (frames == 0 || nb == 0) return hipSuccess;
    hipLaunchKernelGGL(band_linear_sum_kernel, dim3(reduce_grid(frames * nb)), dim3(kThreads), 0, s, d_db, frames, nb, d_bin_sel,
                       d_sums, d_counts);
    return hipGetLastError();
}

hipError_t launch_highpass_regions(const float *d_src, float *d_dst, const HighpassJob *d_jobs, size_t jobs,
                                   const HighpassCoef *d_coefs, bool in_place, hipStream_t s)
{
    if (jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(highpass_regions_kernel, dim3((unsigned int)((jobs + 63) / 64)), dim3(64), 0, s, d_src, d_dst, d_jobs,
                       jobs, d_coefs, in_place ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_zone_hist(const float *d_db, size_t count, uint32_t nb, const uint8_t *d_zone_of_bin,
                            uint32_t num_zones, unsigned int *d_hist, hipStream_t s)
{
    if (count == 0 || num_zones == 0) return hipSuccess;
    size_t blocks = (count + (size_t)kThreads * 16 - 1) / ((size_t)kThreads * 16);
    blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
    const size_t lds = (size_t)num_zones * kZoneHistBins * sizeof(unsigned int);
    hipLaunchKernelGGL(zone_hist_kernel, dim3((unsigned int)blocks), dim3(kThreads), lds, s, d_db, count, nb,
                       d_zone_of_bin, num_zones, d_hist);
    return hipGetLastError();
}

hipError_t sort_floats(void *d_temp, size_t &temp_bytes, const float *d_in, float *d_out, size_t count, hipStream_t s)
{
    return hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_in, d_out, (int)count, 0, 32, s);
}

hipError_t launch_band_active(const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin, uint32_t max_bin,
                              double target, uint8_t *d_active, double *d_band_energy, hipStream_t s)
{
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(band_active_kernel, dim3(rows_grid(frames)), dim3(kThreads), 0, s, d_mags, frames, bins, min_bin,
                       max_bin, target, d_active, d_band_energy);
    return hipGetLastError();
}

hipError_t launch_power_db(const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin, uint32_t nb,
                           uint32_t fft_size, float *d_db, hipStream_t s)
{
    const size_t count = frames * (size_t)nb;
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(power_db_kernel, dim3(rows_grid_capped(frames)), dim3(kThreads), 0, s,
                       d_mags, frames, bins, min_bin, nb, db_law(fft_size), d_db);
    return hipGetLastError();
}

hipError_t launch_db_max(const float *d_db, size_t count, uint32_t *d_max_key, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    size_t blocks = (count + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(db_max_kernel, dim3((unsigned int)blocks), dim3(kThreads), 0, s, d_db, count, d_max_key);
    return hipGetLastError();
}

hipError_t launch_db_active(const float *d_db, size_t frames, uint32_t nb, double threshold_db, uint8_t *d_active,
                            hipStream_t s)
{
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(db_active_kernel, dim3(rows_grid_capped(frames)), dim3(kThreads), 0, s, d_db, frames, nb, threshold_db,
                       d_active);
    return hipGetLastError();
}

hipError_t launch_frame_features(const float *d_db, size_t frames, uint32_t nb, uint32_t peak_b0, uint32_t peak_b1,
                                 uint32_t low_max_bin, double low_thr, uint32_t high_min_bin, double high_thr,
                                 sono_frame_features *d_out, hipStream_t s)
{
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(frame_features_kernel, dim3(rows_grid_capped(frames)), dim3(kThreads), 0, s, d_db, frames, nb, peak_b0,
                       peak_b1, low_max_bin, low_thr, high_min_bin, high_thr, d_out);
    return hipGetLastError();
}

hipError_t launch_frame_peaks(const float *d_db, size_t frames, uint32_t nb, FramePeak *d_out, hipStream_t s)
{
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(frame_peaks_kernel, dim3(rows_grid_capped(frames)), dim3(kThreads), 0, s, d_db, frames, nb, d_out);
    return hipGetLastError();
}

bool db_peaks_fits(uint32_t nb) { return nb <= 64u * kRowRegs; }

hipError_t launch_db_peaks(const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin, uint32_t nb, uint32_t fft_size,
                           FramePeak *d_out, hipStream_t s)
{
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(db_peaks_kernel, dim3(rows_grid_capped(frames)), dim3(kThreads), 0, s, d_mags, frames, bins, min_bin, nb,
                       db_law(fft_size), d_out);
    return hipGetLastError();
}

hipError_t launch_db_peaks_spans(const float *d_mags, const PeakSpan *d_spans, size_t n_spans, uint32_t bins, uint32_t min_bin,
                                 uint32_t nb, uint32_t fft_size, const PeakTable &table, SpanBest *d_span_best, hipStream_t s)
{
    if (n_spans == 0) return hipSuccess;
    hipLaunchKernelGGL(db_peaks_spans_kernel, dim3(rows_grid_capped(n_spans)), dim3(kThreads), 0, s, d_mags, d_spans, n_spans, bins,
                       min_bin, nb, db_law(fft_size), table, d_span_best);
    return hipGetLastError();
}

hipError_t launch_run_edges(const uint8_t *d_active, size_t frames, uint8_t *d_is_start, uint8_t *d_is_end,
                            hipStream_t s)
{
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(run_edges_kernel, dim3((unsigned int)((frames + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       d_active, frames, d_is_start, d_is_end);
    return hipGetLastError();
}

// The common recording: mono 16-bit PCM whose data chunk starts on a multiple of four bytes (the 44-byte header of a
// plain WAV file does).  Eight samples per thread as four dword loads, two 16-byte stores.
__global__ __launch_bounds__(kThreads) void pcm16_mono_to_f32_kernel(const uint32_t *pcm2, size_t frames, float *out)
{
    const size_t groups = frames / kPcmPerThread;            // whole groups of eight; the tail goes to the generic kernel
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < groups; g += (size_t)gridDim.x * kThreads) {
        const uint32_t *p = pcm2 + g * (kPcmPerThread / 2);
        uint32_t w[kPcmPerThread / 2];
#pragma unroll
        for (int k = 0; k < kPcmPerThread / 2; ++k) w[k] = p[k];
        float v[kPcmPerThread];
#pragma unroll
        for (int k = 0; k < kPcmPerThread / 2; ++k) {
            v[2 * k] = (float)(int16_t)(w[k] & 0xFFFFu) * (1.0f / 32768.0f);
            v[2 * k + 1] = (float)(int16_t)(w[k] >> 16) * (1.0f / 32768.0f);
        }
        float *o = out + g * kPcmPerThread;
        *(float4 *)o = make_float4(v[0], v[1], v[2], v[3]);
        *(float4 *)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// The window energies of refineEndUsingOscillogram (batCallDetector.js:1676-1683) for many spans of a recording in
// HBM: a wave per span, a lane per window, the window's squares added in order in double as the host does
// (host_window_sums: the square of a float is exact in a double, so a fused multiply-add rounds like the sum alone).
__global__ __launch_bounds__(kThreads) void window_sums_kernel(const float *x, const WindowJob *jobs, size_t n_jobs, uint32_t window,
                                                               uint32_t hop, double *sums)
{
    const uint32_t lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * kRowsPerBlock;
    for (size_t j = (size_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6); j < n_jobs; j += waves) {
        const WindowJob job = jobs[j];
        for (uint32_t i = lane; i < job.steps; i += 64) {
            const float *p = x + job.first + (size_t)i * hop;
            double acc = 0.0;
            for (uint32_t k = 0; k < window; ++k) { const double v = (double)p[k]; acc = fma(v, v, acc); }
            sums[job.out + i] = acc;
        }
    }
}

hipError_t launch_window_sums(const float *d_x, const WindowJob *d_jobs, size_t n_jobs, uint32_t window, uint32_t hop, double *d_sums,
                              hipStream_t s)
{
    if (n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(window_sums_kernel, dim3(rows_grid_capped(n_jobs)), dim3(kThreads), 0, s, d_x, d_jobs, n_jobs, window, hop, d_sums);
    return hipGetLastError();
}

hipError_t launch_pcm_to_f32(const void *d_pcm, size_t frames, uint32_t channels, uint32_t channel, uint32_t bits,
                             int is_float, float *d_out, hipStream_t s)
{
    if (frames == 0) return hipSuccess;
    const uint32_t bps = bits / 8;
    if (!is_float && bps == 2 && channels == 1 && (uintptr_t)d_pcm % 4 == 0 && (uintptr_t)d_out % 16 == 0 && frames >= 4096) {
        const size_t groups = frames / kPcmPerThread, blocks = (groups + kThreads - 1) / kThreads;
        hipLaunchKernelGGL(pcm16_mono_to_f32_kernel, dim3((unsigned int)(blocks < 65536 ? blocks : 65536)), dim3(kThreads), 0, s,
                           (const uint32_t *)d_pcm, frames, d_out);
        const size_t done = groups * kPcmPerThread;
        if (done == frames) return hipGetLastError();
        // the last few samples: the generic kernel on the tail
        return launch_pcm_to_f32((const uint8_t *)d_pcm + done * 2, frames - done, 1, 0, 16, 0, d_out + done, s);
    }
    const size_t per_block = (size_t)kThreads * kPcmPerThread;
    const size_t blocks = (frames + per_block - 1) / per_block;
    const dim3 grid((unsigned int)(blocks < 65536 ? blocks : 65536)), block(kThreads);
    const bool aligned = ((uintptr_t)d_pcm % (bps == 3 ? 1 : bps)) == 0, wide = ((uintptr_t)d_out % 16) == 0;
    const uint8_t *p = (const uint8_t *)d_pcm;
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, block, 0, s, p, frames, channels, channel, d_out); };
    auto pick = [&](auto bps_, auto flt_) {
        constexpr int B = decltype(bps_)::value;
        constexpr bool F = decltype(flt_)::value;
        if (aligned && wide) go(pcm_to_f32_kernel<B, F, true, true>);
        else if (aligned) go(pcm_to_f32_kernel<B, F, true, false>);
        else if (wide) go(pcm_to_f32_kernel<B, F, false, true>);
        else go(pcm_to_f32_kernel<B, F, false, false>);
    };
    using std::integral_constant;
    if (is_float) pick(integral_constant<int, 4>{}, integral_constant<bool, true>{});
    else if (bps == 1) pick(integral_constant<int, 1>{}, integral_constant<bool, false>{});
    else if (bps == 2) pick(integral_constant<int, 2>{}, integral_constant<bool, false>{});
    else if (bps == 3) pick(integral_constant<int, 3>{}, integral_constant<bool, false>{});
    else pick(integral_constant<int, 4>{}, integral_constant<bool, false>{});
    return hipGetLastError();
}

hipError_t select_flagged(void *d_temp, size_t &temp_bytes, const uint8_t *d_flags, uint64_t *d_out, uint64_t *d_count,
                          size_t frames, hipStream_t s)
{
    hipcub::CountingInputIterator<uint64_t> idx(0);
    return hipcub::DeviceSelect::Flagged(d_temp, temp_bytes, idx, d_flags, d_out, d_count, (int)frames, s);
}

}  // namespace sono
