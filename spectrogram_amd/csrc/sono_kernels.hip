// sono_kernels.hip -- gfx950 kernels of libsonostft and their launchers.
#include <hip/hip_runtime.h>

#include "sono_core.h"
#include "sono_kernels.h"

namespace sono {

// ---------------------------------------------------------------------------
// Generic STFT kernel (all fft sizes 256..4096): LDS-staged Stockham FFT.
// ---------------------------------------------------------------------------
struct LdsMax64 {
    __device__ void operator()(unsigned long long *p, unsigned long long v) const { atomicMax(p, v); }
};
struct GlobalMaxU32 {
    __device__ void operator()(unsigned int *p, unsigned int v) const { atomicMax(p, v); }
};

template <int MODE>
__global__ __launch_bounds__(GEN_THREADS) void stft_generic_kernel(const StftArgs a)
{
    extern __shared__ float2 lds_dyn[];
    __shared__ long long s_off[GEN_MAX_FPB];
    __shared__ unsigned long long s_peak[GEN_MAX_FPB];
    __shared__ float s_dc[GEN_MAX_FPB];

    const int M = (int)(a.N >> 1);
    const int slab = M > 1024 ? M : 1024;
    GenericShared sh{lds_dyn, lds_dyn + slab, s_off, s_peak, s_dc};
    GenericBlock<MODE> blk(a, sh, blockIdx.x);
    const int t = threadIdx.x;

    blk.phase_setup(t);
    __syncthreads();
    blk.phase_load(t);
    if (MODE == MODE_POWER) {
        __syncthreads();
        blk.phase_dc_sum(t);
        __syncthreads();
        blk.phase_dc_sub(t);
    }
    float2 *src = sh.buf0, *dst = sh.buf1;
    int p = 1;
    for (; p * 4 <= M; p <<= 2) {
        __syncthreads();
        blk.phase_r4(t, p, src, dst);
        float2 *tmp = src; src = dst; dst = tmp;
    }
    if (p < M) {
        __syncthreads();
        blk.phase_r2(t, src, dst);
        float2 *tmp = src; src = dst; dst = tmp;
    }
    __syncthreads();
    blk.phase_epilogue(t, src, LdsMax64());
    if (MODE == MODE_U8) {
        __syncthreads();
        blk.phase_features(t, GlobalMaxU32());
    }
}

template <int MODE>
static hipError_t launch_generic_mode(const StftArgs &a, hipStream_t stream)
{
    const unsigned long long total = a.n_streams * a.frames_per_stream;
    if (total == 0) return hipSuccess;
    const int M = (int)(a.N >> 1);
    const int slab = M > 1024 ? M : 1024;
    const int fpb = slab / M;
    const unsigned long long blocks = (total + fpb - 1) / fpb;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * slab * sizeof(float2);
    hipLaunchKernelGGL(stft_generic_kernel<MODE>, dim3((unsigned int)blocks), dim3(GEN_THREADS), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_stft_generic(int mode, const StftArgs &a, hipStream_t stream)
{
    switch (mode) {
    case MODE_MAG: return launch_generic_mode<MODE_MAG>(a, stream);
    case MODE_U8: return launch_generic_mode<MODE_U8>(a, stream);
    case MODE_POWER: return launch_generic_mode<MODE_POWER>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// Power spectrum reduction (lib.rs:1087-1115): per bin, sum the frames' powers
// in frame order (the reference's += order), average, convert to dB.
// ---------------------------------------------------------------------------
__global__ void power_reduce_kernel(const float *power, unsigned long long frames, int bins,
                                    float n_fft, float *out_db)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= bins) return;
    float s = 0.0f;
    for (unsigned long long f = 0; f < frames; ++f) s += power[f * (unsigned long long)bins + k];
    const float avg = s / (float)frames;
    const float rms = sqrtf(avg);
    const float psd = (rms * rms) / n_fft;
    out_db[k] = 10.0f * log10f(psd > 1e-16f ? psd : 1e-16f);
}

hipError_t launch_power_reduce(const float *power, unsigned long long frames, int bins, unsigned int N,
                               float *out_db, hipStream_t stream)
{
    hipLaunchKernelGGL(power_reduce_kernel, dim3((bins + 127) / 128), dim3(128), 0, stream,
                       power, frames, bins, (float)N, out_db);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Filter-bank path (lib.rs:270-274, 310-334).  The reference walks every
// (filter, bin) pair of the flat bank with row stride `bins`, bounds-checked
// against the bank's length, and sums magnitude * weight in bin order, product
// and sum rounded separately (wasm has no fused multiply-add).  A product with a
// zero weight adds an exact zero, so the kernel walks only the bank's NON-ZERO
// taps (bank_taps(): per filter, bins ascending; tap k of filter i is stored at k * filters + i, so
// that the lanes of a wave, which hold neighbouring filters, read neighbouring words) and still produces the
// reference's sums bit for bit; the banks the application builds have two taps
// per filter (spectrogram.esm.js:1197-1232), 256 times less work than the pairs.
// Every wave works on its own: it takes a magnitude row (16-byte loads; the next row's are
// requested before the current one is worked on), keeps it in the wave's part of LDS, finds the
// row's peak from the unfiltered magnitudes on the way, and then its lanes take the filters,
// each summing its filter's taps (staged in LDS when the list is small enough, which the
// application's banks always are) and quantising.  No workgroup barrier inside the loop.
// ---------------------------------------------------------------------------
constexpr int FB_THREADS = 256, FB_ROWS = FB_THREADS / 64;
constexpr unsigned int FB_LDS_TAPS = 4096;       // tap lists up to this many entries (and filters) are staged in LDS

__device__ __forceinline__ void fb_wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// FB_MAXQ: 16-byte pieces of a row per lane (1 up to 256 bins, 2 / 4 / 8 at 512 / 1024 / 2048); a template
// parameter so that the small rows do not pay the registers of the large ones and eight waves fit a SIMD.
template <int FB_MAXQ>
__global__ __launch_bounds__(FB_THREADS) void filterbank_taps_u8_kernel(
    const float *mags, unsigned long long frames, int bins, const unsigned int *tap_start, const unsigned int *tap_bin,
    const float *tap_w, unsigned int taps, int num_filters, int taps_in_lds, float gain_db, float range_db,
    uint8_t *out_u8, uint16_t *peak_bin, float *peak_mag, unsigned int *gmax_bits)
{
    // tap_start[i] = number of taps of filter i; taps = (largest count) x filters = entries of tap_bin / tap_w
    extern __shared__ float rows[];                        // FB_ROWS x bins, then (taps_in_lds) counts / bin / w
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // LDS copies of the tap list, at fixed places behind the rows so that every access below is an LDS
    // instruction (a pointer that may be either kind would make them flat loads)
    unsigned int *const l_start = (unsigned int *)(rows + (size_t)FB_ROWS * bins), *const l_bin = l_start + num_filters;
    float *const l_w = (float *)(l_bin + taps);
    if (taps_in_lds) {
        for (int i = threadIdx.x; i < num_filters; i += FB_THREADS) l_start[i] = tap_start[i];
        for (unsigned int t = threadIdx.x; t < taps; t += FB_THREADS) { l_bin[t] = tap_bin[t]; l_w[t] = tap_w[t]; }
    }
    __syncthreads();
    const BankQuant bq = make_bank_quant(gain_db, range_db);
    const int quads = bins / 4;                            // float4 per row; lane owns quads lane, lane + 64, ...
    float *const row = rows + (size_t)wave * bins;
    const unsigned long long n_waves = (unsigned long long)gridDim.x * FB_ROWS;
    unsigned long long f = (unsigned long long)blockIdx.x * FB_ROWS + wave;
    unsigned int wave_gmax = 0;
    float4 nxt[FB_MAXQ];
    auto request = [&](unsigned long long g) {
        const float4 *src = (const float4 *)(mags + g * (unsigned long long)bins);
#pragma unroll
        for (int k = 0; k < FB_MAXQ; ++k)
            if (64 * k < quads) {                          // the same in every lane
                const int q = lane + 64 * k;
                nxt[k] = q < quads ? src[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
    };
    if (f < frames) request(f);
    for (; f < frames; f += n_waves) {
        fb_wave_fence();                                   // the previous row's sums have read the row
        unsigned long long best = peak_key(0.0f, 0u);
#pragma unroll
        for (int k = 0; k < FB_MAXQ; ++k)
            if (64 * k < quads) {
                const int q = lane + 64 * k;
                if (q < quads) {
                    const float4 m = nxt[k];
                    ((float4 *)row)[q] = m;
                    const float v[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (v[c] > 0.0f) {
                            const unsigned long long key = peak_key(v[c], (unsigned int)(4 * q + c));
                            best = key > best ? key : best;
                        }
                }
            }
        if (f + n_waves < frames) request(f + n_waves);
        // the row's largest magnitude, then the lowest bin among the lanes that hold it (the key's low word is the
        // inverted bin): two reductions of 32-bit words instead of one of 64-bit keys
        unsigned int bits = (unsigned int)(best >> 32);
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned int other = __shfl_xor(bits, o, 64);
            bits = other > bits ? other : bits;
        }
        unsigned int inv = (unsigned int)(best >> 32) == bits ? (unsigned int)(best & 0xFFFFFFFFull) : 0u;
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned int other = __shfl_xor(inv, o, 64);
            inv = other > inv ? other : inv;
        }
        if (lane == 0) {
            if (peak_bin) peak_bin[f] = (uint16_t)(0xFFFFFFFFu - inv);
            if (peak_mag) peak_mag[f] = __uint_as_float(bits);
            wave_gmax = bits > wave_gmax ? bits : wave_gmax;
        }
        fb_wave_fence();                                   // the row is in LDS for every lane
        for (int i = lane; i < num_filters; i += 64) {
            float sum = 0.0f;
            if (taps_in_lds) {
                for (unsigned int k = 0, n = l_start[i], t = (unsigned int)i; k < n; ++k, t += (unsigned int)num_filters)
                    sum = __fadd_rn(sum, __fmul_rn(row[l_bin[t]], l_w[t]));
            } else {
                size_t t = (size_t)i;
                for (unsigned int k = 0, n = tap_start[i]; k < n; ++k, t += (size_t)num_filters)
                    sum = __fadd_rn(sum, __fmul_rn(row[tap_bin[t]], tap_w[t]));
            }
            const unsigned int code = bank_code<0>(sum, bq.a, bq.c);
            out_u8[f * (unsigned long long)num_filters + i] = (uint8_t)code;
        }
    }
    if (lane == 0 && gmax_bits && wave_gmax) atomicMax(gmax_bits, wave_gmax);
}

// The bank's non-zero taps.  count[i] = taps of filter i; tap k of filter i (bins ascending) sits at entry
// k * num_filters + i of bin / w, which have (largest count) x num_filters entries (unused ones zero).
// Weights that compare equal to zero (+0 and -0) are the ones left out; NaN and infinite weights stay.
void bank_taps(const float *bank, size_t bank_len, size_t num_filters, size_t bins, std::vector<unsigned int> &count,
               std::vector<unsigned int> &bin, std::vector<float> &w)
{
    count.assign(num_filters, 0);
    size_t most = 0;
    for (size_t i = 0; i < num_filters; ++i) {
        const size_t r0 = i * bins;
        for (size_t j = 0; j < bins && r0 + j < bank_len; ++j) count[i] += !(bank[r0 + j] == 0.0f);
        most = count[i] > most ? count[i] : most;
    }
    bin.assign(most * num_filters, 0u);
    w.assign(most * num_filters, 0.0f);
    for (size_t i = 0; i < num_filters; ++i) {
        const size_t r0 = i * bins;
        size_t k = 0;
        for (size_t j = 0; j < bins && r0 + j < bank_len; ++j)
            if (!(bank[r0 + j] == 0.0f)) {
                bin[k * num_filters + i] = (unsigned int)j;
                w[k * num_filters + i] = bank[r0 + j];
                ++k;
            }
    }
}

hipError_t launch_filterbank_u8(const float *mags, unsigned long long frames, int bins, const unsigned int *tap_start,
                                const unsigned int *tap_bin, const float *tap_w, unsigned int taps, int num_filters,
                                float gain_db, float range_db, uint8_t *out_u8, uint16_t *peak_bin, float *peak_mag,
                                unsigned int *gmax_bits, hipStream_t stream)
{
    if (frames == 0) return hipSuccess;
    if (bins <= 0 || bins % 4 != 0 || bins > 2048 || num_filters <= 0) return hipErrorInvalidValue;
    // the tap list goes to LDS beside the rows when it is small and the workgroup stays within 60 KB
    const size_t row_bytes = (size_t)FB_ROWS * bins * sizeof(float), tap_bytes = ((size_t)num_filters + 2 * (size_t)taps) * 4;
    const int in_lds = taps <= FB_LDS_TAPS && (unsigned int)num_filters <= FB_LDS_TAPS && row_bytes + tap_bytes <= 60 * 1024;
    const size_t lds = row_bytes + (in_lds ? tap_bytes : 0);
    const unsigned long long groups = (frames + FB_ROWS - 1) / FB_ROWS;
    const unsigned int grid = (unsigned int)(groups < 2048 ? groups : 2048);       // 8 workgroups per CU
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(FB_THREADS), lds, stream, mags, frames, bins, tap_start, tap_bin, tap_w, taps,
                           num_filters, in_lds, gain_db, range_db, out_u8, peak_bin, peak_mag, gmax_bits);
    };
    if (bins <= 256) go(filterbank_taps_u8_kernel<1>);
    else if (bins <= 512) go(filterbank_taps_u8_kernel<2>);
    else if (bins <= 1024) go(filterbank_taps_u8_kernel<4>);
    else go(filterbank_taps_u8_kernel<8>);
    return hipGetLastError();
}

}  // namespace sono
