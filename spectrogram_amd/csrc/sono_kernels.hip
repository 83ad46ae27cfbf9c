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
// Filter-bank path (cold: lib.rs:270-274,310-334).  One workgroup per frame;
// thread i owns filter i and walks the magnitude row in bin order with row
// stride `bins`, bounds-checked against the bank length, as the reference does.
// The same kernel records the frame's peak from the unfiltered magnitudes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void filterbank_u8_kernel(
    const float *mags, unsigned long long frames, int bins, const float *bank,
    unsigned long long bank_len, int num_filters, float gain_db, float range_db,
    uint8_t *out_u8, uint16_t *peak_bin, float *peak_mag, unsigned int *gmax_bits)
{
    extern __shared__ float row[];
    __shared__ unsigned long long s_peak;
    const unsigned long long f = blockIdx.x;
    if (threadIdx.x == 0) s_peak = peak_key(0.0f, 0u);
    __syncthreads();
    unsigned long long best = peak_key(0.0f, 0u);
    for (int j = threadIdx.x; j < bins; j += blockDim.x) {
        const float m = mags[f * (unsigned long long)bins + j];
        row[j] = m;
        if (m > 0.0f) {
            const unsigned long long key = peak_key(m, (unsigned int)j);
            best = key > best ? key : best;
        }
    }
    atomicMax(&s_peak, best);
    __syncthreads();
    const float gain_db_neg = -gain_db;
    const float range_db_reciprocal = 255.0f / range_db;
    for (int i = threadIdx.x; i < num_filters; i += blockDim.x) {
        float sum = 0.0f;
        const unsigned long long r0 = (unsigned long long)i * (unsigned long long)bins;
        if (bank_len != 0)
            for (int j = 0; j < bins; ++j)
                if (r0 + j < bank_len) sum += row[j] * bank[r0 + j];
        const float safe = sum > 1e-10f ? sum : 1e-10f;
        const float db = 20.0f * log10f(safe);
        unsigned int code;
        if (db < gain_db_neg - range_db) code = 0;
        else if (db > gain_db_neg) code = 255;
        else {
            float v = (db - (gain_db_neg - range_db)) * range_db_reciprocal;
            v = v > 0.0f ? v : 0.0f;
            v = v < 255.0f ? v : 255.0f;
            code = (unsigned int)v;
        }
        out_u8[f * (unsigned long long)num_filters + i] = (uint8_t)code;
    }
    if (threadIdx.x == 0) {
        const unsigned int bits = (unsigned int)(s_peak >> 32);
        if (peak_bin) peak_bin[f] = (uint16_t)(0xFFFFFFFFu - (unsigned int)(s_peak & 0xFFFFFFFFull));
        if (peak_mag) peak_mag[f] = __uint_as_float(bits);
        if (gmax_bits) atomicMax(gmax_bits, bits);
    }
}

hipError_t launch_filterbank_u8(const float *mags, unsigned long long frames, int bins, const float *bank,
                                unsigned long long bank_len, int num_filters, float gain_db, float range_db,
                                uint8_t *out_u8, uint16_t *peak_bin, float *peak_mag,
                                unsigned int *gmax_bits, hipStream_t stream)
{
    if (frames == 0) return hipSuccess;
    if (frames > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(filterbank_u8_kernel, dim3((unsigned int)frames), dim3(256), (size_t)bins * sizeof(float),
                       stream, mags, frames, bins, bank, bank_len, num_filters, gain_db, range_db, out_u8,
                       peak_bin, peak_mag, gmax_bits);
    return hipGetLastError();
}

}  // namespace sono
