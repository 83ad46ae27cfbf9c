// sono_core.h -- arithmetic shared by every STFT kernel of libsonostft.
//
// Everything here is written as __host__ __device__ so that the index math and
// the butterflies can be exercised lane by lane in a plain g++ build (see
// tests/host_emul) before a kernel ever runs on a GPU.  The GPU build is the
// only one that ships; the host instantiation exists for that test alone.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <cmath>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SONO_HD __host__ __device__ __forceinline__
#else
#define SONO_HD inline
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint4 { unsigned int x, y, z, w; };
#endif

namespace sono {

enum Mode : int {
    MODE_MAG = 0,    // frames x N/2 linear magnitudes           (lib.rs:113-161)
    MODE_U8 = 1,     // frames x N/2 u8 dB codes + peak features (lib.rs:199-304,349-443)
    MODE_POWER = 2,  // frames x (N/2+1) |X|^2 of mean-removed frames (lib.rs:1058-1094)
    MODE_BAND = 3,   // per frame: sum of mag^2 over a band of bins, compared with a target (batCallDetector.js:913-927);
                     // tiled kernel only
    MODE_SURVEY = 4, // tiled kernel, hop N/4: MODE_U8's peak features without rows AND MODE_BAND's test on every second frame
                     // (the frames of the fast scan's hop N/2): one read of the samples for both (batCallDetector.js:878-949 beside lib.rs:349-443)
    MODE_BANK = 5,   // tiled kernel, fft 1024: MODE_U8 with the rows taken through a sparse filter bank (lib.rs:310-334) in the
                     // epilogue, from the power image in LDS: no magnitude matrix in HBM between transform and bank
};

// Arguments common to the STFT kernels.  Frames are numbered globally:
// g = stream * frames_per_stream + frame; frame f of stream s starts at sample
// s * stream_stride + f * hop.  Output row g is contiguous after row g-1.
struct StftArgs {
    const float *x;               // f32 samples; NULL when the samples are 16-bit PCM (x16)
    const int16_t *x16;           // tiled kernel, MODE_U8 / MODE_BAND: interleaved 16-bit PCM sample frames, sample = value / 32768;
    unsigned int x16_channels;    //   stream offsets, strides and hops count sample frames of x16_channels values,
    unsigned int x16_channel;     //   of which this channel is transformed
    unsigned long long n_streams;
    unsigned long long stream_stride;
    unsigned long long frames_per_stream;
    unsigned int hop;
    unsigned int N;               // fft_size (real samples per frame)
    const float *window;          // N floats
    const float2 *tw_half;        // W_{N/2}^k = exp(-2*pi*i*k/(N/2)), k in [0, N/2)
    const float2 *tw_full;        // W_N^k     = exp(-2*pi*i*k/N),     k in [0, N/2]
    float *out_mag;               // MODE_MAG
    uint8_t *out_u8;              // MODE_U8; NULL = features only (no rows are written)
    uint16_t *peak_bin;           // MODE_U8, optional
    float *peak_mag;              // MODE_U8, optional
    unsigned int *gmax_bits;      // MODE_U8, optional: f32 bits, only ever raised
    float *out_power;             // MODE_POWER
    float inv_n;                  // 1/N: |2X| * inv_n = |X| * 2/N (lib.rs:150)
    float q_a, q_c;               // code = round(q_a * log2(max(|2X|^2, q_floor)) + q_c), saturated to [0,255] (make_quant)
    float q_floor;                // 0 = no floor (it would not change any code, see make_quant)
    unsigned int band_lo, band_hi;   // MODE_BAND: bins band_lo..band_hi (inclusive, band_hi < N/2); MODE_MAG, tiled kernel, band_hi > 0: only the bins the caller will read (more may be written)
    double band_target;              // MODE_BAND: a frame is active iff its band energy > band_target
    // MODE_BANK: the bank's non-zero taps as bank_taps() lists them (tap k of filter i at k * filters + i, filters with
    // fewer taps padded with weight 0), at most kBankLdsTaps entries; rows of `bank_filters` codes go to out_u8
    const unsigned int *bank_tap_bin;
    const float *bank_tap_w;
    unsigned int bank_filters, bank_max_taps;    // filters; largest number of taps of any filter
    float bank_q_a, bank_q_c;                    // make_bank_quant
    uint8_t *band_active;            // MODE_BAND: one flag per frame; MODE_SURVEY: one per EVEN frame of each stream, ceil(frames / 2) per stream
    double *band_energy;             // MODE_BAND / MODE_SURVEY, optional: the sums themselves
};

SONO_HD float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
SONO_HD float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
SONO_HD float2 cmul(float2 a, float2 w) { return float2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
SONO_HD float2 mul_neg_i(float2 a) { return float2{a.y, -a.x}; }   // -i * a

// Real-input split.  Z is the N/2-point DFT of z[n] = x[2n] + i x[2n+1];
// returns 2*X[k] of the N-point DFT of x, given Z[k], Z[(N/2-k) mod N/2] and
// w = W_N^k.  (X[k] = Xe + W_N^k Xo, Xe = (Zk + conj Zp)/2, Xo = -i (Zk - conj Zp)/2.)
SONO_HD float2 real_split2(float2 zk, float2 zp, float2 w)
{
    const float er = zk.x + zp.x, ei = zk.y - zp.y;      // Zk + conj(Zp)
    const float dr = zk.x - zp.x, di = zk.y + zp.y;      // Zk - conj(Zp)
    // 2*Xo = -i * d = (di, -dr);  w * (di, -dr) = (w.x*di + w.y*dr, w.y*di - w.x*dr)
    // written so that each component contracts into two fused multiply-adds
    return float2{(er + w.y * dr) + w.x * di, (ei - w.x * dr) + w.y * di};
}

// Quantiser constants for the dB -> u8 law of lib.rs:277-293, restated on
// p = |2X|^2 = (mag * N)^2:   db = 20 log10(mag) = 10 log10(p) - 20 log10(N)
//   code = trunc((db - (-gain - range)) * 255/range) = trunc(a * log2(p) + c0), saturated to [0, 255],
// with the reference's floor mag >= 1e-10 carried over as p >= (1e-10 * N)^2.
// The kernels form the code with v_cvt_pk_u8_f32, which saturates and rounds to the
// nearest integer (ties to even): trunc(v) for v >= 0 is round(v - 1/2) except where
// v is an integer exactly, so the constant they add is c = c0 - 1/2.  (An exactly
// integral v with an odd value comes out one code lower; v has 16 fractional bits
// at most, and the 1-LSB tolerance of the u8 rows covers it.)
struct Quant { float a, c, floor; };
inline Quant make_quant(unsigned int N, float gain_db, float range_db)
{
    const double k = 255.0 / (double)range_db;
    const double ten_log10_2 = 3.010299956639812;       // 10*log10(2)
    Quant q;
    q.a = (float)(ten_log10_2 * k);
    const double c0 = (-20.0 * std::log10((double)N) + (double)gain_db + (double)range_db) * k;
    q.c = (float)(c0 - 0.5);
    const double fl = 1e-10 * (double)N;
    q.floor = (float)(fl * fl);
    // Where the floor's own code is 0 (the defaults: the floor lies 100 dB below the lowest
    // coded level) the clamp cannot change a code: log2 of anything below the floor, down to
    // log2(0) = -inf, saturates to 0 as well.  floor = 0 tells the kernels to leave it out.
    const double floor_code = (double)q.a * std::log2((double)q.floor) + c0;
    if (q.a > 0.0f && floor_code < -1.0) q.floor = 0.0f;
    return q;
}

// What the tiled kernel's workgroup has LDS left for: the tap list of a sparse filter bank, one entry per (tap, filter).
constexpr unsigned int kBankLdsTaps = 1536;
struct alignas(8) BankTap { float w; unsigned int bin; };

#if defined(__HIP_DEVICE_COMPILE__)
SONO_HD float fast_log2(float p) { return __builtin_amdgcn_logf(p); }   // v_log_f32 (log2, ~1 ulp)
SONO_HD float fast_sqrt(float p) { return __builtin_amdgcn_sqrtf(p); }  // v_sqrt_f32 (~1 ulp)
#else
SONO_HD float fast_log2(float p) { return std::log2(p); }
SONO_HD float fast_sqrt(float p) { return std::sqrt(p); }
#endif

SONO_HD float as_float(unsigned int u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    std::memcpy(&f, &u, 4);
    return f;
#endif
}

// Product and sum rounded on their own (the reference's wasm has no fused multiply-add): the compiler must not contract them.
#if defined(__HIP_DEVICE_COMPILE__)
SONO_HD float fmul_rn(float a, float b) { return __fmul_rn(a, b); }
SONO_HD float fadd_rn(float a, float b) { return __fadd_rn(a, b); }
#else
SONO_HD float fmul_rn(float a, float b) { volatile float p = a * b; return p; }
SONO_HD float fadd_rn(float a, float b) { volatile float s = a + b; return s; }
#endif

// Larger of two values that are +0, positive or NaN (p = |2X|^2 and the floor
// are): their bit patterns order like the values, and an integer maximum needs
// no NaN-quieting instruction in front of it.  A NaN wins.
SONO_HD float max_nonneg(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned int ua = __float_as_uint(a), ub = __float_as_uint(b);
    return __uint_as_float(ua > ub ? ua : ub);
#else
    unsigned int ua, ub;
    std::memcpy(&ua, &a, 4);
    std::memcpy(&ub, &b, 4);
    return ua > ub ? a : b;
#endif
}

// Largest of four such values.
SONO_HD float max_of_4(float a, float b, float c, float d)
{
    return max_nonneg(max_nonneg(a, b), max_nonneg(c, d));
}

// Saturating float -> u8 conversion, round to nearest even, NaN -> 0, placed in byte
// BYTE of `word` (v_cvt_pk_u8_f32: conversion, clamp and byte insert in one instruction).
template <int BYTE>
SONO_HD unsigned int cvt_pk_u8(float v, unsigned int word)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_cvt_pk_u8_f32(v, (unsigned int)BYTE, word);
#else
    unsigned int code = 0;
    if (v >= 255.0f) code = 255;
    else if (v > 0.0f) code = (unsigned int)std::nearbyint(v);      // default rounding mode: ties to even
    return (word & ~(0xFFu << (8 * BYTE))) | (code << (8 * BYTE));
#endif
}

template <int BYTE = 0>
SONO_HD unsigned int bank_code(float sum, float q_a, float q_c, unsigned int word = 0u)
{
    const float safe = sum > 1e-10f ? sum : 1e-10f;                      // lib.rs:280 (strict >; a NaN takes the floor)
    return cvt_pk_u8<BYTE>(__builtin_fmaf(q_a, fast_log2(safe), q_c), word);
}

// Keeps a wave-uniform value in a vector register: a vector instruction with a
// scalar-register operand issues in 3.2 cycles instead of 1.96 on gfx950, so a constant
// that 32 instructions per tile multiply by is worth the register.
SONO_HD float in_vgpr(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// u8 code of a filter's output `sum` (a linear magnitude; the filter-bank paths cannot take the detour over |2X|^2):
// the law of lib.rs:277-293 on the magnitude itself, code = trunc((20 log10(max(sum, 1e-10)) + gain + range) * 255 / range)
// saturated to [0, 255], formed like quantise() below: one hardware log2, one fused multiply-add with
// a = 20 log10(2) * 255 / range and c = (gain + range) * 255 / range - 1/2, one saturating conversion that rounds to
// nearest (equal to the truncation except on exact integers, see make_quant).
struct BankQuant { float a, c; };
SONO_HD BankQuant make_bank_quant(float gain_db, float range_db)
{
    const double k = 255.0 / (double)range_db;
    BankQuant q;
    q.a = (float)(6.020599913279624 * k);
    q.c = (float)(((double)gain_db + (double)range_db) * k - 0.5);
    return q;
}

// u8 code of one bin from p = |2X|^2 (see make_quant), in byte BYTE of `word`.
template <bool FLOOR = true, int BYTE = 0>
SONO_HD unsigned int quantise(float p, float q_a, float q_c, float q_floor, unsigned int word = 0u)
{
    const float pf = FLOOR ? max_nonneg(p, q_floor) : p;
#ifdef SONO_EXP_NOLOG       // diagnostic build: what the kernel costs without its logarithms (wrong codes)
    return cvt_pk_u8<BYTE>(__builtin_fmaf(q_a, pf, q_c), word);
#else
    return cvt_pk_u8<BYTE>(__builtin_fmaf(q_a, fast_log2(pf), q_c), word);
#endif
}

// Ordering key for "largest magnitude, lowest bin on ties" (lib.rs:370-378):
// magnitudes are >= 0, so their bit patterns order like the values.
SONO_HD unsigned long long peak_key(float mag, unsigned int bin)
{
    unsigned int bits;
#if defined(__HIP_DEVICE_COMPILE__)
    bits = __float_as_uint(mag);
#else
    std::memcpy(&bits, &mag, 4);
#endif
    return ((unsigned long long)bits << 32) | (unsigned long long)(0xFFFFFFFFu - bin);
}

// ---------------------------------------------------------------------------
// Generic path: one 256-thread workgroup transforms a slab of max(N/2, 1024)
// complex points (1024/(N/2) frames for N <= 2048, one frame for N = 4096) with
// a Stockham radix-4 (+ one radix-2 pass when log2(N/2) is odd) FFT staged in
// LDS.  Each method is one barrier-delimited phase executed by thread `t`.
// ---------------------------------------------------------------------------
constexpr int GEN_THREADS = 256;
constexpr int GEN_MAX_FPB = 8;      // frames per block at N = 256

struct GenericShared {
    float2 *buf0, *buf1;                       // 2 x slab complex
    long long *frame_off;                      // [GEN_MAX_FPB] sample offset of the frame, -1 = past the end
    unsigned long long *peak;                  // [GEN_MAX_FPB] peak_key per frame
    float *dc;                                 // [GEN_MAX_FPB] mean of the windowed frame (MODE_POWER)
};

template <int MODE>
struct GenericBlock {
    const StftArgs &a;
    GenericShared sh;
    unsigned long long block;                  // blockIdx.x
    int M, logM, slab, fpb;

    SONO_HD GenericBlock(const StftArgs &args, GenericShared s, unsigned long long blk)
        : a(args), sh(s), block(blk)
    {
        M = (int)(a.N >> 1);
        logM = 0;
        while ((1 << logM) < M) ++logM;
        slab = M > 1024 ? M : 1024;
        fpb = slab / M;
    }

    SONO_HD void phase_setup(int t) const
    {
        if (t < fpb) {
            const unsigned long long total = a.n_streams * a.frames_per_stream;
            const unsigned long long g = block * (unsigned long long)fpb + (unsigned long long)t;
            long long off = -1;
            if (g < total) {
                const unsigned long long s = g / a.frames_per_stream;
                const unsigned long long f = g - s * a.frames_per_stream;
                off = (long long)(s * a.stream_stride + f * a.hop);
            }
            sh.frame_off[t] = off;
            sh.peak[t] = peak_key(0.0f, 0u);
            sh.dc[t] = 0.0f;
        }
    }

    // z[n] = (x[2n] w[2n], x[2n+1] w[2n+1])
    SONO_HD void phase_load(int t) const
    {
        for (int idx = t; idx < slab; idx += GEN_THREADS) {
            const int fb = idx >> logM, n = idx & (M - 1);
            const long long off = sh.frame_off[fb];
            float2 z = float2{0.0f, 0.0f};
            if (off >= 0) {
                const float *xp = a.x + off + 2 * n;
                z.x = xp[0] * a.window[2 * n];
                z.y = xp[1] * a.window[2 * n + 1];
            }
            sh.buf0[idx] = z;
        }
    }

    // Mean of the windowed frame, summed in sample order like lib.rs:1069-1073.
    SONO_HD void phase_dc_sum(int t) const
    {
        if (t < fpb && sh.frame_off[t] >= 0) {
            float s = 0.0f;
            const float2 *z = sh.buf0 + t * M;
            for (int n = 0; n < M; ++n) { s += z[n].x; s += z[n].y; }
            sh.dc[t] = s / (float)a.N;
        }
    }
    SONO_HD void phase_dc_sub(int t) const
    {
        for (int idx = t; idx < slab; idx += GEN_THREADS) {
            const float m = sh.dc[idx >> logM];
            float2 z = sh.buf0[idx];
            z.x -= m; z.y -= m;
            sh.buf0[idx] = z;
        }
    }

    // One Stockham radix-4 pass; p = product of the radices already applied.
    SONO_HD void phase_r4(int t, int p, const float2 *src, float2 *dst) const
    {
        const int quarter = M >> 2, logq = logM - 2;
        int logp = 0;
        while ((1 << logp) < p) ++logp;
        const int tshift = logM - 2 - logp;            // twiddle step M/(4p)
        for (int b = t; b < (slab >> 2); b += GEN_THREADS) {
            const int fb = b >> logq, i = b & (quarter - 1);
            const int k = i & (p - 1);
            const int j = ((i - k) << 2) + k;
            const float2 *s = src + fb * M;
            float2 u0 = s[i], u1 = s[i + quarter], u2 = s[i + 2 * quarter], u3 = s[i + 3 * quarter];
            if (p > 1) {
                const int ti = k << tshift;
                u1 = cmul(u1, a.tw_half[ti]);
                u2 = cmul(u2, a.tw_half[2 * ti]);
                u3 = cmul(u3, a.tw_half[3 * ti]);
            }
            const float2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3), v3 = mul_neg_i(csub(u1, u3));
            float2 *d = dst + fb * M + j;
            d[0] = cadd(v0, v2);
            d[p] = cadd(v1, v3);
            d[2 * p] = csub(v0, v2);
            d[3 * p] = csub(v1, v3);
        }
    }

    // Final radix-2 pass (p = M/2) when log2(M) is odd.
    SONO_HD void phase_r2(int t, const float2 *src, float2 *dst) const
    {
        const int half = M >> 1, logh = logM - 1;
        for (int b = t; b < (slab >> 1); b += GEN_THREADS) {
            const int fb = b >> logh, i = b & (half - 1);
            const float2 *s = src + fb * M;
            const float2 u0 = s[i];
            const float2 u1 = cmul(s[i + half], a.tw_half[i]);
            float2 *d = dst + fb * M;
            d[i] = cadd(u0, u1);
            d[i + half] = csub(u0, u1);
        }
    }

    // Real split + output.  Z = finished N/2-point FFTs of the slab.
    template <typename AtomicMax64>
    SONO_HD void phase_epilogue(int t, const float2 *Z, AtomicMax64 atomic_max64) const
    {
        // each thread takes 4 consecutive bins so that u8 codes pack into one dword
        for (int q = t; q < (slab >> 2); q += GEN_THREADS) {
            const int fb = (q << 2) >> logM, k0 = (q << 2) & (M - 1);
            if (sh.frame_off[fb] < 0) continue;
            const unsigned long long g = block * (unsigned long long)fpb + (unsigned long long)fb;
            const float2 *z = Z + fb * M;
            float p[4];
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + r;
                const float2 X2 = real_split2(z[k], z[(M - k) & (M - 1)], a.tw_full[k]);
                p[r] = X2.x * X2.x + X2.y * X2.y;
            }
            if (MODE == MODE_MAG) {
                float4 o;
                o.x = fast_sqrt(p[0]) * a.inv_n; o.y = fast_sqrt(p[1]) * a.inv_n;
                o.z = fast_sqrt(p[2]) * a.inv_n; o.w = fast_sqrt(p[3]) * a.inv_n;
                *(float4 *)(a.out_mag + g * (unsigned long long)M + k0) = o;
            } else if (MODE == MODE_U8) {
                // The peak is located on p = |2X|^2 (monotone in the magnitude, no sqrt per
                // bin); the winner's magnitude is formed once per frame in phase_features.
                unsigned int packed = 0;
                unsigned long long best = peak_key(0.0f, 0u);
                packed = quantise<true, 0>(p[0], a.q_a, a.q_c, a.q_floor, packed);
                packed = quantise<true, 1>(p[1], a.q_a, a.q_c, a.q_floor, packed);
                packed = quantise<true, 2>(p[2], a.q_a, a.q_c, a.q_floor, packed);
                packed = quantise<true, 3>(p[3], a.q_a, a.q_c, a.q_floor, packed);
                for (int r = 0; r < 4; ++r) {
                    if (p[r] > 0.0f) {                      // false for NaN, like `val > acc`
                        const unsigned long long key = peak_key(p[r], (unsigned int)(k0 + r));
                        best = key > best ? key : best;
                    }
                }
                if (a.out_u8) *(unsigned int *)(a.out_u8 + g * (unsigned long long)M + k0) = packed;
                atomic_max64(&sh.peak[fb], best);
            } else {
                float *o = a.out_power + g * (unsigned long long)(M + 1) + k0;
                for (int r = 0; r < 4; ++r) o[r] = 0.25f * p[r];
                if (k0 == 0) {                              // Nyquist bin: X[N/2] = Re Z0 - Im Z0
                    const float ny = z[0].x - z[0].y;
                    o[M] = ny * ny;
                }
            }
        }
    }

    // Per-frame peak features out of the LDS keys (MODE_U8).
    template <typename AtomicMaxU32>
    SONO_HD void phase_features(int t, AtomicMaxU32 atomic_max_u32) const
    {
        if (t < fpb && sh.frame_off[t] >= 0) {
            const unsigned long long g = block * (unsigned long long)fpb + (unsigned long long)t;
            const unsigned long long key = sh.peak[t];
            const unsigned int bits = (unsigned int)(key >> 32);
            const unsigned int bin = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
            float pmax;
#if defined(__HIP_DEVICE_COMPILE__)
            pmax = __uint_as_float(bits);
#else
            std::memcpy(&pmax, &bits, 4);
#endif
            const float mag = fast_sqrt(pmax) * a.inv_n;
            unsigned int mag_bits;
#if defined(__HIP_DEVICE_COMPILE__)
            mag_bits = __float_as_uint(mag);
#else
            std::memcpy(&mag_bits, &mag, 4);
#endif
            if (a.peak_bin) a.peak_bin[g] = (uint16_t)bin;
            if (a.peak_mag) a.peak_mag[g] = mag;
            if (a.gmax_bits) atomic_max_u32(a.gmax_bits, mag_bits);
        }
    }
};

}  // namespace sono
