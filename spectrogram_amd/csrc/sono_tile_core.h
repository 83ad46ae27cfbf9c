// sono_tile_core.h -- per-lane arithmetic of the register-tiled STFT kernel.
//
// One wave64 transforms a tile of 1024 complex points = 2048 real samples' worth
// of packed frames: 64/M1 frames of N = 64*M1 samples (M1 = 16, 8, 4 for
// N = 1024, 512, 256).  A frame is packed as z[n] = x[2n] + i x[2n+1] (M = 32*M1
// complex points) and transformed in two in-register passes with one exchange
// through wave-private LDS in between:
//
//   phase 1  lane = (half h, n2 in [0,32)); registers = (frame of the half, n1):
//            window, radix-M1 DFT over n1 (n = 32*n1 + n2), twiddle W_M^(n2*k1),
//            write image T[frame][k1][n2]
//   phase 2  lane = (frame, k1 in [0,M1)); registers = n2: radix-32 DFT over n2
//            -> Z[k1 + M1*k2]; write image U[frame][k] in natural order
//   phase 3  lane = (frame, l in [0,M1)) owns bins 4l + 4*M1*m + r (m < 8, r < 4):
//            reads Z[k] and Z[M-k], real split, magnitude / dB / u8 / peak.
//
// Everything is __host__ __device__: tests/host_emul runs the three phases lane
// by lane against an ordinary array standing in for LDS.
#pragma once
#include <type_traits>
#include <utility>

#include "sono_core.h"

#if defined(__clang__)
#define SONO_NO_UNROLL _Pragma("clang loop unroll(disable)")
#else
#define SONO_NO_UNROLL
#endif

namespace sono {
namespace tile {

// ---- compile-time loop -----------------------------------------------------
template <int I>
using ic = std::integral_constant<int, I>;

template <int B, int E, class F>
SONO_HD void static_for(F &&f)
{
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(static_cast<F &&>(f));
    }
}

// ---- compile-time twiddles W_32^k = exp(-2*pi*i*k/32) ------------------------
constexpr float kCos32[9] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                             0.70710678118654757f, 0.55557023301960229f, 0.38268343236508984f,
                             0.19509032201612833f, 0.0f};
constexpr float cos32(int k)
{
    k &= 31;
    return k <= 8 ? kCos32[k] : k <= 16 ? -kCos32[16 - k] : k <= 24 ? -kCos32[k - 16] : kCos32[32 - k];
}
constexpr float w32_re(int k) { return cos32(k); }
constexpr float w32_im(int k) { return -cos32(k + 24); }   // -sin(2*pi*k/32)

constexpr int bitrev(int i, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
    return r;
}
constexpr int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

// Decimation-in-time butterfly (a, b) -> (a + w b, a - w b), w = W_32^K32.
// The second output is formed as 2a - (a + w b), which keeps a general butterfly
// at six fused multiply-adds.
template <int K32>
SONO_HD void bfly(float &ar, float &ai, float &br, float &bi)
{
    if constexpr (K32 == 0) {
        const float xr = ar + br, xi = ai + bi;
        br = ar - br; bi = ai - bi; ar = xr; ai = xi;
    } else if constexpr (K32 == 8) {            // w = -i: w b = (b.im, -b.re)
        const float xr = ar + bi, xi = ai - br;
        const float yr = ar - bi, yi = ai + br;
        ar = xr; ai = xi; br = yr; bi = yi;
    } else if constexpr (K32 == 4) {            // w = (1 - i)/sqrt 2: w b = c (b.re + b.im, b.im - b.re)
        constexpr float c = 0.70710678118654757f;
        const float s1 = br + bi, s2 = bi - br;
        const float xr = ar + c * s1, xi = ai + c * s2;
        br = ar - c * s1; bi = ai - c * s2; ar = xr; ai = xi;
    } else if constexpr (K32 == 12) {           // w = (-1 - i)/sqrt 2: w b = c (b.im - b.re, -(b.re + b.im))
        constexpr float c = 0.70710678118654757f;
        const float s1 = bi - br, s2 = br + bi;
        const float xr = ar + c * s1, xi = ai - c * s2;
        br = ar - c * s1; bi = ai + c * s2; ar = xr; ai = xi;
    } else {
        constexpr float wr = w32_re(K32), wi = w32_im(K32);
        const float xr = ar + (wr * br - wi * bi);
        const float xi = ai + (wr * bi + wi * br);
        br = 2.0f * ar - xr; bi = 2.0f * ai - xi; ar = xr; ai = xi;
    }
}

// In-place N-point DFT (N = 2..32).  Input in bit-reversed order
// (re[i] = x[bitrev(i)]), output in natural order.
template <int N>
SONO_HD void dft_bitrev_in(float (&re)[N], float (&im)[N])
{
    constexpr int L = ilog2(N);
    static_for<1, L + 1>([&](auto s_) {
        constexpr int len = 1 << decltype(s_)::value, half = len / 2;
        static_for<0, N / len>([&](auto b_) {
            constexpr int base = decltype(b_)::value * len;
            static_for<0, half>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                bfly<j * (32 / len)>(re[base + j], im[base + j], re[base + j + half], im[base + j + half]);
            });
        });
    });
}

// ---- tile geometry -----------------------------------------------------------
constexpr int kRowStride = 34;                 // complex entries per image row: 32 + 2 pad (16-byte aligned, bank-conflict free)
constexpr int kWaveLdsComplex = 64 * kRowStride;   // one wave's LDS image, in complex entries (17 408 B)

template <int LOG_M1>
struct Geo {
    static constexpr int M1 = 1 << LOG_M1;     // pass-1 radix = lanes per frame
    static constexpr int M = 32 * M1;          // complex points per frame
    static constexpr int N = 2 * M;            // real samples per frame
    static constexpr int FPW = 64 / M1;        // frames per wave
    static constexpr int FPH = FPW / 2;        // frames per half-wave (phase 1)
};

// Per-tile frame descriptor: sample offset of each of the wave's frames
// (negative = frame does not exist), indexed by frame-in-wave.
struct TileFrames {
    long long off[16];
};

// ---- phase 1 -------------------------------------------------------------------
// in:   xr/xi[fr*M1 + n1] = samples 2n, 2n+1 (n = 32*n1 + n2) of frame h*FPH + fr
//       wr/wi[n1]         = window at those positions
//       tw_re/tw_im[k1]   = W_M^(n2*k1)
// out:  image T: entry ((h*FPH + fr)*M1 + k1)*kRowStride + n2
template <int LOG_M1>
SONO_HD void phase1(int lane, const float (&xr)[32], const float (&xi)[32], const float (&wr)[Geo<LOG_M1>::M1],
                    const float (&wi)[Geo<LOG_M1>::M1], const float (&tw_re)[Geo<LOG_M1>::M1],
                    const float (&tw_im)[Geo<LOG_M1>::M1], float2 *T)
{
    using G = Geo<LOG_M1>;
    const int h = lane >> 5, n2 = lane & 31;
    static_for<0, G::FPH>([&](auto fr_) {
        constexpr int fr = decltype(fr_)::value;
        float re[G::M1], im[G::M1];
        static_for<0, G::M1>([&](auto i_) {
            constexpr int i = decltype(i_)::value, n1 = bitrev(i, LOG_M1);
            re[i] = xr[fr * G::M1 + n1] * wr[n1];
            im[i] = xi[fr * G::M1 + n1] * wi[n1];
        });
        dft_bitrev_in<G::M1>(re, im);
        float2 *row = T + ((h * G::FPH + fr) * G::M1) * kRowStride + n2;
        row[0] = float2{re[0], im[0]};
        static_for<1, G::M1>([&](auto k_) {
            constexpr int k1 = decltype(k_)::value;
            const float yr = re[k1] * tw_re[k1] - im[k1] * tw_im[k1];
            const float yi = re[k1] * tw_im[k1] + im[k1] * tw_re[k1];
            row[k1 * kRowStride] = float2{yr, yi};
        });
    });
}

// ---- phase 2 -------------------------------------------------------------------
// lane = frame*M1 + k1 reads its image row (32 entries), runs the radix-32 DFT and
// writes Z[k1 + M1*k2] to the natural-order image U[frame*M + k].
template <int LOG_M1>
SONO_HD void phase2_load(int lane, const float2 *T, float (&re)[32], float (&im)[32])
{
    const float4 *row = (const float4 *)(T + lane * kRowStride);     // row = frame*M1 + k1 = lane
    static_for<0, 16>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        const float4 v = row[j];                                     // entries n2 = 2j, 2j+1
        re[bitrev(2 * j, 5)] = v.x; im[bitrev(2 * j, 5)] = v.y;
        re[bitrev(2 * j + 1, 5)] = v.z; im[bitrev(2 * j + 1, 5)] = v.w;
    });
}

template <int LOG_M1>
SONO_HD void phase2_store(int lane, const float (&re)[32], const float (&im)[32], float2 *U)
{
    using G = Geo<LOG_M1>;
    const int frame = lane >> LOG_M1, k1 = lane & (G::M1 - 1);
    float2 *u = U + frame * G::M + k1;
    static_for<0, 32>([&](auto k_) {
        constexpr int k2 = decltype(k_)::value;
        u[k2 * G::M1] = float2{re[k2], im[k2]};
    });
}

// ---- phase 3 -------------------------------------------------------------------
// |2X[k]|^2 for the four bins k0..k0+3 (k0 = 4l + 4*M1*m) of the lane's frame.
// Zf = U + frame*M; tw = W_N^k table (N/2 entries, 16-byte aligned).
template <int LOG_M1>
SONO_HD void phase3_group(const float2 *Zf, const float2 *tw, int k0, float (&p)[4])
{
    using G = Geo<LOG_M1>;
    const float4 z01 = *(const float4 *)(Zf + k0), z23 = *(const float4 *)(Zf + k0 + 2);
    const float4 w01 = *(const float4 *)(tw + k0), w23 = *(const float4 *)(tw + k0 + 2);
    const int K = G::M - k0;                            // partner of k0 (mod M); partners of k0+1.. are K-1..
    const float2 p0 = Zf[K & (G::M - 1)];
    const float4 p21 = *(const float4 *)(Zf + K - 2);   // entries K-2, K-1 (K-2 is even: 16-byte aligned)
    const float2 p3 = Zf[K - 3];
    const float2 p2 = float2{p21.x, p21.y}, p1 = float2{p21.z, p21.w};
    const float2 x0 = real_split2(float2{z01.x, z01.y}, p0, float2{w01.x, w01.y});
    const float2 x1 = real_split2(float2{z01.z, z01.w}, p1, float2{w01.z, w01.w});
    const float2 x2 = real_split2(float2{z23.x, z23.y}, p2, float2{w23.x, w23.y});
    const float2 x3 = real_split2(float2{z23.z, z23.w}, p3, float2{w23.z, w23.w});
    p[0] = x0.x * x0.x + x0.y * x0.y;
    p[1] = x1.x * x1.x + x1.y * x1.y;
    p[2] = x2.x * x2.x + x2.y * x2.y;
    p[3] = x3.x * x3.x + x3.y * x3.y;
}

// ---- one lane of one wave ------------------------------------------------------
// Tiles never straddle streams: stream s owns tiles [s*tps, (s+1)*tps), tile j of
// a stream covers its frames j*FPW .. j*FPW + FPW-1 (the last tile may be partial).
struct TileIter {
    unsigned int tiles_per_stream;     // ceil(frames_per_stream / FPW)
    unsigned long long n_tiles;        // n_streams * tiles_per_stream
};

template <int LOG_M1, int MODE>
struct TileLane {
    using G = Geo<LOG_M1>;
    const StftArgs &a;
    int lane;
    // per-lane constants
    float wr[G::M1], wi[G::M1];        // window at samples 64*n1 + 2*n2 (+1)
    float t1r[G::M1], t1i[G::M1];      // W_M^(n2*k1)
    // tile inputs (prefetched one tile ahead)
    float xr[32], xi[32];
    // running maximum of |2X|^2 over every frame this lane has seen (MODE_U8)
    float run_max;

    SONO_HD TileLane(const StftArgs &args, int lane_) : a(args), lane(lane_), run_max(0.0f)
    {
        const int n2 = lane & 31;
        static_for<0, G::M1>([&](auto n_) {
            constexpr int n1 = decltype(n_)::value;
            wr[n1] = a.window[64 * n1 + 2 * n2];
            wi[n1] = a.window[64 * n1 + 2 * n2 + 1];
            const float2 w = a.tw_half[(n2 * n1) & (G::M - 1)];
            t1r[n1] = w.x; t1i[n1] = w.y;
        });
    }

    // Issues the loads of `tile` (phase-1 lane layout).  Frames past the end of a
    // stream re-read the stream's frame 0; their results are never stored.
    SONO_HD void load(unsigned long long tile, const TileIter &it)
    {
        const unsigned long long s = tile / it.tiles_per_stream;
        const unsigned int j = (unsigned int)(tile - s * it.tiles_per_stream);
        const int h = lane >> 5, n2 = lane & 31;
        const float *xs = a.x + s * a.stream_stride + 2 * n2;
        static_for<0, G::FPH>([&](auto fr_) {
            constexpr int fr = decltype(fr_)::value;
            unsigned long long f = (unsigned long long)j * G::FPW + (unsigned int)(h * G::FPH + fr);
            if (f >= a.frames_per_stream) f = 0;
            const float2 *src = (const float2 *)(xs + f * a.hop);
            static_for<0, G::M1>([&](auto n_) {
                constexpr int n1 = decltype(n_)::value;
                const float2 v = src[32 * n1];
                xr[fr * G::M1 + n1] = v.x;
                xi[fr * G::M1 + n1] = v.y;
            });
        });
    }

    SONO_HD void run_phase1(float2 *img) { phase1<LOG_M1>(lane, xr, xi, wr, wi, t1r, t1i, img); }

    SONO_HD void run_phase2_load(const float2 *img, float (&re)[32], float (&im)[32]) const
    {
        phase2_load<LOG_M1>(lane, img, re, im);
    }
    SONO_HD void run_phase2_fft_store(float (&re)[32], float (&im)[32], float2 *img) const
    {
        dft_bitrev_in<32>(re, im);
        phase2_store<LOG_M1>(lane, re, im, img);
    }

    // Real split + outputs for the lane's 32 bins.  `peak_slot` is the LDS word of
    // the lane's frame (ordering key of the frame's largest |2X|^2).
    template <typename AtomicMax64>
    SONO_HD void run_phase3(unsigned long long tile, const TileIter &it, const float2 *img, const float2 *tw,
                            unsigned long long *peak_slot, AtomicMax64 atomic_max64)
    {
        const unsigned long long s = tile / it.tiles_per_stream;
        const unsigned int j = (unsigned int)(tile - s * it.tiles_per_stream);
        const int frame = lane >> LOG_M1, l = lane & (G::M1 - 1);
        const unsigned long long f = (unsigned long long)j * G::FPW + (unsigned int)frame;
        const bool valid = f < a.frames_per_stream;
        const unsigned long long g = s * a.frames_per_stream + f;
        const float2 *Zf = img + frame * G::M;
        float best_p = 0.0f;
        unsigned int best_k = 0;
        SONO_NO_UNROLL
        for (int m = 0; m < 8; ++m) {
            const int k0 = 4 * l + 4 * G::M1 * m;
            float p[4];
            phase3_group<LOG_M1>(Zf, tw, k0, p);
            if (MODE == MODE_MAG) {
                if (valid) {
                    float4 o;
                    o.x = fast_sqrt(p[0]) * a.inv_n; o.y = fast_sqrt(p[1]) * a.inv_n;
                    o.z = fast_sqrt(p[2]) * a.inv_n; o.w = fast_sqrt(p[3]) * a.inv_n;
                    *(float4 *)(a.out_mag + g * (unsigned long long)G::M + k0) = o;
                }
            } else {
                unsigned int packed = 0;
                static_for<0, 4>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    packed |= quantise(p[r], a.q_a, a.q_c, a.q_floor) << (8 * r);
                    if (p[r] > best_p) { best_p = p[r]; best_k = (unsigned int)(k0 + r); }   // strict >: first wins
                });
                if (valid) *(unsigned int *)(a.out_u8 + g * (unsigned long long)G::M + k0) = packed;
            }
        }
        if (MODE == MODE_U8 && valid) {
            run_max = best_p > run_max ? best_p : run_max;
            atomic_max64(peak_slot, peak_key(best_p, best_k));
        }
    }

    // After every lane of the frame has contributed: the frame's first lane emits
    // the features and clears the slot for the next tile.
    SONO_HD void run_phase4(unsigned long long tile, const TileIter &it, unsigned long long *peak_slot) const
    {
        const int frame = lane >> LOG_M1, l = lane & (G::M1 - 1);
        if (MODE != MODE_U8 || l != 0) return;
        const unsigned long long s = tile / it.tiles_per_stream;
        const unsigned int j = (unsigned int)(tile - s * it.tiles_per_stream);
        const unsigned long long f = (unsigned long long)j * G::FPW + (unsigned int)frame;
        const unsigned long long key = *peak_slot;
        *peak_slot = peak_key(0.0f, 0u);
        if (f >= a.frames_per_stream) return;
        const unsigned long long g = s * a.frames_per_stream + f;
        unsigned int bits = (unsigned int)(key >> 32);
        float pmax;
#if defined(__HIP_DEVICE_COMPILE__)
        pmax = __uint_as_float(bits);
#else
        std::memcpy(&pmax, &bits, 4);
#endif
        if (a.peak_bin) a.peak_bin[g] = (uint16_t)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
        if (a.peak_mag) a.peak_mag[g] = fast_sqrt(pmax) * a.inv_n;
    }
};

}  // namespace tile
}  // namespace sono
