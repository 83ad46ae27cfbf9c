// sono_tile_core.h -- per-lane arithmetic of the register-tiled STFT kernel.
//
// One wave64 transforms FPW = 128/R1 real frames of N = R1*R2 samples, two frames
// ("a pair": A = 2q, B = 2q+1) at a time, with a two-pass real-input FFT that
// never forms the redundant half of the spectrum.  R2 = 32 with R1 = 64, 32, 16, 8
// serves N = 2048, 1024, 512, 256; R2 = 64 with R1 = 64 serves N = 4096 (and, as an
// alternative that the tests keep alive, R1 = 32 serves N = 2048).
// Written out for R2 = 32:
//
//   X[k1 + R1*k2] = sum_{n2<32} W_32^(n2*k2) * W_N^(n2*k1) * Y[k1][n2],
//   Y[k1][n2]     = sum_{n1<R1} xw[32*n1 + n2] * W_R1^(n1*k1)          (xw = x * window)
//
//   pass 1  lane = (h, n2), R1 register points n1: one real-input DFT-R1 per
//           frame (rdft_bitrev_in: half the work of a complex one) gives Y_A and
//           Y_B for k1 = 0..R1/2.  Rows k1 = 1..R1/2-1 of each frame are multiplied
//           by W_N^(n2*k1); the two real rows k1 = 0 and k1 = R1/2 of both frames
//           ride together as Y_A + i*Y_B (times W_N^(n2*R1/2) for k1 = R1/2).  Each
//           row is written to the wave's LDS image [row][n2], real parts first,
//           imaginary parts in a second round over the same 8 KB.
//   pass 2  lane = row (pair q, rho): rho = k1 in 1..R1/2-1 carries frame A,
//           rho = R1/2 + k1 frame B, rho = 0 and rho = R1/2 the two shared real
//           rows.  One complex DFT-32 over n2 leaves 2*X[k1 + R1*k2] in register
//           k2; registers k2 >= 16 hold the conjugates of bins
//           (R1-k1) + R1*(31-k2), so a lane ends with 32 distinct bins < N/2 of
//           its frame and nothing is computed twice.  The shared rows hold
//           DFT32(Y_A + i*Y_B); |2*X_A|^2 and |2*X_B|^2 of their 16 bins are
//           |C[t] +- conj C[partner(t)]|^2, formed by the same instructions that
//           square the frame rows (see pass2_power).
//   power   every lane scatters its 32 values of |2X|^2 into a bin-major image;
//           the epilogue reads it back four consecutive bins per lane:
//           log2 -> dB -> u8 -> one dword store, running arg-max for the
//           per-frame peak.
//
//   scale   Y_A + i*Y_B in one DFT leaks rounding noise of the louder frame's
//           rows into the quieter one's (about 1e-7 of the LOUDER level).  Only
//           the two shared rows are affected -- every other row of pass 2 holds one
//           frame alone.  Frame B's share of them is therefore multiplied by the
//           power of two that brings its largest entry to the binade of frame
//           A's (exact), and the epilogue multiplies |2X_B|^2 of the bins that came
//           through those rows (the multiples of R1/2) by its inverse square.  A
//           frame whose share is exactly zero gets the factor 0: its bins are
//           exactly zero, as they are in every other row without any help.
//
//   hop     With hop = N/4 or N/2 (HOPQ = 1, 2) frame B's first samples ARE frame
//           A's samples R1/4 (R1/2) rows further down the same column, so only
//           B's last R1/4 (R1/2) rows are loaded; HOPQ = 0 loads both frames whole.
//
//   split   R1 = 64 with R2 = 32 (N = 2048): a lane's 64 registers hold ONE frame's column, so
//           the two frames of the pair sit in the two halves of the wave during pass 1
//           (lanes 0..31 frame A, 32..63 frame B) and every frame is loaded whole.  What the
//           shared rows need of the other frame (Y[0], Y[R1/2], the maxima for the scale)
//           crosses between the halves in the caller (v_permlane32_swap in the kernel).
//           Everything after pass 1 is the same code: 64 rows of 32 columns.
//
// Everything is __host__ __device__: tests/host_emul runs the passes lane by
// lane against an ordinary array standing in for LDS.
#pragma once
#include <type_traits>
#include <utility>

#include "sono_core.h"

#ifndef SONO_EPI_UNROLL
#define SONO_EPI_UNROLL 1
#endif
#ifndef SONO_EPI_BATCH
#define SONO_EPI_BATCH 8          // epilogue reads in flight per lane, fft 1024
#endif
#ifndef SONO_EPI_BATCH_512
#define SONO_EPI_BATCH_512 8      // the same, fft 512 and 256 (4: 1.70 ms, 8: 1.66 ms per hour of audio at 512)
#endif
#ifndef SONO_EPI_BATCH_WIDE
#define SONO_EPI_BATCH_WIDE 8     // the same, fft 2048 (16: 3.46 against 3.42 ms per hour of audio)
#endif
#ifndef SONO_EPI_BATCH_4096
#define SONO_EPI_BATCH_4096 16    // the same, fft 4096 (8: 2.88, 16: 2.82 ms per 10 min at 95 % overlap)
#endif
#define SONO_PRAGMA_(x) _Pragma(#x)
#define SONO_PRAGMA(x) SONO_PRAGMA_(x)
#if defined(__clang__)
#define SONO_EPILOGUE_UNROLL SONO_PRAGMA(clang loop unroll_count(SONO_EPI_UNROLL))
#else
#define SONO_EPILOGUE_UNROLL
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

// ---- compile-time twiddles W_64^k = exp(-2*pi*i*k/64) ------------------------
constexpr float kCos64[17] = {1.0f, 0.99518472667219693f, 0.98078528040323043f, 0.95694033573220882f,
                              0.92387953251128674f, 0.88192126434835505f, 0.83146961230254524f,
                              0.77301045336273699f, 0.70710678118654757f, 0.63439328416364549f,
                              0.55557023301960229f, 0.47139673682599764f, 0.38268343236508984f,
                              0.29028467725446233f, 0.19509032201612833f, 0.098017140329560604f, 0.0f};
constexpr float cos64(int k)
{
    k &= 63;
    return k <= 16 ? kCos64[k] : k <= 32 ? -kCos64[32 - k] : k <= 48 ? -kCos64[k - 32] : kCos64[64 - k];
}
constexpr float w64_re(int k) { return cos64(k); }
constexpr float w64_im(int k) { return -cos64(k + 48); }   // -sin(2*pi*k/64)

constexpr int bitrev(int i, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
    return r;
}
constexpr int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

// Decimation-in-time butterfly (a, b) -> (a + w b, a - w b), w = W_64^K64.
// A general butterfly is six fused multiply-adds: a + w b as two chained ones per
// component, the second output as 2a - (a + w b).  Every constant is a literal of
// the instruction (v_fmamk_f32 / v_fmaak_f32) or an inline constant; no operand
// carries a negation of a product, which would force the constant into a scalar
// register -- and a vector instruction with a scalar-register operand issues in
// 3.2 cycles instead of 1.96 on gfx950 (tools/valubench.hip).
template <int K64>
SONO_HD void bfly(float &ar, float &ai, float &br, float &bi)
{
    if constexpr (K64 == 0) {
        const float xr = ar + br, xi = ai + bi;
        br = ar - br; bi = ai - bi; ar = xr; ai = xi;
    } else if constexpr (K64 == 16) {           // w = -i: w b = (b.im, -b.re)
        const float xr = ar + bi, xi = ai - br;
        const float yr = ar - bi, yi = ai + br;
        ar = xr; ai = xi; br = yr; bi = yi;
    } else if constexpr (K64 == 8) {            // w = (1 - i)/sqrt 2: w b = c (b.re + b.im, b.im - b.re)
        constexpr float c = 0.70710678118654757f, nc = -c;
        const float s1 = br + bi, s2 = bi - br;
        const float xr = __builtin_fmaf(s1, c, ar), xi = __builtin_fmaf(s2, c, ai);
        br = __builtin_fmaf(s1, nc, ar); bi = __builtin_fmaf(s2, nc, ai); ar = xr; ai = xi;
    } else if constexpr (K64 == 24) {           // w = (-1 - i)/sqrt 2: w b = c (b.im - b.re, -(b.re + b.im))
        constexpr float c = 0.70710678118654757f, nc = -c;
        const float s1 = bi - br, s2 = br + bi;
        const float xr = __builtin_fmaf(s1, c, ar), xi = __builtin_fmaf(s2, nc, ai);
        br = __builtin_fmaf(s1, nc, ar); bi = __builtin_fmaf(s2, c, ai); ar = xr; ai = xi;
    } else {
        constexpr float wr = w64_re(K64), wi = w64_im(K64), nwi = -wi;
        const float xr = __builtin_fmaf(br, wr, __builtin_fmaf(bi, nwi, ar));
        const float xi = __builtin_fmaf(bi, wr, __builtin_fmaf(br, wi, ai));
        br = __builtin_fmaf(2.0f, ar, -xr); bi = __builtin_fmaf(2.0f, ai, -xi); ar = xr; ai = xi;
    }
}

// In-place N-point DFT (N = 2..64).  Input in bit-reversed order
// (re[i] = x[bitrev(i)]), output in natural order.
// FIRST = 2: the caller has already done the first stage (pairs 2j, 2j+1).
template <int N, int FIRST = 1>
SONO_HD void dft_bitrev_in(float (&re)[N], float (&im)[N])
{
    constexpr int L = ilog2(N);
    static_for<FIRST, L + 1>([&](auto s_) {
        constexpr int len = 1 << decltype(s_)::value, half = len / 2;
        static_for<0, N / len>([&](auto b_) {
            constexpr int base = decltype(b_)::value * len;
            static_for<0, half>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                bfly<j * (64 / len)>(re[base + j], im[base + j], re[base + j + half], im[base + j + half]);
            });
        });
    });
}

// In-place DFT of N real values (N = 2..64), input in bit-reversed order, output
// in half-complex order: Re X[k] in v[k] (k = 0..N/2), Im X[k] in v[N-k]
// (k = 1..N/2-1).  Decimation in time on real data: a size-M block is combined
// from the half-complex DFTs E (even samples, first half of the block) and O (odd
// samples, second half):
//   X[0] = E0 + O0,  X[M/2] = E0 - O0,  X[M/4] = E[M/4] - i O[M/4]  (all of E0, O0, E[M/4], O[M/4] real; see rdft_flip),
//   X[k] = E[k] + W_M^k O[k],  X[M/2-k] = conj(E[k] - W_M^k O[k])   for 0 < k < M/4,
// about 200 operations at N = 32, 400 for the two frames of a pair, where one
// complex transform of the pair takes 430 and the separation of the two spectra
// 60 more.  FIRST = 4: the size-2 blocks
// (v[2j] + v[2j+1], v[2j] - v[2j+1]; the difference negated in the odd blocks, see rdft_flip)
// have been formed by the caller.
// A block that is the odd-sample half of its parent (rdft_flip) stores its X[M/2]
// negated: the parent wants Im X[M/4] = -O[M/4] in exactly that place.
template <int M, int BASE, int N>
constexpr bool rdft_flip() { return M < N && ((BASE / M) & 1) != 0; }
template <int M, int BASE, int N>
SONO_HD void rdft_combine(float (&v)[N])
{
    constexpr int H = M / 2, Q = M / 4;
    const float e0 = v[BASE], o0 = v[BASE + H];
    v[BASE] = e0 + o0;
    v[BASE + H] = rdft_flip<M, BASE, N>() ? o0 - e0 : e0 - o0;
    static_for<1, Q>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        constexpr float wr = w64_re(k * (64 / M)), wi = w64_im(k * (64 / M));
        constexpr float nwi = -wi;
        const float er = v[BASE + k], ei = v[BASE + H - k], orr = v[BASE + H + k], oi = v[BASE + M - k];
        const float xr = __builtin_fmaf(orr, wr, __builtin_fmaf(oi, nwi, er));
        const float xi = __builtin_fmaf(oi, wr, __builtin_fmaf(orr, wi, ei));
        v[BASE + k] = xr;
        v[BASE + M - k] = xi;
        v[BASE + H - k] = __builtin_fmaf(2.0f, er, -xr);
        v[BASE + H + k] = __builtin_fmaf(-2.0f, ei, xi);
    });
}
template <int N, int FIRST = 2>
SONO_HD void rdft_bitrev_in(float (&v)[N])
{
    constexpr int L = ilog2(N), L0 = ilog2(FIRST);
    static_for<L0, L + 1>([&](auto s_) {
        constexpr int M = 1 << decltype(s_)::value;
        static_for<0, N / M>([&](auto b_) { rdft_combine<M, decltype(b_)::value * M>(v); });
    });
}

// Makes the compiler finish computing v[0..N) here: without it the part of a
// transform that the next few instructions do not need is sunk below them, and its
// inputs (twice as many registers) stay alive instead of its outputs.
template <int G0 = 0, int N>
SONO_HD void pin_registers(float (&v)[N])
{
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(N % 8 == 0, "pinned in groups of eight");
    asm volatile("" : "+v"(v[G0]), "+v"(v[G0 + 1]), "+v"(v[G0 + 2]), "+v"(v[G0 + 3]),
                      "+v"(v[G0 + 4]), "+v"(v[G0 + 5]), "+v"(v[G0 + 6]), "+v"(v[G0 + 7]));
    if constexpr (G0 + 8 < N) pin_registers<G0 + 8>(v);
#else
    (void)v;
#endif
}

// Hides where a value comes from, so that the compiler computes what depends on it at the
// point of use instead of keeping it in registers from before the tile loop.
SONO_HD unsigned int opaque(unsigned int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// dst = src as one move instruction that stays a move: written as a plain assignment under a
// lane-dependent condition it would be turned into a select, which costs twice as much.
SONO_HD void move_in_place(float &dst, float src)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mov_b32 %0, %1" : "+v"(dst) : "v"(src));
#else
    dst = src;
#endif
}

// ---- tile geometry -----------------------------------------------------------
// N = R1 * R2: pass 1 transforms R1 register points per lane, pass 2 R2.  R2 = 32
// serves N = 256..1024 (R1 = 8..32), R2 = 64 serves N = 2048 and 4096 (R1 = 32, 64).
template <int LOG_R1, int LOG_R2>
struct Geo {
    static constexpr int R1 = 1 << LOG_R1;     // pass-1 radix = image rows per frame pair
    static constexpr int R2 = 1 << LOG_R2;     // pass-2 radix = image columns = values per lane in either pass
    static constexpr int N = R1 * R2;          // real samples per frame
    static constexpr int BINS = N / 2;
    static constexpr int HALF = R1 / 2;        // the Nyquist row of pass 1; lanes per frame in the epilogue
    static constexpr int PPW = 64 / R1;        // frame pairs per wave
    static constexpr int FPW = 2 * PPW;        // frames per wave
    static constexpr bool SPLIT = R1 > R2;     // pass 1 holds one frame per lane, the pair's frames in the wave's two halves
    static constexpr int JPL = SPLIT ? 1 : R2 / R1;   // pass-1 jobs (pairs) per lane
    // Floats per frame in the power image.  The pad puts the frames that one 32-lane
    // group of pass 2 stores to on disjoint banks (ds_write_b32: 32 banks).  The
    // epilogue's 16-byte reads go by 64 banks in 16-lane groups that take lanes from
    // two frames; at R1 = 32 they are conflict-free as well if frame f's lanes take
    // their bins ROT*f places further round the frame (bin_lane below).
    static constexpr int PAD = R1 < 64 ? HALF : 0;
    static constexpr int PROW = BINS + PAD;
    static constexpr int ROT = R1 == 32 ? PAD / 4 : 0;
    // Inter-pass twiddles W_N^(n2*k1), one row of R2 entries per k1.  Unfolded: rows
    // k1 = 1..R1/2.  Folded (R1 = 64, where the full table would not fit beside
    // eight wave regions): rows k1 = 1..R1/4, then one row of W_N^(n2*R1/2); the
    // rows above R1/4 come from W^(n2*(R1/2-k)) = W^(n2*R1/2) * conj W^(n2*k).
    static constexpr bool TW_FOLD = R1 == 64 && R2 == 64;
    static constexpr int TW_ROWS = TW_FOLD ? HALF / 2 + 1 : HALF;
    // A wave's LDS region is first the exchange image (64 rows of R2 floats), then
    // the power image (FPW rows of PROW floats).
    static constexpr int REGION = 64 * R2 > FPW * PROW ? 64 * R2 : FPW * PROW;
};

// Exchange image: row L (one per pass-2 lane) keeps its 16-byte chunks in the
// order chunk ^ swizzle(L).  With 32-float rows swizzle = (L >> 1) & 7, with
// 64-float rows L & 15: the 16-lane groups in which ds_read_b128 is serviced then
// touch 16 disjoint 4-bank sets when every lane reads the same chunk of its own
// row, and a pass-1 store (one row, consecutive columns) still covers distinct banks.
template <int LOG_R2>
SONO_HD constexpr int image_swizzle(int row) { return LOG_R2 == 5 ? ((row >> 1) & 7) << 2 : (row & 15) << 2; }
template <int LOG_R2>
SONO_HD constexpr int image_index(int row, int col) { return (row << LOG_R2) + (col ^ image_swizzle<LOG_R2>(row)); }

// Per-wave, per-tile side information handed from pass 1 to the epilogue through
// LDS: [kMetaFix + frame] = the factor for |2X|^2 of the frame's bins that came
// through the rows it shares with the other frame of its pair (see pass1_rows).
constexpr int kMetaFix = 0;
constexpr int kMetaFloats = 16;

#if defined(__HIP_DEVICE_COMPILE__)
SONO_HD int float_exponent(float v) { return __builtin_amdgcn_frexp_expf(v); }
SONO_HD float pow2(int e) { return __builtin_amdgcn_ldexpf(1.0f, e); }
#else
SONO_HD int float_exponent(float v) { int e = 0; (void)std::frexp(v, &e); return e; }
SONO_HD float pow2(int e) { return std::ldexp(1.0f, e); }
#endif

// Per-launch tile walk.  Tiles never straddle streams; tile j of a stream covers
// its frames j*FPW .. j*FPW+FPW-1 (the last tile of a stream may be partial).
// Uniform batch: every stream has frames_per_stream frames and stream_stride
// samples between starts; stream s owns tiles [s*tps, (s+1)*tps).  Ragged batch:
// every stream has its own start, frame count and first output row; a per-tile
// table names the stream of each tile.
struct RaggedStream {
    unsigned long long sample_offset;  // first sample of the stream, relative to StftArgs::x
    unsigned long long frames;         // frames of the stream (0 if it is shorter than a frame)
    unsigned long long row0;           // output row of its frame 0 (rows of all streams are contiguous, in stream order)
    unsigned long long tile0;          // index of its first tile
};
struct TileIter {
    unsigned int tiles_per_stream;     // uniform: ceil(frames_per_stream / FPW)
    unsigned long long n_tiles;        // tiles of the launch
    const RaggedStream *streams;       // ragged: one per stream (device memory); nullptr for a uniform batch
    const unsigned int *tile_stream;   // ragged: stream of every tile (device memory)
    // The launch is cut into one contiguous range of tiles per XCD ("part": workgroup b belongs to part b % parts).
    // part_begin[p] .. part_begin[p + 1] is part p's range when `parts` is not 0 (then it is the number of parts the
    // kernel will find, kTileParts, and n_tiles fits 32 bits); parts = 0: equal ranges, n_tiles * p / parts.
    unsigned int parts;
    unsigned int part_begin[9];
    // Where every workgroup leaves {first start, last end (100 MHz ticks), tiles of its part, `stamp`} when its last
    // wave is done (host memory the device can write; nullptr: nowhere): what the next launch's ranges are cut by.
    unsigned long long *feedback;
    unsigned long long stamp;
};
constexpr unsigned int kTileParts = 8;      // XCDs of the chip
constexpr unsigned int kFeedbackWords = 4;  // 64-bit words per workgroup in TileIter::feedback
// Where one tile reads and writes.
struct TileRef {
    const float *xs;                   // sample 0 of the tile's stream (f32 input)
    unsigned long long xs_off;         // the same as an offset from the start of the input, in samples (sample frames)
    unsigned long long frames;         // frames of that stream
    unsigned long long row0;           // output row of the stream's frame 0
    unsigned int j;                    // the tile's index inside the stream
    unsigned int rem;                  // how many of the tile's FPW frames exist (fewer than FPW only in the last tile of a stream)
};
template <bool RAGGED, int FPW>
SONO_HD TileRef locate(const StftArgs &a, const TileIter &it, unsigned long long tile)
{
    TileRef r;
    if (RAGGED) {
        const RaggedStream st = it.streams[it.tile_stream[tile]];
        r.xs = a.x + st.sample_offset;
        r.xs_off = st.sample_offset;
        r.frames = st.frames;
        r.row0 = st.row0;
        r.j = (unsigned int)(tile - st.tile0);
        const unsigned long long left = st.frames - (unsigned long long)r.j * FPW;
        r.rem = left < (unsigned long long)FPW ? (unsigned int)left : (unsigned int)FPW;
    } else {
        const unsigned long long s = tile / it.tiles_per_stream;
        r.xs = a.x + s * a.stream_stride;
        r.xs_off = s * a.stream_stride;
        r.frames = a.frames_per_stream;
        r.row0 = s * a.frames_per_stream;
        r.j = (unsigned int)(tile - s * it.tiles_per_stream);
        const unsigned long long left = a.frames_per_stream - (unsigned long long)r.j * FPW;
        r.rem = left < (unsigned long long)FPW ? (unsigned int)left : (unsigned int)FPW;
    }
    return r;
}

// ---- one lane of one wave ------------------------------------------------------
// HOPQ: 0 = any hop; 1 = the hop is N/4; 2 = the hop is N/2 (see "hop" above).
// PCM: the samples are 16-bit integers (StftArgs::x16), loaded and converted in
// place of the f32 loads; their factor 1/32768 rides in the window table the
// caller passes, which gives the same bits as f32 samples value/32768 would.
template <int LOG_R1, int LOG_R2, int MODE, int HOPQ = 0, bool PCM = false>
struct TileLane {
    using G = Geo<LOG_R1, LOG_R2>;
    static constexpr int R2 = G::R2;
    // Rows by which frame B's column lies below frame A's (0: not known at compile time).
    static constexpr int SHIFT = HOPQ == 1 ? G::R1 / 4 : HOPQ == 2 ? G::R1 / 2 : 0;
    static constexpr int OWN_B = SHIFT ? G::R1 - SHIFT : 0;   // first row of frame B that is loaded for B itself
    static constexpr bool SPLIT = G::SPLIT;
    static constexpr bool FEATURES = MODE == MODE_U8 || MODE == MODE_SURVEY || MODE == MODE_BANK;      // per-frame peak features and the global maximum
    const StftArgs &a;
    int lane;
    // samples of frames A / B of job jj at [jj*R1 + n1] (B: rows >= OWN_B only); SPLIT: xa = this lane's frame, row n1 at [n1]
    float xa[SPLIT ? G::R1 : R2], xb[SPLIT ? 1 : R2];
    float run_max;                     // running maximum of |2X|^2 over every frame this lane has finished (MODE_U8)

    SONO_HD TileLane(const StftArgs &args, int lane_) : a(args), lane(lane_), run_max(0.0f) {}

    // Pass-1 lane layout: column n2 = lane % R2 of the pairs h*JPL .. h*JPL + JPL-1, h = lane / R2.
    SONO_HD int col() const { return lane & (R2 - 1); }
    SONO_HD int hgroup() const { return lane >> LOG_R2; }

    // Sample (row N1, this lane's column) of frame B of job JJ.
    template <int JJ, int N1>
    SONO_HD float b_at() const
    {
        if constexpr (N1 < OWN_B) return xa[JJ * G::R1 + N1 + SHIFT];
        else return xb[JJ * G::R1 + N1];
    }

    // Issues the loads of `tile` (pass-1 lane layout).  Frames past the end of a
    // stream re-read an existing frame (the tile's first one; 16-bit input: the stream's
    // first one); their results are never stored.
    SONO_HD void load(const TileRef &tr)
    {
        const unsigned int j = tr.j;
        const int h = hgroup(), n2 = col();
        if constexpr (SPLIT) {
            // One frame per lane: frame h of the tile's two (a frame past the end of the stream reads frame 0).
            const unsigned int t = opaque((unsigned int)h);
            if constexpr (PCM) {
                const unsigned long long f = (unsigned long long)j * G::FPW + (t < tr.rem ? t : 0u);
                const unsigned long long c = a.x16_channels, row = (unsigned long long)R2 * c;
                const int16_t *q = a.x16 + (tr.xs_off + (unsigned int)n2 + f * a.hop) * c + a.x16_channel;
                auto rows_at = [&](auto stride_) {
                    constexpr int stride = decltype(stride_)::value;
                    static_for<0, G::R1>([&](auto n_) { xa[decltype(n_)::value] = (float)q[stride * decltype(n_)::value]; });
                };
                if (c == 1) rows_at(ic<R2>{});
                else if (c == 2) rows_at(ic<2 * R2>{});
                else static_for<0, G::R1>([&](auto n_) { xa[decltype(n_)::value] = (float)q[row * decltype(n_)::value]; });
            } else {
                const float *base = tr.xs + (unsigned long long)j * G::FPW * a.hop;
                const unsigned int off = t < tr.rem ? (unsigned int)n2 + t * a.hop : (unsigned int)n2;
                const float *p = base + off;
                static_for<0, G::R1>([&](auto n_) { xa[decltype(n_)::value] = p[R2 * decltype(n_)::value]; });
            }
        } else {
        static_for<0, G::JPL>([&](auto jj_) {
            constexpr int jj = decltype(jj_)::value;
            if constexpr (PCM) {
                // A frame past the end of the stream reads the tile's frame 0, as in the f32 path below: what a
                // frame's absent partner holds reaches the frame's own bins at the level of the last bit, so both
                // paths must substitute the same samples for their results to agree bit for bit.
                const unsigned int ta = (unsigned int)(2 * (h * G::JPL + jj)), tb = ta + 1u;
                const unsigned long long f0 = (unsigned long long)j * G::FPW;
                const unsigned long long fa = f0 + (ta < tr.rem ? ta : 0u), fb = f0 + (tb < tr.rem ? tb : 0u);
                const unsigned long long c = a.x16_channels, row = (unsigned long long)R2 * c;
                const int16_t *qa = a.x16 + (tr.xs_off + (unsigned int)n2 + fa * a.hop) * c + a.x16_channel;
                const int16_t *qb = a.x16 + (tr.xs_off + (unsigned int)n2 + fb * a.hop) * c + a.x16_channel;
                // One or two channels: the rows lie at distances known here, which the loads carry
                // as immediate offsets (a run-time stride costs an address pair per load and halves
                // the band mode's rate).
                auto rows_at = [&](auto stride_) {
                    constexpr int stride = decltype(stride_)::value;
                    static_for<0, G::R1>([&](auto n_) {
                        constexpr int n1 = decltype(n_)::value;
                        xa[jj * G::R1 + n1] = (float)qa[stride * n1];
                        if constexpr (n1 >= OWN_B) xb[jj * G::R1 + n1] = (float)qb[stride * n1];
                    });
                };
                if (c == 1) rows_at(ic<R2>{});
                else if (c == 2) rows_at(ic<2 * R2>{});
                else {
                    static_for<0, G::R1>([&](auto n_) {
                        constexpr int n1 = decltype(n_)::value;
                        xa[jj * G::R1 + n1] = (float)qa[row * n1];
                        if constexpr (n1 >= OWN_B) xb[jj * G::R1 + n1] = (float)qb[row * n1];
                    });
                }
                return;
            }
            // Addresses = the tile's first sample (the same in every lane: a scalar) + a 32-bit lane offset,
            // n2 + (frame inside the tile) * hop, which does not change from tile to tile: hardly any vector
            // instruction forms an address.  A frame past the end of the stream reads the tile's frame 0.
            const float *base = tr.xs + (unsigned long long)j * G::FPW * a.hop;
            const unsigned int rem = tr.rem;
            const unsigned int ta = opaque((unsigned int)(2 * (h * G::JPL + jj))), tb = ta + 1u;      // formed per tile: see opaque()
            const unsigned int off_a = ta < rem ? (unsigned int)n2 + ta * a.hop : (unsigned int)n2;
            const unsigned int off_b = tb < rem ? (unsigned int)n2 + tb * a.hop : (unsigned int)n2;
            const float *pa = base + off_a, *pb = base + off_b;
            static_for<0, G::R1>([&](auto n_) {
                constexpr int n1 = decltype(n_)::value;
#ifdef SONO_EXP_NOLOAD      // diagnostic build: what the kernel costs without its sample loads
                xa[jj * G::R1 + n1] = (float)(lane + n1) * 1e-3f + (float)(j + ta) * 1e-9f;
                if constexpr (n1 >= OWN_B) xb[jj * G::R1 + n1] = (float)(lane - n1) * 1e-3f + (float)(j + tb) * 1e-9f;
#else
                xa[jj * G::R1 + n1] = pa[R2 * n1];
                if constexpr (n1 >= OWN_B) xb[jj * G::R1 + n1] = pb[R2 * n1];
#endif
            });
        });
        }
    }

    // Pass 1, first half: window and real-input DFT-R1 of this lane's column of
    // frames A and B of each of its pairs.  Leaves Y_A / Y_B in half-complex order
    // (Re Y[k1] at [jj*R1 + k1], Im Y[k1] at [jj*R1 + R1 - k1]) and, in pa / pb, the
    // larger of |Y[0]| and |Y[R1/2]| -- the lane's share of the two rows that the
    // frames of a pair send through pass 2 together.
    SONO_HD void pass1_dft(const float *win, float (&ya)[R2], float (&yb)[R2], float (&pa)[G::JPL], float (&pb)[G::JPL]) const
    {
        const int n2 = col();
        static_for<0, G::JPL>([&](auto jj_) {
            constexpr int jj = decltype(jj_)::value;
            // Window and the DFT's first stage in one go: a butterfly's two outputs are
            // a +- b with a = x[n1]*w[n1] and b = x[n1 + R1/2]*w[n1 + R1/2] fused into the sum.
            float va[G::R1], vb[G::R1];
            static_for<0, G::HALF>([&](auto j_) {
                constexpr int i = 2 * decltype(j_)::value, na = bitrev(i, LOG_R1), nb = na + G::HALF;
                static_assert(bitrev(i + 1, LOG_R1) == nb, "stage-1 partner");
                constexpr bool flip = rdft_flip<2, i, G::R1>();      // the difference negated in the odd blocks
                const float wa = win[R2 * na + n2], wb = win[R2 * nb + n2];
                const float a0 = xa[jj * G::R1 + na] * wa, a1 = xa[jj * G::R1 + nb];
                va[i] = __builtin_fmaf(a1, wb, a0);
                va[i + 1] = flip ? __builtin_fmaf(a1, wb, -a0) : __builtin_fmaf(-a1, wb, a0);
                const float b0 = b_at<jj, na>() * wa, b1 = b_at<jj, nb>();
                vb[i] = __builtin_fmaf(b1, wb, b0);
                vb[i + 1] = flip ? __builtin_fmaf(b1, wb, -b0) : __builtin_fmaf(-b1, wb, b0);
            });
            rdft_bitrev_in<G::R1, 4>(va);
            rdft_bitrev_in<G::R1, 4>(vb);
            pin_registers(va);
            pin_registers(vb);
            static_for<0, G::R1>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                ya[jj * G::R1 + i] = va[i];
                yb[jj * G::R1 + i] = vb[i];
            });
            pa[jj] = fmaxf(fabsf(va[0]), fabsf(va[G::HALF]));
            pb[jj] = fmaxf(fabsf(vb[0]), fabsf(vb[G::HALF]));
        });
    }

    // Pass 1, second half: the inter-pass twiddles.  pa / pb = the maxima of
    // pass1_dft's values over all R2 columns.  Leaves the R2 row values of the lane
    // (rows rho of its JPL pairs, column n2) in rr / ri at [jj*R1 + rho]: rows
    // 1..R1/2-1 = frame A's Y[k1] * 2 W_N^(n2*k1), rows R1/2+1..R1-1 frame B's; row 0 =
    // Y_A[0] + i s Y_B[0] and row R1/2 = (Y_A[R1/2] + i s Y_B[R1/2]) * W_N^(n2*R1/2), where the
    // power of two s brings frame B's share of these two rows to the binade of
    // frame A's (see "scale" above).  win = the N window values; tw = the twiddle
    // table described at Geo::TW_FOLD.  The lane of column 0 records the pair's side
    // information in `meta`.
    SONO_HD void pass1_rows(const float2 *tw, const float (&ya)[R2], const float (&yb)[R2], const float (&pa)[G::JPL],
                            const float (&pb)[G::JPL], float (&rr)[R2], float (&ri)[R2], float *meta) const
    {
        const int h = hgroup(), n2 = col();
        static_for<0, G::JPL>([&](auto jj_) {
            constexpr int jj = decltype(jj_)::value, o = jj * G::R1;
            int d = (pa[jj] > 0.0f && pb[jj] > 0.0f) ? float_exponent(pa[jj]) - float_exponent(pb[jj]) : 0;
            d = d > 40 ? 40 : d < -40 ? -40 : d;
            const float sb = pow2(d);
            if (n2 == 0) {
                // What the epilogue multiplies |2X|^2 of the shared rows' bins by: 0 for a
                // frame whose share of them is exactly zero (its bins are then exactly zero,
                // and what the separation leaves of the other frame must not show), else 1
                // for a frame A and s^-2 for a frame B.
                const int q = h * G::JPL + jj;
                const float gb = pow2(-d);
                meta[kMetaFix + 2 * q] = pa[jj] > 0.0f ? 1.0f : 0.0f;
                meta[kMetaFix + 2 * q + 1] = pb[jj] > 0.0f ? gb * gb : 0.0f;
            }
            rr[o] = ya[o];
            ri[o] = yb[o] * sb;
            // rows k1 (frame A) and R1/2 + k1 (frame B), twiddled by t = 2 W_N^(n2*k1)
            auto rows = [&](auto k_, const float2 t) {
                constexpr int k1 = decltype(k_)::value;
                const float ar = ya[o + k1], ai = ya[o + G::R1 - k1], br = yb[o + k1], bi = yb[o + G::R1 - k1];
                rr[o + k1] = ar * t.x - ai * t.y;
                ri[o + k1] = ar * t.y + ai * t.x;
                rr[o + G::HALF + k1] = br * t.x - bi * t.y;
                ri[o + G::HALF + k1] = br * t.y + bi * t.x;
            };
            float2 tn;                                                                       // W_N^(n2*R1/2)
            if constexpr (G::TW_FOLD) {
                constexpr int Q = G::HALF / 2;
                tn = tw[Q * R2 + n2];
                static_for<1, Q>([&](auto k_) {
                    constexpr int k1 = decltype(k_)::value;
                    const float2 t = tw[(k1 - 1) * R2 + n2];
                    rows(k_, t);
                    float2 u;
                    u.x = tn.x * t.x + tn.y * t.y;
                    u.y = tn.y * t.x - tn.x * t.y;
                    rows(ic<G::HALF - k1>{}, u);
                });
                rows(ic<Q>{}, tw[(Q - 1) * R2 + n2]);
            } else {
                tn = tw[(G::HALF - 1) * R2 + n2];
                static_for<1, G::HALF>([&](auto k_) { rows(k_, tw[(decltype(k_)::value - 1) * R2 + n2]); });
            }
            const float cr = ya[o + G::HALF], ci = yb[o + G::HALF] * sb;
            rr[o + G::HALF] = cr * tn.x - ci * tn.y;
            ri[o + G::HALF] = cr * tn.y + ci * tn.x;
        });
    }

    // SPLIT, pass 1, first half: window and real-input DFT-R1 of this lane's column of its frame.
    // Leaves Y in half-complex order in y and, in p, the larger of |Y[0]| and |Y[R1/2]|.
    SONO_HD void pass1_dft_split(const float *win, float (&y)[G::R1], float &p) const
    {
        const int n2 = col();
        static_for<0, G::HALF>([&](auto j_) {
            constexpr int i = 2 * decltype(j_)::value, na = bitrev(i, LOG_R1), nb = na + G::HALF;
            constexpr bool flip = rdft_flip<2, i, G::R1>();
            const float wa = win[R2 * na + n2], wb = win[R2 * nb + n2];
            const float a0 = xa[na] * wa, a1 = xa[nb];
            y[i] = __builtin_fmaf(a1, wb, a0);
            y[i + 1] = flip ? __builtin_fmaf(a1, wb, -a0) : __builtin_fmaf(-a1, wb, a0);
        });
        rdft_bitrev_in<G::R1, 4>(y);
        pin_registers(y);
        p = fmaxf(fabsf(y[0]), fabsf(y[G::HALF]));
    }

    // SPLIT, pass 1, second half.  a0 / b0 = Y[0] of frame A's / frame B's lane of this column, ah / bh
    // = their Y[R1/2], pa / pb = the maxima of pass1_dft_split's p over the columns of frame A / B (all
    // six the same in both lanes of a column).  Leaves this lane's R2 row values in rr / ri: value k1 =
    // 1..R1/2-1 is its frame's Y[k1] * 2 W_N^(n2*k1) (row k1 of frame A, row R1/2 + k1 of frame B);
    // value 0 is the shared row 0, Y_A[0] + i s Y_B[0], in frame A's lanes and the shared row R1/2,
    // (Y_A[R1/2] + i s Y_B[R1/2]) * W_N^(n2*R1/2), in frame B's.
    SONO_HD void pass1_rows_split(const float2 *tw, const float (&y)[G::R1], float a0, float b0, float ah, float bh, float pa,
                                  float pb, float (&rr)[R2], float (&ri)[R2], float *meta) const
    {
        static_assert(G::HALF == R2, "one value per row and lane");
        const int h = hgroup(), n2 = col();
        int d = (pa > 0.0f && pb > 0.0f) ? float_exponent(pa) - float_exponent(pb) : 0;
        d = d > 40 ? 40 : d < -40 ? -40 : d;
        const float sb = pow2(d);
        if (lane == 0) {
            const float gb = pow2(-d);
            meta[kMetaFix] = pa > 0.0f ? 1.0f : 0.0f;
            meta[kMetaFix + 1] = pb > 0.0f ? gb * gb : 0.0f;
        }
        const float2 tn = tw[(G::HALF - 1) * R2 + n2];                                       // W_N^(n2*R1/2)
        const float cr = h ? ah : a0, ci = (h ? bh : b0) * sb;
        const float tx = h ? tn.x : 1.0f, ty = h ? tn.y : 0.0f;
        rr[0] = cr * tx - ci * ty;
        ri[0] = cr * ty + ci * tx;
        static_for<1, G::HALF>([&](auto k_) {
            constexpr int k1 = decltype(k_)::value;
            const float2 t = tw[(k1 - 1) * R2 + n2];
            const float ar = y[k1], ai = y[G::R1 - k1];
            rr[k1] = ar * t.x - ai * t.y;
            ri[k1] = ar * t.y + ai * t.x;
        });
    }

    // One exchange round: every pass-1 lane stores one component of its R2 row
    // values at column n2 (value i = jj*R1 + rho belongs to row R2*h + i) ...
    SONO_HD void exchange_store(const float (&v)[R2], float *img) const
    {
        const int h = hgroup(), n2 = col();
        float *base = img + R2 * R2 * h;
        static_for<0, R2>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            base[R2 * i + (n2 ^ image_swizzle<LOG_R2>(i))] = v[i];    // = img[image_index(R2*h + i, n2)]
        });
    }
    // ... and every pass-2 lane reads its row, placing column n2 where the DFT wants it.
    SONO_HD void exchange_load(const float *img, float (&v)[R2]) const
    {
        const float *row = img + R2 * lane;
        const int f = image_swizzle<LOG_R2>(lane);
        static_for<0, R2 / 4>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const float4 q = *(const float4 *)(row + ((4 * j) ^ f));
            v[bitrev(4 * j, LOG_R2)] = q.x; v[bitrev(4 * j + 1, LOG_R2)] = q.y;
            v[bitrev(4 * j + 2, LOG_R2)] = q.z; v[bitrev(4 * j + 3, LOG_R2)] = q.w;
        });
    }

    // Pass 2 (after both exchange rounds): DFT-R2 over n2, then |2X|^2 of the
    // lane's R2 bins scattered into the bin-major power image P[frame][bin].
    //
    // A frame row (rho = k1 or R1/2 + k1) holds 2X[k1 + R1*t] in register t and
    // conj 2X[(R1-k1) + R1*t] in register R2-1-t.  A shared row holds C = 2X_A' + i 2X_B'
    // of both frames of the pair, where X' is conjugate-symmetric about register 0
    // (rho = 0: bins R1*t) or about register -1/2 (rho = R1/2: bins R1/2 + R1*t):
    //   |2X_A|^2 = |C[t] + conj C[p]|^2,  |2X_B|^2 = |C[t] - conj C[p]|^2,  p = (R2-t)%R2 or R2-1-t.
    // One instruction stream serves all three kinds: with b = conj C[p] and a
    // per-lane s (0 for frame rows, 1 for shared rows) the two stores are
    // |C[t] + s b|^2 and |s C[t] - b|^2.  (Frame B's values carry its scale.)
    SONO_HD void pass2_power(float (&re)[R2], float (&im)[R2], float *P) const
    {
        dft_bitrev_in<R2>(re, im);
        const int q = lane >> LOG_R1, rho = lane & (G::R1 - 1), k1 = rho & (G::HALF - 1);
        const bool shared = k1 == 0, row0 = rho == 0;
        const float s = shared ? 1.0f : 0.0f;
        const int frame = 2 * q + (rho >> (LOG_R1 - 1));
        float *row1 = P + (shared ? 2 * q * G::PROW + rho : frame * G::PROW + k1);
        float *row2 = P + (shared ? (2 * q + 1) * G::PROW + rho : frame * G::PROW + G::R1 - k1);
        // The partner of register t is register R2-1-t in every row but row 0, where it is (R2-t) mod R2:
        // the lanes of row 0 (two of 64) move their upper registers one place down, C[0] into the last,
        // after which one rule serves all.  (Moves under the execution mask: half the cost of the selects
        // that would pick the partner lane by lane.)
        if (row0) {
            static_for<R2 / 2, R2 - 1>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                move_in_place(re[j], re[j + 1]);
                move_in_place(im[j], im[j + 1]);
            });
            move_in_place(re[R2 - 1], re[0]);
            move_in_place(im[R2 - 1], im[0]);
        }
        static_for<0, R2 / 2>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            const float br = re[R2 - 1 - t];
            const float bi = im[R2 - 1 - t];
            const float er = re[t] + s * br, ei = im[t] - s * bi;
            const float dr = s * re[t] - br, di = s * im[t] + bi;
            row1[G::R1 * t] = er * er + ei * ei;
            row2[G::R1 * t] = dr * dr + di * di;
        });
    }

    // Epilogue: lane (frame, l) owns bins 4l + 2*R1*m + r (m < R2/4, r < 4) of its frame.
    // FLOOR = false leaves out the clamp to q_floor (for q_floor = 0).
    // The bins that came through the pair's shared rows are the multiples of R1/2:
    // the first of a lane's four wherever 4l is one.
    SONO_HD int bin_lane() const       // which quarter-row of four bins per 2*R1 this lane takes (see Geo::ROT)
    {
        const int frame = lane >> (LOG_R1 - 1), l = lane & (G::HALF - 1);
        return (l - G::ROT * frame) & (G::HALF - 1);
    }
    SONO_HD float shared_fix(const float *meta) const
    {
        const int frame = lane >> (LOG_R1 - 1), l = bin_lane();
        return ((4 * l) & (G::HALF - 1)) == 0 ? meta[kMetaFix + frame] : 1.0f;
    }
    // best_p / best_k: the largest |2X|^2 among this lane's bins and its bin (the lowest
    // one on a tie; 0 where everything is zero).  The caller reduces them over the frame's
    // R1/2 lanes: largest power, then the lowest bin among the lanes that hold it (MODE_U8).
    template <bool FLOOR>
    SONO_HD void epilogue(const TileRef &tr, const float *P, const float *meta, float &best_p, unsigned int &best_k)
    {
        const int frame = lane >> (LOG_R1 - 1), l = bin_lane();
        const unsigned long long f0 = (unsigned long long)tr.j * G::FPW;          // the tile's first frame: the same in every lane
        const bool valid = (unsigned int)frame < tr.rem;
        // first output element of the tile's first row (a scalar) + this lane's place in the tile (a 32-bit offset)
        const unsigned long long tile_elem = (tr.row0 + f0) * (unsigned long long)G::BINS;
        const unsigned int lane_elem = opaque((unsigned int)frame) * (unsigned int)G::BINS + (unsigned int)(4 * l);
        const float fix = shared_fix(meta);
        const float *row = P + frame * G::PROW + 4 * l;
        // Peak search: per group of four bins only the group's maximum is compared
        // with the running one (strict >, so the lowest group wins a tie); which of
        // the four it was is settled once, after the loop.
        best_p = 0.0f;
        best_k = 0u;
        int best_m = 0;
        // The reads of a batch are issued together: the LDS serves every wave of the
        // workgroup from one queue, and a read that is waited for on its own sits
        // behind whatever bursts the other waves have queued.
        constexpr int NB = R2 / 4, B = LOG_R2 == 6 ? (LOG_R1 == 6 ? SONO_EPI_BATCH_4096 : SONO_EPI_BATCH_WIDE) : LOG_R1 == 5 ? SONO_EPI_BATCH : SONO_EPI_BATCH_512;
        static_assert(NB % B == 0, "the epilogue batch must divide the reads per lane");
        if (!valid) return;          // lanes of frames past the end of the stream have nothing to store
#ifdef SONO_EXP_NOSTORE
        unsigned int nostore_sum = 0;
#endif
        const float qa = in_vgpr(a.q_a), qc = in_vgpr(a.q_c);
        SONO_EPILOGUE_UNROLL
        for (int m0 = 0; m0 < NB; m0 += B) {
            float4 p4[B];
            static_for<0, B>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                p4[i] = *(const float4 *)(row + 2 * G::R1 * (m0 + i));
            });
            static_for<0, B>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                float p[4] = {p4[i].x * fix, p4[i].y, p4[i].z, p4[i].w};
                if (MODE == MODE_MAG) {
                    // a caller that reads a band of the rows only (band_hi > 0) gets the groups of four bins that
                    // touch it; the rest of the matrix keeps what it held
                    const unsigned int k0 = (unsigned int)(4 * l + 2 * G::R1 * (m0 + i));
                    if (a.band_hi == 0u || (k0 + 3u >= a.band_lo && k0 <= a.band_hi)) {
                        float4 o;
                        o.x = fast_sqrt(p[0]) * a.inv_n; o.y = fast_sqrt(p[1]) * a.inv_n;
                        o.z = fast_sqrt(p[2]) * a.inv_n; o.w = fast_sqrt(p[3]) * a.inv_n;
                        *(float4 *)((a.out_mag + tile_elem) + ((size_t)lane_elem + 2 * G::R1 * (m0 + i))) = o;
                    }
                } else {
                    if (MODE == MODE_U8 && a.out_u8) {
                        unsigned int packed = quantise<FLOOR, 0>(p[0], qa, qc, a.q_floor);
                        packed = quantise<FLOOR, 1>(p[1], qa, qc, a.q_floor, packed);
                        packed = quantise<FLOOR, 2>(p[2], qa, qc, a.q_floor, packed);
                        packed = quantise<FLOOR, 3>(p[3], qa, qc, a.q_floor, packed);
#ifdef SONO_EXP_NOSTORE     // diagnostic build: what the kernel costs without its row stores (the codes stay alive in a sum that is never written)
                        if ((nostore_sum += packed) == 0x12345679u && best_m == 77)
#endif
                        *(unsigned int *)((a.out_u8 + tile_elem) + ((size_t)lane_elem + 2 * G::R1 * (m0 + i))) = packed;
                    }
                    const float top = max_of_4(p[0], p[1], p[2], p[3]);
                    if (top > best_p) { best_p = top; best_m = m0 + i; }        // false for a NaN
                }
            });
        }
        if (FEATURES) {
            const float4 w4 = *(const float4 *)(row + 2 * G::R1 * best_m);
            const float w[4] = {w4.x * fix, w4.y, w4.z, w4.w};
            const int r = w[0] == best_p ? 0 : w[1] == best_p ? 1 : w[2] == best_p ? 2 : 3;
            best_k = best_p > 0.0f ? (unsigned int)(4 * l + 2 * G::R1 * best_m + r) : 0u;
            run_max = best_p > run_max ? best_p : run_max;
        }
    }

    // MODE_BAND epilogue, step 1: this lane's share of its frame's band energy --
    // the sum over its bins k in [band_lo, band_hi] of (double)mag * mag, where mag is
    // the f32 magnitude MODE_MAG would store, in increasing k.  Step 2 (the caller's):
    // add up the R1/2 lanes of the frame, pairing lanes l and l ^ o for o = R1/4 ... 1.
    SONO_HD double band_sum(const float *P, const float *meta) const
    {
        const int frame = lane >> (LOG_R1 - 1), l = bin_lane();
        const float fix = shared_fix(meta);
        const float *row = P + frame * G::PROW + 4 * l;
        constexpr int NB = R2 / 4, B = LOG_R2 == 6 ? (LOG_R1 == 6 ? SONO_EPI_BATCH_4096 : SONO_EPI_BATCH_WIDE) : LOG_R1 == 5 ? SONO_EPI_BATCH : SONO_EPI_BATCH_512;
        double acc = 0.0;
        SONO_EPILOGUE_UNROLL
        for (int m0 = 0; m0 < NB; m0 += B) {
            float4 p4[B];
            static_for<0, B>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                p4[i] = *(const float4 *)(row + 2 * G::R1 * (m0 + i));
            });
            static_for<0, B>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                const unsigned int k0 = (unsigned int)(4 * l + 2 * G::R1 * (m0 + i));
                const float p[4] = {p4[i].x * fix, p4[i].y, p4[i].z, p4[i].w};
                static_for<0, 4>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    const double m = (double)(fast_sqrt(p[r]) * a.inv_n);
                    const double t = m * m;
                    acc += (k0 + r >= a.band_lo && k0 + r <= a.band_hi) ? t : 0.0;
                });
            });
        }
        return acc;
    }
    // Step 3: the frame's first lane writes the flag (and the sum).
    SONO_HD void band_emit(const TileRef &tr, double total) const
    {
        const int frame = lane >> (LOG_R1 - 1), l = lane & (G::HALF - 1);
        if (l != 0) return;
        if ((unsigned int)frame >= tr.rem) return;
        const unsigned long long g = tr.row0 + (unsigned long long)tr.j * G::FPW + (unsigned int)frame;
        a.band_active[g] = total > a.band_target ? 1 : 0;
        if (a.band_energy) a.band_energy[g] = total;
    }

    // MODE_BANK, after the feature epilogue: the frame's row of filter outputs (apply_filter_bank, lib.rs:310-334) from
    // the power image, in two steps with a wave fence between them.
    // Step 1: every lane turns its own bins of the image (the epilogue's assignment) into the f32 magnitudes that
    // MODE_MAG stores, in place: a bin feeds two filters on average, so its square root is taken once, not per tap.
    SONO_HD void bank_magnitudes(float *P, const float *meta) const
    {
        const int frame = lane >> (LOG_R1 - 1), l = bin_lane();
        const float fix = shared_fix(meta);
        float *row = P + frame * G::PROW + 4 * l;
        constexpr int NB = R2 / 4;
        float4 p4[NB];
        static_for<0, NB>([&](auto i_) { p4[decltype(i_)::value] = *(const float4 *)(row + 2 * G::R1 * decltype(i_)::value); });
        static_for<0, NB>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            float4 o;
            o.x = fast_sqrt(p4[i].x * fix) * a.inv_n; o.y = fast_sqrt(p4[i].y) * a.inv_n;
            o.z = fast_sqrt(p4[i].z) * a.inv_n; o.w = fast_sqrt(p4[i].w) * a.inv_n;
            *(float4 *)(row + 2 * G::R1 * i) = o;
        });
    }
    // Step 2: lane (frame, l) takes the filters 4l + 2*R1*m + r (r < 4) and writes their four codes as one dword.  A
    // filter's output is the sum over its taps, bins ascending, of magnitude * weight, product and sum rounded
    // separately as in the reference.  taps: the workgroup's LDS copy of the tap list, {weight, bin} of tap k of
    // filter i at k * filters + i, every filter padded to bank_max_taps entries with weight 0 (the reference itself
    // walks every bin of every filter, zero weights included: a zero product adds an exact zero).
    SONO_HD void bank_rows(const TileRef &tr, const float *P, const BankTap *taps) const
    {
        const int frame = lane >> (LOG_R1 - 1), l = lane & (G::HALF - 1);
        if ((unsigned int)frame >= tr.rem) return;
        const float *row = P + frame * G::PROW;
        const unsigned int filters = a.bank_filters;
        const float qa = in_vgpr(a.bank_q_a), qc = in_vgpr(a.bank_q_c);
        uint8_t *out = a.out_u8 + (tr.row0 + (unsigned long long)tr.j * G::FPW + (unsigned int)frame) * (unsigned long long)filters;
        auto emit = [&](unsigned int i0, const float (&sum)[4]) {
            unsigned int packed = bank_code<0>(sum[0], qa, qc);
            packed = bank_code<1>(sum[1], qa, qc, packed);
            packed = bank_code<2>(sum[2], qa, qc, packed);
            packed = bank_code<3>(sum[3], qa, qc, packed);
            *(unsigned int *)(out + i0) = packed;
        };
        if (a.bank_max_taps == 2) {
            // The application's banks (two taps per filter, spectrogram.esm.js:1197-1232): the eight taps of a lane's
            // four filters are four 16-byte reads, issued together, then the eight magnitudes, then the arithmetic --
            // the reads of a lane's next four filters go out before this group's results are waited for.
            SONO_PRAGMA(clang loop unroll_count(2))
            for (unsigned int i0 = (unsigned int)(4 * l); i0 < filters; i0 += (unsigned int)(2 * G::R1)) {
                const uint4 *q0 = (const uint4 *)(taps + i0), *q1 = (const uint4 *)(taps + filters + i0);
                const uint4 t0a = q0[0], t0b = q0[1], t1a = q1[0], t1b = q1[1];     // {w, bin, w, bin} of filters i0, i0+1 | i0+2, i0+3
                const float m0[4] = {row[t0a.y], row[t0a.w], row[t0b.y], row[t0b.w]};
                const float m1[4] = {row[t1a.y], row[t1a.w], row[t1b.y], row[t1b.w]};
                const float w0[4] = {as_float(t0a.x), as_float(t0a.z), as_float(t0b.x), as_float(t0b.z)};
                const float w1[4] = {as_float(t1a.x), as_float(t1a.z), as_float(t1b.x), as_float(t1b.z)};
                float sum[4];
                static_for<0, 4>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    sum[r] = fadd_rn(fadd_rn(0.0f, fmul_rn(m0[r], w0[r])), fmul_rn(m1[r], w1[r]));
                });
                emit(i0, sum);
            }
            return;
        }
        for (unsigned int i0 = (unsigned int)(4 * l); i0 < filters; i0 += (unsigned int)(2 * G::R1)) {
            float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (unsigned int k = 0, t = i0; k < a.bank_max_taps; ++k, t += filters) {
                static_for<0, 4>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    const BankTap tap = taps[t + r];
                    sum[r] = fadd_rn(sum[r], fmul_rn(row[tap.bin], tap.w));
                });
            }
            emit(i0, sum);
        }
    }

    // MODE_SURVEY, after the feature epilogue: the band energy of the tile's EVEN frames (the fast scan's frames: its
    // hop is twice the features' hop, and both start at sample 0).  All 64 lanes share the FPW/2 even frames, R1
    // lanes per frame: lane (e, h, l) takes the bins 4l + 2*R1*(2m + h) + r (m < R2/8, r < 4) of frame 2e.  Step 2 (the
    // caller's): add up the R1 lanes of the frame, pairing lanes by xor R1/2 ... 1.  Each term is MODE_BAND's term
    // (double of the f32 magnitude, squared); only the order of the additions differs from that mode's.
    SONO_HD double survey_band_sum(const float *P, const float *meta) const
    {
        static_assert(R2 % 8 == 0, "half of a frame's reads per lane");
        const int frame = 2 * (lane >> LOG_R1), h = (lane >> (LOG_R1 - 1)) & 1, l0 = lane & (G::HALF - 1);
        const int l = (l0 - G::ROT * frame) & (G::HALF - 1);
        const float fix = ((4 * l) & (G::HALF - 1)) == 0 ? meta[kMetaFix + frame] : 1.0f;
        const float *row = P + frame * G::PROW + 4 * l;
        constexpr int NB = R2 / 8;
        // Read i of every lane covers the bins [2*R1*2i, 2*R1*(2i + 2)): the reads that lie outside the band altogether
        // (a third of them for the detector's 10 ... 128 kHz at 384 kHz) are left out, the same ones in every lane.
        float4 p4[NB];
        static_for<0, NB>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            if ((unsigned int)(4 * G::R1 * i) <= a.band_hi && (unsigned int)(4 * G::R1 * (i + 1)) > a.band_lo)
                p4[i] = *(const float4 *)(row + 2 * G::R1 * (2 * i + h));
        });
        double acc = 0.0;
        static_for<0, NB>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            if ((unsigned int)(4 * G::R1 * i) <= a.band_hi && (unsigned int)(4 * G::R1 * (i + 1)) > a.band_lo) {
                const unsigned int k0 = (unsigned int)(4 * l + 2 * G::R1 * (2 * i + h));
                const float p[4] = {p4[i].x * fix, p4[i].y, p4[i].z, p4[i].w};
                static_for<0, 4>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    const double m = (double)(fast_sqrt(p[r]) * a.inv_n);
                    const double t = m * m;
                    acc += (k0 + r >= a.band_lo && k0 + r <= a.band_hi) ? t : 0.0;
                });
            }
        });
        return acc;
    }
    // Step 3: the frame's first lane writes flag (and sum) number (frame index / 2) of its stream.
    SONO_HD void survey_band_emit(const TileRef &tr, double total) const
    {
        const int frame = 2 * (lane >> LOG_R1);
        if ((lane & (G::R1 - 1)) != 0 || (unsigned int)frame >= tr.rem) return;
        // tr.row0 = the stream's index x frames_per_stream; the scan has ceil(frames / 2) frames per stream
        const unsigned long long stream = a.frames_per_stream ? tr.row0 / a.frames_per_stream : 0ull;
        const unsigned long long g = stream * ((a.frames_per_stream + 1) / 2) + ((unsigned long long)tr.j * G::FPW + (unsigned int)frame) / 2;
        a.band_active[g] = total > a.band_target ? 1 : 0;
        if (a.band_energy) a.band_energy[g] = total;
    }

    // pmax / bin = the frame's largest |2X|^2 and its lowest bin (the epilogue's values
    // reduced over the frame's lanes): the frame's first lane emits the features.
    SONO_HD void features(const TileRef &tr, float pmax, unsigned int bin) const
    {
        const int frame = lane >> (LOG_R1 - 1), l = lane & (G::HALF - 1);
        if (!FEATURES || l != 0) return;
        const unsigned long long f0 = (unsigned long long)tr.j * G::FPW;
        if ((unsigned int)frame >= tr.rem) return;
        const unsigned long long g0 = tr.row0 + f0;           // a scalar; the lane adds its frame
        // The lane's part is formed here, per tile (one shift): kept across tiles as two 64-bit
        // pointers it costs four registers that the tile loop does not have.
        const unsigned int fr = opaque((unsigned int)frame);
        if (a.peak_bin) (a.peak_bin + g0)[(size_t)fr] = (uint16_t)bin;
        if (a.peak_mag) (a.peak_mag + g0)[(size_t)fr] = fast_sqrt(pmax) * a.inv_n;
    }
};

}  // namespace tile
}  // namespace sono
