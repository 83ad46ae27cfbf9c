// valubench.hip -- what the vector ALU of gfx950 charges for the instruction forms the tiled STFT
// kernel is made of, at the kernel's own occupancy (16 waves per CU, 4 per SIMD, every CU busy):
//   * hand-placed register operands: does the number of VGPR sources of an instruction, or the
//     registers' numbers (banks), change its issue cost?
//   * the kernel's own transforms (sono_tile_core.h) in a loop on registers, no LDS, no memory:
//     the arithmetic's own time, against which the whole kernel's time can be held.
// Prints cycles per wave-instruction per SIMD from the in-kernel clock (s_memtime / s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/valubench tools/valubench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../spectrogram_amd/csrc/sono_tile_core.h"

// The complex DFT on (re, im) register pairs with packed f32 instructions: measured here and in
// a build of the tiled kernel, and not used by the library.  A packed instruction does two
// lanes' worth of work per issue slot, but the SIMD charges it as much as the two plain
// instructions it stands for (4.8 against 2 x 2.4 cycles), the dependent ones need wait states
// in between, and register pairs must be aligned: the kernel with its pass 1 packed ran 3 %
// slower (2.85 against 2.77 ms per hour of audio at fft 1024, same box, bit-identical rows).
namespace sono {
namespace tile {
// Two f32 values in an aligned register pair, worked on by the packed instructions
// (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32).  One packed instruction issues in 3.2
// cycles of a SIMD where the two plain ones it stands for take 3.8 - 3.9, and, what
// counts for more, it takes ONE of the issue slots of its wave: a wave issues a vector
// instruction at most every 5 cycles, so a SIMD needs three waves in arithmetic to
// keep its ALU busy with plain instructions and hardly more than one with packed ones.
// Pass 1 keeps (frame A, frame B) of a pair of frames in a register pair, pass 2 the
// (real, imaginary) parts of a complex value; op_sel / neg modifiers of the packed
// instructions do the half swaps and sign flips that complex arithmetic needs.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float pk2 __attribute__((ext_vector_type(2)));
#else
struct pk2 { float x, y; };
#endif
SONO_HD pk2 pk_make(float x, float y) { pk2 r; r.x = x; r.y = y; return r; }
#if defined(__HIP_DEVICE_COMPILE__)
SONO_HD pk2 pk_add(pk2 a, pk2 b) { return a + b; }
SONO_HD pk2 pk_sub(pk2 a, pk2 b) { return a - b; }
SONO_HD pk2 pk_mul(pk2 a, pk2 b) { return a * b; }
SONO_HD pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }
// a + (-i) b = (a.x + b.y, a.y - b.x)  and  a - (-i) b = (a.x - b.y, a.y + b.x), complex values as (re, im)
SONO_HD pk2 pk_add_negi(pk2 a, pk2 b)
{
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
SONO_HD pk2 pk_sub_negi(pk2 a, pk2 b)
{
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + w b for complex a, b as (re, im) and the constant w = (WR, WI) given as the bit patterns of
// its parts: a + b.re (wr, wr) + (b.im, b.re) (-wi, wi).  The constant rides in one scalar register pair.
SONO_HD pk2 pk_add_wmul(pk2 a, pk2 b, unsigned long long w_bits)
{
    pk2 t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(b), "s"(w_bits), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(b), "s"(w_bits), "v"(t));
    return r;
}
// 2 a - x
SONO_HD pk2 pk_twice_minus(pk2 a, pk2 x)
{
    pk2 r;
    asm("v_pk_fma_f32 %0, 2.0, %1, %2 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(x));
    return r;
}
#else
SONO_HD pk2 pk_add(pk2 a, pk2 b) { return pk_make(a.x + b.x, a.y + b.y); }
SONO_HD pk2 pk_sub(pk2 a, pk2 b) { return pk_make(a.x - b.x, a.y - b.y); }
SONO_HD pk2 pk_mul(pk2 a, pk2 b) { return pk_make(a.x * b.x, a.y * b.y); }
SONO_HD pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return pk_make(std::fmaf(a.x, b.x, c.x), std::fmaf(a.y, b.y, c.y)); }
SONO_HD pk2 pk_add_negi(pk2 a, pk2 b) { return pk_make(a.x + b.y, a.y - b.x); }
SONO_HD pk2 pk_sub_negi(pk2 a, pk2 b) { return pk_make(a.x - b.y, a.y + b.x); }
SONO_HD pk2 pk_add_wmul(pk2 a, pk2 b, unsigned long long w_bits)
{
    const unsigned int lo = (unsigned int)w_bits, hi = (unsigned int)(w_bits >> 32);
    float wr, wi;
    std::memcpy(&wr, &lo, 4);
    std::memcpy(&wi, &hi, 4);
    const pk2 t = pk_make(std::fmaf(b.x, wr, a.x), std::fmaf(b.y, wr, a.y));
    return pk_make(std::fmaf(b.y, -wi, t.x), std::fmaf(b.x, wi, t.y));
}
SONO_HD pk2 pk_twice_minus(pk2 a, pk2 x) { return pk_make(std::fmaf(2.0f, a.x, -x.x), std::fmaf(2.0f, a.y, -x.y)); }
#endif

// Bit pattern of a float at compile time (positive normal numbers, zero and their negatives: all
// the twiddle tables hold), for constants that ride in scalar register pairs.
constexpr unsigned int float_bits_ce(float v)
{
    if (v == 0.0f) return 0u;
    const unsigned int sign = v < 0.0f ? 0x80000000u : 0u;
    float m = v < 0.0f ? -v : v;
    int e = 0;
    while (m >= 2.0f) { m *= 0.5f; ++e; }
    while (m < 1.0f) { m *= 2.0f; --e; }
    const unsigned int frac = (unsigned int)((m - 1.0f) * 8388608.0f);       // exact: m - 1 has 23 fraction bits
    return sign | ((unsigned int)(e + 127) << 23) | frac;
}
constexpr unsigned long long w64_bits(int k) { return ((unsigned long long)float_bits_ce(w64_im(k)) << 32) | float_bits_ce(w64_re(k)); }

// Decimation-in-time butterfly on complex values held as (re, im) pairs:
// (a, b) -> (a + w b, a - w b), w = W_64^K64.  Two packed instructions where w is 1 or -i, three otherwise.
template <int K64>
SONO_HD void bfly_pk(pk2 &a, pk2 &b)
{
    if constexpr (K64 == 0) {
        const pk2 x = pk_add(a, b);
        b = pk_sub(a, b); a = x;
    } else if constexpr (K64 == 16) {
        const pk2 x = pk_add_negi(a, b);
        b = pk_sub_negi(a, b); a = x;
    } else {
        constexpr unsigned long long w = w64_bits(K64);      // evaluated here, at compile time
        const pk2 x = pk_add_wmul(a, b, w);
        b = pk_twice_minus(a, x); a = x;
    }
}

// In-place N-point DFT (N = 2..64) of complex values held as (re, im) pairs, input in
// bit-reversed order, output in natural order.
template <int N>
SONO_HD void cdft_bitrev_in(pk2 (&z)[N])
{
    constexpr int L = ilog2(N);
    static_for<1, L + 1>([&](auto s_) {
        constexpr int len = 1 << decltype(s_)::value, half = len / 2;
        static_for<0, N / len>([&](auto b_) {
            constexpr int base = decltype(b_)::value * len;
            static_for<0, half>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                bfly_pk<j * (64 / len)>(z[base + j], z[base + j + half]);
            });
        });
    });
}

}  // namespace tile
}  // namespace sono

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define R4(s) s s s s
#define R8(s) R4(s) R4(s)

struct Stamp { unsigned long long cyc, real, start, where; };

// XCC id in bits 32.., HW_ID (wave, SIMD, CU, SH, SE ...) in the low word
__device__ __forceinline__ unsigned long long where_am_i()
{
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((unsigned long long)xcc << 32) | hw;
}

// ---- hand-placed operands -------------------------------------------------------------------
// 32 instructions per loop round, destinations v40..v71 (never read inside the loop), sources as named.
#define LOOP_BEGIN "s_mov_b32 s40, %0\n 1:\n"
#define LOOP_END "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"
#define CLOBBERS "s40", "scc", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
    "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"
#define INIT "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 2.0\n v_mov_b32 v11, 4.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 0.5\n" \
             "v_mov_b32 v14, 2.0\n v_mov_b32 v15, 4.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 0.5\n v_mov_b32 v18, 2.0\n v_mov_b32 v19, 4.0\n" \
             "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 2.0\n v_mov_b32 v23, 4.0\n"
// D(i): the i-th destination
#define BODY8(op, srcs, d0) op " v" #d0 ", " srcs "\n"

template <int KIND>
__global__ __launch_bounds__(1024, 4) void asm_probe(Stamp *stamps, int iters)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define DST32(op, srcs) \
    op " v40, " srcs "\n" op " v41, " srcs "\n" op " v42, " srcs "\n" op " v43, " srcs "\n" op " v44, " srcs "\n" op " v45, " srcs "\n" op " v46, " srcs "\n" op " v47, " srcs "\n" \
    op " v48, " srcs "\n" op " v49, " srcs "\n" op " v50, " srcs "\n" op " v51, " srcs "\n" op " v52, " srcs "\n" op " v53, " srcs "\n" op " v54, " srcs "\n" op " v55, " srcs "\n" \
    op " v56, " srcs "\n" op " v57, " srcs "\n" op " v58, " srcs "\n" op " v59, " srcs "\n" op " v60, " srcs "\n" op " v61, " srcs "\n" op " v62, " srcs "\n" op " v63, " srcs "\n" \
    op " v64, " srcs "\n" op " v65, " srcs "\n" op " v66, " srcs "\n" op " v67, " srcs "\n" op " v68, " srcs "\n" op " v69, " srcs "\n" op " v70, " srcs "\n" op " v71, " srcs "\n"
    if (KIND == 0) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v9, v10") LOOP_END : : "s"(iters) : CLOBBERS);          // 3 VGPRs, numbers 0,1,2 mod 4
    if (KIND == 1) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v12, v16") LOOP_END : : "s"(iters) : CLOBBERS);         // 3 VGPRs, all 0 mod 4
    if (KIND == 2) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v8, v8") LOOP_END : : "s"(iters) : CLOBBERS);           // one VGPR three times
    if (KIND == 3) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v9, 2.0") LOOP_END : : "s"(iters) : CLOBBERS);          // 2 VGPRs + inline constant
    if (KIND == 4) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, s41, v9") LOOP_END : : "s"(iters) : CLOBBERS, "s41");   // 2 VGPRs + SGPR
    if (KIND == 5) asm volatile(INIT LOOP_BEGIN DST32("v_add_f32", "v8, v9") LOOP_END : : "s"(iters) : CLOBBERS);               // 2 VGPRs (VOP2)
    if (KIND == 6) asm volatile(INIT LOOP_BEGIN DST32("v_add_f32", "v8, v12") LOOP_END : : "s"(iters) : CLOBBERS);              // 2 VGPRs, same number mod 4
    if (KIND == 7) asm volatile(INIT LOOP_BEGIN DST32("v_mul_f32", "2.0, v8") LOOP_END : : "s"(iters) : CLOBBERS);              // 1 VGPR + inline constant
    if (KIND == 8) asm volatile(INIT LOOP_BEGIN DST32("v_mul_f32", "0x3f3504f3, v8") LOOP_END : : "s"(iters) : CLOBBERS);       // 1 VGPR + 32-bit literal
    if (KIND == 9) asm volatile(INIT LOOP_BEGIN DST32("v_fmac_f32", "v8, v9") LOOP_END : : "s"(iters) : CLOBBERS);              // VOP2 fmac: dst += a*b (3 reads)
    if (KIND == 10) asm volatile(INIT LOOP_BEGIN DST32("v_fmamk_f32", "v8, 0x3f3504f3, v9") LOOP_END : : "s"(iters) : CLOBBERS); // a*K + b
    if (KIND == 11) asm volatile(INIT LOOP_BEGIN DST32("v_mov_b32", "v8") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 12) asm volatile(INIT LOOP_BEGIN DST32("v_max3_f32", "v8, v9, v10") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 13) asm volatile(INIT LOOP_BEGIN DST32("v_cndmask_b32", "v8, v9, vcc") LOOP_END : : "s"(iters) : CLOBBERS, "vcc");
    if (KIND == 14) asm volatile(INIT LOOP_BEGIN DST32("v_perm_b32", "v8, v9, v10") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 15) asm volatile(INIT LOOP_BEGIN DST32("v_cvt_u32_f32", "v8") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 16) asm volatile(INIT LOOP_BEGIN DST32("v_med3_f32", "v8, 0, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 17) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v13, v18") LOOP_END : : "s"(iters) : CLOBBERS);        // 3 VGPRs 0,1,2 mod 4, spread
    if (KIND == 18) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v9, v12") LOOP_END : : "s"(iters) : CLOBBERS);         // 0,1,0 mod 4
    if (KIND == 19) asm volatile(INIT LOOP_BEGIN DST32("v_log_f32", "v8") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 20) asm volatile(INIT "s_mov_b64 s[42:43], 0x5555\n" LOOP_BEGIN DST32("v_cndmask_b32_e64", "v8, v9, s[42:43]") LOOP_END : : "s"(iters) : CLOBBERS, "s42", "s43");
    if (KIND == 21) asm volatile(INIT LOOP_BEGIN DST32("v_max_f32", "v8, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 22) asm volatile(INIT LOOP_BEGIN DST32("v_lshl_or_b32", "v8, 8, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 23) asm volatile(INIT LOOP_BEGIN DST32("v_or3_b32", "v8, v9, v10") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 24) asm volatile(INIT LOOP_BEGIN R8(R4("v_cmp_lt_f32 vcc, v8, v9\n")) LOOP_END : : "s"(iters) : CLOBBERS, "vcc");
    if (KIND == 25) asm volatile(INIT LOOP_BEGIN DST32("v_cvt_f32_i32", "v8") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 26) asm volatile(INIT "s_mov_b32 s41, 0x3f000000\n" LOOP_BEGIN DST32("v_mul_f32", "s41, v8") LOOP_END : : "s"(iters) : CLOBBERS, "s41");
    if (KIND == 27) asm volatile(INIT LOOP_BEGIN DST32("v_mul_f32", "v8, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 28) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v9, -v10") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 29) asm volatile(INIT LOOP_BEGIN DST32("v_mov_b32_dpp", "v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 30) asm volatile(INIT LOOP_BEGIN DST32("v_sqrt_f32", "v8") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 31) asm volatile(INIT LOOP_BEGIN DST32("v_add_u32", "v8, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 32) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v9, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 33) asm volatile(INIT LOOP_BEGIN DST32("v_fma_f32", "v8, v12, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 34) asm volatile(INIT LOOP_BEGIN DST32("v_max3_u32", "v8, v9, v10") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 35) asm volatile(INIT LOOP_BEGIN DST32("v_cvt_pk_u8_f32", "v8, 1, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 36) asm volatile(INIT LOOP_BEGIN DST32("v_fmaak_f32", "v8, v9, 0x3f3504f3") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 37) asm volatile(INIT LOOP_BEGIN DST32("v_sub_f32", "v8, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 38) asm volatile(INIT LOOP_BEGIN DST32("v_max_u32", "v8, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 40) asm volatile(INIT LOOP_BEGIN DST32("v_mul_f32_e64", "v8, v9") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 41) asm volatile(INIT LOOP_BEGIN DST32("v_bfe_u32", "v8, 8, 8") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 42) asm volatile(INIT LOOP_BEGIN DST32("v_cvt_u32_f32_sdwa", "v8 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:DWORD") LOOP_END : : "s"(iters) : CLOBBERS);
    // packed f32: 16 instructions per round on the pairs v[40:41] ... v[70:71]
#define PK16(op, srcs) \
    op " v[40:41], " srcs "\n" op " v[42:43], " srcs "\n" op " v[44:45], " srcs "\n" op " v[46:47], " srcs "\n" op " v[48:49], " srcs "\n" op " v[50:51], " srcs "\n" op " v[52:53], " srcs "\n" op " v[54:55], " srcs "\n" \
    op " v[56:57], " srcs "\n" op " v[58:59], " srcs "\n" op " v[60:61], " srcs "\n" op " v[62:63], " srcs "\n" op " v[64:65], " srcs "\n" op " v[66:67], " srcs "\n" op " v[68:69], " srcs "\n" op " v[70:71], " srcs "\n"
    if (KIND == 60) asm volatile(INIT LOOP_BEGIN PK16("v_pk_fma_f32", "v[8:9], v[10:11], v[12:13]") PK16("v_pk_fma_f32", "v[8:9], v[10:11], v[12:13]") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 61) asm volatile(INIT LOOP_BEGIN PK16("v_pk_add_f32", "v[8:9], v[10:11]") PK16("v_pk_add_f32", "v[8:9], v[10:11]") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 62) asm volatile(INIT LOOP_BEGIN PK16("v_pk_mul_f32", "v[8:9], v[10:11]") PK16("v_pk_mul_f32", "v[8:9], v[10:11]") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 63) asm volatile(INIT LOOP_BEGIN PK16("v_pk_fma_f32", "v[8:9], v[10:11], v[12:13] op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]") PK16("v_pk_fma_f32", "v[8:9], v[10:11], v[12:13] op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]") LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 64) asm volatile(INIT "s_mov_b32 s42, 0x3f3504f3\n s_mov_b32 s43, 0x3f3504f3\n" LOOP_BEGIN PK16("v_pk_fma_f32", "v[8:9], s[42:43], v[12:13]") PK16("v_pk_fma_f32", "v[8:9], s[42:43], v[12:13]") LOOP_END : : "s"(iters) : CLOBBERS, "s42", "s43");
    if (KIND == 65) asm volatile(INIT LOOP_BEGIN PK16("v_pk_fma_f32", "v[8:9], v[10:11], v[12:13]") DST32("v_fma_f32", "v8, v9, v10") LOOP_END : : "s"(iters) : CLOBBERS);   // 16 packed + 32 plain per round
    if (KIND == 66) asm volatile(INIT LOOP_BEGIN PK16("v_pk_mul_f32", "v[8:9], v[10:11] op_sel_hi:[1,0]") PK16("v_pk_mul_f32", "v[8:9], v[10:11] op_sel_hi:[1,0]") LOOP_END : : "s"(iters) : CLOBBERS);
    // compare-and-select sequences as the compiler writes them: 16 pairs = 32 instructions per round
#define CS16(cmp, sel) cmp(40) sel(40) cmp(41) sel(41) cmp(42) sel(42) cmp(43) sel(43) cmp(44) sel(44) cmp(45) sel(45) cmp(46) sel(46) cmp(47) sel(47) \
                       cmp(48) sel(48) cmp(49) sel(49) cmp(50) sel(50) cmp(51) sel(51) cmp(52) sel(52) cmp(53) sel(53) cmp(54) sel(54) cmp(55) sel(55)
#define CMP_VCC(d) "v_cmp_lt_f32_e32 vcc, v8, v" #d "\n"
#define SEL_VCC(d) "v_cndmask_b32_e32 v" #d ", v8, v9, vcc\n"
#define CMP_S(d) "v_cmp_lt_f32_e64 s[42:43], v8, v" #d "\n"
#define SEL_S(d) "v_cndmask_b32_e64 v" #d ", v8, v9, s[42:43]\n"
#define CMP_S2(d) "v_cmp_lt_f32_e64 s[44:45], v8, v" #d "\n"
#define MAXDPP(d) "v_max_f32_dpp v" #d ", v8, v9 row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define MINDPP(d) "v_min_u32_dpp v" #d ", v8, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define NOTHING(d)
    if (KIND == 50) asm volatile(INIT LOOP_BEGIN CS16(CMP_VCC, SEL_VCC) LOOP_END : : "s"(iters) : CLOBBERS, "vcc");
    if (KIND == 51) asm volatile(INIT LOOP_BEGIN CS16(CMP_S, SEL_S) LOOP_END : : "s"(iters) : CLOBBERS, "s42", "s43");
    if (KIND == 52) asm volatile(INIT "s_mov_b64 s[42:43], 0x5555\n" LOOP_BEGIN CS16(CMP_S2, SEL_S) LOOP_END : : "s"(iters) : CLOBBERS, "s42", "s43", "s44", "s45");   // select on an old mask
    if (KIND == 53) asm volatile(INIT LOOP_BEGIN CS16(MAXDPP, MAXDPP) LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 54) asm volatile(INIT LOOP_BEGIN CS16(MINDPP, MINDPP) LOOP_END : : "s"(iters) : CLOBBERS);
    if (KIND == 55) asm volatile(INIT "s_mov_b64 vcc, 0x5555\n" LOOP_BEGIN CS16(SEL_VCC, SEL_VCC) LOOP_END : : "s"(iters) : CLOBBERS, "vcc");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0, r0, where_am_i()};
}

// ---- the kernel's own transforms on registers -----------------------------------------------------
// KIND 0: 64 multiplies by a constant (what keeps the values finite; subtracted from the others)
// KIND 1: + complex DFT-32 (pass 2)      KIND 2: + two real-input DFT-32 (pass 1, both frames of a pair)
// KIND 3: + 32 quantised bins: the epilogue's arithmetic      KIND 4: + complex DFT-32 with packed instructions
template <int KIND>
__global__ __launch_bounds__(1024, 4) void core_probe(Stamp *stamps, float *sink, int iters)
{
    float re[32], im[32];
    for (int i = 0; i < 32; ++i) { re[i] = (float)((threadIdx.x * 7 + i * 13) % 101) * 1e-2f - 0.5f; im[i] = (float)((threadIdx.x * 3 + i * 29) % 103) * 1e-2f - 0.5f; }
    unsigned int acc = 0;
    const float qa = sono::in_vgpr(9.6f), qc = sono::in_vgpr(100.0f);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 1) sono::tile::dft_bitrev_in<32>(re, im);
        if (KIND == 2) { sono::tile::rdft_bitrev_in<32>(re); sono::tile::rdft_bitrev_in<32>(im); }
        if (KIND == 4) {
            sono::tile::pk2 z[32];
            for (int i = 0; i < 32; ++i) z[i] = sono::tile::pk_make(re[i], im[i]);
            sono::tile::cdft_bitrev_in<32>(z);
            for (int i = 0; i < 32; ++i) { re[i] = z[i].x; im[i] = z[i].y; }
        }
        if (KIND == 3) {
            for (int i = 0; i < 32; i += 4) {
                unsigned int packed = 0;
                for (int r = 0; r < 4; ++r) packed |= sono::quantise<false>(re[i + r] * re[i + r], qa, qc, 0.0f) << (8 * r);
                acc ^= packed;
            }
        }
        for (int i = 0; i < 32; ++i) { re[i] *= 0.173f; im[i] *= 0.171f; }
        sono::tile::pin_registers(re);
        sono::tile::pin_registers(im);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)acc;
    for (int i = 0; i < 32; ++i) s += re[i] + im[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0, r0, where_am_i()};
}

static void report(const char *name, std::vector<Stamp> &h, float ms, double instr_per_wave)
{
    std::vector<double> cyc, clk;
    for (auto &s : h) { cyc.push_back((double)s.cyc); clk.push_back((double)s.cyc / (double)s.real * 0.1); }   // GHz: realtime ticks at 100 MHz
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double c = cyc[cyc.size() / 2], g = clk[clk.size() / 2];
    // 4 waves share a SIMD: cycles per wave-instruction per SIMD = wave cycles / (4 * instructions per wave)
    unsigned long long first = ~0ull, last = 0;
    std::vector<unsigned long long> cu;
    for (auto &s : h) {
        first = std::min(first, s.start); last = std::max(last, s.start);
        // HW_ID: [3:0] wave, [5:4] SIMD, [11:8] CU, [12] SH, [15:13] SE; XCC_ID [3:0]
        cu.push_back(((s.where >> 32) & 15) << 16 | (s.where & 0xFF00));
    }
    std::sort(cu.begin(), cu.end());
    const size_t distinct = std::unique(cu.begin(), cu.end()) - cu.begin();
    printf("%-46s clock %.3f GHz  wave cycles min %.0f med %.0f max %.0f  starts spread %.1f us  CUs used %zu  launch %.3f ms", name, g, cyc.front(), c, cyc.back(),
           (double)(last - first) * 0.01, distinct, ms);
    if (instr_per_wave > 0) printf("  -> %.3f cycles per wave-instruction per SIMD", c / (4.0 * instr_per_wave));
    printf("\n");
}

template <int KIND>
static void run_asm(const char *name, int cus)
{
    const int iters = 20000;
    Stamp *d; CHECK(hipMalloc(&d, (size_t)cus * 16 * sizeof(Stamp)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(asm_probe<KIND>, dim3(cus), dim3(1024), 0, 0, d, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(asm_probe<KIND>, dim3(cus), dim3(1024), 0, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h((size_t)cus * 16);
    CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    report(name, h, ms, 32.0 * iters);
    CHECK(hipFree(d));
}

template <int KIND>
static void run_core(const char *name, int cus)
{
    const int iters = 3000;
    Stamp *d; float *sink;
    CHECK(hipMalloc(&d, (size_t)cus * 16 * sizeof(Stamp))); CHECK(hipMalloc(&sink, (size_t)cus * 1024 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(core_probe<KIND>, dim3(cus), dim3(1024), 0, 0, d, sink, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(core_probe<KIND>, dim3(cus), dim3(1024), 0, 0, d, sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h((size_t)cus * 16);
    CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> cyc;
    for (auto &s : h) cyc.push_back((double)s.cyc);
    std::sort(cyc.begin(), cyc.end());
    report(name, h, ms, 0.0);
    printf("    = %.1f cycles per loop round per wave (4 waves per SIMD: %.1f SIMD cycles per round)\n", cyc[cyc.size() / 2] / iters, cyc[cyc.size() / 2] / iters / 4.0);
    CHECK(hipFree(d)); CHECK(hipFree(sink));
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s CUs %d; one workgroup of 16 waves per CU\n", p.gcnArchName, cus);
    run_asm<0>("v_fma_f32 d, v8, v9, v10", cus);
    run_asm<17>("v_fma_f32 d, v8, v13, v18", cus);
    run_asm<1>("v_fma_f32 d, v8, v12, v16 (same mod 4)", cus);
    run_asm<18>("v_fma_f32 d, v8, v9, v12", cus);
    run_asm<2>("v_fma_f32 d, v8, v8, v8", cus);
    run_asm<3>("v_fma_f32 d, v8, v9, 2.0", cus);
    run_asm<4>("v_fma_f32 d, v8, s41, v9", cus);
    run_asm<9>("v_fmac_f32 d, v8, v9", cus);
    run_asm<10>("v_fmamk_f32 d, v8, K, v9", cus);
    run_asm<5>("v_add_f32 d, v8, v9", cus);
    run_asm<6>("v_add_f32 d, v8, v12 (same mod 4)", cus);
    run_asm<7>("v_mul_f32 d, 2.0, v8", cus);
    run_asm<8>("v_mul_f32 d, literal, v8", cus);
    run_asm<11>("v_mov_b32 d, v8", cus);
    run_asm<12>("v_max3_f32 d, v8, v9, v10", cus);
    run_asm<13>("v_cndmask_b32 d, v8, v9, vcc", cus);
    run_asm<14>("v_perm_b32 d, v8, v9, v10", cus);
    run_asm<15>("v_cvt_u32_f32 d, v8", cus);
    run_asm<16>("v_med3_f32 d, v8, 0, v9", cus);
    run_asm<19>("v_log_f32 d, v8", cus);
    run_asm<30>("v_sqrt_f32 d, v8", cus);
    run_asm<20>("v_cndmask_b32_e64 d, v8, v9, s[42:43]", cus);
    run_asm<21>("v_max_f32 d, v8, v9", cus);
    run_asm<38>("v_max_u32 d, v8, v9", cus);
    run_asm<34>("v_max3_u32 d, v8, v9, v10", cus);
    run_asm<22>("v_lshl_or_b32 d, v8, 8, v9", cus);
    run_asm<23>("v_or3_b32 d, v8, v9, v10", cus);
    run_asm<41>("v_bfe_u32 d, v8, 8, 8", cus);
    run_asm<24>("v_cmp_lt_f32 vcc, v8, v9", cus);
    run_asm<25>("v_cvt_f32_i32 d, v8", cus);
    run_asm<42>("v_cvt_u32_f32_sdwa d, v8 (byte 1)", cus);
    run_asm<35>("v_cvt_pk_u8_f32 d, v8, 1, v9", cus);
    run_asm<26>("v_mul_f32 d, s41, v8", cus);
    run_asm<27>("v_mul_f32 d, v8, v9", cus);
    run_asm<40>("v_mul_f32_e64 d, v8, v9", cus);
    run_asm<37>("v_sub_f32 d, v8, v9", cus);
    run_asm<31>("v_add_u32 d, v8, v9", cus);
    run_asm<28>("v_fma_f32 d, v8, v9, -v10", cus);
    run_asm<32>("v_fma_f32 d, v8, v9, v9", cus);
    run_asm<33>("v_fma_f32 d, v8, v12, v9", cus);
    run_asm<36>("v_fmaak_f32 d, v8, v9, K", cus);
    run_asm<29>("v_mov_b32_dpp d, v8 quad_perm", cus);
    run_asm<60>("v_pk_fma_f32 (32 per round)", cus);
    run_asm<63>("v_pk_fma_f32 op_sel swap + neg_lo", cus);
    run_asm<64>("v_pk_fma_f32 with s[42:43]", cus);
    run_asm<61>("v_pk_add_f32", cus);
    run_asm<62>("v_pk_mul_f32", cus);
    run_asm<66>("v_pk_mul_f32 broadcast low half of src1", cus);
    run_asm<65>("16 v_pk_fma_f32 + 32 v_fma_f32 per round (48 instructions; prints per 32)", cus);
    run_asm<50>("v_cmp_lt_f32 vcc + v_cndmask_e32 vcc (pairs)", cus);
    run_asm<51>("v_cmp_lt_f32_e64 s[] + v_cndmask_e64 s[] (pairs)", cus);
    run_asm<52>("v_cmp_e64 s[44:45] + v_cndmask_e64 s[42:43] (independent)", cus);
    run_asm<53>("v_max_f32_dpp row_ror", cus);
    run_asm<54>("v_min_u32_dpp quad_perm", cus);
    run_asm<55>("v_cndmask_e32 vcc, vcc set once", cus);
    run_core<0>("64 multiplies", cus);
    run_core<1>("64 multiplies + complex DFT-32", cus);
    run_core<2>("64 multiplies + 2 real-input DFT-32", cus);
    run_core<3>("64 multiplies + 32 quantised bins", cus);
    run_core<4>("64 multiplies + complex DFT-32 on (re, im) pairs", cus);
    return 0;
}
