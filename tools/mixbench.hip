// mixbench.hip -- do two kinds of instruction share one pipe of a gfx950 SIMD, or do they run side by side?
//   * plain vector arithmetic (v_fma_f32) next to the half-rate class (v_max_f32, v_cvt_pk_u8_f32, v_log_f32):
//     if a round of 16 + 16 takes the sum of the two kinds' own times they share a pipe, if it takes the larger they do not;
//   * f32 matrix instructions (v_mfma_f32_16x16x4_f32, 32 MACs per cycle and SIMD -- the vector ALU's own rate) next to
//     v_fma_f32 of the same and of other waves: can the matrix pipe take work off a kernel that is bound by vector issue?
// 16 waves per CU (4 per SIMD), every CU busy, as the tiled STFT kernel runs.  Prints SIMD cycles per round.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mixbench tools/mixbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long cyc, real; };

#define LOOP_BEGIN "s_mov_b32 s40, %0\n 1:\n"
#define LOOP_END "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"
#define CLOBBERS "s40", "scc", "vcc", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", \
    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
    "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", \
    "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87"
#define INIT "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 2.0\n v_mov_b32 v11, 4.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 0.5\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n" \
    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n" \
    "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"

// destinations v56..v87 for the vector instructions (never read inside the loop)
#define F(d) "v_fma_f32 v" #d ", v8, v9, v10\n"
#define M(d) "v_max_f32 v" #d ", v8, v9\n"
#define C(d) "v_cvt_pk_u8_f32 v" #d ", v8, 1, v9\n"
#define L(d) "v_log_f32 v" #d ", v8\n"
#define X4(op, a) op(a) op(a + 1) op(a + 2) op(a + 3)
#define F4a F(56) F(57) F(58) F(59)
#define F4b F(60) F(61) F(62) F(63)
#define F4c F(64) F(65) F(66) F(67)
#define F4d F(68) F(69) F(70) F(71)
#define F4e F(72) F(73) F(74) F(75)
#define F4f F(76) F(77) F(78) F(79)
#define F4g F(80) F(81) F(82) F(83)
#define F4h F(84) F(85) F(86) F(87)
#define F16 F4a F4b F4c F4d
#define F32 F16 F4e F4f F4g F4h
#define M16 M(72) M(73) M(74) M(75) M(76) M(77) M(78) M(79) M(80) M(81) M(82) M(83) M(84) M(85) M(86) M(87)
#define C16 C(72) C(73) C(74) C(75) C(76) C(77) C(78) C(79) C(80) C(81) C(82) C(83) C(84) C(85) C(86) C(87)
#define L16 L(72) L(73) L(74) L(75) L(76) L(77) L(78) L(79) L(80) L(81) L(82) L(83) L(84) L(85) L(86) L(87)
// alternating one of each
#define FM(a, b) F(a) M(b)
#define FM16 FM(56, 72) FM(57, 73) FM(58, 74) FM(59, 75) FM(60, 76) FM(61, 77) FM(62, 78) FM(63, 79) FM(64, 80) FM(65, 81) FM(66, 82) FM(67, 83) FM(68, 84) FM(69, 85) FM(70, 86) FM(71, 87)
#define FL(a, b) F(a) L(b)
#define FFFL(a, b, c, d) F(a) F(b) F(c) L(d)
// matrix instructions: four independent accumulators
#define MF(a) "v_mfma_f32_16x16x4_f32 v[" #a ":" #a "+3], v12, v13, v[" #a ":" #a "+3]\n"
#define MFA "v_mfma_f32_16x16x4_f32 v[40:43], v12, v13, v[40:43]\n"
#define MFB "v_mfma_f32_16x16x4_f32 v[44:47], v12, v13, v[44:47]\n"
#define MFC "v_mfma_f32_16x16x4_f32 v[48:51], v12, v13, v[48:51]\n"
#define MFD "v_mfma_f32_16x16x4_f32 v[52:55], v12, v13, v[52:55]\n"
// packed f32: destinations the pairs v[40:41] ... v[54:55] (P8) and v[72:73] ... v[86:87] (Q8)
#define PK(a) "v_pk_fma_f32 v[" #a ":" #a "+1], v[8:9], v[10:11], v[12:13]\n"
#define P8 PK(40) PK(42) PK(44) PK(46) PK(48) PK(50) PK(52) PK(54)
#define Q8 PK(72) PK(74) PK(76) PK(78) PK(80) PK(82) PK(84) PK(86)
#define PF(a, b) PK(a) F(b)
#define PF8a PF(40, 56) PF(42, 57) PF(44, 58) PF(46, 59) PF(48, 60) PF(50, 61) PF(52, 62) PF(54, 63)
#define PF8b PF(72, 64) PF(74, 65) PF(76, 66) PF(78, 67) PF(80, 68) PF(82, 69) PF(84, 70) PF(86, 71)
#define PFF(a, b, c) PK(a) F(b) F(c)
#define PFF8a PFF(40, 56, 57) PFF(42, 58, 59) PFF(44, 60, 61) PFF(46, 62, 63) PFF(48, 64, 65) PFF(50, 66, 67) PFF(52, 68, 69) PFF(54, 70, 71)

template <int KIND>
__global__ __launch_bounds__(1024, 4) void probe(Stamp *stamps, int iters)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (KIND == 0) asm volatile(INIT LOOP_BEGIN F32 LOOP_END : : "s"(iters) : CLOBBERS);                       // 32 fma
    if (KIND == 1) asm volatile(INIT LOOP_BEGIN M16 M16 LOOP_END : : "s"(iters) : CLOBBERS);                   // 32 max
    if (KIND == 2) asm volatile(INIT LOOP_BEGIN F16 M16 LOOP_END : : "s"(iters) : CLOBBERS);                   // 16 fma then 16 max
    if (KIND == 3) asm volatile(INIT LOOP_BEGIN FM16 LOOP_END : : "s"(iters) : CLOBBERS);                      // fma, max alternating
    if (KIND == 4) asm volatile(INIT LOOP_BEGIN F16 C16 LOOP_END : : "s"(iters) : CLOBBERS);                   // 16 fma + 16 cvt_pk_u8
    if (KIND == 5) asm volatile(INIT LOOP_BEGIN F16 L16 LOOP_END : : "s"(iters) : CLOBBERS);                   // 16 fma + 16 log
    if (KIND == 6) asm volatile(INIT LOOP_BEGIN L16 L16 LOOP_END : : "s"(iters) : CLOBBERS);                   // 32 log
    if (KIND == 7) asm volatile(INIT LOOP_BEGIN F32 F16 L16 LOOP_END : : "s"(iters) : CLOBBERS);               // 48 fma + 16 log
    if (KIND == 20) asm volatile(INIT LOOP_BEGIN P8 Q8 P8 Q8 LOOP_END : : "s"(iters) : CLOBBERS);                 // 32 packed fma
    if (KIND == 21) asm volatile(INIT LOOP_BEGIN PF8a PF8b LOOP_END : : "s"(iters) : CLOBBERS);                 // 16 x (packed fma, fma)
    if (KIND == 22) asm volatile(INIT LOOP_BEGIN P8 Q8 F16 LOOP_END : : "s"(iters) : CLOBBERS);                 // 16 packed fma, then 16 fma
    if (KIND == 23) asm volatile(INIT LOOP_BEGIN PFF8a PFF8a LOOP_END : : "s"(iters) : CLOBBERS);               // 16 x (packed fma, fma, fma)
    if (KIND == 24) asm volatile(INIT LOOP_BEGIN F32 F16 LOOP_END : : "s"(iters) : CLOBBERS);                   // 48 fma
    if (KIND == 25) asm volatile(INIT LOOP_BEGIN P8 Q8 LOOP_END : : "s"(iters) : CLOBBERS);                     // 16 packed fma
    if (KIND == 10) asm volatile(INIT LOOP_BEGIN MFA MFB MFC MFD LOOP_END : : "s"(iters) : CLOBBERS);          // 4 mfma
    if (KIND == 11) asm volatile(INIT LOOP_BEGIN MFA F4a MFB F4b MFC F4c MFD F4d LOOP_END : : "s"(iters) : CLOBBERS);                   // 4 mfma + 16 fma
    if (KIND == 12) asm volatile(INIT LOOP_BEGIN MFA F4a F4e MFB F4b F4f MFC F4c F4g MFD F4d F4h LOOP_END : : "s"(iters) : CLOBBERS);   // 4 mfma + 32 fma
    if (KIND == 13) asm volatile(INIT LOOP_BEGIN MFA F16 MFB F16 MFC F16 MFD F16 LOOP_END : : "s"(iters) : CLOBBERS);                   // 4 mfma + 64 fma
    if (KIND == 14) asm volatile(INIT LOOP_BEGIN MFA MFB MFC MFD F32 F32 LOOP_END : : "s"(iters) : CLOBBERS);                           // 4 mfma, then 64 fma
    if (KIND == 15) asm volatile(INIT LOOP_BEGIN F32 F32 LOOP_END : : "s"(iters) : CLOBBERS);                                           // 64 fma
    if (KIND == 16) {       // waves 0..7 of the workgroup only matrix instructions, waves 8..15 only fma (two of each kind per SIMD)
        if ((threadIdx.x >> 6) & 1) asm volatile(INIT LOOP_BEGIN MFA MFB MFC MFD LOOP_END : : "s"(iters) : CLOBBERS);
        else asm volatile(INIT LOOP_BEGIN F32 F32 LOOP_END : : "s"(iters) : CLOBBERS);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0};
}

// Occupancy sweep: workgroups of four waves (one per SIMD), each claiming 1/W of the CU's LDS, so that exactly W of them
// -- W waves per SIMD -- are resident on a CU.  32 independent v_fma_f32 per round.
__global__ __launch_bounds__(256) void occ_probe(Stamp *stamps, int iters)
{
    extern __shared__ float lds[];
    if (iters < 0) lds[threadIdx.x] = 0.0f;       // never true: keeps the allocation
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile(INIT LOOP_BEGIN F32 LOOP_END : : "s"(iters) : CLOBBERS);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0};
}

static void run_occ(int w, int cus)
{
    const int iters = 20000;
    const size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)1023;
    CHECK(hipFuncSetAttribute((const void *)occ_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Stamp *d; CHECK(hipMalloc(&d, (size_t)cus * w * 4 * sizeof(Stamp)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(occ_probe, dim3(cus * w), dim3(256), lds, 0, d, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(occ_probe, dim3(cus * w), dim3(256), lds, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h((size_t)cus * w * 4);
    CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (auto &s : h) { clk.push_back((double)s.cyc / (double)s.real * 0.1); cyc.push_back((double)s.cyc); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double g = clk[clk.size() / 2];
    printf("%d wave(s) per SIMD, 32 v_fma_f32 per round: clock %.3f GHz  launch %.3f ms  -> %.2f SIMD cycles per instruction  (a wave's own: min %.2f med %.2f max %.2f)\n",
           w, g, ms, ms * 1e-3 * g * 1e9 / iters / w / 32.0, cyc.front() / iters / 32.0, cyc[cyc.size() / 2] / iters / 32.0, cyc.back() / iters / 32.0);
    CHECK(hipFree(d));
}

template <int KIND>
static void run(const char *name, int cus)
{
    const int iters = 20000;
    Stamp *d; CHECK(hipMalloc(&d, (size_t)cus * 16 * sizeof(Stamp)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(cus), dim3(1024), 0, 0, d, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<KIND>, dim3(cus), dim3(1024), 0, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h((size_t)cus * 16);
    CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (auto &s : h) { cyc.push_back((double)s.cyc); clk.push_back((double)s.cyc / (double)s.real * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    // a SIMD's four waves run the loop side by side: SIMD cycles per round of one wave = wave cycles / rounds / 4
    // The launch's own time is the robust figure: the SIMD favours its oldest wave, so the waves of a SIMD do not run side by
    // side all the way and a single wave's cycle count says little.  SIMD cycles per wave-round = launch time x clock / rounds / 4.
    printf("%-62s clock %.3f GHz  launch %.3f ms  -> %.1f SIMD cycles per wave-round  (wave cycles per round: min %.0f med %.0f max %.0f)\n", name,
           clk[clk.size() / 2], ms, ms * 1e-3 * clk[clk.size() / 2] * 1e9 / iters / 4.0, cyc.front() / iters, cyc[cyc.size() / 2] / iters, cyc.back() / iters);
    CHECK(hipFree(d));
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s CUs %d; one workgroup of 16 waves per CU\n", p.gcnArchName, cus);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run_occ(w, cus);
    run<0>("32 v_fma_f32", cus);
    run<1>("32 v_max_f32", cus);
    run<2>("16 v_fma_f32, then 16 v_max_f32", cus);
    run<3>("16 x (v_fma_f32, v_max_f32)", cus);
    run<4>("16 v_fma_f32 + 16 v_cvt_pk_u8_f32", cus);
    run<6>("32 v_log_f32", cus);
    run<5>("16 v_fma_f32 + 16 v_log_f32", cus);
    run<7>("48 v_fma_f32 + 16 v_log_f32", cus);
    run<15>("64 v_fma_f32", cus);
    run<24>("48 v_fma_f32", cus);
    run<25>("16 v_pk_fma_f32", cus);
    run<20>("32 v_pk_fma_f32", cus);
    run<21>("16 x (v_pk_fma_f32, v_fma_f32)  [work of 48 fma]", cus);
    run<22>("16 v_pk_fma_f32, then 16 v_fma_f32  [work of 48 fma]", cus);
    run<23>("16 x (v_pk_fma_f32, v_fma_f32, v_fma_f32)  [work of 64 fma]", cus);
    run<10>("4 v_mfma_f32_16x16x4_f32", cus);
    run<11>("4 x (mfma, 4 fma)", cus);
    run<12>("4 x (mfma, 8 fma)", cus);
    run<13>("4 x (mfma, 16 fma)", cus);
    run<14>("4 mfma, then 64 fma", cus);
    run<16>("half the waves 4 mfma, the other half 64 fma (per round)", cus);
    return 0;
}
