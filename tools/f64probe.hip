// f64probe.hip -- what the vector ALU of gfx950 charges for the double-precision instruction forms the detector's dB
// conversion and the measurement kernel are made of, next to a plain f32 fma: loops of 32 independent instructions,
// every CU busy with 16 waves (4 per SIMD).  Prints the time per wave-instruction and SIMD relative to v_fma_f32's.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/f64probe tools/f64probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum Kind { FMA32, FMA64, ADD64, MUL64, MAX64, CVT_64_32, CVT_32_64, CVT_64_I32, CMP_SEL64, RCP64, SQRT64, LDEXP64, SHL64, DIV64, FMA64_CHAIN, KINDS };
static const char *names[KINDS] = {"v_fma_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_max_f64", "v_cvt_f64_f32", "v_cvt_f32_f64",
                                   "v_cvt_f64_i32", "v_cmp_gt_f64 + 2 v_cndmask", "v_rcp_f64", "v_sqrt_f64", "v_ldexp_f64", "v_lshlrev_b64",
                                   "a / b in double (the compiler's sequence)", "v_fma_f64, each reading the one before"};

template <int KIND> __device__ __forceinline__ void step(double (&d)[32], float (&s)[32], double a, double b, float fa, float fb, int ia)
{
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (KIND == FMA32) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(s[i]) : "v"(fa), "v"(fb));
        if (KIND == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(d[i]) : "v"(a), "v"(b));
        if (KIND == ADD64) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a), "v"(b));
        if (KIND == MUL64) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a), "v"(b));
        if (KIND == MAX64) asm volatile("v_max_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a), "v"(b));
        if (KIND == CVT_64_32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(fa));
        if (KIND == CVT_32_64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s[i]) : "v"(a));
        if (KIND == CVT_64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(ia));
        if (KIND == CMP_SEL64) d[i] = a > b + (double)i ? a : b;                 // the compiler's choice of compare and selects
        if (KIND == RCP64) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(a));
        if (KIND == SQRT64) asm volatile("v_sqrt_f64 %0, %1" : "=v"(d[i]) : "v"(a));
        if (KIND == LDEXP64) asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a), "v"(ia));
        if (KIND == SHL64) asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(d[i]) : "v"(ia), "v"(a));
        if (KIND == DIV64) d[i] = (a + (double)i) / b;
        if (KIND == FMA64_CHAIN) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d[i]) : "v"(a), "v"(b), "v"(d[(i + 31) & 31]));
    }
}

template <int KIND> __global__ __launch_bounds__(256) void probe(int rounds, double a, double b, double *sink)
{
    double d[32];
    float s[32];
    for (int i = 0; i < 32; ++i) { d[i] = a; s[i] = (float)b; }
    const float fa = (float)a, fb = (float)b;
    const int ia = (int)threadIdx.x & 3;
    for (int r = 0; r < rounds; ++r) step<KIND>(d, s, a + (KIND == DIV64 || KIND == CMP_SEL64 ? (double)r : 0.0), b, fa, fb, ia);
    double sum = 0;
    for (int i = 0; i < 32; ++i) sum += d[i] + (double)s[i];
    if (sum == 12345.678) sink[0] = sum;                                         // never: keeps the results alive
}

template <int KIND> static double run(int rounds, double *sink)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int cus = 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    hipLaunchKernelGGL(probe<KIND>, dim3(cus * 4), dim3(256), 0, 0, rounds / 10, 1.25, 0.75, sink);     // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<KIND>, dim3(cus * 4), dim3(256), 0, 0, rounds, 1.25, 0.75, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    // per SIMD: 4 waves x rounds x 32 instructions
    return (double)best * 1e6 / (4.0 * rounds * 32.0);                          // ns per wave-instruction and SIMD
}

int main()
{
    double *sink;
    CHECK(hipMalloc(&sink, 64));
    const int rounds = 20000;
    double ns[KINDS];
    ns[FMA32] = run<FMA32>(rounds, sink); ns[FMA64] = run<FMA64>(rounds, sink); ns[ADD64] = run<ADD64>(rounds, sink);
    ns[MUL64] = run<MUL64>(rounds, sink); ns[MAX64] = run<MAX64>(rounds, sink); ns[CVT_64_32] = run<CVT_64_32>(rounds, sink);
    ns[CVT_32_64] = run<CVT_32_64>(rounds, sink); ns[CVT_64_I32] = run<CVT_64_I32>(rounds, sink); ns[CMP_SEL64] = run<CMP_SEL64>(rounds, sink);
    ns[RCP64] = run<RCP64>(rounds, sink); ns[SQRT64] = run<SQRT64>(rounds, sink); ns[LDEXP64] = run<LDEXP64>(rounds, sink);
    ns[SHL64] = run<SHL64>(rounds, sink); ns[DIV64] = run<DIV64>(rounds / 8, sink); ns[FMA64_CHAIN] = run<FMA64_CHAIN>(rounds, sink);
    printf("%-46s %10s %12s\n", "instruction (32 independent per loop round)", "ns / instr", "x v_fma_f32");
    for (int k = 0; k < KINDS; ++k) printf("%-46s %10.3f %12.2f\n", names[k], ns[k], ns[k] / ns[FMA32]);
    return 0;
}
