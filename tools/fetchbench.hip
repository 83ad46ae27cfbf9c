// fetchbench.hip -- does straight-line code size limit VALU issue on gfx950?
// Runs the same number of v_fma_f32 (VOP3, 8 bytes) / v_add_f32 (VOP2, 4 bytes)
// instructions as a short loop body (256 instructions) and as a long one (4096,
// 32 KB for VOP3), at several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define REP4(s) s s s s
#define REP16(s) REP4(s) REP4(s) REP4(s) REP4(s)
#define REP64(s) REP16(s) REP16(s) REP16(s) REP16(s)
#define REP256(s) REP64(s) REP64(s) REP64(s) REP64(s)
#define REP1024(s) REP256(s) REP256(s) REP256(s) REP256(s)

#define FMA4 "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
#define ADD4 "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
#define FMAC4 "v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n"

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *sink, int iters)
{
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(REP64(FMA4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));        // 256 instr
        if (KIND == 1) asm volatile(REP1024(FMA4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));      // 4096 instr
        if (KIND == 2) asm volatile(REP64(ADD4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (KIND == 3) asm volatile(REP1024(ADD4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (KIND == 4) asm volatile(REP64(FMAC4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (KIND == 5) asm volatile(REP1024(FMAC4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int KIND>
static void run(const char *name, int per_iter, int blocks_per_cu, int cus)
{
    const int total_instr = 1 << 20;              // per wave
    const int iters = total_instr / per_iter;
    const int blocks = cus * blocks_per_cu;
    float *d_sink; CHECK(hipMalloc(&d_sink, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_sink, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd_ns = ms * 1e6 / ((double)blocks * 4 * total_instr / (cus * 4.0));
    printf("%-28s body %5d instr  waves/SIMD=%d  %.3f ns per wave-instr per SIMD\n", name, per_iter, blocks_per_cu, per_simd_ns);
    CHECK(hipFree(d_sink));
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    for (int w : {2, 4, 8}) {
        run<0>("v_fma_f32 (VOP3, 8 B)", 256, w, cus);
        run<1>("v_fma_f32 (VOP3, 8 B)", 4096, w, cus);
        run<2>("v_add_f32 (VOP2, 4 B)", 256, w, cus);
        run<3>("v_add_f32 (VOP2, 4 B)", 4096, w, cus);
        run<4>("v_fmac_f32 (VOP2, 4 B)", 256, w, cus);
        run<5>("v_fmac_f32 (VOP2, 4 B)", 4096, w, cus);
    }
    return 0;
}
