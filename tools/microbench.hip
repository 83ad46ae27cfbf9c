// microbench.hip -- instruction-rate probes for gfx950 that decide how the tiled
// STFT kernel is written (VALU rate at several occupancies, packed math, DPP
// operands, transcendental rate, LDS exchange primitives, cvt_pk_u8 rounding).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITER = 2000;

#define REP8(s) s s s s s s s s
#define REP32(s) REP8(s) REP8(s) REP8(s) REP8(s)

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long *cycles, float *sink)
{
    __shared__ float lds[4096 + 64];
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) lds[i] = i;
    __syncthreads();
    unsigned int addr = (threadIdx.x & 63) * 8;        // byte address, conflict-free for b64
    unsigned int baddr = ((threadIdx.x * 7) & 63) * 4;   // bpermute: some permutation of lanes
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {   // v_fma_f32, 8 independent chains
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n") 
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 1) {  // v_add_f32
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 2) {  // v_pk_fma_f32 on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (KIND == 3) {  // v_add_f32 with a DPP row_mirror operand
            asm volatile(REP8("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %2, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %3, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %0, %3 row_mirror row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 4) {  // v_log_f32
            asm volatile(REP8("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 5) {  // v_sqrt_f32
            asm volatile(REP8("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 6) {  // mix: 3 fma : 1 log
            asm volatile(REP8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_log_f32 %3, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        } else if (KIND == 7) {  // ds_bpermute_b32
            asm volatile(REP8("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(baddr));
        } else if (KIND == 8) {  // ds_write_b64 (conflict-free)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1};
            asm volatile(REP32("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(p0) : "memory");
        } else if (KIND == 9) {  // ds_read_b64
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0, p1, p2, p3;
            asm volatile(REP8("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(addr) : "memory");
            a0 += p0.x + p1.x + p2.x + p3.x;
        } else if (KIND == 10) {  // ds_write_b128
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 p0 = {a0, a1, a2, a3};
            unsigned int ad = (threadIdx.x & 63) * 16;
            asm volatile(REP32("ds_write_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(ad), "v"(p0) : "memory");
        } else if (KIND == 11) {  // v_mov_b32_dpp row_ror:1
            asm volatile(REP8("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 12) {  // v_pk_add_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b};
            asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (KIND == 13) {  // v_cvt_pk_u8_f32
            unsigned int u0 = 0, u1 = 0, u2 = 0, u3 = 0;
            asm volatile(REP8("v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_cvt_pk_u8_f32 %1, %5, 1, %1\n v_cvt_pk_u8_f32 %2, %6, 2, %2\n v_cvt_pk_u8_f32 %3, %7, 3, %3\n")
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            a4 += __uint_as_float(u0 ^ u1 ^ u2 ^ u3);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[threadIdx.x];
}

__global__ void cvt_probe(const float *in, unsigned int *out_pk, unsigned int *out_cvt, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned int r = 0;
    float v = in[i];
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(v));
    out_pk[i] = r;
    out_cvt[i] = (unsigned int)v;   // v_cvt_u32_f32
}

template <int KIND>
static void run(const char *name, int instr_per_iter, int blocks_per_cu)
{
    int blocks = 256 * blocks_per_cu;
    unsigned long long *d_cyc; float *d_sink;
    CHECK(hipMalloc(&d_cyc, blocks * 4 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_sink, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    double per_instr_wave = med / ((double)ITER * instr_per_iter);      // memtime ticks (100 MHz) per wave-instruction
    // wall-based: total wave-instr / time / SIMDs
    double total = (double)blocks * 4 * ITER * instr_per_iter;
    double per_simd_ns = ms * 1e6 / (total / 1024.0);
    printf("%-34s waves/SIMD=%d  wall: %.3f ns per wave-instr per SIMD (=%.2f cyc @2.4GHz)  memtime ticks/instr/wave %.4f\n",
           name, blocks_per_cu, per_simd_ns, per_simd_ns * 2.4, per_instr_wave);
    CHECK(hipFree(d_cyc)); CHECK(hipFree(d_sink));
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 32, w);
        run<1>("v_add_f32", 32, w);
        run<2>("v_pk_fma_f32", 32, w);
        run<12>("v_pk_add_f32", 32, w);
        run<3>("v_add_f32_dpp row_mirror", 32, w);
        run<11>("v_mov_b32_dpp row_ror:1", 32, w);
        run<4>("v_log_f32", 32, w);
        run<5>("v_sqrt_f32", 32, w);
        run<6>("3 fma : 1 log", 32, w);
        run<13>("v_cvt_pk_u8_f32", 32, w);
        run<7>("ds_bpermute_b32", 32, w);
        run<8>("ds_write_b64", 32, w);
        run<10>("ds_write_b128", 32, w);
        run<9>("ds_read_b64", 32, w);
    }
    // rounding of v_cvt_pk_u8_f32 vs truncation
    std::vector<float> vals = {-1.5f, -0.5f, 0.0f, 0.25f, 0.5f, 0.75f, 0.999f, 1.0f, 1.5f, 2.5f, 3.5f, 126.5f, 127.5f, 254.4f, 254.5f, 254.6f, 254.999f, 255.0f, 255.5f, 256.0f, 300.0f, 1e9f, -1e9f, NAN, INFINITY, -INFINITY};
    float *d_in; unsigned int *d_a, *d_b;
    CHECK(hipMalloc(&d_in, vals.size() * 4)); CHECK(hipMalloc(&d_a, vals.size() * 4)); CHECK(hipMalloc(&d_b, vals.size() * 4));
    CHECK(hipMemcpy(d_in, vals.data(), vals.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, d_in, d_a, d_b, (int)vals.size());
    std::vector<unsigned int> ha(vals.size()), hb(vals.size());
    CHECK(hipMemcpy(ha.data(), d_a, vals.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hb.data(), d_b, vals.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < vals.size(); ++i) printf("cvt %12g -> pk_u8 %3u   cvt_u32 %u\n", vals[i], ha[i] & 0xff, hb[i]);
    return 0;
}
