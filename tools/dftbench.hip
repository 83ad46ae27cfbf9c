// dftbench.hip -- VALU issue rate of the in-register DFT-32 code alone (no memory, no LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../spectrogram_amd/csrc/sono_tile_core.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int WPS>
__global__ __launch_bounds__(256, WPS) void probe(float *sink, int iters)
{
    float re[32], im[32];
    for (int i = 0; i < 32; ++i) { re[i] = threadIdx.x * 1e-3f + i; im[i] = threadIdx.x * 2e-3f - i; }
    for (int it = 0; it < iters; ++it) {
        sono::tile::dft_bitrev_in<32>(re, im);
        for (int i = 0; i < 32; ++i) { re[i] *= 0.17f; }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += re[i] + im[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WPS>
static void run(int cus, int iters)
{
    const int blocks = cus * WPS;
    float *d_sink; CHECK(hipMalloc(&d_sink, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<WPS>, dim3(blocks), dim3(256), 0, 0, d_sink, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<WPS>, dim3(blocks), dim3(256), 0, 0, d_sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("DFT-32 loop  waves/SIMD=%d  %.3f ns per DFT-32 per SIMD (per wave-DFT)\n", WPS, ms * 1e6 / ((double)WPS * iters));
    CHECK(hipFree(d_sink));
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    run<1>(cus, 2000); run<2>(cus, 2000); run<3>(cus, 2000); run<4>(cus, 2000);
    return 0;
}
