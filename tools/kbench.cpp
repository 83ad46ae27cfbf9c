// kbench.cpp -- times sono_stft_u8_dev of a given libsonostft build on device-resident
// noise, without Python: `kbench <path/to/libsonostft.so> [fft] [noverlap] [seconds] [reps]`.
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/kbench tools/kbench.cpp -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "../include/sonostft.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void fill_noise(float *x, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned int v = (unsigned int)i * 2654435761u;
        v ^= v >> 15; v *= 0x2c1b3c6du; v ^= v >> 12;
        x[i] = ((float)(v >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.01f + ((i % 38400) < 1920 ? 0.5f * __sinf((float)(i % 38400) * 0.9f) : 0.0f);
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) { printf("usage: kbench lib.so [fft] [noverlap] [seconds] [reps]\n"); return 2; }
    const unsigned int N = argc > 2 ? atoi(argv[2]) : 1024, V = argc > 3 ? atoi(argv[3]) : 768;
    const size_t seconds = argc > 4 ? atoi(argv[4]) : 3600;
    const int reps = argc > 5 ? atoi(argv[5]) : 10;
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    auto create = (decltype(&sono_engine_create))dlsym(h, "sono_engine_create");
    auto u8dev = (decltype(&sono_stft_u8_dev))dlsym(h, "sono_stft_u8_dev");
    auto frames_for = (decltype(&sono_frames_for))dlsym(h, "sono_frames_for");
    auto last_error = (decltype(&sono_last_error))dlsym(h, "sono_last_error");
    const size_t n = seconds * 384000;
    const size_t F = frames_for(n, N, V);
    float *x; uint8_t *out; uint16_t *pb; float *pm; uint32_t *gm;
    CHECK(hipMalloc(&x, n * 4)); CHECK(hipMalloc(&out, F * (N / 2))); CHECK(hipMalloc(&pb, F * 2));
    CHECK(hipMalloc(&pm, F * 4)); CHECK(hipMalloc(&gm, 4));
    fill_noise<<<4096, 256>>>(x, n);
    CHECK(hipMemset(gm, 0, 4));
    CHECK(hipDeviceSynchronize());
    sono_engine *e = nullptr;
    if (create(N, "hann", 0, 0.0f, 0, &e)) { printf("create: %s\n", last_error()); return 1; }
    hipStream_t s; CHECK(hipStreamCreate(&s));
    sono_batch b{x, 1, n, n};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // warm-up launches, each waited for: an engine that adapts to what it measured on the launch before gets to do so
    for (int i = 0; i < 4; ++i) {
        if (u8dev(e, &b, V, 20.0f, 80.0f, out, pb, pm, gm, s)) { printf("launch: %s\n", last_error()); return 1; }
        CHECK(hipStreamSynchronize(s));
    }
    std::vector<float> ms(reps);
    typedef void (*stamps_fn0)(unsigned long long *);
    stamps_fn0 st0 = (stamps_fn0)dlsym(h, "sono_debug_stamps");
    typedef void (*wg_fn)(unsigned long long *, int);
    wg_fn wg = (wg_fn)dlsym(h, "sono_debug_wg_ends");
    std::vector<unsigned long long> wgv(4 * 2048);
    for (int i = 0; i < reps; ++i) {
        if (st0 && i == reps - 1) { unsigned long long junk[16]; st0(junk); if (wg) wg(wgv.data(), 2048); }
        CHECK(hipEventRecord(e0, s));
        u8dev(e, &b, V, 20.0f, 80.0f, out, pb, pm, gm, s);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    typedef void (*stamps_fn)(unsigned long long *);
    if (stamps_fn st = (stamps_fn)dlsym(h, "sono_debug_stamps")) {
        unsigned long long v[16];
        st(v);
        const double waves = (double)v[10], launches = 1;
        printf("stamps over %g wave-launches: clock %.3f GHz; per wave per launch cycles: load %.0f pass1 %.0f exchange %.0f pass2+power %.0f epilogue %.0f total %.0f\n",
               waves, (double)v[8] / (double)v[9] * 0.1, v[0] / waves, v[1] / waves, v[2] / waves, v[3] / waves, v[4] / waves, v[8] / waves);
        printf("  last launch: longest wave %.0f cycles; first start -> last start %.1f us, first end -> last end %.1f us, first start -> last end %.1f us\n",
               (double)v[11], (v[14] - v[13]) * 0.01, (v[12] - v[15]) * 0.01, (v[12] - v[13]) * 0.01);
        (void)launches;
    }
    if (wg) {
        // the last launch, workgroup by workgroup: when each ended relative to the first start, by XCC
        wg(wgv.data(), 2048);
        unsigned long long t0 = ~0ull;
        int groups = 0;
        for (int i = 0; i < 2048; ++i) if (wgv[4 * i + 1]) { ++groups; if (wgv[4 * i] < t0) t0 = wgv[4 * i]; }
        printf("workgroups %d; per XCC: groups, tiles, end of the earliest / median / latest workgroup (us after the first start)\n", groups);
        for (unsigned long long x = 0; x < 16; ++x) {
            std::vector<double> ends; unsigned long long tiles = 0;
            for (int i = 0; i < 2048; ++i) if (wgv[4 * i + 1] && wgv[4 * i + 2] == x) { ends.push_back((wgv[4 * i + 1] - t0) * 0.01); tiles += wgv[4 * i + 3]; }
            if (ends.empty()) continue;
            std::sort(ends.begin(), ends.end());
            printf("  xcc %llu: %zu groups %llu tiles  ends %.1f / %.1f / %.1f\n", x, ends.size(), tiles, ends.front(), ends[ends.size() / 2], ends.back());
        }
    }
    float best = ms[0], sum = 0;
    for (float m : ms) { best = m < best ? m : best; sum += m; }
    uint32_t g; CHECK(hipMemcpy(&g, gm, 4, hipMemcpyDeviceToHost));
    std::vector<uint8_t> probe(N / 2);
    CHECK(hipMemcpy(probe.data(), out + (F / 2) * (N / 2), N / 2, hipMemcpyDeviceToHost));
    unsigned int cs = 0; for (uint8_t c : probe) cs = cs * 31 + c;
    printf("%s N=%u V=%u frames=%zu  avg %.3f ms  best %.3f ms  (%.2f Gframes/s best)  gmax_bits=%08x probe=%08x\n",
           argv[1], N, V, F, sum / reps, best, F / best / 1e6, g, cs);
    return 0;
}
