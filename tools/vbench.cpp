// vbench.cpp -- times ONE named case of the library's device entry points on device-resident
// synthetic input and prints a JSON line: average / best launch time (HIP events on the launch
// stream), the case's algorithmic HBM bytes, and the rate they give.  Built against the in-tree
// library; run under rocprofv3 (tools/profile_variants.sh) for the per-kernel picture.
//   vbench <case> [reps] [seconds]     vbench list
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/vbench tools/vbench.cpp -Lspectrogram_amd/lib -lsonostft \
//        -Wl,-rpath,'$ORIGIN/../spectrogram_amd/lib'
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../include/sonostft.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define SONO(x) do { int r_ = (x); if (r_ != SONO_OK) { printf("library status %d at %s:%d: %s\n", r_, __FILE__, __LINE__, sono_last_error()); exit(1); } } while (0)

// Low noise with a short tone burst every 100 ms: most frames quiet, some loud, like a recording.
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
__global__ void to_pcm16(const float *x, int16_t *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (int16_t)(x[i] * 32767.0f);
}

template <typename T> static T *dalloc(size_t count)
{
    T *p = nullptr;
    CHECK(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
    return p;
}

struct Case {
    std::string name, what;
    double bytes = 0;              // algorithmic HBM bytes of one launch
    size_t units = 0;              // frames (or elements) one launch processes
    std::function<void()> launch;
};

static const double SR = 384000;

int main(int argc, char **argv)
{
    static const char *const names[] = {"u8_1024", "u8_4096_blackman95", "u8_2048", "u8_512", "u8_256", "u8_1024_pcm16",
                                        "features_1024", "band_1024_50", "mag_1024", "mag_1024_hop32", "filterbank_1024", "survey_1024_pcm16",
                                        "power_db", "db_max", "db_active", "frame_features", "zonal_floors", "band_active_2pass",
                                        "runs", "runs_static", "pcm_to_f32", "sustained_u8_1024"};
    if (argc < 2 || !strcmp(argv[1], "list")) {
        for (const char *n : names) printf("%s\n", n);
        return argc < 2 ? 2 : 0;
    }
    const std::string name = argv[1];
    int reps = argc > 2 ? atoi(argv[2]) : 10;
    const size_t seconds_arg = argc > 3 ? (size_t)atoi(argv[3]) : 0;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    sono_engine *e = nullptr;
    Case c;
    c.name = name;
    auto stft_case = [&](uint32_t N, const char *window, int has_alpha, float alpha, uint32_t V, size_t seconds, int mode) {
        // mode 0 u8 + features, 1 features only, 2 magnitudes, 3 band flags, 4 u8 from 16-bit PCM,
        // 5 filter bank (the application's mel bank) through the host-pointer call: only that form applies a loaded bank, so the
        // event time includes both PCIe copies and the kernels' own times have to be read from the profiler
        if (seconds_arg) seconds = seconds_arg;
        const size_t n = seconds * (size_t)SR, F = sono_frames_for(n, N, V), hop = N - V, bins = N / 2;
        float *x = dalloc<float>(n);
        fill_noise<<<4096, 256, 0, s>>>(x, n);
        SONO(sono_engine_create(N, window, has_alpha, alpha, -1, &e));
        // the application's bank (spectrogram.esm.js:415-418, 1197-1232): N/2 filters on the mel scale, two taps each,
        // rows of N/2 + 1 weights flattened -- which the engine then reads with stride N/2 (lib.rs:317-326)
        const uint32_t nf = bins;
        if (mode == 5) {
            const size_t rowlen = bins + 1;
            std::vector<float> w((size_t)nf * rowlen, 0.0f);
            const double mel_max = 2595.0 * std::log10(1.0 + (SR / 2) / 700.0), hz_per_bin = SR / N;
            for (uint32_t f = 0; f < nf; ++f) {
                const double hz = 700.0 * (std::pow(10.0, (double)f / nf * mel_max / 2595.0) - 1.0);
                const size_t o = (size_t)std::floor(hz / hz_per_bin);
                const double c = (hz - o * hz_per_bin) / hz_per_bin;
                if (o < rowlen) w[f * rowlen + o] = (float)(1 - c);
                if (o + 1 < rowlen) w[f * rowlen + o + 1] = (float)c;
            }
            SONO(sono_engine_load_filter_bank(e, w.data(), w.size(), nf));
        }
        uint8_t *out = mode == 0 || mode == 4 ? dalloc<uint8_t>(F * bins) : nullptr;
        float *x_host = nullptr;
        uint8_t *out_host = nullptr;
        if (mode == 5) {
            x_host = (float *)malloc(n * sizeof(float));
            out_host = (uint8_t *)malloc(F * nf);
            if (!x_host || !out_host) { printf("out of host memory\n"); exit(1); }
            CHECK(hipStreamSynchronize(s));
            CHECK(hipMemcpy(x_host, x, n * sizeof(float), hipMemcpyDeviceToHost));
        }
        float *mags = mode == 2 ? dalloc<float>(F * bins) : nullptr;
        uint8_t *active = mode == 3 || mode == 6 ? dalloc<uint8_t>(F) : nullptr;
        uint16_t *pb = dalloc<uint16_t>(F);
        float *pm = dalloc<float>(F);
        uint32_t *gm = dalloc<uint32_t>(1);
        CHECK(hipMemsetAsync(gm, 0, 4, s));
        int16_t *pcm = nullptr;
        if (mode == 4 || mode == 6) {
            pcm = dalloc<int16_t>(n);
            to_pcm16<<<4096, 256, 0, s>>>(x, pcm, n);
        }
        const sono_batch b{x, 1, n, n};
        const uint32_t lo = (uint32_t)std::floor(10000.0 / (SR / N)), hi = std::min<uint32_t>(bins - 1, (uint32_t)std::floor(128000.0 / (SR / N)));
        char buf[256];
        snprintf(buf, sizeof buf, "%zu s at %g Hz, FFT %u %s, noverlap %u (hop %zu), %zu frames", seconds, SR, N, window, V, hop, F);
        c.what = buf;
        c.units = F;
        switch (mode) {
        case 0: c.bytes = (double)F * (hop * 4 + bins + 6); c.launch = [=] { SONO(sono_stft_u8_dev(e, &b, V, 20.0f, 80.0f, out, pb, pm, gm, s)); }; break;
        case 1: c.bytes = (double)F * (hop * 4 + 6); c.launch = [=] { SONO(sono_stft_features_dev(e, &b, V, pb, pm, gm, s)); }; break;
        case 2: c.bytes = (double)F * (hop * 4 + bins * 4); c.launch = [=] { SONO(sono_stft_mag_f32_dev(e, &b, V, mags, s)); }; break;
        case 3: c.bytes = (double)F * (hop * 4 + 1);
                c.launch = [=] { SONO(sono_stft_band_active_dev(e, &b, V, lo, hi, 1e-3, active, nullptr, s)); }; break;
        case 4: c.bytes = (double)F * (hop * 2 + bins + 6);
                c.launch = [=] { SONO(sono_stft_u8_pcm16_dev(e, pcm, 1, n, n, 1, 0, V, 20.0f, 80.0f, out, pb, pm, gm, s)); }; break;
        case 6: c.bytes = (double)F * (hop * 2 + 6) + (double)((F + 1) / 2);      // survey pass, 16-bit samples: features + every second frame's flag
                c.launch = [=] { SONO(sono_stft_survey_pcm16_dev(e, pcm, 1, n, n, 1, 0, lo, hi, 1e-3, pb, pm, gm, active, nullptr, s)); }; break;
        case 5: c.bytes = (double)F * (hop * 4 + nf + 6);
                c.launch = [=] { size_t frames = 0; SONO(sono_stft_u8(e, x_host, n, V, 20.0f, 80.0f, out_host, F * nf, &frames)); }; break;
        }
    };
    // The detector's matrix passes on `frames` rows of the band 10..128 kHz of an FFT-1024 analysis at 384 kHz.
    auto detect_case = [&](int which) {
        const uint32_t N = 1024, bins = 512, V = 992;
        const size_t seconds = seconds_arg ? seconds_arg : 80;
        const size_t n = seconds * (size_t)SR, F = sono_frames_for(n, N, V);
        const uint32_t lo = 26, hi = 341, nb = hi - lo + 1;
        float *x = dalloc<float>(n);
        fill_noise<<<4096, 256, 0, s>>>(x, n);
        SONO(sono_engine_create(N, "hann", 0, 0.0f, -1, &e));
        float *mags = dalloc<float>(F * bins), *db = dalloc<float>(F * nb), *mx = dalloc<float>(1);
        uint8_t *active = dalloc<uint8_t>(F);
        const sono_batch b{x, 1, n, n};
        SONO(sono_stft_mag_f32_dev(e, &b, V, mags, s));
        SONO(sono_detect_power_db_dev(e, mags, F, bins, lo, nb, db, s));
        SONO(sono_detect_db_active_dev(e, db, F, nb, -60.0, active, s));
        char buf[256];
        snprintf(buf, sizeof buf, "%zu frames x %u bins (band 10-128 kHz of FFT 1024 at %g Hz)", F, nb, SR);
        c.what = buf;
        c.units = F;
        std::vector<float> fb(nb);
        for (uint32_t i = 0; i < nb; ++i) fb[i] = (float)((lo + i) * (SR / N));
        switch (which) {
        case 0: c.bytes = (double)F * nb * 8; c.launch = [=] { SONO(sono_detect_power_db_dev(e, mags, F, bins, lo, nb, db, s)); }; break;
        case 1: c.bytes = (double)F * nb * 4; c.launch = [=] { SONO(sono_detect_db_max_dev(e, db, F * nb, mx, s)); }; break;
        case 2: c.bytes = (double)F * (nb * 4 + 1); c.launch = [=] { SONO(sono_detect_db_active_dev(e, db, F, nb, -60.0, active, s)); }; break;
        case 3: {
            sono_frame_features *ff = dalloc<sono_frame_features>(F);
            c.bytes = (double)F * (nb * 4 + sizeof(sono_frame_features));
            c.launch = [=] { SONO(sono_detect_frame_features_dev(e, db, F, nb, 0, nb - 1, nb - 1, -50.0, 0, -50.0, ff, s)); };
            break;
        }
        case 4: {
            c.bytes = (double)F * nb * 4;
            c.launch = [=] {
                double floors[SONO_MAX_ZONES];
                size_t zones = 0;
                SONO(sono_zonal_noise_floors_dev(e, db, F, nb, fb.data(), 0, F - 1, floors, SONO_MAX_ZONES, &zones, s));
            };
            break;
        }
        case 5: c.bytes = (double)F * (nb * 4 + 1);
                c.launch = [=] { SONO(sono_detect_band_active_dev(e, mags, F, bins, lo, hi, 1e-3, active, nullptr, s)); }; break;
        case 6: case 7: {
            uint64_t *first = dalloc<uint64_t>(F / 2 + 1), *last = dalloc<uint64_t>(F / 2 + 1), *count = dalloc<uint64_t>(1);
            c.bytes = (double)F * 2;                                   // the flags, read by the counting and the writing pass
            if (which == 6) c.launch = [=] { SONO(sono_detect_runs_dev(e, active, F, first, last, count, s)); };
            else c.launch = [=] { SONO(sono_detect_runs_static_dev(e, active, F, first, last, count, s)); };
            break;
        }
        }
    };
    if (name == "u8_1024" || name == "sustained_u8_1024") stft_case(1024, "hann", 0, 0, 768, 3600, 0);
    else if (name == "u8_4096_blackman95") stft_case(4096, "blackman", 1, 0.16f, 3891, 600, 0);
    else if (name == "u8_2048") stft_case(2048, "hann", 0, 0, 1536, 3600, 0);
    else if (name == "u8_512") stft_case(512, "hann", 0, 0, 384, 3600, 0);
    else if (name == "u8_256") stft_case(256, "hann", 0, 0, 192, 1800, 0);
    else if (name == "u8_1024_pcm16") stft_case(1024, "hann", 0, 0, 768, 3600, 4);
    else if (name == "features_1024") stft_case(1024, "hann", 0, 0, 768, 3600, 1);
    else if (name == "band_1024_50") stft_case(1024, "hann", 0, 0, 512, 3600, 3);
    else if (name == "mag_1024") stft_case(1024, "hann", 0, 0, 768, 900, 2);
    else if (name == "mag_1024_hop32") stft_case(1024, "hann", 0, 0, 992, 120, 2);
    else if (name == "filterbank_1024") stft_case(1024, "hann", 0, 0, 768, 600, 5);
    else if (name == "survey_1024_pcm16") stft_case(1024, "hann", 0, 0, 768, 3600, 6);
    else if (name == "power_db") detect_case(0);
    else if (name == "db_max") detect_case(1);
    else if (name == "db_active") detect_case(2);
    else if (name == "frame_features") detect_case(3);
    else if (name == "zonal_floors") detect_case(4);
    else if (name == "band_active_2pass") detect_case(5);
    else if (name == "runs") detect_case(6);
    else if (name == "runs_static") detect_case(7);
    else if (name == "pcm_to_f32") {
        const size_t n = (seconds_arg ? seconds_arg : 3600) * (size_t)SR;
        float *x = dalloc<float>(n);
        int16_t *pcm = dalloc<int16_t>(n);
        fill_noise<<<4096, 256, 0, s>>>(x, n);
        to_pcm16<<<4096, 256, 0, s>>>(x, pcm, n);
        SONO(sono_engine_create(1024, "hann", 0, 0.0f, -1, &e));
        c.what = std::to_string(n) + " mono 16-bit samples";
        c.units = n;
        c.bytes = (double)n * 6;
        c.launch = [=] { SONO(sono_pcm_to_f32_dev(e, pcm, n, 1, 0, 16, 0, x, s)); };
    } else {
        printf("unknown case %s (vbench list)\n", name.c_str());
        return 2;
    }
    CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < 3; ++i) c.launch();
    CHECK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    if (name == "sustained_u8_1024") {
        // Back-to-back launches for at least `reps` seconds of wall time, no host synchronisation in between
        // except once per block of 50: the launch time per block shows whether the rate holds once the
        // card has warmed up and the clock has settled.
        const double want_s = reps > 0 ? reps : 6;
        const int block = 50;
        std::vector<float> per_block;
        const auto t0 = std::chrono::steady_clock::now();
        double elapsed = 0;
        while (elapsed < want_s) {
            CHECK(hipEventRecord(e0, s));
            for (int i = 0; i < block; ++i) c.launch();
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            per_block.push_back(ms / block);
            elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        float lo = per_block[0], hi = per_block[0];
        double sum = 0;
        for (float m : per_block) { lo = std::min(lo, m); hi = std::max(hi, m); sum += m; }
        printf("{\"case\": \"%s\", \"what\": \"%s\", \"wall_s\": %.2f, \"launches\": %zu, \"ms_per_launch_mean\": %.4f, "
               "\"ms_per_launch_first_block\": %.4f, \"ms_per_launch_last_block\": %.4f, \"ms_per_launch_min_block\": %.4f, "
               "\"ms_per_launch_max_block\": %.4f, \"block\": %d, \"per_block_ms\": [",
               name.c_str(), c.what.c_str(), elapsed, per_block.size() * block, sum / per_block.size(), per_block.front(), per_block.back(),
               lo, hi, block);
        for (size_t i = 0; i < per_block.size(); ++i) printf("%s%.4f", i ? ", " : "", per_block[i]);
        printf("]}\n");
        return 0;
    }
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0, s));
        c.launch();
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    float best = ms[0];
    double sum = 0;
    for (float m : ms) { best = std::min(best, m); sum += m; }
    const double avg = sum / reps;
    printf("{\"case\": \"%s\", \"what\": \"%s\", \"reps\": %d, \"ms_avg\": %.4f, \"ms_best\": %.4f, \"units\": %zu, "
           "\"algorithmic_bytes\": %.0f, \"GBps_avg\": %.1f, \"frac_of_8TBps\": %.4f, \"units_per_s\": %.4g}\n",
           name.c_str(), c.what.c_str(), reps, avg, best, c.units, c.bytes, c.bytes / avg / 1e6, c.bytes / avg / 1e6 / 8000.0,
           c.units / (avg * 1e-3));
    return 0;
}
