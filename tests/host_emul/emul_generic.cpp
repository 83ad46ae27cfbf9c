// Host emulation of the generic STFT workgroup (sono_core.h GenericBlock), one
// barrier phase at a time with the 256 "threads" run in a loop.  Test-only: it
// lets the CPU suite check the kernel's index math and butterflies without a GPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../spectrogram_amd/csrc/sono_core.h"
#include "../../spectrogram_amd/csrc/sono_tables.h"

using namespace sono;

template <int MODE>
static void run_mode(StftArgs a)
{
    const int M = (int)(a.N >> 1);
    const int slab = M > 1024 ? M : 1024;
    const int fpb = slab / M;
    const unsigned long long total = a.n_streams * a.frames_per_stream;
    const unsigned long long blocks = (total + fpb - 1) / fpb;
    std::vector<float2> lds(2 * slab);
    long long off[GEN_MAX_FPB];
    unsigned long long peak[GEN_MAX_FPB];
    float dc[GEN_MAX_FPB];
    auto max64 = [](unsigned long long *p, unsigned long long v) { if (v > *p) *p = v; };
    auto max32 = [](unsigned int *p, unsigned int v) { if (v > *p) *p = v; };
    for (unsigned long long b = 0; b < blocks; ++b) {
        GenericShared sh{lds.data(), lds.data() + slab, off, peak, dc};
        GenericBlock<MODE> blk(a, sh, b);
        for (int t = 0; t < GEN_THREADS; ++t) blk.phase_setup(t);
        for (int t = 0; t < GEN_THREADS; ++t) blk.phase_load(t);
        if (MODE == MODE_POWER) {
            for (int t = 0; t < GEN_THREADS; ++t) blk.phase_dc_sum(t);
            for (int t = 0; t < GEN_THREADS; ++t) blk.phase_dc_sub(t);
        }
        float2 *src = sh.buf0, *dst = sh.buf1;
        int p = 1;
        for (; p * 4 <= M; p <<= 2) {
            for (int t = 0; t < GEN_THREADS; ++t) blk.phase_r4(t, p, src, dst);
            std::swap(src, dst);
        }
        if (p < M) {
            for (int t = 0; t < GEN_THREADS; ++t) blk.phase_r2(t, src, dst);
            std::swap(src, dst);
        }
        for (int t = 0; t < GEN_THREADS; ++t) blk.phase_epilogue(t, src, max64);
        if (MODE == MODE_U8)
            for (int t = 0; t < GEN_THREADS; ++t) blk.phase_features(t, max32);
    }
}

extern "C" int emul_generic(int mode, const float *x, unsigned long long n_streams,
                            unsigned long long stream_stride, unsigned long long frames_per_stream,
                            unsigned int hop, unsigned int N, const float *window, float gain_db,
                            float range_db, float *out_mag, uint8_t *out_u8, uint16_t *peak_bin,
                            float *peak_mag, unsigned int *gmax_bits, float *out_power)
{
    if (!supported_fft_size(N)) return 1;
    std::vector<float2> twh = make_twiddles(N / 2, N / 2), twf = make_twiddles(N, N / 2 + 1);
    StftArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.n_streams = n_streams; a.stream_stride = stream_stride;
    a.frames_per_stream = frames_per_stream; a.hop = hop; a.N = N; a.window = window;
    a.tw_half = twh.data(); a.tw_full = twf.data();
    a.out_mag = out_mag; a.out_u8 = out_u8; a.peak_bin = peak_bin; a.peak_mag = peak_mag;
    a.gmax_bits = gmax_bits; a.out_power = out_power;
    a.inv_n = 1.0f / (float)N;
    const Quant q = make_quant(N, gain_db, range_db);
    a.q_a = q.a; a.q_c = q.c; a.q_floor = q.floor;
    switch (mode) {
    case MODE_MAG: run_mode<MODE_MAG>(a); break;
    case MODE_U8: run_mode<MODE_U8>(a); break;
    case MODE_POWER: run_mode<MODE_POWER>(a); break;
    default: return 1;
    }
    return 0;
}
