// Host emulation of one wave of the register-tiled STFT kernel
// (sono_tile_core.h): the phases run lane by lane against a plain array that
// stands in for the wave's LDS image.  Test-only.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../spectrogram_amd/csrc/sono_tables.h"
#include "../../spectrogram_amd/csrc/sono_tile_core.h"

using namespace sono;

template <int LOG_M1, int MODE>
static void run_tiles(const StftArgs &a, const std::vector<float2> &tw)
{
    using G = tile::Geo<LOG_M1>;
    tile::TileIter it;
    const unsigned long long tps = (a.frames_per_stream + G::FPW - 1) / G::FPW;
    it.tiles_per_stream = (unsigned int)tps;
    it.n_tiles = a.n_streams * tps;
    std::vector<float2> img(tile::kWaveLdsComplex);
    unsigned long long peak[16];
    for (auto &p : peak) p = peak_key(0.0f, 0u);
    auto max64 = [](unsigned long long *p, unsigned long long v) { if (v > *p) *p = v; };
    std::vector<tile::TileLane<LOG_M1, MODE>> lanes;
    for (int l = 0; l < 64; ++l) lanes.emplace_back(a, l);
    static float re[64][32], im[64][32];
    for (unsigned long long t = 0; t < it.n_tiles; ++t) {
        for (int l = 0; l < 64; ++l) lanes[l].load(t, it);
        for (int l = 0; l < 64; ++l) lanes[l].run_phase1(img.data());
        for (int l = 0; l < 64; ++l) lanes[l].run_phase2_load(img.data(), re[l], im[l]);
        for (int l = 0; l < 64; ++l) lanes[l].run_phase2_fft_store(re[l], im[l], img.data());
        for (int l = 0; l < 64; ++l) lanes[l].run_phase3(t, it, img.data(), tw.data(), &peak[l >> LOG_M1], max64);
        for (int l = 0; l < 64; ++l) lanes[l].run_phase4(t, it, &peak[l >> LOG_M1]);
    }
    if (MODE == MODE_U8 && a.gmax_bits) {
        float m = 0.0f;
        for (int l = 0; l < 64; ++l) m = lanes[l].run_max > m ? lanes[l].run_max : m;
        const float mag = fast_sqrt(m) * a.inv_n;
        unsigned int bits;
        std::memcpy(&bits, &mag, 4);
        if (bits > *a.gmax_bits) *a.gmax_bits = bits;
    }
}

extern "C" int emul_tile(int mode, const float *x, unsigned long long n_streams,
                         unsigned long long stream_stride, unsigned long long frames_per_stream,
                         unsigned int hop, unsigned int N, const float *window, float gain_db,
                         float range_db, float *out_mag, uint8_t *out_u8, uint16_t *peak_bin,
                         float *peak_mag, unsigned int *gmax_bits, float *out_power)
{
    (void)out_power;
    if (N != 1024 && N != 512 && N != 256) return 1;
    if (mode != MODE_MAG && mode != MODE_U8) return 1;
    std::vector<float2> twh = make_twiddles(N / 2, N / 2), twf = make_twiddles(N, N / 2 + 1);
    StftArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.n_streams = n_streams; a.stream_stride = stream_stride;
    a.frames_per_stream = frames_per_stream; a.hop = hop; a.N = N; a.window = window;
    a.tw_half = twh.data(); a.tw_full = twf.data();
    a.out_mag = out_mag; a.out_u8 = out_u8; a.peak_bin = peak_bin; a.peak_mag = peak_mag;
    a.gmax_bits = gmax_bits;
    a.inv_n = 1.0f / (float)N;
    const Quant q = make_quant(N, gain_db, range_db);
    a.q_a = q.a; a.q_c = q.c; a.q_floor = q.floor;
#define RUN(L) do { if (mode == MODE_MAG) run_tiles<L, MODE_MAG>(a, twf); else run_tiles<L, MODE_U8>(a, twf); } while (0)
    if (N == 1024) RUN(4); else if (N == 512) RUN(3); else RUN(2);
#undef RUN
    return 0;
}
