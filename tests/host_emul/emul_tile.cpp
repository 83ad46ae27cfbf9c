// Host emulation of one wave of the register-tiled STFT kernel
// (sono_tile_core.h): the passes run lane by lane against a plain array that
// stands in for the wave's LDS region.  Test-only.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../spectrogram_amd/csrc/sono_tables.h"
#include "../../spectrogram_amd/csrc/sono_tile_core.h"

using namespace sono;

static const BankTap *g_bank_taps = nullptr;      // MODE_BANK: the tap list as the kernel holds it in LDS
static int g_ragged = 0;      // 1 = walk the tiles through the ragged-batch tables
extern "C" void emul_tile_set_ragged(int v) { g_ragged = v; }

template <int LOG_R1, int LOG_R2, int MODE, int HOPQ>
static void run_tiles(const StftArgs &a)
{
    using G = tile::Geo<LOG_R1, LOG_R2>;
    constexpr int R2 = G::R2, HW = 64 / R2;      // HW = groups of R2 lanes that share a pass-1 maximum
    if (a.N != (unsigned int)G::N) return;
    tile::TileIter it;
    const unsigned long long tps = (a.frames_per_stream + G::FPW - 1) / G::FPW;
    it.tiles_per_stream = (unsigned int)tps;
    it.n_tiles = a.n_streams * tps;
    it.streams = nullptr;
    it.tile_stream = nullptr;
    // ragged form of the same batch (every stream gets its own table entry): must give identical results
    std::vector<tile::RaggedStream> rs;
    std::vector<unsigned int> ts;
    if (g_ragged) {
        for (unsigned long long s = 0; s < a.n_streams; ++s) {
            rs.push_back({s * a.stream_stride, a.frames_per_stream, s * a.frames_per_stream, s * tps});
            for (unsigned long long j = 0; j < tps; ++j) ts.push_back((unsigned int)s);
        }
        it.streams = rs.data();
        it.tile_stream = ts.data();
    }
    const std::vector<float2> tw = make_tile_twiddles(a.N, R2);
    alignas(16) static float img[G::REGION];
    std::vector<tile::TileLane<LOG_R1, LOG_R2, MODE, HOPQ>> lanes;
    for (int l = 0; l < 64; ++l) lanes.emplace_back(a, l);
    static float rr[64][R2], ri[64][R2], re[64][R2], im[64][R2], ya[64][R2], yb[64][R2], ma[64][G::JPL], mb[64][G::JPL];
    float meta[tile::kMetaFloats];
    for (unsigned long long t = 0; t < it.n_tiles; ++t) {
        const tile::TileRef tr = g_ragged ? tile::locate<true, G::FPW>(a, it, t) : tile::locate<false, G::FPW>(a, it, t);
        for (int l = 0; l < 64; ++l) lanes[l].load(tr);
        if constexpr (G::SPLIT) {
            // one frame per lane: frame A in lanes 0..31, frame B in lanes 32..63; what crosses between the halves
            // in the kernel (v_permlane32_swap) is read from the other lane's arrays here
            static float y[64][G::R1], p[64];
            for (int l = 0; l < 64; ++l) lanes[l].pass1_dft_split(a.window, y[l], p[l]);
            float pa = 0.0f, pb = 0.0f;
            for (int l = 0; l < 32; ++l) { pa = std::fmax(pa, p[l]); pb = std::fmax(pb, p[32 + l]); }
            for (int l = 0; l < 64; ++l) {
                const int c = l & 31;
                lanes[l].pass1_rows_split(tw.data(), y[l], y[c][0], y[32 + c][0], y[c][G::HALF], y[32 + c][G::HALF], pa, pb, rr[l], ri[l], meta);
            }
        } else {
        for (int l = 0; l < 64; ++l) lanes[l].pass1_dft(a.window, ya[l], yb[l], ma[l], mb[l]);
        for (int h = 0; h < HW; ++h)
            for (int jj = 0; jj < G::JPL; ++jj) {
                float va = 0.0f, vb = 0.0f;
                for (int l = R2 * h; l < R2 * h + R2; ++l) { va = std::fmax(va, ma[l][jj]); vb = std::fmax(vb, mb[l][jj]); }
                for (int l = R2 * h; l < R2 * h + R2; ++l) { ma[l][jj] = va; mb[l][jj] = vb; }
            }
        for (int l = 0; l < 64; ++l) lanes[l].pass1_rows(tw.data(), ya[l], yb[l], ma[l], mb[l], rr[l], ri[l], meta);
        }
        for (int l = 0; l < 64; ++l) lanes[l].exchange_store(rr[l], img);
        for (int l = 0; l < 64; ++l) lanes[l].exchange_load(img, re[l]);
        for (int l = 0; l < 64; ++l) lanes[l].exchange_store(ri[l], img);
        for (int l = 0; l < 64; ++l) lanes[l].exchange_load(img, im[l]);
        for (int l = 0; l < 64; ++l) lanes[l].pass2_power(re[l], im[l], img);
        if (MODE == MODE_BAND) {
            double e[64], e2[64];
            for (int l = 0; l < 64; ++l) e[l] = lanes[l].band_sum(img, meta);
            for (int o = G::HALF / 2; o > 0; o >>= 1) {
                for (int l = 0; l < 64; ++l) e2[l] = e[l] + e[l ^ o];
                std::memcpy(e, e2, sizeof(e));
            }
            for (int l = 0; l < 64; ++l) lanes[l].band_emit(tr, e[l]);
            continue;
        }
        float bp[64], pmax[64];
        unsigned int bk[64], bin[64];
        for (int l = 0; l < 64; ++l) {
            if (MODE == MODE_U8 && a.q_floor > 0.0f) lanes[l].template epilogue<true>(tr, img, meta, bp[l], bk[l]);
            else lanes[l].template epilogue<false>(tr, img, meta, bp[l], bk[l]);
        }
        for (int l = 0; l < 64; ++l) {                      // over the lanes of each frame: largest power, then its lowest bin
            const int f0 = l & ~(G::HALF - 1);
            pmax[l] = bp[f0];
            for (int m = f0; m < f0 + G::HALF; ++m) pmax[l] = bp[m] > pmax[l] ? bp[m] : pmax[l];
            bin[l] = 0xFFFFFFFFu;
            for (int m = f0; m < f0 + G::HALF; ++m)
                if (bp[m] == pmax[l] && bk[m] < bin[l]) bin[l] = bk[m];
        }
        for (int l = 0; l < 64; ++l) lanes[l].features(tr, pmax[l], bin[l]);
        if constexpr (MODE == MODE_SURVEY) {                // the band test on the tile's even frames, R1 lanes per frame
            double e[64], e2[64];
            for (int l = 0; l < 64; ++l) e[l] = lanes[l].survey_band_sum(img, meta);
            for (int o = G::HALF; o > 0; o >>= 1) {
                for (int l = 0; l < 64; ++l) e2[l] = e[l] + e[l ^ o];
                std::memcpy(e, e2, sizeof(e));
            }
            for (int l = 0; l < 64; ++l) lanes[l].survey_band_emit(tr, e[l]);
        }
        if constexpr (MODE == MODE_BANK) {                  // the filter bank from the power image: magnitudes in place, then the rows
            for (int l = 0; l < 64; ++l) lanes[l].bank_magnitudes(img, meta);
            for (int l = 0; l < 64; ++l) lanes[l].bank_rows(tr, img, g_bank_taps);
        }
    }
    if ((MODE == MODE_U8 || MODE == MODE_SURVEY || MODE == MODE_BANK) && a.gmax_bits) {
        float m = 0.0f;
        for (int l = 0; l < 64; ++l) m = lanes[l].run_max > m ? lanes[l].run_max : m;
        const float mag = fast_sqrt(m) * a.inv_n;
        unsigned int bits;
        std::memcpy(&bits, &mag, 4);
        if (bits > *a.gmax_bits) *a.gmax_bits = bits;
    }
}

static int g_log_r2 = 0;      // 0 = the geometry the kernel launcher picks for N
extern "C" void emul_tile_set_log_r2(int v) { g_log_r2 = v; }

static int g_any_hop = 0;     // 1 = the any-hop form even where the launcher would pick the shared-load form
extern "C" void emul_tile_set_any_hop(int v) { g_any_hop = v; }

static unsigned int g_band_lo = 0, g_band_hi = 0;
static double g_band_target = 0.0;
static uint8_t *g_band_active = nullptr;
static double *g_band_energy = nullptr;
// Parameters of the next emul_tile call with mode 3 (band energy).
extern "C" void emul_tile_set_band(unsigned int lo, unsigned int hi, double target, uint8_t *active, double *energy)
{
    g_band_lo = lo; g_band_hi = hi; g_band_target = target; g_band_active = active; g_band_energy = energy;
}

static const float *g_bank = nullptr;
static size_t g_bank_len = 0, g_bank_filters = 0;
// The filter bank of the next emul_tile call with mode 5 (flat weights with row stride fft/2, lib.rs:317-326).
extern "C" void emul_tile_set_bank(const float *w, size_t len, size_t filters) { g_bank = w; g_bank_len = len; g_bank_filters = filters; }

extern "C" int emul_tile(int mode, const float *x, unsigned long long n_streams,
                         unsigned long long stream_stride, unsigned long long frames_per_stream,
                         unsigned int hop, unsigned int N, const float *window, float gain_db,
                         float range_db, float *out_mag, uint8_t *out_u8, uint16_t *peak_bin,
                         float *peak_mag, unsigned int *gmax_bits, float *out_power)
{
    (void)out_power;
    if (N != 4096 && N != 2048 && N != 1024 && N != 512 && N != 256) return 1;
    if (mode != MODE_MAG && mode != MODE_U8 && mode != MODE_BAND && mode != MODE_SURVEY && mode != MODE_BANK) return 1;
    if ((mode == MODE_SURVEY || mode == MODE_BANK) && (N != 1024 || (mode == MODE_SURVEY && hop * 4u != N))) return 1;
    StftArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.n_streams = n_streams; a.stream_stride = stream_stride;
    a.frames_per_stream = frames_per_stream; a.hop = hop; a.N = N; a.window = window;
    a.out_mag = out_mag; a.out_u8 = out_u8; a.peak_bin = peak_bin; a.peak_mag = peak_mag;
    a.gmax_bits = gmax_bits;
    a.inv_n = 1.0f / (float)N;
    a.band_lo = g_band_lo; a.band_hi = g_band_hi; a.band_target = g_band_target;
    a.band_active = g_band_active; a.band_energy = g_band_energy;
    const Quant q = make_quant(N, gain_db, range_db);
    a.q_a = q.a; a.q_c = q.c; a.q_floor = q.floor;
    std::vector<BankTap> taps;
    if (mode == MODE_BANK) {
        // the bank's tap list exactly as the library builds and stages it (bank_taps_host: sono_kernels.hip's bank_taps)
        if (!g_bank || g_bank_filters == 0 || g_bank_filters % 4) return 1;
        const size_t bins = N / 2;
        size_t most = 0;
        std::vector<std::vector<BankTap>> per(g_bank_filters);
        for (size_t i = 0; i < g_bank_filters; ++i) {
            for (size_t j = 0; j < bins && i * bins + j < g_bank_len; ++j)
                if (!(g_bank[i * bins + j] == 0.0f)) per[i].push_back(BankTap{g_bank[i * bins + j], (unsigned int)j});
            most = std::max(most, per[i].size());
        }
        if (most == 0 || most * g_bank_filters > kBankLdsTaps) return 1;
        taps.assign(most * g_bank_filters, BankTap{0.0f, 0u});
        for (size_t i = 0; i < g_bank_filters; ++i)
            for (size_t k = 0; k < per[i].size(); ++k) taps[k * g_bank_filters + i] = per[i][k];
        g_bank_taps = taps.data();
        a.bank_filters = (unsigned int)g_bank_filters;
        a.bank_max_taps = (unsigned int)most;
        const BankQuant bq = make_bank_quant(gain_db, range_db);
        a.bank_q_a = bq.a; a.bank_q_c = bq.c;
    }
    if (mode == MODE_SURVEY) { if (g_any_hop) return 1; run_tiles<5, 5, MODE_SURVEY, 1>(a); return 0; }
    if (mode == MODE_BANK) { if (!g_any_hop && hop * 4u == N) run_tiles<5, 5, MODE_BANK, 1>(a); else run_tiles<5, 5, MODE_BANK, 0>(a); return 0; }
    const bool quarter = !g_any_hop && hop * 4u == N, half = !g_any_hop && hop * 2u == N;
#define RUN(L1, L2) do { \
        if (mode == MODE_MAG) { if (quarter) run_tiles<L1, L2, MODE_MAG, 1>(a); else run_tiles<L1, L2, MODE_MAG, 0>(a); } \
        else if (mode == MODE_BAND) { if (half) run_tiles<L1, L2, MODE_BAND, 2>(a); else run_tiles<L1, L2, MODE_BAND, 0>(a); } \
        else { if (quarter) run_tiles<L1, L2, MODE_U8, 1>(a); else run_tiles<L1, L2, MODE_U8, 0>(a); } } while (0)
    if (g_log_r2 == 6 && N == 1024) RUN(4, 6);
    else if (g_log_r2 == 6 && N == 512) RUN(3, 6);
    else if (g_log_r2 == 5 && N == 2048) RUN(6, 5);
    else if (N == 4096) RUN(6, 6); else if (N == 2048) RUN(5, 6); else if (N == 1024) RUN(5, 5); else if (N == 512) RUN(4, 5); else RUN(3, 5);
#undef RUN
    return 0;
}

// The per-frame peak of ONE frame given as 512 values of |2X|^2 (fft 1024 geometry): the epilogue's search over the
// frame's 16 lanes and the reduction over them, as the kernel does it.  For the test of the tie rule.
extern "C" int emul_peak_of_powers(const float *p512, unsigned int *bin, float *mag)
{
    using G = tile::Geo<5, 5>;
    StftArgs a;
    std::memset(&a, 0, sizeof(a));
    a.N = 1024; a.hop = 256; a.frames_per_stream = 1; a.n_streams = 1; a.inv_n = 1.0f / 1024.0f;
    const Quant q = make_quant(1024, 20.0f, 80.0f);
    a.q_a = q.a; a.q_c = q.c; a.q_floor = q.floor;
    alignas(16) static float img[G::REGION];
    std::memset(img, 0, sizeof(img));
    for (int k = 0; k < G::BINS; ++k) img[k] = p512[k];                      // frame 0 of the power image
    float meta[tile::kMetaFloats];
    for (float &m : meta) m = 1.0f;
    tile::TileRef tr;
    std::memset(&tr, 0, sizeof(tr));
    tr.frames = 1; tr.rem = 1;
    float best = -1.0f;
    unsigned int best_bin = 0xFFFFFFFFu;
    for (int l = 0; l < G::HALF; ++l) {                                      // the 16 lanes of frame 0
        tile::TileLane<5, 5, MODE_U8, 1> lane(a, l);
        float bp; unsigned int bk;
        lane.template epilogue<false>(tr, img, meta, bp, bk);
        if (bp > best) { best = bp; best_bin = bk; }
        else if (bp == best && bk < best_bin) best_bin = bk;
    }
    *bin = best_bin;
    *mag = fast_sqrt(best) * a.inv_n;
    return 0;
}

// The quantiser on its own, for the test that pins its rounding rule.  emul_code_of(v): the code the kernels form
// for v = a * log2(p) + c0, the value the reference truncates (lib.rs:277-293): one conversion of v - 1/2, round
// to nearest even, saturating, NaN -> 0.  emul_quantise: the whole law from a power p = |2X|^2.
extern "C" unsigned int emul_code_of(float v) { return cvt_pk_u8<0>(v - 0.5f, 0u) & 0xFFu; }
extern "C" unsigned int emul_quantise(float p, unsigned int N, float gain_db, float range_db)
{
    const Quant q = make_quant(N, gain_db, range_db);
    return (q.floor > 0.0f ? quantise<true, 0>(p, q.a, q.c, q.floor) : quantise<false, 0>(p, q.a, q.c, q.floor)) & 0xFFu;
}
