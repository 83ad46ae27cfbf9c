// sono_tile.hip -- host side of the register-tiled STFT kernel: twiddle tables, which launches it
// serves, dispatch to the per-mode translation units.
#include "sono_tile.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sono_tables.h"
#include "sono_tile_core.h"

namespace sono {

hipError_t launch_tile_u8(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream);
hipError_t launch_tile_mag(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream);
hipError_t launch_tile_band(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream);
hipError_t launch_tile_survey(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream);
hipError_t launch_tile_bank(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream);

namespace {


hipError_t launch_any(int mode, const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    return mode == MODE_BANK ? launch_tile_bank(a, t, plan, stream)
         : mode == MODE_SURVEY ? launch_tile_survey(a, t, plan, stream)
         : mode == MODE_MAG ? launch_tile_mag(a, t, plan, stream)
         : mode == MODE_BAND ? launch_tile_band(a, t, plan, stream)
                             : launch_tile_u8(a, t, plan, stream);
}

}  // namespace

hipError_t tile_tables_build(unsigned int N, TileTables &t)
{
    t.N = N;
    t.d_blob = nullptr;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 == hipSuccess) e0 = hipDeviceGetAttribute(&t.cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e0 != hipSuccess) return e0;
    if (N != 4096 && N != 2048 && N != 1024 && N != 512 && N != 256) return hipSuccess;
    const std::vector<float2> rows = make_tile_twiddles(N, 1u << (N == 256 ? 5 : tile_log_r2(N)));
    hipError_t e = hipMalloc(&t.d_blob, rows.size() * sizeof(float2));
    if (e != hipSuccess) { t.d_blob = nullptr; return e; }
    const size_t fb_bytes = (size_t)TileBalance::kMaxGroups * tile::kFeedbackWords * sizeof(unsigned long long);
    e = hipHostMalloc((void **)&t.balance.feedback, fb_bytes, hipHostMallocDefault);
    if (e != hipSuccess) { t.balance.feedback = nullptr; return e; }
    std::memset(t.balance.feedback, 0, fb_bytes);
    return hipMemcpy(t.d_blob, rows.data(), rows.size() * sizeof(float2), hipMemcpyHostToDevice);
}

void tile_tables_release(TileTables &t)
{
    if (t.d_blob) (void)hipFree(t.d_blob);
    if (t.balance.feedback) (void)hipHostFree(t.balance.feedback);
    t.d_blob = nullptr;
    t.balance.feedback = nullptr;
}

void tile_balance_plan(TileBalance *b, unsigned long long blocks, unsigned long long resident, unsigned int waves_per_block, tile::TileIter &it)
{
    it.parts = 0;
    it.feedback = nullptr;
    it.stamp = 0;
    for (unsigned int &v : it.part_begin) v = 0;
    // Large launches only: the whole chip is busy, the workgroups divide evenly among the parts, every wave has
    // at least kMinTilesPerWave tiles, tile numbers fit 32 bits.
    constexpr unsigned long long kMinTilesPerWave = 64;
    constexpr unsigned int P = tile::kTileParts;
    if (!b || !b->feedback || b->fixed) return;
    // Workgroup g runs on XCD g % 8 only when the device is one whole chip of eight XCDs of 32 CUs: 256 workgroups
    // (one per CU) or 512 (two).  Anything else -- a partition of the chip, a part with another number of XCDs --
    // keeps equal ranges.
    if (blocks != resident || (blocks != 256 && blocks != 512) || blocks > TileBalance::kMaxGroups) return;
    std::lock_guard<std::mutex> lock(b->mu);
    if (it.n_tiles >= (1ull << 32) || it.n_tiles < blocks * waves_per_block * kMinTilesPerWave) return;
    // Reports of the last large launch that has finished: usable when every workgroup's record carries one stamp,
    // newer than the one taken up last (a launch still running, or records of two launches mixed, are left alone).
    // The records are read while a later launch may be rewriting them: a writer voids a record's stamp, writes the
    // fields, then the new stamp, and the reader takes a record only when its stamp is the expected one before AND
    // after the fields were read.
    const volatile unsigned long long *fb = b->feedback;
    const unsigned long long s0 = fb[3];
    bool whole = s0 != 0 && s0 > b->seen && s0 <= b->stamp;
    unsigned long long first[P], last[P], tiles[P];
    for (unsigned int p = 0; p < P; ++p) { first[p] = ~0ull; last[p] = 0; tiles[p] = 0; }
    for (unsigned long long g = 0; whole && g < blocks; ++g) {
        const volatile unsigned long long *r = fb + g * tile::kFeedbackWords;
        if (r[3] != s0) { whole = false; break; }
        std::atomic_thread_fence(std::memory_order_acquire);
        const unsigned long long t0 = r[0], t1 = r[1], n = r[2];
        std::atomic_thread_fence(std::memory_order_acquire);
        if (r[3] != s0 || t1 <= t0) { whole = false; break; }
        const unsigned int p = (unsigned int)(g % P);
        if (t0 < first[p]) first[p] = t0;
        if (t1 > last[p]) last[p] = t1;
        tiles[p] = n;
    }
    if (whole) {
        double speed[P], sum = 0.0;
        for (unsigned int p = 0; p < P; ++p) {
            speed[p] = tiles[p] > 0 ? (double)tiles[p] / (double)(last[p] - first[p]) : 0.0;
            sum += speed[p];
        }
        bool sane = sum > 0.0;
        for (unsigned int p = 0; sane && p < P; ++p) sane = speed[p] * P > 0.5 * sum && speed[p] * P < 2.0 * sum;
        if (sane) {
            // half way from the old cut to the one these speeds ask for, no part's range more than a fifth off the mean
            double total = 0.0;
            for (unsigned int p = 0; p < P; ++p) {
                double s = 0.5 * b->share[p] + 0.5 * (speed[p] * P / sum);
                s = s < 0.8 ? 0.8 : s > 1.2 ? 1.2 : s;
                b->share[p] = s;
                total += s;
            }
            for (unsigned int p = 0; p < P; ++p) b->share[p] *= P / total;
        }
        b->seen = s0;
        if (b->debug) {
            std::fprintf(stderr, "tile balance, launch %llu: part times (us)", s0);
            for (unsigned int p = 0; p < P; ++p) std::fprintf(stderr, " %.1f", (double)(last[p] - first[p]) * 0.01);
            std::fprintf(stderr, "; next shares");
            for (unsigned int p = 0; p < P; ++p) std::fprintf(stderr, " %.4f", b->share[p]);
            std::fprintf(stderr, "\n");
        }
    }
    double cum = 0.0;
    it.part_begin[0] = 0;
    for (unsigned int p = 0; p + 1 < P; ++p) {
        cum += b->share[p];
        unsigned long long edge = (unsigned long long)((double)it.n_tiles * cum / P + 0.5);
        if (edge < it.part_begin[p]) edge = it.part_begin[p];
        if (edge > it.n_tiles) edge = it.n_tiles;
        it.part_begin[p + 1] = (unsigned int)edge;
    }
    it.part_begin[P] = (unsigned int)it.n_tiles;
    it.parts = P;
    it.feedback = b->feedback;
    it.stamp = ++b->stamp;
}

bool tile_supports(int mode, const StftArgs &a, const TileTables &t)
{
    if (mode != MODE_MAG && mode != MODE_U8 && mode != MODE_BAND && mode != MODE_SURVEY && mode != MODE_BANK) return false;
    if (mode == MODE_BANK) {        // fft 1024, a tap list that fits the workgroup's LDS, rows written as dwords
        if (a.N != 1024 || a.x16 || !a.out_u8 || !a.bank_tap_bin || !a.bank_tap_w) return false;
        if (a.bank_filters == 0 || a.bank_filters % 4 != 0) return false;
        if (a.bank_max_taps == 0 || (unsigned long long)a.bank_max_taps * a.bank_filters > kBankLdsTaps) return false;
        if (reinterpret_cast<uintptr_t>(a.out_u8) & 3u) return false;
    }
    if (mode == MODE_SURVEY && (a.N != 1024 || a.hop * 4u != a.N)) return false;      // the one form the fused survey pass is built in
    if (a.x16 && (mode == MODE_MAG || a.x16_channels == 0 || a.x16_channel >= a.x16_channels)) return false;
    if ((mode == MODE_BAND || mode == MODE_SURVEY) && (!a.band_active || a.band_hi >= a.N / 2 || a.band_lo > a.band_hi)) return false;
    if (a.N != 4096 && a.N != 2048 && a.N != 1024 && a.N != 512 && a.N != 256) return false;
    if (!t.d_blob || t.N != a.N) return false;
    const unsigned long long tps = (a.frames_per_stream + 1) / 2;
    if (tps >= 0xFFFFFFFFull) return false;
    if (mode == MODE_MAG && (reinterpret_cast<uintptr_t>(a.out_mag) & 15u)) return false;
    if (mode == MODE_U8 && (reinterpret_cast<uintptr_t>(a.out_u8) & 3u)) return false;
    return true;
}

hipError_t launch_stft_tile(int mode, const StftArgs &a, const TileTables &t, hipStream_t stream)
{
    return launch_any(mode, a, t, nullptr, stream);
}

unsigned int tile_frames_per_wave(unsigned int N)
{
    switch (N) {
    case 4096: return tile::Geo<6, 6>::FPW;
    case 2048: return tile::Geo<11 - tile_log_r2(2048), tile_log_r2(2048)>::FPW;
    case 1024: return tile::Geo<10 - tile_log_r2(1024), tile_log_r2(1024)>::FPW;
    case 512: return tile::Geo<9 - tile_log_r2(512), tile_log_r2(512)>::FPW;
    case 256: return tile::Geo<3, 5>::FPW;
    default: return 0;
    }
}

static_assert(sizeof(RaggedStreamEntry) == sizeof(tile::RaggedStream), "host and kernel views of a stream entry differ");

hipError_t launch_stft_tile_ragged(int mode, const StftArgs &a, const TileTables &t, const RaggedPlan &plan,
                                   hipStream_t stream)
{
    return launch_any(mode, a, t, &plan, stream);
}

}  // namespace sono

// Test hook (not part of the C ABI of include/sonostft.h): runs tile_balance_plan on records the caller supplies, so that
// the host logic -- which reports are taken up, how the cut moves, that the ranges tile the launch -- can be checked
// without a GPU.  records: blocks x 4 words {start, end, tiles, stamp} as the kernel leaves them; share: 8 values in and
// out; stamp / seen: the engine's counters in and out; part_begin: 9 values out.  Returns the number of parts (0: equal ranges).
extern "C" unsigned int sono_debug_tile_balance(const unsigned long long *records, unsigned int blocks, unsigned int resident,
                                                unsigned int waves_per_block, unsigned long long n_tiles, double *share,
                                                unsigned long long *stamp, unsigned long long *seen, unsigned int *part_begin)
{
    std::vector<unsigned long long> fb((size_t)sono::TileBalance::kMaxGroups * sono::tile::kFeedbackWords, 0ull);
    if (blocks <= sono::TileBalance::kMaxGroups)
        std::memcpy(fb.data(), records, (size_t)blocks * sono::tile::kFeedbackWords * sizeof(unsigned long long));
    sono::TileBalance b;
    b.feedback = fb.data();
    const char *fixed = std::getenv("SONO_FIXED_TILES");      // this hook only: an engine takes the switch as an option
    b.fixed = fixed && fixed[0] && fixed[0] != '0';
    for (int p = 0; p < 8; ++p) b.share[p] = share[p];
    b.stamp = *stamp;
    b.seen = *seen;
    sono::tile::TileIter it;
    it.n_tiles = n_tiles;
    sono::tile_balance_plan(&b, blocks, resident, waves_per_block, it);
    for (int p = 0; p < 8; ++p) share[p] = b.share[p];
    for (int p = 0; p < 9; ++p) part_begin[p] = it.part_begin[p];
    *stamp = b.stamp;
    *seen = b.seen;
    return it.parts;
}

