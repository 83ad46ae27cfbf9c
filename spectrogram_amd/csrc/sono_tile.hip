// sono_tile.hip -- host side of the register-tiled STFT kernel: twiddle tables, which launches it
// serves, dispatch to the per-mode translation units.
#include "sono_tile.h"

#include <vector>

#include "sono_tables.h"
#include "sono_tile_core.h"

namespace sono {

hipError_t launch_tile_u8(const StftArgs &a, const float2 *tw_rows, const RaggedPlan *plan, hipStream_t stream);
hipError_t launch_tile_mag(const StftArgs &a, const float2 *tw_rows, const RaggedPlan *plan, hipStream_t stream);
hipError_t launch_tile_band(const StftArgs &a, const float2 *tw_rows, const RaggedPlan *plan, hipStream_t stream);

namespace {

#ifndef SONO_WIDE_FROM
#define SONO_WIDE_FROM 2048
#endif
constexpr int tile_log_r2(unsigned int N) { return N >= SONO_WIDE_FROM ? 6 : 5; }

hipError_t launch_any(int mode, const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    const float2 *tw_rows = (const float2 *)t.d_blob;
    return mode == MODE_MAG ? launch_tile_mag(a, tw_rows, plan, stream)
         : mode == MODE_BAND ? launch_tile_band(a, tw_rows, plan, stream)
                             : launch_tile_u8(a, tw_rows, plan, stream);
}

}  // namespace

hipError_t tile_tables_build(unsigned int N, TileTables &t)
{
    t.N = N;
    t.d_blob = nullptr;
    if (N != 4096 && N != 2048 && N != 1024 && N != 512 && N != 256) return hipSuccess;
    const std::vector<float2> rows = make_tile_twiddles(N, 1u << (N == 256 ? 5 : tile_log_r2(N)));
    hipError_t e = hipMalloc(&t.d_blob, rows.size() * sizeof(float2));
    if (e != hipSuccess) { t.d_blob = nullptr; return e; }
    return hipMemcpy(t.d_blob, rows.data(), rows.size() * sizeof(float2), hipMemcpyHostToDevice);
}

void tile_tables_release(TileTables &t) { if (t.d_blob) (void)hipFree(t.d_blob); t.d_blob = nullptr; }

bool tile_supports(int mode, const StftArgs &a, const TileTables &t)
{
    if (mode != MODE_MAG && mode != MODE_U8 && mode != MODE_BAND) return false;
    if (a.x16 && (mode == MODE_MAG || a.x16_channels == 0 || a.x16_channel >= a.x16_channels)) return false;
    if (mode == MODE_BAND && (!a.band_active || a.band_hi >= a.N / 2 || a.band_lo > a.band_hi)) return false;
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
    case 2048: return tile::Geo<5, 6>::FPW;
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
