// sono_tile.h -- the register-tiled STFT kernel (sono_tile.hip): interface.
#pragma once
#include <hip/hip_runtime.h>

#include "sono_core.h"

namespace sono {

// Device tables private to the tiled kernel (none beyond the engine's common
// tables at present; kept as a hook for size-specific tables).
struct TileTables {
    unsigned int N = 0;
    void *d_blob = nullptr;
};

hipError_t tile_tables_build(unsigned int N, const float *window, TileTables &t);
void tile_tables_release(TileTables &t);

// True when the tiled kernel can run this launch: fft size 256/512/1024, mode
// MAG or U8, and 8-byte-aligned frame starts (even hop, even stream stride,
// 8-byte-aligned sample pointer) because lanes load float2 pairs.
bool tile_supports(int mode, const StftArgs &a, const TileTables &t);

hipError_t launch_stft_tile(int mode, const StftArgs &a, const TileTables &t, hipStream_t stream);

}  // namespace sono
