// sono_tile.h -- the register-tiled STFT kernel (sono_tile.hip): interface.
#pragma once
#include <hip/hip_runtime.h>

#include "sono_core.h"

namespace sono {

// Device tables private to the tiled kernel (built once per engine).
struct TileTables {
    unsigned int N = 0;
    void *d_blob = nullptr;      // one allocation holding every table below
};

hipError_t tile_tables_build(unsigned int N, const float *window, TileTables &t);
void tile_tables_release(TileTables &t);

// True when the tiled kernel can run this launch (fft size, mode, alignment).
bool tile_supports(int mode, const StftArgs &a, const TileTables &t);

hipError_t launch_stft_tile(int mode, const StftArgs &a, const TileTables &t, hipStream_t stream);

}  // namespace sono
