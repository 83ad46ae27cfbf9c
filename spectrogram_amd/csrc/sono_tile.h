// sono_tile.h -- the register-tiled STFT kernel (sono_tile.hip): interface.
#pragma once
#include <hip/hip_runtime.h>

#include "sono_core.h"

namespace sono {

// Device table private to the tiled kernel: the inter-pass twiddle rows of
// make_tile_twiddles(N) (empty for fft sizes the kernel does not cover).
struct TileTables {
    unsigned int N = 0;
    void *d_blob = nullptr;
};

hipError_t tile_tables_build(unsigned int N, TileTables &t);
void tile_tables_release(TileTables &t);

// True when the tiled kernel can run this launch: fft size 256/512/1024, mode
// MAG or U8, output rows aligned for its 16-byte (MAG) / 4-byte (U8) stores.
// Samples are read one float per lane, so any hop and sample alignment works.
bool tile_supports(int mode, const StftArgs &a, const TileTables &t);

hipError_t launch_stft_tile(int mode, const StftArgs &a, const TileTables &t, hipStream_t stream);

}  // namespace sono
