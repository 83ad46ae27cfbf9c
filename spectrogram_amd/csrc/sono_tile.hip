// sono_tile.hip -- register-tiled STFT kernel (placeholder: no size enabled yet,
// every launch takes the generic path).
#include "sono_tile.h"

namespace sono {

hipError_t tile_tables_build(unsigned int N, const float *, TileTables &t) { t.N = N; return hipSuccess; }
void tile_tables_release(TileTables &t) { if (t.d_blob) (void)hipFree(t.d_blob); t.d_blob = nullptr; }
bool tile_supports(int, const StftArgs &, const TileTables &) { return false; }
hipError_t launch_stft_tile(int, const StftArgs &, const TileTables &, hipStream_t) { return hipErrorNotSupported; }

}  // namespace sono
