// sono_tile_band.hip -- the MODE_BAND instantiations of the tiled STFT kernel (sono_tile_kernel.h).
#include "sono_tile_kernel.h"

namespace sono {

hipError_t launch_tile_band(const StftArgs &a, const float2 *tw_rows, const RaggedPlan *plan, TileBalance *balance, hipStream_t stream)
{
    return launch_mode<MODE_BAND>(a, tw_rows, plan, balance, stream);
}

}  // namespace sono
