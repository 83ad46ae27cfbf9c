// sono_tile_band.hip -- the MODE_BAND instantiations of the tiled STFT kernel (sono_tile_kernel.h).
#include "sono_tile_kernel.h"

namespace sono {

hipError_t launch_tile_band(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    return launch_mode<MODE_BAND>(a, t, plan, stream);
}

}  // namespace sono
