// sono_tile_mag.hip -- the MODE_MAG instantiations of the tiled STFT kernel (sono_tile_kernel.h).
#include "sono_tile_kernel.h"

namespace sono {

hipError_t launch_tile_mag(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    return launch_mode<MODE_MAG>(a, t, plan, stream);
}

}  // namespace sono
