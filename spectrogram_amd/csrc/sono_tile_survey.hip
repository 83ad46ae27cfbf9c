// sono_tile_survey.hip -- the MODE_SURVEY and MODE_BANK instantiations of the tiled STFT kernel (sono_tile_kernel.h):
// peak features at hop N/4 and the fast scan's band test on every second frame in one pass over the samples; u8 rows
// through a sparse filter bank applied in the epilogue.
#include "sono_tile_kernel.h"

namespace sono {

hipError_t launch_tile_survey(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    return launch_mode<MODE_SURVEY>(a, t, plan, stream);
}

hipError_t launch_tile_bank(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    return launch_mode<MODE_BANK>(a, t, plan, stream);
}

}  // namespace sono
