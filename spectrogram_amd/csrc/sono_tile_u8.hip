// sono_tile_u8.hip -- the MODE_U8 instantiations of the tiled STFT kernel (sono_tile_kernel.h).
#include "sono_tile_kernel.h"

namespace sono {

hipError_t launch_tile_u8(const StftArgs &a, const float2 *tw_rows, const RaggedPlan *plan, hipStream_t stream)
{
    return launch_mode<MODE_U8>(a, tw_rows, plan, stream);
}

#ifdef SONO_EXP_STAMPS
extern "C" void sono_debug_stamps(unsigned long long *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
    unsigned long long z[16] = {0};
    z[13] = z[15] = ~0ull;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
}
#endif

}  // namespace sono
