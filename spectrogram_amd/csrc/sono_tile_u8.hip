// sono_tile_u8.hip -- the MODE_U8 instantiations of the tiled STFT kernel (sono_tile_kernel.h).
#include "sono_tile_kernel.h"

namespace sono {

hipError_t launch_tile_u8(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    return launch_mode<MODE_U8>(a, t, plan, stream);
}

#ifdef SONO_EXP_STAMPS
extern "C" void sono_debug_stamps(unsigned long long *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
    unsigned long long z[16] = {0};
    z[13] = z[15] = ~0ull;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
}
// Per workgroup of the launches since the last call: {first start, last end (100 MHz ticks), XCC id, tiles done}; then cleared.
extern "C" void sono_debug_wg_ends(unsigned long long *out, int groups)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_ends), sizeof(unsigned long long) * 4 * groups);
    static unsigned long long z[4 * 2048];
    for (int i = 0; i < 2048; ++i) { z[4 * i] = ~0ull; z[4 * i + 1] = z[4 * i + 2] = z[4 * i + 3] = 0; }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_ends), z, sizeof(z));
}
#endif

}  // namespace sono
