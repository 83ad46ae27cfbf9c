// sono_kernels.h -- launchers implemented in sono_kernels.hip / sono_tile.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "sono_core.h"

namespace sono {

// Generic LDS-staged path, any fft size 256..4096.  mode is a sono::Mode.
hipError_t launch_stft_generic(int mode, const StftArgs &a, hipStream_t stream);

hipError_t launch_power_reduce(const float *power, unsigned long long frames, int bins, unsigned int N,
                               float *out_db, hipStream_t stream);

hipError_t launch_filterbank_u8(const float *mags, unsigned long long frames, int bins, const float *bank,
                                unsigned long long bank_len, int num_filters, float gain_db, float range_db,
                                uint8_t *out_u8, uint16_t *peak_bin, float *peak_mag,
                                unsigned int *gmax_bits, hipStream_t stream);

}  // namespace sono
