// sono_kernels.h -- launchers implemented in sono_kernels.hip / sono_tile.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "sono_core.h"

namespace sono {

// Generic LDS-staged path, any fft size 256..4096.  mode is a sono::Mode.
hipError_t launch_stft_generic(int mode, const StftArgs &a, hipStream_t stream);

hipError_t launch_power_reduce(const float *power, unsigned long long frames, int bins, unsigned int N,
                               float *out_db, hipStream_t stream);

// apply_filter_bank (lib.rs:310-334) over the bank's non-zero taps, which bank_taps() lists per filter with bins
// ascending (row stride `bins`, bounds-checked against bank_len like the reference's loop): count[i] taps for filter i,
// tap k of filter i at entry k * num_filters + i of bin / w.  tap_* are device copies, taps = entries of bin / w.
void bank_taps(const float *bank, size_t bank_len, size_t num_filters, size_t bins, std::vector<unsigned int> &count,
               std::vector<unsigned int> &bin, std::vector<float> &w);
hipError_t launch_filterbank_u8(const float *mags, unsigned long long frames, int bins, const unsigned int *tap_start,
                                const unsigned int *tap_bin, const float *tap_w, unsigned int taps, int num_filters,
                                float gain_db, float range_db, uint8_t *out_u8, uint16_t *peak_bin, float *peak_mag,
                                unsigned int *gmax_bits, hipStream_t stream);

}  // namespace sono
