// sono_host.h -- host-only parts of libsonostft (no GPU work): window tables,
// the waveform helpers and the power-spectrum peak search.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace sono {

// lib.rs:698-776; alpha is the Blackman parameter.
void host_make_window(const char *name, unsigned int size, float alpha, float *out);

// lib.rs:1031-1038: hop of compute_power_spectrum.
size_t host_power_hop(size_t fft_size, bool has_overlap, float overlap_percent);

// lib.rs:1132-1179
float host_find_peak_frequency(const float *spectrum, size_t len, uint32_t sample_rate,
                               size_t fft_size, float flow_hz, float fhigh_hz);

// lib.rs:794-824; returns the number of peaks written
size_t host_wave_peaks(const float *x, size_t n, size_t num_peaks, float *out);

// lib.rs:834-840
float host_find_global_max(const float *x, size_t n);

// lib.rs:904-955 over one channel; writes max(target_width, 1) floats
size_t host_peaks_in_range(const float *x, size_t data_len, size_t start_sample, size_t end_sample,
                           size_t target_width, float *out);

// lib.rs:494-616 given the frames' linear magnitudes (frames x bins): alternative
// dB law of lib.rs:659-672, optional filter bank with row stride bins+1
// (lib.rs:642-654), bilinear resample, colour lookup.
void host_render_image(const float *mags, size_t frames, size_t bins, const float *bank, size_t bank_len,
                       size_t num_filters, bool use_bank, const uint32_t *color_map, size_t width,
                       size_t height, float gain_db, float range_db, uint8_t *out);

}  // namespace sono
