// sono_detect.h -- kernels of the detector front end: the array passes of
// modules/batCallDetector.js that sit directly on the STFT output.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <cstring>

#include "../../include/sonostft.h"

namespace sono {

// batCallDetector.js:913-927.  The reference indexes the flat magnitude array, so a
// band that runs past a frame's last bin continues into the next frame's first
// bins (and stops at the end of the array); kept as is.
hipError_t launch_band_active(const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin, uint32_t max_bin,
                              double target_energy_sum, uint8_t *d_active, double *d_band_energy, hipStream_t s);

// batCallDetector.js:1347-1363: dB = 10 log10(max(mag^2 / fft_size, 1e-16)) in f64, stored as f32.
hipError_t launch_power_db(const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin, uint32_t nb,
                           uint32_t fft_size, float *d_db, hipStream_t s);

// batCallDetector.js:1544-1550.  *d_max_key must hold sono_detect_key(-inf) before the launch.
hipError_t launch_db_max(const float *d_db, size_t count, uint32_t *d_max_key, hipStream_t s);

// batCallDetector.js:1556-1567.
hipError_t launch_db_active(const float *d_db, size_t frames, uint32_t nb, double threshold_db, uint8_t *d_active,
                            hipStream_t s);

// batCallDetector.js:3520-3527,3612-3630 (peak), 2509-2522 (lowest bin above), 2002-2038 (highest bin above).
hipError_t launch_frame_features(const float *d_db, size_t frames, uint32_t nb, uint32_t peak_b0, uint32_t peak_b1,
                                 uint32_t low_max_bin, double low_threshold_db, uint32_t high_min_bin,
                                 double high_threshold_db, sono_frame_features *d_out, hipStream_t s);

// Per row: first bin of the largest value, the value, and its two neighbours (NaN at the ends of the row) --
// what the parabolic peak interpolation of measureFrequencyParameters needs (batCallDetector.js:3010-3034).
// lo12 / hi12: lowest and highest bin of the row above (its peak - 12 dB): the "instantaneous bandwidth" of the
// vertical-click filter (batCallDetector.js:3059-3088).
struct FramePeak { int32_t bin; float db, left, right; int32_t lo12, hi12; };
hipError_t launch_frame_peaks(const float *d_db, size_t frames, uint32_t nb, FramePeak *d_out, hipStream_t s);
// launch_power_db + launch_frame_peaks in one pass over the magnitudes, the dB values staying in registers, for bands
// of up to 512 bins (db_peaks_fits): the same records.
bool db_peaks_fits(uint32_t nb);
hipError_t launch_db_peaks(const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin, uint32_t nb, uint32_t fft_size,
                           FramePeak *d_out, hipStream_t s);

// launch_db_peaks for the rows of regions, filed where the regions' table wants them: span i covers `rows` consecutive
// rows of the magnitude matrix from first_row on, and their records go to entries table_row ... of the six arrays of
// `table` (device memory).  d_span_best[i] (may be nullptr) = the span's largest peak_db in row order, first one
// kept (strict >, from -infinity; bin 0 if none), with its peak bin: the automatic high-pass's question about a region
// (batCallDetector.js:635-644), answered span by span.  A region longer than kPeakSpanRows rows is cut into several spans.
struct PeakSpan { unsigned long long first_row, table_row; uint32_t rows, pad; };
struct PeakTable { float *db; int32_t *bin; float *left, *right; int32_t *lo12, *hi12; };
struct SpanBest { float db; int32_t bin; };
constexpr uint32_t kPeakSpanRows = 16;      // (64 left the launch's last waves 64 rows each to finish alone: 0.67 against 0.57 ms per launch; 8 and 4 the same as 16)
hipError_t launch_db_peaks_spans(const float *d_mags, const PeakSpan *d_spans, size_t n_spans, uint32_t bins, uint32_t min_bin,
                                 uint32_t nb, uint32_t fft_size, const PeakTable &table, SpanBest *d_span_best, hipStream_t s);

// First / last frame of every maximal run of non-zero flags, as two flag arrays.
hipError_t launch_run_edges(const uint8_t *d_active, size_t frames, uint8_t *d_is_start, uint8_t *d_is_end,
                            hipStream_t s);

// Ordered compaction: indices f (ascending) with d_flags[f] != 0 -> d_out, their number -> *d_count.
// Call with d_temp == nullptr to get the temporary size.
hipError_t select_flagged(void *d_temp, size_t &temp_bytes, const uint8_t *d_flags, uint64_t *d_out,
                          uint64_t *d_count, size_t frames, hipStream_t s);

// calculateZonalNoiseFloors' histogram pass (batCallDetector.js:1849-1879) over count = rows*nb values that
// start at d_db: value i belongs to bin i % nb, d_zone_of_bin[bin] is its 10 kHz zone (255 = not counted),
// d_hist[zone*141 + (floor(max(dB, -100)) clamped to -120..20) + 120] counts it.  The caller zeroes d_hist.
constexpr int kZoneHistBins = 141;
hipError_t launch_zone_hist(const float *d_db, size_t count, uint32_t nb, const uint8_t *d_zone_of_bin,
                            uint32_t num_zones, unsigned int *d_hist, hipStream_t s);

// The same histogram pass for many regions at once, mode included: region r covers rows
// d_first_row[r] .. + d_rows_used[r] - 1 of d_db; d_floors[r*zone_stride + z] = noise floor of zone z
// (batCallDetector.js:1886-1921: lowest dB wins a tie, -2 dB offset, -102 for an empty zone).
hipError_t launch_zone_floors_regions(const float *d_db, const unsigned long long *d_first_row,
                                      const unsigned int *d_rows_used, size_t regions, uint32_t nb,
                                      const uint8_t *d_zone_of_bin, uint32_t num_zones, uint32_t zone_stride,
                                      double *d_floors, hipStream_t s);
// ... from the magnitude matrix (rows of `bins` entries, the band starting at min_bin), every value converted on the way in
hipError_t launch_zone_floors_regions_mags(const float *d_mags, uint32_t bins, uint32_t min_bin, uint32_t fft_size,
                                           const unsigned long long *d_first_row, const unsigned int *d_rows_used, size_t regions,
                                           uint32_t nb, const uint8_t *d_zone_of_bin, uint32_t num_zones, uint32_t zone_stride,
                                           double *d_floors, hipStream_t s);

// Butterworth high-pass of many regions at once, one lane per region
// (applyHighpassFilter at its default order 4 = two identical second-order sections,
// batCallDetector.js:4045-4140): doubles, every product and sum rounded on its own, the
// output of each section rounded to float like the reference's Float32Array.
struct HighpassCoef { double b0, b1, b2, a1, a2; };
struct HighpassJob {
    unsigned long long src, dst, len;   // sample offsets into d_src / d_dst, number of samples
    unsigned int coef, pad;             // index into the coefficient table
};
// in_place: the filtered samples also overwrite the regions' own samples in d_src (disjoint regions only).
hipError_t launch_highpass_regions(const float *d_src, float *d_dst, const HighpassJob *d_jobs, size_t jobs,
                                   const HighpassCoef *d_coefs, bool in_place, hipStream_t s);

// calculateRMSbasedSNR's matrix passes (batCallDetector.js:436-538).  Box = rows t0..t1 (inclusive) x bins with
// d_bin_in_box[b] != 0.  Pass 1: largest / smallest dB value inside the box, as detect_key()s in d_keys[0..1]
// (the caller sets them to 0 and 0xFFFFFFFF).  Pass 2: d_sums[0] += 10^(dB/10) over box values > threshold,
// d_counts[0] their number; if outside_is_noise, d_sums[1] / d_counts[1] the same over every value outside the box.
hipError_t launch_box_minmax(const float *d_db, size_t frames, uint32_t nb, size_t t0, size_t t1,
                             const uint8_t *d_bin_in_box, uint32_t *d_keys, hipStream_t s);
hipError_t launch_box_sums(const float *d_db, size_t frames, uint32_t nb, size_t t0, size_t t1, const uint8_t *d_bin_in_box,
                           double threshold_db, int outside_is_noise, double *d_sums, unsigned long long *d_counts,
                           hipStream_t s);
// d_sums[1] += 10^(dB/10) over the columns with d_bin_sel[b] != 0 of a whole matrix, d_counts[1] their number (js:505-516).
hipError_t launch_band_linear_sum(const float *d_db, size_t frames, uint32_t nb, const uint8_t *d_bin_sel, double *d_sums,
                                  unsigned long long *d_counts, hipStream_t s);

// Ascending sort of `count` finite floats (d_in -> d_out).  d_temp == nullptr: size query.
hipError_t sort_floats(void *d_temp, size_t &temp_bytes, const float *d_in, float *d_out, size_t count, hipStream_t s);

// Interleaved PCM -> one f32 channel.  Integer samples are divided by 2^(bits-1); 8-bit WAV is unsigned.
// One span of a recording whose end the oscillogram refinement looks for: its windows' energies go to
// sums[out .. out + steps); window i covers samples first + i * hop .. + window - 1 of the recording.
struct WindowJob { uint64_t first, out; uint32_t steps, pad; };
hipError_t launch_window_sums(const float *d_x, const WindowJob *d_jobs, size_t n_jobs, uint32_t window, uint32_t hop, double *d_sums,
                              hipStream_t s);

hipError_t launch_pcm_to_f32(const void *d_pcm, size_t frames, uint32_t channels, uint32_t channel, uint32_t bits,
                             int is_float, float *d_out, hipStream_t s);

// Order-preserving map f32 -> u32 (so that atomicMax on the key is a float max) and back.
__host__ __device__ inline uint32_t detect_key(float v)
{
    uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = __float_as_uint(v);
#else
    std::memcpy(&b, &v, 4);
#endif
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float detect_unkey(uint32_t k)
{
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float v;
    std::memcpy(&v, &b, 4);
    return v;
}

}  // namespace sono
