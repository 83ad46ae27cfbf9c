// sono_framing.h -- the serial, per-call parts of the detector's call-framing
// stage (host only).  JavaScript numbers are IEEE doubles: everything here is
// computed in double, and values that the reference keeps in a Float32Array are
// rounded to float at the same places.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <vector>

namespace sono {

// savitzkyGolay (batCallDetector.js:1610-1640), window 5, order 2.  out may alias nothing; n < 5 copies.
void host_savitzky_golay(const double *in, size_t n, double *out);

// refineEndUsingOscillogram (batCallDetector.js:1646-1803).  Returns false when the
// sample rate is too low for the 0.1 ms RMS window to have a hop of at least one sample.
bool host_refine_end(const float *audio, size_t n, double sample_rate, int64_t start_sample, int64_t end_sample,
                     int64_t *refined);
// The same in its three parts, for callers whose samples lie elsewhere (the GPU sums the windows of a whole
// recording's calls in one launch): the windows of a span, their energies, and the walk over them.
struct RefinePlan {
    int64_t safe_start = 0;                  // first sample of the first window, in the audio the span counts in
    int64_t window = 0, hop = 0, sustained = 0;
    size_t steps = 0;                        // windows: window i covers samples safe_start + i * hop ... + window - 1
};
// 1: p is filled and has at least one window; 0: nothing to refine, the end stays where it is; -1: the sample rate is
// too low for the window to have a hop.  n: samples of the audio.
int host_refine_plan(size_t n, double sample_rate, int64_t start_sample, int64_t end_sample, RefinePlan &p);
// sums[i] = the squares of window i's samples, added in order in double (js:1676-1683).  audio: sample 0 of the plan's count.
void host_window_sums(const float *audio, const RefinePlan &p, double *sums);
// The refined end sample (end_sample where the walk finds no earlier end), js:1685-1803.
int64_t host_refine_from_sums(const RefinePlan &p, const double *sums, int64_t end_sample);

// calculateAutoHighpassFilterFreq (batCallDetector.js:4028-4035), kHz.
double host_auto_highpass_khz(double peak_khz);

// applyHighpassFilter + _applyButterworthStage (batCallDetector.js:4045-4140).
void host_highpass(const float *in, size_t n, double filter_hz, double sample_rate, double order, float *out);

// _scanSpectrumForFrequency (batCallDetector.js:2936-2987).  bin = -1 (and freq_hz = -1) if no bin qualifies.
void host_scan_spectrum(const float *row, const float *freq_bins, size_t n, double threshold_db, bool high_to_low,
                        double *freq_hz, int64_t *bin);

struct CallFrame {
    uint64_t start_frame, end_frame;
    double start_time_s, end_time_s, duration_ms;
};
struct FrameRun { uint64_t start, end; };
// The per-segment part of detectCalls (batCallDetector.js:1128-1208; padding 3 ms, both duration filters) or,
// with full_file_rules, of processFullFile (batCallDetector.js:739-810; padding 5 ms, only the <= 1 ms discard).
// refine (or nullptr: host_refine_end on audio): answers refineEndUsingOscillogram for a span of the n samples.
using RefineFn = std::function<bool(int64_t start_sample, int64_t end_sample, int64_t *refined)>;
bool host_frame_calls(const FrameRun *segs, size_t count, const double *time_frames, size_t frames, const float *audio,
                      size_t n, double sample_rate, uint32_t fft_size, double padding_ms, double min_call_duration_ms,
                      bool full_file_rules, std::vector<CallFrame> &out, const RefineFn *refine = nullptr);

// Peak frequency of measureFrequencyParameters' first step (batCallDetector.js:3021-3036), kHz: the peak bin's
// centre, moved by a parabola through the peak and its two neighbours where it has both (interior = true).
double host_peak_freq_khz(float left_db, float peak_db, float right_db, bool interior, float bin_freq_hz, double bin_width_hz);

// The vertical-click filter of measureFrequencyParameters (batCallDetector.js:3059-3088): instantaneous bandwidth in
// kHz of the peak frame from its lowest / highest bin above peak - 12 dB (-1 / -1 if none: bandwidth 0); a call is
// discarded when it exceeds 20 kHz.
double host_instant_bandwidth_khz(int32_t lo_bin, int32_t hi_bin, double freq_resolution_hz);

// Zone bookkeeping of calculateZonalNoiseFloors (batCallDetector.js:1837-1839, 1863-1864):
// zone index of every bin, number of zones.
size_t host_zone_map(const float *freq_bins, size_t n, std::vector<uint8_t> &zone_of_bin);
// Mode of every zone's histogram (141 counts for -120 .. +20 dB), lowest dB on ties, -2 dB offset (:1886-1921).
void host_zone_modes(const unsigned int *hist, size_t num_zones, double *floors);

}  // namespace sono
