// sono_measure.h -- the two threshold searches of the detector's measurement
// stage (host only, serial per call): findOptimalLowFrequencyThreshold and
// findOptimalHighFrequencyThreshold of batCallDetector.js.  Doubles throughout,
// like JavaScript numbers; the dB matrix and the frequency axis are float.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace sono {

struct CallMatrix {
    const float *db;          // frames x bins, row-major: the call's slice of the analysis spectrogram
    size_t frames;
    uint32_t bins;
    const float *freq_bins;   // bins centre frequencies, Hz
    const double *zonal;      // noise floor of every 10 kHz zone, or nullptr: computed from the slice as the reference does
    size_t num_zones;
    bool debug_mode = false;  // the detector's debugMode switch (setDebugMode, batCallDetector.js:359)
    bool time_expansion = false;  // getTimeExpansionMode() (fileState.js:158-166): the recording plays 10 x slowed down
};

struct LowResult {
    double threshold_db = -24, low_freq_hz = 0, end_freq_hz = 0;
    int64_t frame = 0, bin = -1;
    bool has_freq = false, has_frame = false, warning = false, discard = false, reference_throws = false;
};
// batCallDetector.js:2413-2909.  limit_frame < 0 = none.
LowResult find_optimal_low_threshold(const CallMatrix &m, double peak_power_db, int64_t peak_frame, int64_t limit_frame);

struct HighResult {
    double threshold_db = -24, high_freq_hz = 0, start_freq_hz = 0;
    int64_t bin = 0, frame = 0, final_search_limit_frame = 0;
    bool has_freq = false, has_start = false, has_limit = false, warning = false, is_cf = false, reference_throws = false;
};
// batCallDetector.js:1931-2401.  time_frames: m.frames centre times, seconds.
HighResult find_optimal_high_threshold(const CallMatrix &m, const double *time_frames, double peak_power_db, int64_t peak_frame);

// What measureFrequencyParameters writes into a BatCall.  Doubles are NaN, indices -1, where the reference leaves null
// (or never sets the field).  start_time_s, end_time_s and duration_ms come in as the framing stage set them.
struct CallMeasure {
    double start_time_s, end_time_s, duration_ms;
    double peak_freq_khz, peak_freq_ms, peak_power_db;
    double high_freq_khz, high_freq_ms;
    double start_freq_khz, start_freq_ms, start_freq_time_s;
    double end_freq_khz, end_freq_ms, end_freq_time_s;
    double low_freq_khz, low_freq_ms;
    double characteristic_freq_khz, characteristic_freq_ms;
    double knee_freq_khz, knee_freq_ms, heel_freq_khz, heel_freq_ms;
    double bandwidth_khz, high_threshold_db_used, low_threshold_db_used;
    int64_t peak_frame, high_freq_frame, start_freq_frame, start_freq_bin, end_frame_for_low_freq, knee_frame, heel_frame;
    bool discarded, reference_throws;
};
// measureFrequencyParameters (batCallDetector.js:2994-3894).  time_frames holds num_times centre times: the call's
// frames and, unless the call ends the analysis matrix, one more (js:787, 1201).  With m.time_expansion the call that
// reaches the end of the measurement is rescaled by 10 as BatCall.applyTimeExpansion does (js:185-278, 3890-3893).
constexpr double kTimeExpansionFactor = 10;
void apply_time_expansion(CallMeasure &c, double factor);
void measure_frequency_parameters(const CallMatrix &m, const double *time_frames, size_t num_times, double fhigh_khz,
                                  double freq_resolution_hz, CallMeasure &c);

// calculateZonalNoiseFloors (batCallDetector.js:1818-1924) on a host matrix: floors of the zones, their number returned.
size_t host_zonal_floors(const float *db, size_t frames, uint32_t bins, const float *freq_bins, int64_t start_frame,
                         int64_t end_frame, double *floors, size_t cap);

}  // namespace sono
