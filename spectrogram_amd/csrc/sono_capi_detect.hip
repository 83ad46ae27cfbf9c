// sono_capi_detect.hip -- C ABI of the detector front end (include/sonostft.h):
// the array passes of modules/batCallDetector.js that follow the STFT.
#include <hip/hip_runtime.h>

#include <cstddef>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/sonostft.h"
#include "sono_detect.h"
#include "sono_engine.h"
#include "sono_framing.h"
#include "sono_measure.h"

using sono::DeviceGuard;
using sono::fail;
using sono::hip_fail;

namespace {

hipStream_t pick(sono_engine *e, void *hip_stream) { return hip_stream ? (hipStream_t)hip_stream : e->stream; }

// Runs of non-zero flags -> host vectors of first / last frame.
int runs_to_host(sono_engine *e, const uint8_t *d_active, size_t frames, std::vector<uint64_t> &first,
                 std::vector<uint64_t> &last)
{
    first.clear(); last.clear();
    if (frames == 0) return SONO_OK;
    const size_t cap = frames / 2 + 1;
    HIP_TRY(e->d_runs.reserve(2 * cap * sizeof(uint64_t)));
    HIP_TRY(e->d_count.reserve(2 * sizeof(uint64_t)));
    uint64_t *d_first = (uint64_t *)e->d_runs.p, *d_last = d_first + cap;
    int rc = sono_detect_runs_dev(e, d_active, frames, d_first, d_last, (uint64_t *)e->d_count.p, nullptr);
    if (rc) return rc;
    uint64_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, e->d_count.p, sizeof(n), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    first.resize(n); last.resize(n);
    if (n) {
        HIP_TRY(hipMemcpyAsync(first.data(), d_first, n * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(last.data(), d_last, n * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return SONO_OK;
}

}  // namespace

extern "C" {

int sono_detect_band_active_dev(sono_engine *e, const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin,
                                uint32_t max_bin, double target, uint8_t *d_active, double *d_band_energy,
                                void *hip_stream)
{
    if (!e || (frames && (!d_mags || !d_active))) return fail(SONO_EINVAL, "sono_detect_band_active_dev: NULL argument");
    if (bins == 0) return fail(SONO_EINVAL, "sono_detect_band_active_dev: bins is 0");
    DeviceGuard guard(e->device);
    HIP_TRY(sono::launch_band_active(d_mags, frames, bins, min_bin, max_bin, target, d_active, d_band_energy,
                                     pick(e, hip_stream)));
    return SONO_OK;
}

int sono_detect_power_db_dev(sono_engine *e, const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin,
                             uint32_t num_bins, float *d_db, void *hip_stream)
{
    if (!e || (frames && num_bins && (!d_mags || !d_db))) return fail(SONO_EINVAL, "sono_detect_power_db_dev: NULL argument");
    if ((uint64_t)min_bin + num_bins > bins) return fail(SONO_EINVAL, "sono_detect_power_db_dev: band exceeds the bins");
    DeviceGuard guard(e->device);
    HIP_TRY(sono::launch_power_db(d_mags, frames, bins, min_bin, num_bins, e->N, d_db, pick(e, hip_stream)));
    return SONO_OK;
}

int sono_detect_frame_peaks_dev(sono_engine *e, const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin,
                                uint32_t num_bins, sono_frame_peak *d_out, int form, void *hip_stream)
{
    static_assert(sizeof(sono_frame_peak) == sizeof(sono::FramePeak) && offsetof(sono_frame_peak, db) == offsetof(sono::FramePeak, db) &&
                      offsetof(sono_frame_peak, lo12_bin) == offsetof(sono::FramePeak, lo12), "the record of the kernels is the record of the header");
    if (!e || (frames && (!d_mags || !d_out))) return fail(SONO_EINVAL, "sono_detect_frame_peaks_dev: NULL argument");
    if (num_bins == 0 || (uint64_t)min_bin + num_bins > bins) return fail(SONO_EINVAL, "sono_detect_frame_peaks_dev: band exceeds the bins");
    if (form != 0 && form != 1) return fail(SONO_EINVAL, "sono_detect_frame_peaks_dev: form must be 0 or 1");
    if (form != 0 && !sono::db_peaks_fits(num_bins)) return fail(SONO_EINVAL, "sono_detect_frame_peaks_dev: the one-pass form takes bands of up to 512 bins");
    DeviceGuard guard(e->device);
    hipStream_t s = pick(e, hip_stream);
    if (form == 0) {
        HIP_TRY(e->d_db.reserve((frames ? frames : 1) * (size_t)num_bins * sizeof(float)));
        HIP_TRY(sono::launch_power_db(d_mags, frames, bins, min_bin, num_bins, e->N, (float *)e->d_db.p, s));
        HIP_TRY(sono::launch_frame_peaks((const float *)e->d_db.p, frames, num_bins, (sono::FramePeak *)d_out, s));
    } else {
        HIP_TRY(sono::launch_db_peaks(d_mags, frames, bins, min_bin, num_bins, e->N, (sono::FramePeak *)d_out, s));
    }
    return SONO_OK;
}

int sono_detect_db_max_dev(sono_engine *e, const float *d_db, size_t count, float *d_max, void *hip_stream)
{
    if (!e || !d_max || (count && !d_db)) return fail(SONO_EINVAL, "sono_detect_db_max_dev: NULL argument");
    DeviceGuard guard(e->device);
    hipStream_t s = pick(e, hip_stream);
    // the running maximum is kept as an order-preserving integer key in the caller's word and decoded in place
    const uint32_t floor_key = sono::detect_key(-INFINITY);
    HIP_TRY(hipMemcpyAsync(d_max, &floor_key, sizeof(floor_key), hipMemcpyHostToDevice, s));
    HIP_TRY(sono::launch_db_max(d_db, count, (uint32_t *)d_max, s));
    uint32_t key = 0;
    HIP_TRY(hipMemcpyAsync(&key, d_max, sizeof(key), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const float v = sono::detect_unkey(key);
    HIP_TRY(hipMemcpyAsync(d_max, &v, sizeof(v), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SONO_OK;
}

int sono_detect_db_active_dev(sono_engine *e, const float *d_db, size_t frames, uint32_t num_bins, double threshold_db,
                              uint8_t *d_active, void *hip_stream)
{
    if (!e || (frames && (!d_db || !d_active))) return fail(SONO_EINVAL, "sono_detect_db_active_dev: NULL argument");
    DeviceGuard guard(e->device);
    HIP_TRY(sono::launch_db_active(d_db, frames, num_bins, threshold_db, d_active, pick(e, hip_stream)));
    return SONO_OK;
}

int sono_detect_frame_features_dev(sono_engine *e, const float *d_db, size_t frames, uint32_t num_bins,
                                   uint32_t peak_b0, uint32_t peak_b1, uint32_t low_max_bin, double low_thr,
                                   uint32_t high_min_bin, double high_thr, sono_frame_features *d_out,
                                   void *hip_stream)
{
    if (!e || (frames && (!d_db || !d_out))) return fail(SONO_EINVAL, "sono_detect_frame_features_dev: NULL argument");
    DeviceGuard guard(e->device);
    HIP_TRY(sono::launch_frame_features(d_db, frames, num_bins, peak_b0, peak_b1, low_max_bin, low_thr, high_min_bin,
                                        high_thr, d_out, pick(e, hip_stream)));
    return SONO_OK;
}

int sono_detect_runs_dev(sono_engine *e, const uint8_t *d_active, size_t frames, uint64_t *d_first, uint64_t *d_last,
                         uint64_t *d_count, void *hip_stream)
{
    if (!e || !d_count || (frames && (!d_active || !d_first || !d_last)))
        return fail(SONO_EINVAL, "sono_detect_runs_dev: NULL argument");
    if (frames > 0x7FFFFFFFull) return fail(SONO_EINVAL, "sono_detect_runs_dev: more than 2^31-1 frames");
    DeviceGuard guard(e->device);
    hipStream_t s = pick(e, hip_stream);
    if (frames == 0) {
        HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint64_t), s));
        return SONO_OK;
    }
    HIP_TRY(e->d_edges.reserve(2 * frames));
    uint8_t *d_is_start = (uint8_t *)e->d_edges.p, *d_is_end = d_is_start + frames;
    HIP_TRY(sono::launch_run_edges(d_active, frames, d_is_start, d_is_end, s));
    size_t temp = 0;
    HIP_TRY(sono::select_flagged(nullptr, temp, d_is_start, d_first, d_count, frames, s));
    HIP_TRY(e->d_select_tmp.reserve(temp ? temp : 1));
    HIP_TRY(sono::select_flagged(e->d_select_tmp.p, temp, d_is_start, d_first, d_count, frames, s));
    HIP_TRY(sono::select_flagged(e->d_select_tmp.p, temp, d_is_end, d_last, d_count, frames, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SONO_OK;
}

int sono_fast_scan_segments(sono_engine *e, const float *x, size_t n, double sample_rate, double flow_khz,
                            double fhigh_khz, double threshold_db, sono_segment *out, size_t cap, size_t *count)
{
    if (count) *count = 0;
    if (!e || (!x && n)) return fail(SONO_EINVAL, "sono_fast_scan_segments: NULL argument");
    const uint32_t N = e->N, hop = N / 2, noverlap = N - hop;                    // js:882-886
    const size_t F = sono_frames_for(n, N, noverlap);
    if (F == 0) return SONO_OK;
    DeviceGuard guard(e->device);
    const size_t used = (F - 1) * (size_t)hop + N;
    HIP_TRY(e->d_in.reserve(used * sizeof(float)));
    const int rc = sono::upload_host_array(e, e->d_in.p, x, used * sizeof(float), e->stream);
    if (rc) return rc;
    return sono::fast_scan_resident(e, (const float *)e->d_in.p, n, sample_rate, flow_khz, fhigh_khz, threshold_db, out, cap, count);
}

}  // extern "C"

// fastScanSegmentsWasm (batCallDetector.js:878-949) on a recording that already lies in HBM at d_x (n samples).
int sono::fast_scan_resident(sono_engine *e, const float *d_x, size_t n, double sample_rate, double flow_khz, double fhigh_khz,
                             double threshold_db, sono_segment *out, size_t cap, size_t *count)
{
    if (count) *count = 0;
    const uint32_t N = e->N, hop = N / 2, noverlap = N - hop, bins = N / 2;      // js:882-886
    const double freq_res = sample_rate / (double)N;                             // js:889
    const double lo = std::floor(flow_khz * 1000.0 / freq_res), hi = std::ceil(fhigh_khz * 1000.0 / freq_res);
    const size_t F = sono_frames_for(n, N, noverlap);
    if (F == 0) return SONO_OK;
    if (!(hi >= lo) || hi < 0.0) return SONO_OK;                                 // empty band: no frame can be active
    const uint32_t min_bin = lo > 0.0 ? (lo > 4.0e9 ? 0xFFFFFFFFu : (uint32_t)lo) : 0u;
    const uint32_t max_bin = hi > 4.0e9 ? 0xFFFFFFFFu : (uint32_t)hi;
    const double target = std::pow(10.0, threshold_db / 10.0) * (double)N * 0.25;   // js:906-909
    DeviceGuard guard(e->device);
    const size_t used = (F - 1) * (size_t)hop + N;
    HIP_TRY(e->d_flags.reserve(F));
    sono_batch b{d_x, 1, used, used};
    int rc;
    if (max_bin < bins) {
        // the usual case: one fused launch, the magnitudes never leave the chip
        rc = sono_stft_band_active_dev(e, &b, noverlap, min_bin, max_bin, target, (uint8_t *)e->d_flags.p, nullptr, nullptr);
    } else {
        // a band that reaches the last bin: the reference's flat index runs into the next frame's row (js:918-921),
        // which needs the whole magnitude matrix
        HIP_TRY(e->d_mags.reserve(F * bins * sizeof(float)));
        rc = sono_stft_mag_f32_dev(e, &b, noverlap, (float *)e->d_mags.p, nullptr);
        if (rc) return rc;
        rc = sono_detect_band_active_dev(e, (const float *)e->d_mags.p, F, bins, min_bin, max_bin, target,
                                         (uint8_t *)e->d_flags.p, nullptr, nullptr);
    }
    if (rc) return rc;
    std::vector<uint64_t> first, last;
    rc = runs_to_host(e, (const uint8_t *)e->d_flags.p, F, first, last);
    if (rc) return rc;
    if (first.size() > cap || (first.size() && !out)) return fail(SONO_ECAPACITY, "sono_fast_scan_segments: output buffer too small");
    for (size_t i = 0; i < first.size(); ++i) {
        out[i].start = first[i] * hop;                                           // js:930-933
        out[i].end = last[i] + 1 < F ? (last[i] + 1) * hop + N : (uint64_t)n;    // js:937,945
    }
    if (count) *count = first.size();
    return SONO_OK;
}

extern "C" {

int sono_generate_spectrogram(sono_engine *e, const float *x, size_t n, double sample_rate, double flow_khz,
                              double fhigh_khz, double hop_percent, float *out, size_t cap, size_t *frames,
                              uint32_t *min_bin_out, uint32_t *num_bins_out)
{
    if (frames) *frames = 0;
    if (min_bin_out) *min_bin_out = 0;
    if (num_bins_out) *num_bins_out = 0;
    if (!e || (!x && n)) return fail(SONO_EINVAL, "sono_generate_spectrogram: NULL argument");
    const uint32_t N = e->N, bins = N / 2;
    const double hop_d = std::floor((double)N * (hop_percent / 100.0));           // js:1306
    if (!(hop_d >= 1.0) || (size_t)N > n) return SONO_OK;                         // js:1309-1312 -> null
    const uint32_t hop = hop_d >= (double)N ? N : (uint32_t)hop_d;
    if (hop >= N + 1u || N - hop >= N) return SONO_OK;
    const uint32_t noverlap = N - hop;                                            // hop == N gives noverlap 0
    const double freq_res = sample_rate / (double)N;
    const size_t F = sono_frames_for(n, N, noverlap);
    if (F < 1) return SONO_OK;
    const double lo = std::max(0.0, std::floor(flow_khz * 1000.0 / freq_res));   // js:1328-1330
    const double hi = std::min((double)(bins - 1), std::floor(fhigh_khz * 1000.0 / freq_res));
    if (!(hi - lo + 1.0 > 0.0)) return SONO_OK;                                   // js:1332-1335 -> null
    const uint32_t min_bin = (uint32_t)lo, nb = (uint32_t)(hi - lo + 1.0);
    if (!out || cap < F * nb) return fail(SONO_ECAPACITY, "sono_generate_spectrogram: output buffer too small");
    DeviceGuard guard(e->device);
    const size_t used = (F - 1) * (size_t)hop + N;
    HIP_TRY(e->d_in.reserve(used * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(e->d_in.p, x, used * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e->d_mags.reserve(F * bins * sizeof(float)));
    HIP_TRY(e->d_db.reserve(F * nb * sizeof(float)));
    sono_batch b{(const float *)e->d_in.p, 1, used, used};
    int rc = sono_stft_mag_f32_dev(e, &b, noverlap, (float *)e->d_mags.p, nullptr);
    if (rc) return rc;
    rc = sono_detect_power_db_dev(e, (const float *)e->d_mags.p, F, bins, min_bin, nb, (float *)e->d_db.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, e->d_db.p, F * nb * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (frames) *frames = F;
    if (min_bin_out) *min_bin_out = min_bin;
    if (num_bins_out) *num_bins_out = nb;
    return SONO_OK;
}

int sono_detect_call_segments(sono_engine *e, const float *power_db, size_t frames, uint32_t num_bins,
                              double call_threshold_db, sono_segment *out, size_t cap, size_t *count)
{
    if (count) *count = 0;
    if (!e || (frames && num_bins && !power_db)) return fail(SONO_EINVAL, "sono_detect_call_segments: NULL argument");
    if (frames == 0) return SONO_OK;
    DeviceGuard guard(e->device);
    const size_t total = frames * (size_t)num_bins;
    HIP_TRY(e->d_db.reserve((total ? total : 1) * sizeof(float)));
    HIP_TRY(e->d_flags.reserve(frames));
    HIP_TRY(e->d_gmax.reserve(sizeof(uint32_t)));
    if (total) HIP_TRY(hipMemcpyAsync(e->d_db.p, power_db, total * sizeof(float), hipMemcpyHostToDevice, e->stream));
    int rc = sono_detect_db_max_dev(e, (const float *)e->d_db.p, total, (float *)e->d_gmax.p, nullptr);
    if (rc) return rc;
    float gmax = 0.0f;
    HIP_TRY(hipMemcpy(&gmax, e->d_gmax.p, sizeof(gmax), hipMemcpyDeviceToHost));
    const double threshold = (double)gmax + call_threshold_db;                   // js:1553
    rc = sono_detect_db_active_dev(e, (const float *)e->d_db.p, frames, num_bins, threshold, (uint8_t *)e->d_flags.p,
                                   nullptr);
    if (rc) return rc;
    std::vector<uint64_t> first, last;
    rc = runs_to_host(e, (const uint8_t *)e->d_flags.p, frames, first, last);
    if (rc) return rc;
    if (first.size() > cap || (first.size() && !out)) return fail(SONO_ECAPACITY, "sono_detect_call_segments: output buffer too small");
    for (size_t i = 0; i < first.size(); ++i) { out[i].start = first[i]; out[i].end = last[i]; }   // js:1580-1594
    if (count) *count = first.size();
    return SONO_OK;
}

int sono_frame_features_db(sono_engine *e, const float *power_db, size_t frames, uint32_t num_bins, uint32_t peak_b0,
                           uint32_t peak_b1, uint32_t low_max_bin, double low_thr, uint32_t high_min_bin,
                           double high_thr, sono_frame_features *out, size_t cap)
{
    if (!e || (frames && (!power_db || !out))) return fail(SONO_EINVAL, "sono_frame_features_db: NULL argument");
    if (cap < frames) return fail(SONO_ECAPACITY, "sono_frame_features_db: output buffer too small");
    if (frames == 0) return SONO_OK;
    DeviceGuard guard(e->device);
    const size_t total = frames * (size_t)num_bins;
    HIP_TRY(e->d_db.reserve((total ? total : 1) * sizeof(float)));
    HIP_TRY(e->d_out.reserve(frames * sizeof(sono_frame_features)));
    if (total) HIP_TRY(hipMemcpyAsync(e->d_db.p, power_db, total * sizeof(float), hipMemcpyHostToDevice, e->stream));
    int rc = sono_detect_frame_features_dev(e, (const float *)e->d_db.p, frames, num_bins, peak_b0, peak_b1, low_max_bin,
                                            low_thr, high_min_bin, high_thr, (sono_frame_features *)e->d_out.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, e->d_out.p, frames * sizeof(sono_frame_features), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SONO_OK;
}

// ---- call framing -------------------------------------------------------------

int sono_zonal_noise_floors_dev(sono_engine *e, const float *d_db, size_t frames, uint32_t num_bins, const float *freq_bins,
                                size_t start_frame, size_t end_frame, double *floors, size_t cap, size_t *num_zones,
                                void *hip_stream)
{
    if (num_zones) *num_zones = 0;
    if (!e || !freq_bins || num_bins == 0 || (frames && !d_db))
        return fail(SONO_EINVAL, "sono_zonal_noise_floors: NULL argument or empty bin list");
    std::vector<uint8_t> zone_of_bin;
    const size_t zones = sono::host_zone_map(freq_bins, num_bins, zone_of_bin);
    if (zones == 0) return fail(SONO_EINVAL, "sono_zonal_noise_floors: frequency axis out of range");
    if (!floors || cap < zones) return fail(SONO_ECAPACITY, "sono_zonal_noise_floors: output buffer too small");
    std::vector<unsigned int> hist(zones * sono::kZoneHistBins, 0u);
    const size_t last = frames ? (end_frame < frames - 1 ? end_frame : frames - 1) : 0;      // js:1850: stop at the matrix end
    if (frames && start_frame <= last) {
        DeviceGuard guard(e->device);
        hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
        const size_t hist_bytes = hist.size() * sizeof(unsigned int);
        HIP_TRY(e->d_edges.reserve(hist_bytes + num_bins));
        unsigned int *d_hist = (unsigned int *)e->d_edges.p;
        uint8_t *d_zone = (uint8_t *)e->d_edges.p + hist_bytes;
        HIP_TRY(hipMemsetAsync(d_hist, 0, hist_bytes, s));
        HIP_TRY(hipMemcpyAsync(d_zone, zone_of_bin.data(), num_bins, hipMemcpyHostToDevice, s));
        HIP_TRY(sono::launch_zone_hist(d_db + start_frame * (size_t)num_bins, (last - start_frame + 1) * (size_t)num_bins,
                                       num_bins, d_zone, (uint32_t)zones, d_hist, s));
        HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, hist_bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    sono::host_zone_modes(hist.data(), zones, floors);
    if (num_zones) *num_zones = zones;
    return SONO_OK;
}

int sono_zonal_noise_floors(sono_engine *e, const float *power_db, size_t frames, uint32_t num_bins, const float *freq_bins,
                            size_t start_frame, size_t end_frame, double *floors, size_t cap, size_t *num_zones)
{
    if (num_zones) *num_zones = 0;
    if (!e || (frames && num_bins && !power_db)) return fail(SONO_EINVAL, "sono_zonal_noise_floors: NULL argument");
    DeviceGuard guard(e->device);
    const size_t total = frames * (size_t)num_bins;
    HIP_TRY(e->d_db.reserve((total ? total : 1) * sizeof(float)));
    if (total) HIP_TRY(hipMemcpyAsync(e->d_db.p, power_db, total * sizeof(float), hipMemcpyHostToDevice, e->stream));
    return sono_zonal_noise_floors_dev(e, (const float *)e->d_db.p, frames, num_bins, freq_bins, start_frame, end_frame,
                                       floors, cap, num_zones, nullptr);
}

int sono_db_percentile_dev(sono_engine *e, const float *d_db, size_t count, double fraction, float *value, void *hip_stream)
{
    if (!e || !value || (count && !d_db)) return fail(SONO_EINVAL, "sono_db_percentile: NULL argument");
    if (count == 0 || !(fraction >= 0.0) || fraction > 1.0) return fail(SONO_EINVAL, "sono_db_percentile: empty input or fraction outside [0, 1]");
    if (count > 0x7FFFFFFFull) return fail(SONO_EINVAL, "sono_db_percentile: more than 2^31 - 1 values");
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    size_t tmp = 0;
    HIP_TRY(sono::sort_floats(nullptr, tmp, d_db, nullptr, count, s));
    HIP_TRY(e->d_select_tmp.reserve(tmp ? tmp : 1));
    HIP_TRY(e->d_runs.reserve(count * sizeof(float)));
    HIP_TRY(sono::sort_floats(e->d_select_tmp.p, tmp, d_db, (float *)e->d_runs.p, count, s));
    size_t k = (size_t)std::floor((double)count * fraction);                     // js:1232
    if (k >= count) k = count - 1;
    HIP_TRY(hipMemcpyAsync(value, (const float *)e->d_runs.p + k, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SONO_OK;
}

int sono_db_percentile(sono_engine *e, const float *power_db, size_t count, double fraction, float *value)
{
    if (!e || !value || (count && !power_db)) return fail(SONO_EINVAL, "sono_db_percentile: NULL argument");
    if (count == 0) return fail(SONO_EINVAL, "sono_db_percentile: empty input");
    DeviceGuard guard(e->device);
    HIP_TRY(e->d_db.reserve(count * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(e->d_db.p, power_db, count * sizeof(float), hipMemcpyHostToDevice, e->stream));
    return sono_db_percentile_dev(e, (const float *)e->d_db.p, count, fraction, value, nullptr);
}

int sono_rms_snr(sono_engine *e, const float *power_db, size_t frames, uint32_t nb, const float *freq_bins, int64_t start_idx,
                 int64_t end_idx, double low_hz, double high_hz, const float *noise_db, size_t noise_frames, uint32_t noise_nb,
                 const float *noise_freq_bins, double sel_low_hz, double sel_high_hz, sono_snr_result *out)
{
    if (!e || !out || (frames && nb && (!power_db || !freq_bins)))
        return fail(SONO_EINVAL, "sono_rms_snr: NULL argument");
    const bool external = noise_db && noise_frames && noise_nb;
    if (external && !noise_freq_bins) return fail(SONO_EINVAL, "sono_rms_snr: noise_freq_bins is NULL");
    std::memset(out, 0, sizeof(*out));
    out->snr_db = out->signal_mean_db = out->noise_mean_db = NAN;
    out->threshold_db = out->signal_max_db = out->signal_min_db = NAN;
    out->status = 1;
    if (frames == 0 || nb == 0 || end_idx < 0 || end_idx < start_idx || start_idx >= (int64_t)frames) return SONO_OK;
    const size_t t0 = start_idx > 0 ? (size_t)start_idx : 0;
    const size_t t1 = (uint64_t)end_idx < frames - 1 ? (size_t)end_idx : frames - 1;      // js:437: stop at the matrix end
    std::vector<uint8_t> in_box(nb), sel(external ? noise_nb : 0);
    bool any = false;
    for (uint32_t b = 0; b < nb; ++b) {
        const double f = (double)freq_bins[b];
        in_box[b] = f >= low_hz && f <= high_hz;                                  // js:442
        any = any || in_box[b];
    }
    if (!any) return SONO_OK;
    for (uint32_t b = 0; b < (external ? noise_nb : 0u); ++b) {
        const double f = (double)noise_freq_bins[b];
        sel[b] = f >= sel_low_hz && f <= sel_high_hz;                             // js:510
    }
    DeviceGuard guard(e->device);
    hipStream_t s = e->stream;
    const size_t total = frames * (size_t)nb, ntotal = external ? noise_frames * (size_t)noise_nb : 0;
    HIP_TRY(e->d_db.reserve((total + ntotal) * sizeof(float)));
    float *d_db = (float *)e->d_db.p, *d_noise = d_db + total;
    HIP_TRY(hipMemcpyAsync(d_db, power_db, total * sizeof(float), hipMemcpyHostToDevice, s));
    if (external) HIP_TRY(hipMemcpyAsync(d_noise, noise_db, ntotal * sizeof(float), hipMemcpyHostToDevice, s));
    // scratch: [2 doubles sums][2 u64 counts][2 u32 keys][nb box flags][noise_nb selection flags]
    HIP_TRY(e->d_edges.reserve(48 + nb + sel.size() + 16));
    char *base = (char *)e->d_edges.p;
    double *d_sums = (double *)base;
    unsigned long long *d_counts = (unsigned long long *)(base + 16);
    uint32_t *d_keys = (uint32_t *)(base + 32);
    uint8_t *d_box = (uint8_t *)(base + 48), *d_sel = d_box + nb;
    struct { double sums[2]; unsigned long long counts[2]; uint32_t keys[2]; uint32_t pad[2]; } init = {{0, 0}, {0, 0}, {0u, 0xFFFFFFFFu}, {0, 0}};
    HIP_TRY(hipMemcpyAsync(base, &init, 48, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_box, in_box.data(), nb, hipMemcpyHostToDevice, s));
    if (external) HIP_TRY(hipMemcpyAsync(d_sel, sel.data(), sel.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(sono::launch_box_minmax(d_db, frames, nb, t0, t1, d_box, d_keys, s));
    uint32_t keys[2];
    HIP_TRY(hipMemcpyAsync(keys, d_keys, sizeof(keys), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (keys[0] == 0u || keys[1] == 0xFFFFFFFFu) return SONO_OK;                  // nothing comparable in the box
    const double smax = (double)sono::detect_unkey(keys[0]), smin = (double)sono::detect_unkey(keys[1]);
    if (smax == -INFINITY) return SONO_OK;                                        // js:452
    const double threshold = smin + (smax - smin) * 0.25;                         // js:460-462
    out->threshold_db = threshold; out->signal_max_db = smax; out->signal_min_db = smin;
    HIP_TRY(sono::launch_box_sums(d_db, frames, nb, t0, t1, d_box, threshold, external ? 0 : 1, d_sums, d_counts, s));
    if (external) HIP_TRY(sono::launch_band_linear_sum(d_noise, noise_frames, noise_nb, d_sel, d_sums, d_counts, s));
    double sums[2];
    unsigned long long counts[2];
    HIP_TRY(hipMemcpyAsync(sums, d_sums, sizeof(sums), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (counts[0] == 0) { out->status = 2; return SONO_OK; }                      // js:543
    out->status = 0;
    if (counts[1] == 0) { out->snr_db = INFINITY; return SONO_OK; }               // js:548
    const double s_mean = sums[0] / (double)counts[0], n_mean = sums[1] / (double)counts[1];
    out->signal_mean_db = 10 * std::log10(s_mean > 1e-16 ? s_mean : 1e-16);       // js:557-558
    out->noise_mean_db = 10 * std::log10(n_mean > 1e-16 ? n_mean : 1e-16);
    out->signal_count = counts[0]; out->noise_count = counts[1];
    out->snr_db = n_mean < 1e-16 ? INFINITY : 10 * std::log10(s_mean / n_mean);  // js:562-568
    return SONO_OK;
}

namespace {

int check_matrix(const sono_call_matrix *m, bool need_times, const char *who)
{
    if (!m) return fail(SONO_EINVAL, std::string(who) + ": NULL argument");
    if (m->frames && m->num_bins && (!m->power_db || !m->freq_bins)) return fail(SONO_EINVAL, std::string(who) + ": NULL matrix or axis");
    if (need_times && m->frames && !m->time_frames) return fail(SONO_EINVAL, std::string(who) + ": time_frames is NULL");
    if (m->zonal && m->num_zones > 256) return fail(SONO_EINVAL, std::string(who) + ": more than 256 zones");
    return SONO_OK;
}

}  // namespace

int sono_find_optimal_low_frequency_threshold(const sono_call_matrix *m, double peak, int64_t peak_frame, int64_t limit_frame,
                                              sono_low_threshold *out)
{
    if (!out) return fail(SONO_EINVAL, "sono_find_optimal_low_frequency_threshold: NULL argument");
    int rc = check_matrix(m, false, "sono_find_optimal_low_frequency_threshold");
    if (rc) return rc;
    if (peak_frame < 0) return fail(SONO_EINVAL, "sono_find_optimal_low_frequency_threshold: peak_frame < 0");
    const sono::CallMatrix cm{m->power_db, m->frames, m->num_bins, m->freq_bins, m->zonal, m->num_zones, (m->debug_mode & SONO_DEBUG_MODE) != 0,
                              (m->debug_mode & SONO_TIME_EXPANSION) != 0};
    const sono::LowResult r = sono::find_optimal_low_threshold(cm, peak, peak_frame, limit_frame);
    *out = {r.threshold_db, r.low_freq_hz, r.end_freq_hz, r.frame, r.bin, r.has_freq, r.has_frame, r.warning, r.discard,
            r.reference_throws, 0};
    return SONO_OK;
}

int sono_find_optimal_high_frequency_threshold(const sono_call_matrix *m, double peak, int64_t peak_frame, sono_high_threshold *out)
{
    if (!out) return fail(SONO_EINVAL, "sono_find_optimal_high_frequency_threshold: NULL argument");
    int rc = check_matrix(m, true, "sono_find_optimal_high_frequency_threshold");
    if (rc) return rc;
    if (peak_frame < 0) return fail(SONO_EINVAL, "sono_find_optimal_high_frequency_threshold: peak_frame < 0");
    const sono::CallMatrix cm{m->power_db, m->frames, m->num_bins, m->freq_bins, m->zonal, m->num_zones, (m->debug_mode & SONO_DEBUG_MODE) != 0,
                              (m->debug_mode & SONO_TIME_EXPANSION) != 0};
    const sono::HighResult r = sono::find_optimal_high_threshold(cm, m->time_frames, peak, peak_frame);
    *out = {r.threshold_db, r.high_freq_hz, r.start_freq_hz, r.bin, r.frame, r.final_search_limit_frame, r.has_freq, r.has_start,
            r.has_limit, r.warning, r.is_cf, r.reference_throws};
    return SONO_OK;
}

int sono_measure_frequency_parameters(const sono_call_matrix *m, size_t num_times, double fhigh_khz, double freq_res,
                                      sono_call_measurement *io)
{
    if (!io) return fail(SONO_EINVAL, "sono_measure_frequency_parameters: NULL argument");
    int rc = check_matrix(m, true, "sono_measure_frequency_parameters");
    if (rc) return rc;
    if (num_times < m->frames) return fail(SONO_EINVAL, "sono_measure_frequency_parameters: fewer times than frames");
    if (!(freq_res > 0)) return fail(SONO_EINVAL, "sono_measure_frequency_parameters: freq_resolution_hz must be positive");
    const sono::CallMatrix cm{m->power_db, m->frames, m->num_bins, m->freq_bins, m->zonal, m->num_zones, (m->debug_mode & SONO_DEBUG_MODE) != 0,
                              (m->debug_mode & SONO_TIME_EXPANSION) != 0};
    sono::CallMeasure c;
    c.start_time_s = io->start_time_s; c.end_time_s = io->end_time_s; c.duration_ms = io->duration_ms;
    sono::measure_frequency_parameters(cm, m->time_frames, num_times, fhigh_khz, freq_res, c);
    *io = {c.start_time_s, c.end_time_s, c.duration_ms, c.peak_freq_khz, c.peak_freq_ms, c.peak_power_db, c.high_freq_khz,
           c.high_freq_ms, c.start_freq_khz, c.start_freq_ms, c.start_freq_time_s, c.end_freq_khz, c.end_freq_ms,
           c.end_freq_time_s, c.low_freq_khz, c.low_freq_ms, c.characteristic_freq_khz, c.characteristic_freq_ms,
           c.knee_freq_khz, c.knee_freq_ms, c.heel_freq_khz, c.heel_freq_ms, c.bandwidth_khz, c.high_threshold_db_used,
           c.low_threshold_db_used, c.peak_frame, c.high_freq_frame, c.start_freq_frame, c.start_freq_bin,
           c.end_frame_for_low_freq, c.knee_frame, c.heel_frame, c.discarded, c.reference_throws};
    return SONO_OK;
}

int sono_refine_end_oscillogram(const float *audio, size_t n, double sample_rate, int64_t start_sample, int64_t end_sample,
                                int64_t *refined)
{
    if (!refined || (n && !audio)) return fail(SONO_EINVAL, "sono_refine_end_oscillogram: NULL argument");
    if (!sono::host_refine_end(audio, n, sample_rate, start_sample, end_sample, refined))
        return fail(SONO_EINVAL, "sono_refine_end_oscillogram: sample rate below 20 kHz (the 0.1 ms window has no hop)");
    return SONO_OK;
}

int sono_frame_calls(const sono_segment *segs, size_t count, const double *time_frames, size_t frames, const float *audio,
                     size_t n, double sample_rate, uint32_t fft_size, double padding_ms, double min_call_duration_ms,
                     int full_file_rules, sono_call_frame *out, size_t cap, size_t *out_count)
{
    if (out_count) *out_count = 0;
    if ((count && !segs) || (frames && !time_frames) || (n && !audio) || !(sample_rate > 0) || fft_size == 0)
        return fail(SONO_EINVAL, "sono_frame_calls: NULL or non-positive argument");
    std::vector<sono::FrameRun> runs(count);
    for (size_t i = 0; i < count; ++i) { runs[i].start = segs[i].start; runs[i].end = segs[i].end; }
    std::vector<sono::CallFrame> calls;
    if (!(padding_ms >= 0)) return fail(SONO_EINVAL, "sono_frame_calls: padding_ms must be >= 0");
    if (!sono::host_frame_calls(runs.data(), count, time_frames, frames, audio, n, sample_rate, fft_size, padding_ms,
                                min_call_duration_ms, full_file_rules != 0, calls))
        return fail(SONO_EINVAL, "sono_frame_calls: sample rate below 20 kHz (the 0.1 ms window has no hop)");
    if (calls.size() > cap || (calls.size() && !out)) return fail(SONO_ECAPACITY, "sono_frame_calls: output buffer too small");
    for (size_t i = 0; i < calls.size(); ++i) {
        out[i].start_frame = calls[i].start_frame; out[i].end_frame = calls[i].end_frame;
        out[i].start_time_s = calls[i].start_time_s; out[i].end_time_s = calls[i].end_time_s;
        out[i].duration_ms = calls[i].duration_ms;
    }
    if (out_count) *out_count = calls.size();
    return SONO_OK;
}

int sono_scan_spectrum_for_frequency(const float *row_db, const float *freq_bins, size_t num_bins, double threshold_db,
                                     int high_to_low, double *freq_hz, int64_t *bin)
{
    if (!freq_hz || !bin || (num_bins && (!row_db || !freq_bins)))
        return fail(SONO_EINVAL, "sono_scan_spectrum_for_frequency: NULL argument");
    sono::host_scan_spectrum(row_db, freq_bins, num_bins, threshold_db, high_to_low != 0, freq_hz, bin);
    return SONO_OK;
}

int sono_peak_frequency_khz(const float *row_db, const float *freq_bins, size_t num_bins, int64_t bin, double *khz)
{
    if (!row_db || !freq_bins || !khz || bin < 0 || (uint64_t)bin >= num_bins)
        return fail(SONO_EINVAL, "sono_peak_frequency_khz: NULL argument or bin outside the row");
    const bool interior = bin > 0 && (uint64_t)bin + 1 < num_bins;
    const double width = num_bins > 1 ? (double)freq_bins[1] - (double)freq_bins[0] : 0.0;          // js:3031
    *khz = sono::host_peak_freq_khz(interior ? row_db[bin - 1] : 0.0f, row_db[bin], interior ? row_db[bin + 1] : 0.0f, interior,
                                    freq_bins[bin], width);
    return SONO_OK;
}

int sono_instant_bandwidth_khz(const float *row_db, size_t num_bins, double peak_db, double freq_resolution_hz, double *khz)
{
    if (!khz || (num_bins && !row_db)) return fail(SONO_EINVAL, "sono_instant_bandwidth_khz: NULL argument");
    int32_t lo = -1, hi = -1;
    const double thr = peak_db - 12;                                             // js:3061
    for (size_t b = 0; b < num_bins; ++b)
        if ((double)row_db[b] > thr) { if (lo < 0) lo = (int32_t)b; hi = (int32_t)b; }
    *khz = sono::host_instant_bandwidth_khz(lo, hi, freq_resolution_hz);
    return SONO_OK;
}

double sono_auto_highpass_khz(double peak_freq_khz) { return sono::host_auto_highpass_khz(peak_freq_khz); }

int sono_highpass_filter(const float *in, size_t n, double filter_hz, double sample_rate, double order, float *out)
{
    if (n && (!in || !out)) return fail(SONO_EINVAL, "sono_highpass_filter: NULL argument");
    if (!(sample_rate > 0)) return fail(SONO_EINVAL, "sono_highpass_filter: sample_rate must be positive");
    sono::host_highpass(in, n, filter_hz, sample_rate, order, out);
    return SONO_OK;
}

int sono_savitzky_golay(const double *in, size_t n, double *out)
{
    if (n && (!in || !out)) return fail(SONO_EINVAL, "sono_savitzky_golay: NULL argument");
    if (in == out && n >= 5) return fail(SONO_EINVAL, "sono_savitzky_golay: in and out must not alias");
    sono::host_savitzky_golay(in, n, out);
    return SONO_OK;
}

int sono_wav_parse(const uint8_t *b, size_t len, sono_wav_info *info)
{
    if (!b || !info) return fail(SONO_EINVAL, "sono_wav_parse: NULL argument");
    std::memset(info, 0, sizeof(*info));
    if (len < 12 || std::memcmp(b, "RIFF", 4) != 0 || std::memcmp(b + 8, "WAVE", 4) != 0)
        return fail(SONO_EINVAL, "sono_wav_parse: not a RIFF/WAVE buffer");
    auto u16 = [&](size_t p) { return (uint32_t)b[p] | ((uint32_t)b[p + 1] << 8); };
    auto u32 = [&](size_t p) { return u16(p) | (u16(p + 2) << 16); };
    bool have_fmt = false, have_data = false;
    size_t pos = 12;                                                             // fileLoader.js:11
    // The reference stops at the first 'data' chunk (js:50-53) and takes the sample rate from the first 'fmt '
    // chunk (js:20-22).  This walk keeps both choices -- later 'fmt ' and 'data' chunks change nothing -- but goes
    // on past 'data', on purpose: a GUANO chunk usually follows the samples.
    while (pos + 8 < len) {                                                      // fileLoader.js:12: pos < byteLength - 8
        const uint32_t size = u32(pos + 4);
        if (std::memcmp(b + pos, "fmt ", 4) == 0 && !have_fmt && !have_data && pos + 8 + 16 <= len) {
            info->format_tag = u16(pos + 8);
            info->channels = u16(pos + 10);                                      // fileLoader.js:45
            info->sample_rate = u32(pos + 12);                                   // fileLoader.js:20,46
            info->bits_per_sample = u16(pos + 22);                               // fileLoader.js:47
            if (info->format_tag == 0xFFFEu && size >= 26 && pos + 8 + 26 <= len) info->format_tag = u16(pos + 8 + 24);
            have_fmt = true;
        } else if (std::memcmp(b + pos, "data", 4) == 0 && !have_data) {
            info->data_offset = pos + 8;
            info->data_bytes = std::min<uint64_t>(size, len - (pos + 8));
            have_data = true;
        } else if (std::memcmp(b + pos, "guan", 4) == 0) {
            info->guano_offset = pos + 8;
            info->guano_bytes = std::min<uint64_t>(size, len - (pos + 8));
        }
        pos += 8 + (size_t)size;                                                 // fileLoader.js:22-23
        if (size % 2 == 1) pos += 1;
    }
    if (!have_fmt || !have_data) return fail(SONO_EINVAL, "sono_wav_parse: fmt or data chunk missing");
    const uint32_t bytes = info->bits_per_sample / 8;
    if (info->channels == 0 || bytes == 0 || bytes > 4 || (info->format_tag != 1 && info->format_tag != 3) ||
        (info->format_tag == 3 && bytes != 4))
        return fail(SONO_EINVAL, "sono_wav_parse: unsupported sample format");
    info->frames = info->data_bytes / ((uint64_t)info->channels * bytes);         // fileLoader.js:55-57
    return SONO_OK;
}

int sono_pcm_to_f32_dev(sono_engine *e, const void *d_pcm, size_t frames, uint32_t channels, uint32_t channel,
                        uint32_t bits, int is_float, float *d_out, void *hip_stream)
{
    if (!e || (frames && (!d_pcm || !d_out))) return fail(SONO_EINVAL, "sono_pcm_to_f32_dev: NULL argument");
    if (channels == 0 || channel >= channels || bits % 8 || bits == 0 || bits > 32 || (is_float && bits != 32))
        return fail(SONO_EINVAL, "sono_pcm_to_f32_dev: bad sample format or channel");
    DeviceGuard guard(e->device);
    HIP_TRY(sono::launch_pcm_to_f32(d_pcm, frames, channels, channel, bits, is_float, d_out, pick(e, hip_stream)));
    return SONO_OK;
}

int sono_wav_decode_channel(sono_engine *e, const uint8_t *bytes, size_t len, uint32_t channel, float *out, size_t cap,
                            size_t *count, sono_wav_info *info_out)
{
    if (count) *count = 0;
    if (!e) return fail(SONO_EINVAL, "sono_wav_decode_channel: NULL engine");
    sono_wav_info info;
    int rc = sono_wav_parse(bytes, len, &info);
    if (rc) return rc;
    if (info_out) *info_out = info;
    if (channel >= info.channels) return fail(SONO_EINVAL, "sono_wav_decode_channel: no such channel");
    if (info.frames == 0) return SONO_OK;
    if (!out || cap < info.frames) return fail(SONO_ECAPACITY, "sono_wav_decode_channel: output buffer too small");
    DeviceGuard guard(e->device);
    const size_t used = info.frames * info.channels * (info.bits_per_sample / 8);
    HIP_TRY(e->d_in.reserve(used));
    HIP_TRY(e->d_out.reserve(info.frames * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(e->d_in.p, bytes + info.data_offset, used, hipMemcpyHostToDevice, e->stream));
    rc = sono_pcm_to_f32_dev(e, e->d_in.p, info.frames, info.channels, channel, info.bits_per_sample,
                             info.format_tag == 3, (float *)e->d_out.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, e->d_out.p, info.frames * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (count) *count = info.frames;
    return SONO_OK;
}

int sono_merge_and_pad_segments(sono_segment *segs, size_t count, uint64_t total_samples, double sample_rate,
                                double padding_ms, sono_segment *out, size_t cap, size_t *out_count)
{
    if (out_count) *out_count = 0;
    if (count == 0) return SONO_OK;                                              // js:1029
    if (!segs || !out) return fail(SONO_EINVAL, "sono_merge_and_pad_segments: NULL argument");
    const double pad_d = std::round((padding_ms / 1000.0) * sample_rate);        // Math.round, js:1031
    const int64_t pad = (int64_t)pad_d;
    std::stable_sort(segs, segs + count, [](const sono_segment &a, const sono_segment &b) { return a.start < b.start; });
    auto lo = [&](const sono_segment &s) { const int64_t v = (int64_t)s.start - pad; return v > 0 ? v : (int64_t)0; };
    auto hi = [&](const sono_segment &s) { const int64_t v = (int64_t)s.end + pad; return v < (int64_t)total_samples ? v : (int64_t)total_samples; };
    size_t k = 0;
    int64_t cur_start = lo(segs[0]), cur_end = hi(segs[0]);
    for (size_t i = 1; i < count; ++i) {
        const int64_t ns = lo(segs[i]), ne = hi(segs[i]);
        if (ns <= cur_end) {
            cur_end = ne > cur_end ? ne : cur_end;
        } else {
            if (k >= cap) return fail(SONO_ECAPACITY, "sono_merge_and_pad_segments: output buffer too small");
            out[k].start = (uint64_t)cur_start; out[k].end = (uint64_t)cur_end; ++k;
            cur_start = ns; cur_end = ne;
        }
    }
    if (k >= cap) return fail(SONO_ECAPACITY, "sono_merge_and_pad_segments: output buffer too small");
    out[k].start = (uint64_t)cur_start; out[k].end = (uint64_t)cur_end; ++k;
    if (out_count) *out_count = k;
    return SONO_OK;
}

}  // extern "C"
