// sono_capi_rois.hip -- C ABI of the batched region analysis: processFullFile's
// per-region work (batCallDetector.js:609-711) for all regions of a recording at
// once.  Thousands of regions share each launch (ragged STFT batch, dB crop,
// per-frame peaks, per-region noise-floor histograms, per-region high-pass);
// only one record per frame and a few numbers per region return to the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sonostft.h"
#include "sono_detect.h"
#include "sono_engine.h"
#include "sono_framing.h"
#include "sono_measure.h"
#include "sono_measure_dev.h"
#include "sono_tile.h"

using sono::DeviceGuard;
using sono::fail;

namespace {

constexpr size_t kBatchFrames = (size_t)1 << 21;      // frames whose magnitudes are in HBM at one time (4 GiB at fft 1024)
constexpr size_t kMeasureFrames = (size_t)1 << 17;    // frames whose dB matrices cross to the host at one time
constexpr size_t kDeviceMeasureFrames = (size_t)1 << 21;   // frames of one batch measured on the GPU (4 GiB of magnitudes at fft 1024)
constexpr uint64_t kDeviceCallFrames = 4096;          // longest call the GPU measures (0.34 s at 384 kHz and hop 32); longer ones on the host

struct Band { uint32_t hop, noverlap, min_bin, nb; bool ok; };

// Hop and band of generateSpectrogramWasm (batCallDetector.js:1306-1335); ok = false where it would return null.
Band band_of(uint32_t N, double sample_rate, double flow_khz, double fhigh_khz, double hop_percent)
{
    Band b{0, 0, 0, 0, false};
    const uint32_t bins = N / 2;
    const double hop_d = std::floor((double)N * (hop_percent / 100.0));
    if (!(hop_d >= 1.0)) return b;
    b.hop = hop_d >= (double)N ? N : (uint32_t)hop_d;
    b.noverlap = N - b.hop;
    const double res = sample_rate / (double)N;
    const double lo = std::max(0.0, std::floor(flow_khz * 1000.0 / res));
    const double hi = std::min((double)(bins - 1), std::floor(fhigh_khz * 1000.0 / res));
    if (!(hi - lo + 1.0 > 0.0)) return b;
    b.min_bin = (uint32_t)lo;
    b.nb = (uint32_t)(hi - lo + 1.0);
    b.ok = true;
    return b;
}

// SONO_TIMING=1 prints the wall time of the stages of the whole-file entry points to stderr.
struct StageTimer {
    const char *who;
    bool on = sono::stage_timing();
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    explicit StageTimer(const char *w) : who(w) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[%s] %-28s %8.2f ms\n", who, what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

// Many memcpys at once, shared out over host threads piece by piece: one thread moves about 10 GB/s and pays every
// first-touch page fault of a fresh destination alone; the passes below move gigabytes between the recording and the
// staging areas.
struct CopyJob { void *dst; const void *src; size_t bytes; };
void parallel_copy(const std::vector<CopyJob> &jobs)
{
    size_t total = 0;
    for (const CopyJob &j : jobs) total += j.bytes;
    unsigned int threads = sono::host_threads();
    if (total < ((size_t)8 << 20)) threads = 1;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < jobs.size(); i = next.fetch_add(1))
            std::memcpy(jobs[i].dst, jobs[i].src, jobs[i].bytes);
    };
    std::vector<std::thread> pool;
    for (unsigned int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (std::thread &th : pool) th.join();
}

// Where region r's samples are read from after sono_analyze_rois: its x_filtered holds the regions that went through the
// automatic high-pass (cutoff > 0 and below Nyquist, js:4058) and nothing else.
const float *region_audio(const float *x, const float *x_filtered, const sono_roi_table *t, size_t r, double sample_rate)
{
    const bool filtered = x_filtered && t->highpass_khz && t->highpass_khz[r] > 0 &&
                          (double)t->highpass_khz[r] * 1000.0 / (sample_rate / 2) < 1;
    return filtered ? x_filtered : x;
}

// The recording as the caller's host memory holds it: f32 samples (x) or interleaved 16-bit sample frames (q), of
// which `channel` is meant (the value / 32768, as the browser's decoder hands 16-bit WAV data to the reference).
struct HostSamples {
    const float *x = nullptr;
    const int16_t *q = nullptr;
    uint32_t channels = 1, channel = 0;
};
// -> sample 0 of region r as floats: the high-passed copy where the region went through the filter, the caller's
// floats, or the region's 16-bit samples converted into `tmp` (exact: a division by a power of two).
const float *region_samples(const HostSamples &hs, const float *x_filtered, const sono_segment *rois, const sono_roi_table *t,
                            size_t r, double sample_rate, std::vector<float> &tmp)
{
    const float *src = region_audio(hs.x, x_filtered, t, r, sample_rate);
    if (src) return src + rois[r].start;
    const size_t len = (size_t)(rois[r].end - rois[r].start);
    tmp.resize(len);
    const int16_t *q = hs.q + (size_t)rois[r].start * hs.channels + hs.channel;
    for (size_t i = 0; i < len; ++i) tmp[i] = (float)q[i * hs.channels] * (1.0f / 32768.0f);
    return tmp.data();
}

// One pass over a set of regions whose samples sit at d_src + off[i]: magnitudes -> dB crop -> per-frame peak, in two
// halves so that the next batch can be on the GPU while the host files this one's records.
// peaks_enqueue: the launches and the copy of the records into `records` (pinned), nothing waited for.
int peaks_enqueue(sono_engine *e, const float *d_src, const std::vector<uint64_t> &off, const std::vector<uint64_t> &len,
                  const Band &b, size_t frames, sono::HostBuf &records)
{
    const uint32_t N = e->N, bins = N / 2;
    HIP_TRY(e->d_mags.reserve(frames * bins * sizeof(float)));
    HIP_TRY(e->d_out.reserve(frames * sizeof(sono::FramePeak)));
    sono_ragged_batch rb{d_src, off.size(), off.data(), len.data()};
    int rc = sono::stft_mag_ragged_band(e, &rb, b.noverlap, b.min_bin, b.min_bin + b.nb - 1, (float *)e->d_mags.p);
    if (rc) return rc;
    if (sono::db_peaks_fits(b.nb)) {
        // nobody reads the dB matrix of a region but the peak records (and the noise floors, which convert their six
        // frames per region themselves): it is never written
        HIP_TRY(sono::launch_db_peaks((const float *)e->d_mags.p, frames, bins, b.min_bin, b.nb, N, (sono::FramePeak *)e->d_out.p, e->stream));
    } else {
        HIP_TRY(e->d_db.reserve(frames * b.nb * sizeof(float)));
        rc = sono_detect_power_db_dev(e, (const float *)e->d_mags.p, frames, bins, b.min_bin, b.nb, (float *)e->d_db.p, nullptr);
        if (rc) return rc;
        HIP_TRY(sono::launch_frame_peaks((const float *)e->d_db.p, frames, b.nb, (sono::FramePeak *)e->d_out.p, e->stream));
    }
    if (!records.reserve((frames ? frames : 1) * sizeof(sono::FramePeak)))
        return fail(SONO_ENOMEM, "sono_analyze_rois: cannot allocate host staging memory");
    HIP_TRY(hipMemcpyAsync(records.p, e->d_out.p, frames * sizeof(sono::FramePeak), hipMemcpyDeviceToHost, e->stream));
    return SONO_OK;
}

// peaks_enqueue for bands that db_peaks_fits: the records go straight into the regions' table in HBM (`table`), span by
// span (spans[i]: rows of the batch's matrix -> entries of the table; built by the caller into slot.h_spans), and what
// the automatic high-pass asks about each span comes back in slot.h_out when `want_best`.  Nothing waited for.
int peaks_enqueue_spans(sono_engine *e, const float *d_src, const std::vector<uint64_t> &off, const std::vector<uint64_t> &len,
                        const Band &b, size_t frames, size_t n_spans, const sono::PeakTable &table, sono_engine::MeasureSlot &slot,
                        bool want_best)
{
    const uint32_t N = e->N, bins = N / 2;
    HIP_TRY(e->d_mags.reserve(frames * bins * sizeof(float)));
    sono_ragged_batch rb{d_src, off.size(), off.data(), len.data()};
    int rc = sono::stft_mag_ragged_band(e, &rb, b.noverlap, b.min_bin, b.min_bin + b.nb - 1, (float *)e->d_mags.p);
    if (rc) return rc;
    const size_t sb = n_spans * sizeof(sono::PeakSpan), bb = n_spans * sizeof(sono::SpanBest);
    HIP_TRY(slot.d_spans.reserve(sb + bb + 8));
    sono::PeakSpan *d_spans = (sono::PeakSpan *)slot.d_spans.p;
    sono::SpanBest *d_best = (sono::SpanBest *)((char *)slot.d_spans.p + sb);
    HIP_TRY(hipMemcpyAsync(d_spans, slot.h_spans.p, sb, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(sono::launch_db_peaks_spans((const float *)e->d_mags.p, d_spans, n_spans, bins, b.min_bin, b.nb, N, table,
                                        want_best ? d_best : nullptr, e->stream));
    if (want_best) {
        if (!slot.h_out.reserve((bb ? bb : 1))) return fail(SONO_ENOMEM, "sono_analyze_rois: cannot allocate host staging memory");
        HIP_TRY(hipMemcpyAsync(slot.h_out.p, d_best, bb, hipMemcpyDeviceToHost, e->stream));
    }
    return SONO_OK;
}

// The spans of a run of regions: region i (rows[i] = its first entry in the table, F(i) rows) starts at row `row` of the
// batch's matrix; spans of at most kPeakSpanRows rows, region after region.  region_of (optional): span -> i.
template <typename FramesOf>
bool build_spans(size_t cnt, const std::vector<uint64_t> &rows, FramesOf &&frames_of, sono::HostBuf &h_spans, size_t *n_spans,
                 std::vector<uint32_t> *region_of)
{
    size_t n = 0;
    for (size_t i = 0; i < cnt; ++i) n += (frames_of(i) + sono::kPeakSpanRows - 1) / sono::kPeakSpanRows;
    if (!h_spans.reserve((n ? n : 1) * sizeof(sono::PeakSpan))) return false;
    sono::PeakSpan *sp = (sono::PeakSpan *)h_spans.p;
    if (region_of) region_of->resize(n);
    size_t row = 0, k = 0;
    for (size_t i = 0; i < cnt; ++i) {
        const size_t F = frames_of(i);
        for (size_t f = 0; f < F; f += sono::kPeakSpanRows, ++k) {
            sp[k] = sono::PeakSpan{(unsigned long long)(row + f), (unsigned long long)(rows[i] + f),
                                   (uint32_t)std::min<size_t>(sono::kPeakSpanRows, F - f), 0u};
            if (region_of) (*region_of)[k] = (uint32_t)i;
        }
        row += F;
    }
    *n_spans = n;
    return true;
}

// peaks_collect: the records (arrived: the caller has waited) -> the table's six arrays, the regions shared out over
// host threads.  feat_rows[i] = where region i's records go in the host arrays.
void peaks_collect(sono_engine *e, const std::vector<uint64_t> &len, const std::vector<uint64_t> &feat_rows, const Band &b,
                   size_t frames, const sono_roi_table *t, const sono::HostBuf &records)
{
    const uint32_t N = e->N;
    const sono::FramePeak *feat = (const sono::FramePeak *)records.p;
    std::vector<size_t> first(len.size() + 1, 0);
    for (size_t i = 0; i < len.size(); ++i) first[i + 1] = first[i] + sono_frames_for((size_t)len[i], N, b.noverlap);
    std::atomic<size_t> next{0};
    const size_t step = 64;                                  // regions per grab
    auto work = [&]() {
        for (size_t i0 = next.fetch_add(step); i0 < len.size(); i0 = next.fetch_add(step))
            for (size_t i = i0; i < std::min(i0 + step, len.size()); ++i) {
                const size_t row = first[i], F = first[i + 1] - first[i];
                for (size_t f = 0; f < F; ++f) {
                    t->peak_db[feat_rows[i] + f] = feat[row + f].db;
                    t->peak_bin[feat_rows[i] + f] = feat[row + f].bin;
                    t->peak_left_db[feat_rows[i] + f] = feat[row + f].left;
                    t->peak_right_db[feat_rows[i] + f] = feat[row + f].right;
                    t->inst_lo_bin[feat_rows[i] + f] = feat[row + f].lo12;
                    t->inst_hi_bin[feat_rows[i] + f] = feat[row + f].hi12;
                }
            }
    };
    const unsigned int threads = frames < 100000 ? 1u : sono::host_threads();
    std::vector<std::thread> pool;
    for (unsigned int w = 1; w < threads; ++w) pool.emplace_back(work);
    work();
    for (std::thread &th : pool) th.join();
}

}  // namespace

extern "C" {

int sono_roi_rows(const sono_segment *rois, size_t n_rois, uint32_t fft_size, double hop_percent, uint64_t *row0)
{
    if (!row0 || (n_rois && !rois)) return fail(SONO_EINVAL, "sono_roi_rows: NULL argument");
    const Band b = band_of(fft_size, 2.0 * fft_size, 0.0, 1e9, hop_percent);      // only the hop matters here
    uint64_t rows = 0;
    for (size_t r = 0; r < n_rois; ++r) {
        row0[r] = rows;
        if (rois[r].end < rois[r].start) return fail(SONO_EINVAL, "sono_roi_rows: region with end < start");
        if (b.hop) rows += sono_frames_for((size_t)(rois[r].end - rois[r].start), fft_size, b.noverlap);
    }
    row0[n_rois] = rows;
    return SONO_OK;
}

}  // extern "C"

// sono_analyze_rois.  d_rec: the whole recording in HBM (sono_process_full_file keeps it there from the fast scan on), or
// nullptr.  With it the regions are transformed where they lie -- nothing is packed or uploaded -- and the automatic
// high-pass ALSO replaces the samples of the regions it filters in d_rec, so that whatever reads a region there
// afterwards gets what region_audio() would pick on the host.
static int analyze_rois(sono_engine *e, const float *x, size_t n, const sono_segment *rois, size_t n_rois, double sample_rate,
                        double flow_khz, double fhigh_khz, double hop_percent, int auto_highpass, sono_roi_table *out,
                        size_t frames_cap, float *x_filtered, float *d_rec)
{
    if (!e || !out || (n && !x && !d_rec) || (n_rois && !rois)) return fail(SONO_EINVAL, "sono_analyze_rois: NULL argument");
    if (!(sample_rate > 0)) return fail(SONO_EINVAL, "sono_analyze_rois: sample_rate must be positive");
    if (n_rois && (!out->row0 || !out->zonal || !out->highpass_khz))
        return fail(SONO_EINVAL, "sono_analyze_rois: NULL output array");
    const uint32_t N = e->N;
    const Band b = band_of(N, sample_rate, flow_khz, fhigh_khz, hop_percent);
    out->num_zones = 0; out->min_bin = b.min_bin; out->num_bins = b.nb; out->hop = b.hop;
    for (size_t r = 0; r < n_rois; ++r) {
        if (rois[r].end < rois[r].start || rois[r].end > n) return fail(SONO_EINVAL, "sono_analyze_rois: region outside the recording");
        if (r && rois[r].start < rois[r - 1].end) return fail(SONO_EINVAL, "sono_analyze_rois: regions must be sorted and disjoint");
        out->highpass_khz[r] = 0.0f;
    }
    if (!b.ok) {                                             // no region has a spectrogram (js:1309-1335)
        for (size_t r = 0; r <= n_rois && out->row0; ++r) out->row0[r] = 0;
        return SONO_OK;
    }
    int rc = sono_roi_rows(rois, n_rois, N, hop_percent, out->row0);
    if (rc) return rc;
    const size_t total = (size_t)out->row0[n_rois];
    if (total > frames_cap || (total && (!out->peak_db || !out->peak_bin || !out->peak_left_db || !out->peak_right_db || !out->inst_lo_bin ||
                                    !out->inst_hi_bin)))
        return fail(SONO_ECAPACITY, "sono_analyze_rois: per-frame output arrays too small");
    // frequency axis and its 10 kHz zones (js:1342-1344, 1837-1839)
    std::vector<float> freq_bins(b.nb);
    const double res = sample_rate / (double)N;
    for (uint32_t i = 0; i < b.nb; ++i) freq_bins[i] = (float)((double)(b.min_bin + i) * res);
    std::vector<uint8_t> zone_of_bin;
    const size_t zones = sono::host_zone_map(freq_bins.data(), b.nb, zone_of_bin);
    if (zones == 0 || zones > SONO_MAX_ZONES) return fail(SONO_EINVAL, "sono_analyze_rois: frequency axis needs more than SONO_MAX_ZONES zones");
    out->num_zones = (uint32_t)zones;
    // the three corner frequencies of calculateAutoHighpassFilterFreq (js:4028-4035), order 4 (js:4045, 660)
    sono::HighpassCoef coefs[3];
    const double corners_khz[3] = {20.0, 25.0, 30.0};
    bool corner_ok[3];
    for (int c = 0; c < 3; ++c) {
        const double norm = corners_khz[c] * 1000.0 / (sample_rate / 2);
        corner_ok[c] = norm < 1;                                                  // js:4058: otherwise the audio passes unchanged
        const double wc = std::tan(M_PI * norm / 2), wc2 = wc * wc, s2 = std::sqrt(2.0) * wc, den = wc2 + s2 + 1;
        coefs[c] = {1 / den, -2 / den, 1 / den, (2 * (wc2 - 1)) / den, (wc2 - s2 + 1) / den};      // js:4117-4124
    }

    DeviceGuard guard(e->device);
    hipStream_t s = e->stream;
    StageTimer timer("sono_analyze_rois");
    // Bands of up to 512 bins: the GPU files the per-frame records into the six arrays of the table itself (in HBM),
    // the second pass overwrites the filtered regions' entries there, and the finished arrays reach `out` in six plain
    // copies at the end; of the first pass only a few numbers per region come back (noise floors; the loudest frame's
    // bin, for the automatic high-pass).  Wider bands go through the dB matrix and per-row records the host files.
    const bool on_device = sono::db_peaks_fits(b.nb);
    sono::PeakTable table{};
    if (on_device && total) {
        HIP_TRY(e->d_frame_table.reserve(6 * total * sizeof(float)));
        float *base = (float *)e->d_frame_table.p;
        table = sono::PeakTable{base, (int32_t *)(base + total), base + 2 * total, base + 3 * total, (int32_t *)(base + 4 * total),
                                (int32_t *)(base + 5 * total)};
    }
    auto download_table = [&]() -> int {
        if (!on_device || !total) return SONO_OK;
        const size_t bytes = total * sizeof(float);
        HIP_TRY(hipMemcpyAsync(out->peak_db, table.db, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->peak_bin, table.bin, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->peak_left_db, table.left, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->peak_right_db, table.right, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->inst_lo_bin, table.lo12, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->inst_hi_bin, table.hi12, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        timer.lap("table to the host");
        return SONO_OK;
    };
    // All regions' samples stay in HBM for the whole call (packed, region after region): the first pass works through
    // them in batches of at most kBatchFrames frames, the automatic high-pass then filters every region that needs it
    // in ONE launch -- the filter is a serial recurrence per region, one lane each, so its time is that of the longest
    // region however many regions run beside it --, and the second pass goes through the filtered regions in batches.
    size_t total_samples = 0;
    std::vector<uint64_t> src_off(n_rois);
    for (size_t r = 0; r < n_rois; ++r) {
        src_off[r] = d_rec ? rois[r].start : total_samples;
        total_samples += (size_t)(rois[r].end - rois[r].start);
    }
    if (!d_rec) HIP_TRY(e->d_in.reserve((total_samples ? total_samples : 1) * sizeof(float)));
    float *d_all = d_rec ? d_rec : (float *)e->d_in.p;
    std::vector<sono::HighpassJob> jobs;                     // every region that gets the high-pass
    std::vector<size_t> job_region;
    size_t filtered_samples = 0;
    // Two batches are in flight: while the GPU works on one, the host files the records of the one before and decides
    // about its regions' high-pass.  (Without the resident recording a batch's samples are packed and uploaded first,
    // through one staging area, and the batches follow each other.)
    struct PeakBatch {
        size_t r0 = 0, r1 = 0, frames = 0, samples = 0;
        std::vector<uint64_t> off, len, rows;
        std::vector<unsigned long long> first_row;           // of each region in the batch's matrices
        std::vector<unsigned int> used;                      // frames of each region that count for its noise floors (js:622-627)
        size_t n_spans = 0;                                  // on_device: the batch's row spans, and the region of each
        std::vector<uint32_t> span_region;
    };
    std::vector<PeakBatch> plan;
    for (size_t r0 = 0; r0 < n_rois;) {
        PeakBatch pb;
        pb.r0 = pb.r1 = r0;
        while (pb.r1 < n_rois && (pb.r1 == r0 || pb.frames + (out->row0[pb.r1 + 1] - out->row0[pb.r1]) <= kBatchFrames)) {
            pb.frames += (size_t)(out->row0[pb.r1 + 1] - out->row0[pb.r1]);
            pb.samples += (size_t)(rois[pb.r1].end - rois[pb.r1].start);
            ++pb.r1;
        }
        const size_t cnt = pb.r1 - r0;
        pb.off.resize(cnt); pb.len.resize(cnt); pb.rows.resize(cnt); pb.first_row.resize(cnt); pb.used.resize(cnt);
        for (size_t i = 0; i < cnt; ++i) {
            const sono_segment &g = rois[r0 + i];
            const size_t F = (size_t)(out->row0[r0 + i + 1] - out->row0[r0 + i]);
            pb.off[i] = src_off[r0 + i]; pb.len[i] = g.end - g.start; pb.rows[i] = out->row0[r0 + i];
            pb.first_row[i] = out->row0[r0 + i] - out->row0[r0];
            pb.used[i] = (unsigned int)(F < 6 ? F : 6);
        }
        r0 = pb.r1;
        plan.push_back(std::move(pb));
    }
    for (sono_engine::MeasureSlot &slot : e->measure_slot)          // the measurement stage's slots are free until it starts
        if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    auto enqueue = [&](size_t k) -> int {
        PeakBatch &pb = plan[k];
        sono_engine::MeasureSlot &slot = e->measure_slot[k & 1];
        const size_t cnt = pb.r1 - pb.r0;
        if (!d_rec) {
            // pack the batch's samples and upload them in one copy
            if (!e->h_in.reserve((pb.samples ? pb.samples : 1) * sizeof(float)))
                return fail(SONO_ENOMEM, "sono_analyze_rois: cannot allocate host staging memory");
            float *packed = (float *)e->h_in.p;
            size_t pos = 0;
            std::vector<CopyJob> copies(cnt);
            for (size_t i = 0; i < cnt; ++i) {
                copies[i] = {packed + pos, x + rois[pb.r0 + i].start, (size_t)pb.len[i] * sizeof(float)};
                pos += (size_t)pb.len[i];
            }
            parallel_copy(copies);
            HIP_TRY(hipMemcpyAsync(d_all + src_off[pb.r0], packed, pb.samples * sizeof(float), hipMemcpyHostToDevice, s));
        }
        if (pb.frames == 0) return SONO_OK;
        int rc;
        if (on_device) {
            if (!build_spans(cnt, pb.rows, [&](size_t i) { return (size_t)(out->row0[pb.r0 + i + 1] - out->row0[pb.r0 + i]); },
                             slot.h_spans, &pb.n_spans, &pb.span_region))
                return fail(SONO_ENOMEM, "sono_analyze_rois: cannot allocate host staging memory");
            rc = peaks_enqueue_spans(e, d_all, pb.off, pb.len, b, pb.frames, pb.n_spans, table, slot, true);
        } else {
            rc = peaks_enqueue(e, d_all, pb.off, pb.len, b, pb.frames, slot.h_out);
        }
        if (rc) return rc;
        // ---- noise floors of every region's first six frames (js:622-627), from the unfiltered spectrogram ----
        const size_t tb = cnt * sizeof(unsigned long long), ub = cnt * sizeof(unsigned int);
        const size_t fb = cnt * SONO_MAX_ZONES * sizeof(double);
        HIP_TRY(e->d_edges.reserve(tb + ub + b.nb + 8 + fb));
        if (!slot.h_jobs.reserve(fb)) return fail(SONO_ENOMEM, "sono_analyze_rois: cannot allocate host staging memory");
        char *base = (char *)e->d_edges.p;
        double *d_floors = (double *)base;
        unsigned long long *d_fr = (unsigned long long *)(base + fb);
        unsigned int *d_used = (unsigned int *)(base + fb + tb);
        uint8_t *d_zone = (uint8_t *)(base + fb + tb + ub);
        HIP_TRY(hipMemcpyAsync(d_fr, pb.first_row.data(), tb, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_used, pb.used.data(), ub, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_zone, zone_of_bin.data(), b.nb, hipMemcpyHostToDevice, s));
        if (on_device)
            HIP_TRY(sono::launch_zone_floors_regions_mags((const float *)e->d_mags.p, N / 2, b.min_bin, N, d_fr, d_used, cnt, b.nb,
                                                          d_zone, (uint32_t)zones, SONO_MAX_ZONES, d_floors, s));
        else
            HIP_TRY(sono::launch_zone_floors_regions((const float *)e->d_db.p, d_fr, d_used, cnt, b.nb, d_zone, (uint32_t)zones,
                                                     SONO_MAX_ZONES, d_floors, s));
        HIP_TRY(hipMemcpyAsync(slot.h_jobs.p, d_floors, fb, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(slot.done, s));
        return SONO_OK;
    };
    auto collect = [&](size_t k) -> int {
        PeakBatch &pb = plan[k];
        sono_engine::MeasureSlot &slot = e->measure_slot[k & 1];
        const size_t cnt = pb.r1 - pb.r0;
        if (pb.frames == 0) {
            HIP_TRY(hipStreamSynchronize(s));                // the staging area is reused by the next batch
            for (size_t r = pb.r0; r < pb.r1; ++r) for (size_t z = 0; z < SONO_MAX_ZONES; ++z) out->zonal[r * SONO_MAX_ZONES + z] = -102.0;
            return SONO_OK;
        }
        HIP_TRY(hipEventSynchronize(slot.done));
        timer.lap("wait for STFT, dB, peaks, noise floors");
        std::vector<float> span_db;                          // on_device: every region's loudest frame, joined from its spans in order
        std::vector<int32_t> span_bin;
        if (on_device) {
            span_db.assign(cnt, -INFINITY); span_bin.assign(cnt, 0);
            const sono::SpanBest *sb = (const sono::SpanBest *)slot.h_out.p;
            for (size_t k2 = 0; k2 < pb.n_spans; ++k2) {
                const uint32_t i = pb.span_region[k2];
                if (sb[k2].db > span_db[i]) { span_db[i] = sb[k2].db; span_bin[i] = sb[k2].bin; }
            }
        } else {
            peaks_collect(e, pb.len, pb.rows, b, pb.frames, out, slot.h_out);
        }
        std::memcpy(out->zonal + pb.r0 * SONO_MAX_ZONES, slot.h_jobs.p, cnt * SONO_MAX_ZONES * sizeof(double));
        // ---- automatic high-pass (js:631-671): regions whose loudest bin lies at 30 kHz or above ----
        for (size_t i = 0; auto_highpass && i < cnt; ++i) {
            const size_t F = (size_t)(out->row0[pb.r0 + i + 1] - out->row0[pb.r0 + i]);
            if (F == 0) continue;
            float best = -INFINITY;
            int32_t bin = 0;
            if (on_device) bin = span_bin[i];
            else
                for (size_t f = 0; f < F; ++f)                                     // first maximum in frame order (js:635-644)
                    if (out->peak_db[pb.rows[i] + f] > best) { best = out->peak_db[pb.rows[i] + f]; bin = out->peak_bin[pb.rows[i] + f]; }
            if (bin < 0) bin = 0;
            const double cutoff = sono::host_auto_highpass_khz((double)freq_bins[bin] / 1000);          // js:646-647
            out->highpass_khz[pb.r0 + i] = (float)cutoff;
            if (!(cutoff > 0)) continue;
            const int c = cutoff == 20.0 ? 0 : cutoff == 25.0 ? 1 : 2;
            if (!corner_ok[c]) continue;
            jobs.push_back({pb.off[i], filtered_samples, pb.len[i], (unsigned int)c, 0u});
            job_region.push_back(pb.r0 + i);
            filtered_samples += (size_t)pb.len[i];
        }
        timer.lap("records to tables, high-pass choice");
        return SONO_OK;
    };
    const bool overlap = d_rec != nullptr;
    if (overlap && !plan.empty()) { rc = enqueue(0); if (rc) { (void)hipStreamSynchronize(s); return rc; } }
    for (size_t k = 0; k < plan.size(); ++k) {
        if (!overlap) rc = enqueue(k);
        else rc = k + 1 < plan.size() ? enqueue(k + 1) : SONO_OK;
        if (rc == SONO_OK) rc = collect(k);
        if (rc) { (void)hipStreamSynchronize(s); return rc; }
    }
    if (jobs.empty()) return download_table();
    // ---- the high-pass of all regions that need it, one launch; the filtered samples packed job after job ----
    const size_t jb = jobs.size() * sizeof(sono::HighpassJob);
    HIP_TRY(e->d_runs.reserve(filtered_samples * sizeof(float)));
    HIP_TRY(e->d_flags.reserve(jb + sizeof(coefs)));
    sono::HighpassJob *d_jobs = (sono::HighpassJob *)e->d_flags.p;
    sono::HighpassCoef *d_coefs = (sono::HighpassCoef *)((char *)e->d_flags.p + jb);
    HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), jb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_coefs, coefs, sizeof(coefs), hipMemcpyHostToDevice, s));
    HIP_TRY(sono::launch_highpass_regions(d_all, (float *)e->d_runs.p, d_jobs, jobs.size(), d_coefs, d_rec != nullptr, s));
    // Where the caller wants them (x_filtered; a whole-file detection does not: it frames its calls on the resident
    // recording), the filtered samples go back to the host (in pieces of whole regions through the pinned staging area,
    // then to their places in x_filtered) on a thread and a stream of their own while this thread runs the second pass.
    std::thread copier;
    int copy_status = SONO_OK;
    hipStream_t down = nullptr;
    hipEvent_t filtered = nullptr;
    if (x_filtered) {
        HIP_TRY(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&filtered, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(filtered, s));
        HIP_TRY(hipStreamWaitEvent(down, filtered, 0));
        const int device = e->device;
        copier = std::thread([&, device]() {
            if (hipSetDevice(device) != hipSuccess) { copy_status = SONO_EDEVICE; return; }
            const size_t piece = (size_t)16 << 20;               // samples per piece (a region longer than that goes alone)
            size_t k0 = 0;
            while (k0 < jobs.size()) {
                size_t k1 = k0, count = 0;
                while (k1 < jobs.size() && (k1 == k0 || count + jobs[k1].len <= piece)) count += (size_t)jobs[k1++].len;
                if (!e->h_in.reserve(count * sizeof(float))) { copy_status = SONO_ENOMEM; return; }
                const float *filt = (const float *)e->h_in.p;
                if (hipMemcpyAsync(e->h_in.p, (const float *)e->d_runs.p + jobs[k0].dst, count * sizeof(float), hipMemcpyDeviceToHost,
                                   down) != hipSuccess || hipStreamSynchronize(down) != hipSuccess) {
                    (void)hipGetLastError();
                    copy_status = SONO_EDEVICE;
                    return;
                }
                std::vector<CopyJob> copies(k1 - k0);
                for (size_t k = k0; k < k1; ++k)
                    copies[k - k0] = {x_filtered + rois[job_region[k]].start, filt + (jobs[k].dst - jobs[k0].dst),
                                      (size_t)jobs[k].len * sizeof(float)};
                parallel_copy(copies);
                k0 = k1;
            }
        });
    }
    // ---- second pass over the filtered regions, in batches, two in flight like the first ----
    struct FilteredBatch { std::vector<uint64_t> off, len, rows; std::vector<size_t> region_frames; size_t frames = 0; };
    std::vector<FilteredBatch> plan2;
    for (size_t k0 = 0; k0 < jobs.size();) {
        FilteredBatch fbatch;
        size_t k1 = k0;
        while (k1 < jobs.size()) {
            const size_t r = job_region[k1], F = (size_t)(out->row0[r + 1] - out->row0[r]);
            if (k1 > k0 && fbatch.frames + F > kBatchFrames) break;
            fbatch.off.push_back(jobs[k1].dst); fbatch.len.push_back(jobs[k1].len); fbatch.rows.push_back(out->row0[r]);
            fbatch.region_frames.push_back(F);
            fbatch.frames += F;
            ++k1;
        }
        k0 = k1;
        plan2.push_back(std::move(fbatch));
    }
    auto enqueue2 = [&](size_t k) -> int {
        sono_engine::MeasureSlot &slot = e->measure_slot[k & 1];
        int rc2;
        if (on_device) {
            // the filtered regions' records replace the first pass's in the table
            size_t n_spans = 0;
            if (!build_spans(plan2[k].off.size(), plan2[k].rows, [&](size_t i) { return plan2[k].region_frames[i]; }, slot.h_spans, &n_spans, nullptr))
                return fail(SONO_ENOMEM, "sono_analyze_rois: cannot allocate host staging memory");
            rc2 = peaks_enqueue_spans(e, (const float *)e->d_runs.p, plan2[k].off, plan2[k].len, b, plan2[k].frames, n_spans, table, slot, false);
        } else {
            rc2 = peaks_enqueue(e, (const float *)e->d_runs.p, plan2[k].off, plan2[k].len, b, plan2[k].frames, slot.h_out);
        }
        if (rc2) return rc2;
        HIP_TRY(hipEventRecord(slot.done, s));
        return SONO_OK;
    };
    rc = plan2.empty() ? SONO_OK : enqueue2(0);
    for (size_t k = 0; k < plan2.size() && rc == SONO_OK; ++k) {
        if (k + 1 < plan2.size()) rc = enqueue2(k + 1);
        if (rc == SONO_OK && hipEventSynchronize(e->measure_slot[k & 1].done) != hipSuccess) rc = sono::hip_fail(hipGetLastError(), "hipEventSynchronize");
        if (rc == SONO_OK && !on_device) peaks_collect(e, plan2[k].len, plan2[k].rows, b, plan2[k].frames, out, e->measure_slot[k & 1].h_out);
    }
    if (rc) (void)hipStreamSynchronize(s);
    timer.lap("second pass");
    if (rc == SONO_OK) rc = download_table();
    if (copier.joinable()) copier.join();
    if (filtered) (void)hipEventDestroy(filtered);
    if (down) (void)hipStreamDestroy(down);
    timer.lap("filtered samples on the host");
    if (rc) return rc;
    if (copy_status == SONO_ENOMEM) return fail(SONO_ENOMEM, "sono_analyze_rois: cannot allocate host staging memory");
    if (copy_status) return fail(copy_status, "sono_analyze_rois: copying the filtered samples to the host failed");
    return SONO_OK;
}

extern "C" {

int sono_analyze_rois(sono_engine *e, const float *x, size_t n, const sono_segment *rois, size_t n_rois, double sample_rate,
                      double flow_khz, double fhigh_khz, double hop_percent, int auto_highpass, sono_roi_table *out,
                      size_t frames_cap, float *x_filtered)
{
    return analyze_rois(e, x, n, rois, n_rois, sample_rate, flow_khz, fhigh_khz, hop_percent, auto_highpass, out, frames_cap,
                        x_filtered, nullptr);
}


// Where the oscillogram refinement finds a recording that lies in HBM (sono_process_full_file: the resident recording
// with the high-passed regions in place): the engine on whose stream and buffers the windows are summed.
struct ResidentRecording { sono_engine *e; const float *d_rec; };

static int frame_roi_calls(const HostSamples &hs, const float *x_filtered, size_t n, const sono_segment *rois, size_t n_rois,
                           const sono_roi_table *t, double sample_rate, uint32_t fft_size, double call_threshold_db,
                           double min_call_duration_ms, sono_roi_call *out, size_t cap, size_t *count,
                           std::vector<sono_roi_call> *grown = nullptr, const ResidentRecording *resident = nullptr)
{
    // grown: the calls go into this vector, sized to what was found, instead of out[0..cap)
    // resident: the samples are read in HBM, not on the host.  The calls' spans follow from the per-frame records alone,
    // so the regions are walked twice: once to list the spans the refinement will ask about, then -- the window
    // energies of all of them summed in one launch, 8 bytes per window back instead of the samples themselves --
    // for the calls.
    if (count) *count = 0;
    if (!t || (n && !hs.x && !hs.q && !resident) || (n_rois && (!rois || !t->row0)) || !(sample_rate > 0) || fft_size == 0)
        return fail(SONO_EINVAL, "sono_frame_roi_calls: NULL or non-positive argument");
    for (size_t r = 0; r < n_rois; ++r)
        if (t->row0[r + 1] > t->row0[r] && (rois[r].end > n || rois[r].end < rois[r].start))
            return fail(SONO_EINVAL, "sono_frame_roi_calls: region outside the recording");
    struct Cand { uint64_t s, e, peak_frame; double power, time; };
    // Regions are independent: they are shared out over host threads and their calls joined in region order.
    // What the threads find goes into lists of their own (a list per region would be tens of thousands of small
    // allocations, made by one thread and released by another); a region knows whose list holds its entries and where.
    const unsigned int nt = (unsigned int)std::min<size_t>(sono::host_threads(), std::max<size_t>(n_rois / 64, 1));
    struct Found {
        std::vector<sono_roi_call> calls;
        std::vector<Cand> kept;                              // resident: the candidates the echo filter kept ...
        std::vector<sono::RefinePlan> spans;                 // ... and the spans with windows to sum, in the order the refinement asks
    };
    struct Where { uint32_t owner = 0, calls_owner = 0; size_t calls_at = 0, calls_n = 0, kept_at = 0, kept_n = 0, spans_at = 0, spans_n = 0; };
    std::vector<Found> found(nt);
    std::vector<Where> where(n_rois);
    std::atomic<size_t> next{0};
    std::atomic<bool> low_rate{false};
    std::vector<size_t> sums_at(resident ? n_rois + 1 : 0);  // resident: where each region's first span's sums begin
    const double *sums = nullptr;
    bool listing = resident != nullptr;                      // first walk: list the spans, make no calls
    auto work = [&](unsigned int me) {
        std::vector<double> tf;
        std::vector<float> converted;
        std::vector<sono::CallFrame> framed;
        std::vector<Cand> cands, kept;
        Found &mine = found[me];
        for (size_t r = next.fetch_add(1); r < n_rois; r = next.fetch_add(1)) {
            const size_t row = (size_t)t->row0[r], F = (size_t)(t->row0[r + 1] - t->row0[r]);
            if (F == 0) continue;
            const float *pdb = t->peak_db + row;
            tf.resize(F);
            for (size_t f = 0; f < F; ++f) tf[f] = ((double)f * (double)t->hop + (double)fft_size / 2) / sample_rate;    // js:1350-1351
            Where &here = where[r];
            if (resident && !listing) {                                                 // found on the first walk
                const Cand *first = found[here.owner].kept.data() + here.kept_at;
                kept.assign(first, first + here.kept_n);
            } else {
                float gmax = -INFINITY;
                for (size_t f = 0; f < F; ++f) if (pdb[f] > gmax) gmax = pdb[f];       // js:1544-1550
                const double threshold = (double)gmax + call_threshold_db;              // js:1553
                cands.clear();
                size_t f = 0;
                while (f < F) {                                                         // maximal runs of active frames, js:1556-1595
                    if (!((double)pdb[f] > threshold)) { ++f; continue; }
                    size_t g = f;
                    while (g + 1 < F && (double)pdb[g + 1] > threshold) ++g;
                    if (!(tf[g] - tf[f] < min_call_duration_ms / 1000)) {               // js:688-689
                        size_t pk = f;
                        for (size_t k = f; k <= g; ++k) if (pdb[k] > pdb[pk]) pk = k;   // first maximum, js:695-703
                        cands.push_back({f, g, pk, (double)pdb[pk], tf[pk]});
                    }
                    f = g + 1;
                }
                std::stable_sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.power > b.power; });   // js:714
                kept.clear();
                for (const Cand &c : cands) {                                           // js:720-734
                    bool close = false;
                    for (const Cand &k : kept) if (std::fabs(c.time - k.time) < 0.030) { close = true; break; }
                    if (!close) kept.push_back(c);
                }
                if (resident) {
                    here.owner = me;
                    here.kept_at = mine.kept.size(); here.kept_n = kept.size();
                    mine.kept.insert(mine.kept.end(), kept.begin(), kept.end());
                    here.spans_at = mine.spans.size(); here.spans_n = 0;
                }
            }
            const sono::RefinePlan *listed = resident && !listing ? found[here.owner].spans.data() + here.spans_at : nullptr;
            if (!listing) { here.calls_owner = me; here.calls_at = mine.calls.size(); here.calls_n = 0; }
            const double offset = (double)rois[r].start / sample_rate;                  // js:737
            const size_t region_len = (size_t)(rois[r].end - rois[r].start);
            const float *audio = kept.empty() || resident ? nullptr : region_samples(hs, x_filtered, rois, t, r, sample_rate, converted);
            size_t asked = 0;                                // spans with windows met so far in this region
            const sono::RefineFn refine = [&](int64_t start_sample, int64_t end_sample, int64_t *refined) {
                *refined = end_sample;
                sono::RefinePlan plan;
                const int planned = sono::host_refine_plan(region_len, sample_rate, start_sample, end_sample, plan);
                if (planned <= 0) return planned == 0;
                if (listing) { mine.spans.push_back(plan); ++here.spans_n; return true; }
                size_t at = sums_at[r];
                for (size_t k = 0; k < asked; ++k) at += listed[k].steps;
                ++asked;
                *refined = sono::host_refine_from_sums(plan, sums + at, end_sample);
                return true;
            };
            for (const Cand &c : kept) {
                const sono::FrameRun run{c.s, c.e};
                if (!sono::host_frame_calls(&run, 1, tf.data(), F, audio, region_len, sample_rate,
                                            fft_size, 5.0, min_call_duration_ms, true, framed, resident ? &refine : nullptr)) {   // js:743-810
                    low_rate = true;
                    return;
                }
                if (listing) continue;
                for (const sono::CallFrame &k : framed) {
                    size_t pk = (size_t)k.start_frame;
                    for (size_t q = pk; q <= (size_t)k.end_frame; ++q) if (pdb[q] > pdb[pk]) pk = q;
                    mine.calls.emplace_back();
                    ++here.calls_n;
                    sono_roi_call &o = mine.calls.back();
                    o.roi = r; o.start_frame = k.start_frame; o.end_frame = k.end_frame; o.peak_frame = pk;
                    o.start_time_s = k.start_time_s + offset;                           // js:822-823
                    o.end_time_s = k.end_time_s + offset;
                    o.duration_ms = k.duration_ms;
                    o.peak_time_s = tf[pk] + offset;
                    o.peak_power_db = pdb[pk];
                    o.peak_bin = t->peak_bin[row + pk];
                    const int32_t pb = o.peak_bin;
                    const bool interior = pb > 0 && (uint32_t)pb + 1 < t->num_bins;                      // js:3023
                    const double res = sample_rate / (double)fft_size;
                    const float f_bin = (float)((double)(t->min_bin + (uint32_t)(pb < 0 ? 0 : pb)) * res);
                    const double width = (double)(float)((double)(t->min_bin + 1) * res) - (double)(float)((double)t->min_bin * res);
                    o.peak_freq_khz = sono::host_peak_freq_khz(t->peak_left_db[row + pk], pdb[pk], t->peak_right_db[row + pk], interior,
                                                               f_bin, width);
                    o.peak_freq_ms = (tf[pk] - tf[(size_t)k.start_frame]) * 1000;                        // js:3041-3043
                    o.inst_bandwidth_khz = sono::host_instant_bandwidth_khz(t->inst_lo_bin[row + pk], t->inst_hi_bin[row + pk], res);
                    o.discarded = o.inst_bandwidth_khz > 20.0 ? 1 : 0;                                   // js:3080
                    o.pad = 0;
                }
            }
        }
    };
    auto walk = [&]() {
        next = 0;
        std::vector<std::thread> pool;
        for (unsigned int w = 1; w < nt; ++w) pool.emplace_back(work, w);
        work(0);
        for (std::thread &th : pool) th.join();
    };
    StageTimer timer("sono_frame_roi_calls");
    timer.lap("set up");
    walk();
    if (low_rate) return fail(SONO_EINVAL, "sono_frame_roi_calls: sample rate below 20 kHz (the 0.1 ms window has no hop)");
    if (resident) {
        timer.lap("spans listed");
        // the spans' windows, summed where the samples lie
        sono_engine *e = resident->e;
        size_t n_spans = 0, windows = 0;
        auto spans_of = [&](size_t r) { return found[where[r].owner].spans.data() + where[r].spans_at; };
        for (size_t r = 0; r < n_rois; ++r) {
            sums_at[r] = windows;
            n_spans += where[r].spans_n;
            for (size_t k = 0; k < where[r].spans_n; ++k) windows += spans_of(r)[k].steps;
        }
        sums_at[n_rois] = windows;
        if (n_spans) {
            DeviceGuard guard(e->device);
            const size_t jobs_bytes = (n_spans * sizeof(sono::WindowJob) + 255) / 256 * 256, sums_bytes = windows * sizeof(double);
            if (!e->h_refine.reserve(jobs_bytes + sums_bytes))
                return fail(SONO_ENOMEM, "sono_frame_roi_calls: cannot allocate host staging memory");
            HIP_TRY(e->d_refine.reserve(jobs_bytes + sums_bytes));
            sono::WindowJob *jobs = (sono::WindowJob *)e->h_refine.p;
            uint32_t window = 0, hop = 0;
            size_t k = 0, at = 0;
            for (size_t r = 0; r < n_rois; ++r)
                for (size_t q = 0; q < where[r].spans_n; ++q) {
                    const sono::RefinePlan &p = spans_of(r)[q];
                    const uint64_t first = rois[r].start + (uint64_t)p.safe_start;
                    if (first + (uint64_t)(p.steps - 1) * (uint64_t)p.hop + (uint64_t)p.window > rois[r].end)
                        return fail(SONO_EINVAL, "sono_frame_roi_calls: a span's windows leave its region");
                    jobs[k++] = {first, at, (uint32_t)p.steps, 0u};
                    at += p.steps;
                    window = (uint32_t)p.window; hop = (uint32_t)p.hop;          // the same for every span: they follow from the sample rate
                }
            hipStream_t s = e->stream;
            char *d = (char *)e->d_refine.p;
            HIP_TRY(hipMemcpyAsync(d, jobs, n_spans * sizeof(sono::WindowJob), hipMemcpyHostToDevice, s));
            HIP_TRY(sono::launch_window_sums(resident->d_rec, (const sono::WindowJob *)d, n_spans, window, hop, (double *)(d + jobs_bytes), s));
            HIP_TRY(hipMemcpyAsync((char *)e->h_refine.p + jobs_bytes, d + jobs_bytes, sums_bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            sums = (const double *)((const char *)e->h_refine.p + jobs_bytes);
        }
        timer.lap("windows summed (GPU)");
        listing = false;
        walk();
        timer.lap("calls framed");
        if (low_rate) return fail(SONO_EINVAL, "sono_frame_roi_calls: sample rate below 20 kHz (the 0.1 ms window has no hop)");
    }
    size_t made = 0;
    if (grown) {
        size_t total = 0;
        for (const Found &f : found) total += f.calls.size();
        grown->resize(total);
        out = grown->data();
        cap = total;
    }
    for (size_t r = 0; r < n_rois; ++r) {                    // in the order of the regions
        const Where &here = where[r];
        if (made + here.calls_n > cap || (!out && here.calls_n))
            return fail(SONO_ECAPACITY, "sono_frame_roi_calls: output buffer too small");
        const sono_roi_call *first = found[here.calls_owner].calls.data() + here.calls_at;
        for (size_t k = 0; k < here.calls_n; ++k) out[made++] = first[k];
    }
    timer.lap("calls joined");
    if (count) *count = made;
    return SONO_OK;
}

int sono_frame_roi_calls(const float *x, const float *x_filtered, size_t n, const sono_segment *rois, size_t n_rois, const sono_roi_table *t,
                         double sample_rate, uint32_t fft_size, double call_threshold_db, double min_call_duration_ms,
                         sono_roi_call *out, size_t cap, size_t *count)
{
    HostSamples hs;
    hs.x = x;
    return frame_roi_calls(hs, x_filtered, n, rois, n_rois, t, sample_rate, fft_size, call_threshold_db, min_call_duration_ms, out, cap, count);
}

// sono_measure_roi_calls; `keep` (or nullptr) receives, for every call that processFullFile goes on to keep (not
// discarded, longer than 1 ms: js:802-810), the call's own rows of the dB matrix and the number of its centre times.
// d_rec (or nullptr): the recording in HBM with the filtered regions already in place (analyze_rois): the calls' frames
// are then transformed where they lie instead of being packed on the host and uploaded.
static int measure_roi_calls(sono_engine *e, const float *x, const float *x_filtered, size_t n, const sono_segment *rois, size_t n_rois,
                             const sono_roi_table *t, const sono_roi_call *calls, size_t n_calls, double sample_rate, double flow_khz,
                             double fhigh_khz, double hop_percent, int debug_mode, unsigned int threads, sono_call_measurement *out,
                             std::vector<sono::FullFileSlice> *keep, const float *d_rec)
{
    if (!e || !t || (n && !x && !d_rec) || (n_rois && (!rois || !t->row0 || !t->zonal)) || (n_calls && (!calls || !out)))
        return fail(SONO_EINVAL, "sono_measure_roi_calls: NULL argument");
    if (!(sample_rate > 0)) return fail(SONO_EINVAL, "sono_measure_roi_calls: sample_rate must be positive");
    if (n_calls == 0) return SONO_OK;
    const uint32_t N = e->N, bins = N / 2;
    const Band b = band_of(N, sample_rate, flow_khz, fhigh_khz, hop_percent);
    if (!b.ok || b.hop != t->hop || b.nb != t->num_bins || b.min_bin != t->min_bin)
        return fail(SONO_EINVAL, "sono_measure_roi_calls: band or hop differ from the table's");
    std::vector<float> freq_bins(b.nb);
    const double res = sample_rate / (double)N;
    for (uint32_t i = 0; i < b.nb; ++i) freq_bins[i] = (float)((double)(b.min_bin + i) * res);
    // A call's frames are transformed again from the (filtered) recording.  The slice starts on a tile boundary of its
    // region and runs to one (or to the region's end), so every frame meets the same partner in the kernel's frame
    // pairs as it did in the region's own transform and the dB values are the same bit for bit.
    unsigned int align = sono::tile_frames_per_wave(N);
    if (align == 0) align = 16;
    struct Slice { uint64_t first, count, lead; };          // first frame transformed, frames transformed, frames before the call
    std::vector<Slice> slices(n_calls);
    for (size_t c = 0; c < n_calls; ++c) {
        const sono_roi_call &k = calls[c];
        if (k.roi >= n_rois) return fail(SONO_EINVAL, "sono_measure_roi_calls: call names a region that does not exist");
        const uint64_t F = t->row0[k.roi + 1] - t->row0[k.roi];
        if (k.end_frame < k.start_frame || k.end_frame >= F || rois[k.roi].end > n || rois[k.roi].end < rois[k.roi].start)
            return fail(SONO_EINVAL, "sono_measure_roi_calls: call outside its region");
        const uint64_t first = k.start_frame - k.start_frame % align;
        uint64_t last = (k.end_frame / align + 1) * align;                          // one past the last frame transformed
        if (last > F) last = F;
        slices[c] = {first, last - first, k.start_frame - first};
    }
    DeviceGuard guard(e->device);
    hipStream_t s = e->stream;
    if (threads == 0) threads = sono::host_threads();
    StageTimer timer("sono_measure_roi_calls");
    // Where the calls are measured.  On the GPU (sono_measure_dev.hip, one wave per call) the dB matrices stay in HBM
    // and 280 bytes per call come back; on host threads every matrix crosses the link first -- twice the bytes of
    // the recording itself for an hour of bat passes -- which is what the stage then waits for.  The host does it when
    // the caller keeps the matrices anyway, asks for it (engine option "host_measure"), or a call is longer than the
    // GPU's per-wave work area is laid out for.
    uint64_t longest = 0;
    for (size_t c = 0; c < n_calls; ++c) longest = std::max<uint64_t>(longest, calls[c].end_frame - calls[c].start_frame + 1);
    const bool on_gpu = !keep && !e->host_measure && longest <= kDeviceCallFrames && t->num_zones <= SONO_MAX_ZONES;
    // a wave per call: batches of tens of thousands of calls keep every wave slot of the GPU busy; towards the host the
    // batch is what two slots of pinned staging memory hold
    const size_t batch_frames = on_gpu ? kDeviceMeasureFrames : kMeasureFrames;
    const float *d_freq = nullptr;
    const double *d_zonal = nullptr;
    if (on_gpu) {
        const size_t fbytes = ((size_t)b.nb * sizeof(float) + 255) / 256 * 256, zbytes = n_rois * SONO_MAX_ZONES * sizeof(double);
        HIP_TRY(e->d_measure_tables.reserve(fbytes + zbytes));
        d_freq = (const float *)e->d_measure_tables.p;
        d_zonal = (const double *)((const char *)e->d_measure_tables.p + fbytes);
        HIP_TRY(hipMemcpyAsync((void *)d_freq, freq_bins.data(), (size_t)b.nb * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync((void *)d_zonal, t->zonal, zbytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));                    // both sources are the caller's pageable memory
    }
    // Batches of at most kMeasureFrames frames go through two slots of staging memory: while the host threads measure
    // the calls of one batch, the next one is packed, uploaded, transformed and copied back.
    struct Batch {
        size_t c0 = 0, c1 = 0, frames = 0, samples = 0;
        std::vector<uint64_t> off, len, row;
    };
    for (sono_engine::MeasureSlot &slot : e->measure_slot)
        if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    auto prepare = [&](Batch &bt, size_t c0, sono_engine::MeasureSlot &slot) -> int {
        bt.c0 = bt.c1 = c0; bt.frames = bt.samples = 0;
        while (bt.c1 < n_calls && (bt.c1 == c0 || bt.frames + slices[bt.c1].count <= batch_frames)) {
            bt.frames += (size_t)slices[bt.c1].count;
            bt.samples += (size_t)(slices[bt.c1].count - 1) * b.hop + N;
            ++bt.c1;
        }
        const size_t cnt = bt.c1 - c0;
        if (!slot.h_out.reserve(on_gpu ? cnt * sizeof(sono_call_measurement) : bt.frames * b.nb * sizeof(float)) ||
            (!d_rec && !slot.h_in.reserve(bt.samples * sizeof(float))) || (on_gpu && !slot.h_jobs.reserve(cnt * sizeof(sono::MeasureJob))))
            return fail(SONO_ENOMEM, "sono_measure_roi_calls: cannot allocate host staging memory");
        bt.off.resize(cnt); bt.len.resize(cnt); bt.row.resize(cnt);
        size_t pos = 0, rows = 0;
        std::vector<CopyJob> copies(d_rec ? 0 : cnt);
        for (size_t i = 0; i < cnt; ++i) {
            const sono_roi_call &k = calls[c0 + i];
            const Slice &sl = slices[c0 + i];
            bt.len[i] = (sl.count - 1) * b.hop + N; bt.row[i] = rows;
            if (d_rec) bt.off[i] = rois[k.roi].start + sl.first * b.hop;
            else {
                bt.off[i] = pos;
                copies[i] = {(float *)slot.h_in.p + pos,
                             region_audio(x, x_filtered, t, (size_t)k.roi, sample_rate) + rois[k.roi].start + sl.first * b.hop,
                             (size_t)bt.len[i] * sizeof(float)};
            }
            pos += (size_t)bt.len[i];
            rows += (size_t)sl.count;
        }
        HIP_TRY(e->d_mags.reserve(bt.frames * bins * sizeof(float)));
        HIP_TRY(slot.d_db.reserve(bt.frames * b.nb * sizeof(float)));
        if (!d_rec) {
            parallel_copy(copies);
            HIP_TRY(slot.d_in.reserve(bt.samples * sizeof(float)));
            HIP_TRY(hipMemcpyAsync(slot.d_in.p, slot.h_in.p, bt.samples * sizeof(float), hipMemcpyHostToDevice, s));
        }
        sono_ragged_batch rb{d_rec ? d_rec : (const float *)slot.d_in.p, cnt, bt.off.data(), bt.len.data()};
        int rc = sono::stft_mag_ragged_band(e, &rb, b.noverlap, b.min_bin, b.min_bin + b.nb - 1, (float *)e->d_mags.p);
        if (rc) return rc;
        rc = sono_detect_power_db_dev(e, (const float *)e->d_mags.p, bt.frames, bins, b.min_bin, b.nb, (float *)slot.d_db.p, nullptr);
        if (rc) return rc;
        if (!on_gpu) {
            HIP_TRY(hipMemcpyAsync(slot.h_out.p, slot.d_db.p, bt.frames * b.nb * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(slot.done, s));
            return SONO_OK;
        }
        // the batch's calls, measured where their matrices lie (js:786-797)
        sono::MeasureJob *jobs = (sono::MeasureJob *)slot.h_jobs.p;
        unsigned int most = 1;
        for (size_t i = 0; i < cnt; ++i) {
            const sono_roi_call &k = calls[c0 + i];
            const uint64_t F = t->row0[k.roi + 1] - t->row0[k.roi];
            const uint64_t call_frames = k.end_frame - k.start_frame + 1;
            jobs[i] = {bt.row[i] + slices[c0 + i].lead, k.start_frame, (unsigned int)call_frames,
                       (unsigned int)(k.end_frame + 1 < F ? call_frames + 1 : call_frames), (unsigned int)k.roi, 0u, k.duration_ms};
            most = std::max(most, (unsigned int)call_frames);
        }
        const unsigned int waves = sono::measure_waves(cnt, e->tables.tile.cus);
        const size_t per_wave = sono::measure_scratch_per_wave(most, b.nb);
        HIP_TRY(slot.d_jobs.reserve(cnt * sizeof(sono::MeasureJob)));
        HIP_TRY(slot.d_results.reserve(cnt * sizeof(sono_call_measurement)));
        HIP_TRY(e->d_measure_scratch.reserve((size_t)waves * per_wave));
        HIP_TRY(hipMemcpyAsync(slot.d_jobs.p, jobs, cnt * sizeof(sono::MeasureJob), hipMemcpyHostToDevice, s));
        sono::MeasureArgs ma{(const float *)slot.d_db.p, b.nb, d_freq, d_zonal, t->num_zones, SONO_MAX_ZONES,
                             (const sono::MeasureJob *)slot.d_jobs.p, cnt, (double)b.hop, (double)N / 2, sample_rate, fhigh_khz, res,
                             (debug_mode & SONO_DEBUG_MODE) != 0, (debug_mode & SONO_TIME_EXPANSION) != 0,
                             (char *)e->d_measure_scratch.p, per_wave, most, (sono_call_measurement *)slot.d_results.p};
        HIP_TRY(sono::launch_measure_calls(ma, waves, s));
        HIP_TRY(hipMemcpyAsync(slot.h_out.p, slot.d_results.p, cnt * sizeof(sono_call_measurement), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(slot.done, s));
        return SONO_OK;
    };
    // measureFrequencyParameters per call (js:786-797), the batch's calls shared out over host threads
    auto measure = [&](const Batch &bt, const float *db) {
        const size_t c0 = bt.c0, cnt = bt.c1 - bt.c0;
        if (on_gpu) {                                        // what came back is the measurements themselves
            std::memcpy(out + c0, db, cnt * sizeof(sono_call_measurement));
            return;
        }
        std::atomic<size_t> next{0};
        auto work = [&]() {
            std::vector<double> tf;
            for (size_t i = next.fetch_add(1); i < cnt; i = next.fetch_add(1)) {
                const sono_roi_call &k = calls[c0 + i];
                const uint64_t F = t->row0[k.roi + 1] - t->row0[k.roi];
                const size_t call_frames = (size_t)(k.end_frame - k.start_frame + 1);
                const size_t times = k.end_frame + 1 < F ? call_frames + 1 : call_frames;          // js:787
                tf.resize(times);
                for (size_t f = 0; f < times; ++f)
                    tf[f] = ((double)(k.start_frame + f) * (double)b.hop + (double)N / 2) / sample_rate;     // js:1350-1351
                const sono::CallMatrix m{db + (bt.row[i] + (size_t)slices[c0 + i].lead) * b.nb, call_frames, b.nb, freq_bins.data(),
                                         t->zonal + (size_t)k.roi * SONO_MAX_ZONES, t->num_zones, (debug_mode & SONO_DEBUG_MODE) != 0,
                                         (debug_mode & SONO_TIME_EXPANSION) != 0};
                sono::CallMeasure cm;
                cm.start_time_s = tf[0];                                                           // js:791-794
                cm.end_time_s = tf[times - 1];
                cm.duration_ms = k.duration_ms;
                sono::measure_frequency_parameters(m, tf.data(), times, fhigh_khz, res, cm);
                out[c0 + i] = {cm.start_time_s, cm.end_time_s, cm.duration_ms, cm.peak_freq_khz, cm.peak_freq_ms, cm.peak_power_db,
                               cm.high_freq_khz, cm.high_freq_ms, cm.start_freq_khz, cm.start_freq_ms, cm.start_freq_time_s,
                               cm.end_freq_khz, cm.end_freq_ms, cm.end_freq_time_s, cm.low_freq_khz, cm.low_freq_ms,
                               cm.characteristic_freq_khz, cm.characteristic_freq_ms, cm.knee_freq_khz, cm.knee_freq_ms,
                               cm.heel_freq_khz, cm.heel_freq_ms, cm.bandwidth_khz, cm.high_threshold_db_used, cm.low_threshold_db_used,
                               cm.peak_frame, cm.high_freq_frame, cm.start_freq_frame, cm.start_freq_bin, cm.end_frame_for_low_freq,
                               cm.knee_frame, cm.heel_frame, cm.discarded, cm.reference_throws};
                if (keep && !cm.discarded && !cm.reference_throws && cm.duration_ms > 1.0) {
                    sono::FullFileSlice &sl = (*keep)[c0 + i];
                    sl.db.assign(m.db, m.db + call_frames * b.nb);                               // js:785
                    sl.times = (uint32_t)times;
                }
            }
        };
        const unsigned int nt = (unsigned int)std::min<size_t>(threads, cnt);
        std::vector<std::thread> pool;
        for (unsigned int w = 1; w < nt; ++w) pool.emplace_back(work);
        work();
        for (std::thread &th : pool) th.join();
    };
    Batch batch[2];
    int rc = prepare(batch[0], 0, e->measure_slot[0]);
    if (rc) return rc;
    timer.lap("first batch queued");
    for (unsigned int cur = 0;; cur ^= 1u) {
        const Batch &bt = batch[cur];
        const bool more = bt.c1 < n_calls;
        if (more) {
            rc = prepare(batch[cur ^ 1u], bt.c1, e->measure_slot[cur ^ 1u]);
            if (rc) { (void)hipStreamSynchronize(s); return rc; }
            timer.lap("next batch packed and queued");
        }
        HIP_TRY(hipEventSynchronize(e->measure_slot[cur].done));
        timer.lap("wait for the batch's dB");
        measure(bt, (const float *)e->measure_slot[cur].h_out.p);
        timer.lap("measure on host threads");
        if (!more) break;
    }
    return SONO_OK;
}

int sono_measure_roi_calls(sono_engine *e, const float *x, const float *x_filtered, size_t n, const sono_segment *rois, size_t n_rois, const sono_roi_table *t,
                           const sono_roi_call *calls, size_t n_calls, double sample_rate, double flow_khz, double fhigh_khz,
                           double hop_percent, int debug_mode, unsigned int threads, sono_call_measurement *out)
{
    return measure_roi_calls(e, x, x_filtered, n, rois, n_rois, t, calls, n_calls, sample_rate, flow_khz, fhigh_khz, hop_percent,
                             debug_mode, threads, out, nullptr, nullptr);
}

// What processFullFile does with regions once it has them (batCallDetector.js:609-827): analysis, call framing,
// measurement, follow-up.  The regions are rois[0 .. n_rois) of a recording that lies in HBM at d_rec (at least up to
// the last region's end); a kept call names its region as roi_base + its index here.  Appends to the engine's call list;
// calls_seen counts the framed calls of the recording so far (the reference's TypeError is reported by that number).
struct StageTimes { double analysis = 0, framing = 0, measurement = 0; };
// What is left to do for a group of regions once they are analysed -- the framing of their calls, the calls'
// measurement and what follows it -- with everything it reads: it can run beside the next group's analysis.  The
// per-frame records (t's arrays) lie in one of the engine's two sets of tables; the next group's analysis writes the
// other.
struct GroupTail {
    std::vector<sono_segment> rois;
    std::vector<uint64_t> row0;
    std::vector<float> hp;
    std::vector<double> zonal;
    sono_roi_table t{};
    size_t roi_base = 0;
    std::vector<sono_roi_call> framed;       // what the framing found, for the measurement
    size_t n_calls = 0;
};

// Analysis of the group of regions g.rois (js:609-671) on engine e, into buffer set `set` of engine `keeper`; the rest of the group's way is left in `g`.
static int analyze_regions(sono_engine *e, sono_engine *keeper, const HostSamples &hs, size_t n, double sample_rate, double flow_khz,
                           double fhigh_khz, const sono_full_file_options *opt, float *d_rec, unsigned int set, StageTimes &times, GroupTail &g)
{
    const uint32_t N = e->N;
    const auto t0 = std::chrono::steady_clock::now();
    const sono_segment *rois = g.rois.data();                // (the group's regions and roi_base are the caller's)
    const size_t n_rois = g.rois.size();
    // per-region analysis, all regions at once (js:609-711)
    g.row0.resize(n_rois + 1);
    int rc = sono_roi_rows(rois, n_rois, N, opt->hop_percent, g.row0.data());
    if (rc) return rc;
    const size_t total = (size_t)g.row0[n_rois];
    // six per-frame arrays, every entry of which the analysis writes: one block the engine keeps (fresh pages again)
    sono::HostBuf &tables = keeper->frame_tables[set];
    if (!tables.reserve(6 * (total + 1) * sizeof(uint32_t))) return fail(SONO_ENOMEM, "sono_process_full_file: cannot allocate the regions' tables");
    float *peak_db = (float *)tables.p, *left = peak_db + (total + 1), *right = left + (total + 1);
    int32_t *peak_bin = (int32_t *)(right + (total + 1)), *ilo = peak_bin + (total + 1), *ihi = ilo + (total + 1);
    g.hp.resize(n_rois + 1);
    g.zonal.resize((n_rois + 1) * SONO_MAX_ZONES);
    g.t = sono_roi_table{g.row0.data(), peak_db, peak_bin, left, right, ilo, ihi, g.zonal.data(), g.hp.data(), 0, 0, 0, 0};
    // The automatic high-pass filters its regions where they lie in the resident recording, and there they stay: the
    // framing stage has the GPU sum the windows it asks about, and no sample returns to the host.
    rc = analyze_rois(e, hs.x, n, rois, n_rois, sample_rate, flow_khz, fhigh_khz, opt->hop_percent, 1, &g.t, total, nullptr, d_rec);
    if (rc) return rc;
    static std::mutex clock;                                 // (two groups may be analysed side by side)
    std::lock_guard<std::mutex> l(clock);
    times.analysis += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SONO_OK;
}

// The rest of an analysed group's way, first half: the framing of its calls (js:677-794, host threads; the windows of
// the oscillogram refinement summed on engine ef: its stream, its buffers).
static int frame_group(sono_engine *e, sono_engine *ef, const HostSamples &hs, size_t n, double sample_rate,
                       const sono_full_file_options *opt, GroupTail &g, float *d_rec, StageTimes &times)
{
    const auto t0 = std::chrono::steady_clock::now();
    // the list is sized by the framing: a bound from the frame count would be hundreds of MB of fresh pages
    g.n_calls = 0;
    const ResidentRecording resident{ef, d_rec};
    int rc = frame_roi_calls(hs, nullptr, n, g.rois.data(), g.rois.size(), &g.t, sample_rate, e->N, opt->call_threshold_db,
                             opt->min_call_duration_ms, nullptr, 0, &g.n_calls, &g.framed, &resident);
    times.framing += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// Second half: the measurement (js:797) of the framed calls on engine em (its stream, its buffers) and what follows it
// (js:802-827); the calls kept join the list of engine e.
static int measure_group(sono_engine *e, sono_engine *em, const HostSamples &hs, size_t n, double sample_rate, double flow_khz,
                         double fhigh_khz, const sono_full_file_options *opt, GroupTail &g, size_t *calls_seen, float *d_rec,
                         StageTimes &times)
{
    const auto t0 = std::chrono::steady_clock::now();
    const size_t n_calls = g.n_calls;
    const std::vector<sono_roi_call> &framed = g.framed;
    std::vector<sono_call_measurement> meas(n_calls + 1);
    const bool keep_slices = (opt->debug_mode & SONO_KEEP_CALL_SPECTROGRAMS) != 0;
    std::vector<sono::FullFileSlice> slices(keep_slices ? n_calls : 0);
    int rc = measure_roi_calls(em, hs.x, nullptr, n, g.rois.data(), g.rois.size(), &g.t, framed.data(), n_calls, sample_rate, flow_khz,
                               fhigh_khz, opt->hop_percent, opt->debug_mode, opt->threads, meas.data(), keep_slices ? &slices : nullptr, d_rec);
    if (rc) return rc;
    e->full_file_axes = {g.t.hop, g.t.min_bin, g.t.num_bins, sample_rate};
    std::vector<sono_full_file_call> &kept = e->full_file_calls;
    for (size_t i = 0; i < n_calls; ++i) {
        sono_call_measurement &m = meas[i];
        if (m.reference_throws) {
            kept.clear();
            e->full_file_slices.clear();
            return fail(SONO_EREFERENCE, "sono_process_full_file: the reference raises a TypeError on call " + std::to_string(*calls_seen + i) +
                                             " of this recording outside its debug mode (batCallDetector.js:2261, 2815)");
            }
        if (m.discarded || m.duration_ms <= 1.0) continue;                          // js:802-810
        const double offset = (double)g.rois[framed[i].roi].start / sample_rate;    // js:737, 822-825
        m.start_time_s += offset;
        m.end_time_s += offset;
        if (m.start_freq_time_s == m.start_freq_time_s) m.start_freq_time_s += offset;
        if (m.end_freq_time_s == m.end_freq_time_s) m.end_freq_time_s += offset;
        kept.emplace_back();
        if (keep_slices) e->full_file_slices.push_back(std::move(slices[i]));
        sono_full_file_call &o = kept.back();
        o.m = m;
        o.roi = g.roi_base + framed[i].roi; o.start_frame = framed[i].start_frame; o.end_frame = framed[i].end_frame;
        o.highpass_khz = g.hp[framed[i].roi];
        o.flow_hz = m.low_freq_khz * 1000;                                          // js:813-814
        o.fhigh_khz = m.high_freq_khz;
        o.snr_db = m.peak_power_db - (-80);                                         // js:818
        const double bw = m.bandwidth_khz;                                          // CallTypeClassifier.classify, js:48-56
        o.call_type = (!(bw == bw) || bw == 0 || bw < 5) ? SONO_CALL_CF : bw > 20 ? SONO_CALL_FM : SONO_CALL_CF_FM;
        o.quality = o.snr_db < 10 ? 0 : o.snr_db < 15 ? 1 : o.snr_db < 20 ? 2 : o.snr_db < 30 ? 3 : 4;    // getQualityRating, js:375-387
    }
    *calls_seen += n_calls;
    times.measurement += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SONO_OK;
}

// Both halves in line, on one engine.
static int finish_regions(sono_engine *e, sono_engine *em, const HostSamples &hs, size_t n, double sample_rate, double flow_khz,
                          double fhigh_khz, const sono_full_file_options *opt, GroupTail &g, size_t *calls_seen, float *d_rec,
                          StageTimes &times)
{
    int rc = frame_group(e, em, hs, n, sample_rate, opt, g, d_rec, times);
    if (rc) return rc;
    return measure_group(e, em, hs, n, sample_rate, flow_khz, fhigh_khz, opt, g, calls_seen, d_rec, times);
}

// The groups' way from their regions to their calls on threads of their own, beside the scan of the pieces that
// follow: group k is analysed (on the analysing engine) while group k - 1's calls are framed (host threads; the window
// sums on the framing engine) and group k - 2's are measured (on the measuring engine) -- each kind of stage one group
// at a time and in the groups' order, so the calls join the list in the order of the recording.  A group's thread
// analyses as soon as the group two before it has been analysed (two analysing engines take the groups in turn: one
// group's waits for the GPU are the other's work) and the group four before it has ended (whose set of tables it writes), frames as soon as the group before it has been framed, and measures as soon as that group has
// ended; a failure anywhere ends the groups behind it without further work.
struct TailPipeline {
    struct Tail {
        std::unique_ptr<GroupTail> group;
        std::thread worker;
        std::mutex m;
        std::condition_variable cv;
        bool analysed = false, framed = false, ended = false;
        int rc = SONO_OK;
        std::string error;
        void mark(bool Tail::*flag) { { std::lock_guard<std::mutex> l(m); this->*flag = true; } cv.notify_all(); }
        void await(bool Tail::*flag) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return this->*flag; }); }
    };
    static constexpr unsigned int kAnalysers = 2;    // groups analysed side by side (one analysing engine each)
    static constexpr unsigned int kSets = kAnalysers + 2;    // sets of tables: being analysed, being framed, being measured
    std::vector<std::unique_ptr<Tail>> tails;        // in the groups' order; ended ones stay (their rc is read at the end)
    std::atomic<long long> us{0};                    // the threads' own time in their stages, beside whatever else went on
    size_t joined = 0;                               // tails before this one have been joined
    using Stage = std::function<int(GroupTail &)>;
    using Analysis = std::function<int(GroupTail &, unsigned int k)>;      // k: the group's number (engine k % kAnalysers, set k % kSets)
    void start(std::unique_ptr<GroupTail> g, Analysis analyse, Stage frame, Stage measure)
    {
        const size_t k = tails.size();
        for (size_t j = joined; j < k; ++j) {                // groups that have ended give their threads back (in order: they end in order)
            { std::lock_guard<std::mutex> l(tails[j]->m); if (!tails[j]->ended) break; }
            if (tails[j]->worker.joinable()) tails[j]->worker.join();
            joined = j + 1;
        }
        Tail *prev = k ? tails[k - 1].get() : nullptr, *owner = k >= kSets ? tails[k - kSets].get() : nullptr;
        Tail *before = k >= kAnalysers ? tails[k - kAnalysers].get() : nullptr;      // the group that had this group's analysing engine
        tails.emplace_back(new Tail());
        Tail *t = tails.back().get();
        t->group = std::move(g);
        t->worker = std::thread([this, t, prev, owner, before, k, analyse, frame, measure]() {
            int rc = SONO_OK;
            auto run = [&](const std::function<int()> &stage) {
                if (rc != SONO_OK) return;
                const auto t0 = std::chrono::steady_clock::now();
                rc = stage();
                if (rc != SONO_OK) t->error = sono_last_error();
                us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            };
            if (before) before->await(&Tail::analysed);
            if (owner) owner->await(&Tail::ended);
            run([&] { return analyse(*t->group, (unsigned int)k); });
            t->mark(&Tail::analysed);
            if (prev) prev->await(&Tail::framed);
            run([&] { return frame(*t->group); });
            t->mark(&Tail::framed);
            if (prev) {
                prev->await(&Tail::ended);
                if (rc == SONO_OK && prev->rc != SONO_OK) { rc = prev->rc; t->error = prev->error; }
            }
            run([&] { return measure(*t->group); });
            t->rc = rc;
            t->group.reset();
            t->mark(&Tail::ended);
        });
    }
    // -> once every group has ended; the first failure (they carry it down the line) becomes this thread's
    int finish()
    {
        for (auto &t : tails) if (t->worker.joinable()) t->worker.join();
        for (auto &t : tails) if (t->rc != SONO_OK) return fail(t->rc, t->error);
        return SONO_OK;
    }
    ~TailPipeline() { for (auto &t : tails) if (t->worker.joinable()) t->worker.join(); }
};

// The recording's way to the GPU, piece by piece, on a thread and a stream of its own: piece p has arrived (and, for
// 16-bit samples, been converted into the resident float recording) when arrived[p] has completed.
struct PieceUpload {
    sono_engine *e;
    const HostSamples &hs;
    size_t n, piece;                         // samples of the recording; samples per piece (a multiple of the fft size)
    float *d_rec;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> arrived;
    std::atomic<size_t> queued{0};           // pieces whose event has been recorded
    std::atomic<int> status{SONO_OK};
    std::thread worker;
    PieceUpload(sono_engine *e_, const HostSamples &hs_, size_t n_, size_t piece_, float *d) : e(e_), hs(hs_), n(n_), piece(piece_), d_rec(d) {}
    size_t pieces() const { return (n + piece - 1) / piece; }
    int start()
    {
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        arrived.assign(pieces(), nullptr);
        for (hipEvent_t &ev : arrived) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        if (hs.q) HIP_TRY(e->d_in.reserve(std::min(n, piece) * (size_t)hs.channels * sizeof(int16_t)));
        const int device = e->device;
        // half the process's host threads fill the pinned pieces (enough for the link's rate): the other stages' host
        // work runs beside this thread's
        const unsigned int copy_threads = std::max(1u, sono::host_threads() / 2);
        worker = std::thread([this, device, copy_threads]() {
            if (hipSetDevice(device) != hipSuccess) { status = SONO_EDEVICE; queued = pieces(); return; }
            for (size_t p = 0; p < pieces(); ++p) {
                const size_t f0 = p * piece, cnt = std::min(piece, n - f0);
                int rc;
                if (hs.q) {
                    // 16-bit sample frames cross the link as they are (half the bytes of the float recording) and become
                    // the resident float recording on the device: everything reads the same floats as the f32 entry
                    // point would have uploaded (value / 32768 is exact).  One staging buffer: copy and conversion follow
                    // each other on this stream, so a piece is converted before the next one overwrites it.
                    const size_t frame_bytes = (size_t)hs.channels * sizeof(int16_t);
                    rc = sono::upload_host_array(e, e->d_in.p, hs.q + f0 * hs.channels, cnt * frame_bytes, stream, copy_threads);
                    if (rc == SONO_OK && sono::launch_pcm_to_f32(e->d_in.p, cnt, hs.channels, hs.channel, 16, 0, d_rec + f0, stream) != hipSuccess)
                        rc = SONO_EDEVICE;
                } else {
                    rc = sono::upload_host_array(e, d_rec + f0, hs.x + f0, cnt * sizeof(float), stream, copy_threads);
                }
                if (rc == SONO_OK && hipEventRecord(arrived[p], stream) != hipSuccess) rc = SONO_EDEVICE;
                if (rc != SONO_OK) { (void)hipGetLastError(); status = rc; queued = pieces(); return; }
                queued = p + 1;
            }
        });
        return SONO_OK;
    }
    // -> once piece p's event has been recorded (or the upload has failed): makes stream `s` wait for it
    int wait_on(size_t p, hipStream_t s)
    {
        while (queued.load() <= p) std::this_thread::sleep_for(std::chrono::microseconds(50));     // a piece takes milliseconds
        if (status.load() != SONO_OK) return fail(status.load(), "sono_process_full_file: the upload of the recording failed");
        HIP_TRY(hipStreamWaitEvent(s, arrived[p], 0));
        return SONO_OK;
    }
    ~PieceUpload()
    {
        if (worker.joinable()) worker.join();
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        for (hipEvent_t ev : arrived) if (ev) (void)hipEventDestroy(ev);
    }
};

static int process_full_file(sono_engine *e, const HostSamples &hs, size_t n, double sample_rate, double flow_khz, double fhigh_khz,
                             const sono_full_file_options *opt, sono_full_file_call *out, size_t cap, size_t *count)
{
    if (count) *count = 0;
    if (!e || !opt || (n && !hs.x && !hs.q) || (cap && !out)) return fail(SONO_EINVAL, "sono_process_full_file: NULL argument");
    if (hs.q && (hs.channels == 0 || hs.channel >= hs.channels)) return fail(SONO_EINVAL, "sono_process_full_file_pcm16: channel out of range");
    e->full_file_calls.clear();
    e->full_file_slices.clear();
    if (!(sample_rate > 0)) return fail(SONO_EINVAL, "sono_process_full_file: sample_rate must be positive");
    const uint32_t N = e->N, hop = N / 2, bins = N / 2;                              // the fast scan's hop, js:882-886
    StageTimer timer("sono_process_full_file");
    if (n < N) return SONO_OK;                                                      // no frame, no segment (js:595-598)
    // The fast scan's band and threshold (fastScanSegmentsWasm, js:889-909)
    const double freq_res = sample_rate / (double)N;
    const double lo = std::floor(flow_khz * 1000.0 / freq_res), hi = std::ceil(fhigh_khz * 1000.0 / freq_res);
    const size_t F = sono_frames_for(n, N, N - hop);
    if (!(hi >= lo) || hi < 0.0) return SONO_OK;                                    // empty band: no frame can be active
    const uint32_t min_bin = lo > 0.0 ? (lo > 4.0e9 ? 0xFFFFFFFFu : (uint32_t)lo) : 0u;
    const uint32_t max_bin = hi > 4.0e9 ? 0xFFFFFFFFu : (uint32_t)hi;
    const double target = std::pow(10.0, opt->threshold_db / 10.0) * (double)N * 0.25;
    DeviceGuard guard(e->device);
    HIP_TRY(e->d_rec.reserve(n * sizeof(float)));
    float *d_rec = (float *)e->d_rec.p;
    StageTimes times;
    size_t calls_seen = 0;
    int rc;
    // The recording goes to HBM once and stays there: the fast scan, the regions' and the calls' transforms all read it
    // where it lies.  It goes there in pieces, and what can be done with the pieces that have arrived is done while the
    // next ones cross the link: the scan's flags are local to a frame, its runs and the merged, padded regions are final
    // as soon as the scan has moved a padding's length past them, and a region's analysis, framing and measurement need
    // that region's samples only.  (A band that reaches the last bin makes the reference's flat index run into the next
    // frame's row, js:918-921: that scan wants the whole matrix, and the recording then goes up in one piece.)
    const bool whole = e->no_upload_overlap || max_bin >= bins;
    // samples per piece, a multiple of four hops of the scan (a wave of the scan's kernel takes four frames): 128 MB of
    // 16-bit samples; in one go, still no more than 512 MB at a time through the staging buffer
    const size_t piece = e->small_upload_pieces ? (size_t)64 * N : whole ? (size_t)1 << 28 : (size_t)1 << 26;
    // A group of regions is analysed (GPU, with this thread filing what comes back), its calls are framed (host
    // threads) and measured (GPU).  When another group is to follow, framing and measurement run on a thread of their
    // own -- the measurement on the engine's companion: a stream and buffers of its own -- while this thread goes on
    // with the next pieces' scan and the next group's analysis; the two groups' records and filtered samples lie in the
    // engine's two sets of buffers.  One such tail at a time, in the groups' order: the calls join the list in the order
    // of the recording.
    sono_engine *ea[2] = {nullptr, nullptr}, *ef = nullptr, *em = nullptr;
    if (!whole && !e->no_measure_overlap) {
        for (sono_engine **companion : {&e->analysis_engine[0], &e->analysis_engine[1], &e->frame_engine, &e->measure_engine}) {
            if (!*companion && sono_engine_create(N, e->window_name.c_str(), 1, e->alpha, e->device, companion) != SONO_OK)
                *companion = nullptr;                        // without them the groups are worked on in line
            if (*companion) {
                (*companion)->host_measure = e->host_measure;
                (*companion)->tables.force_generic = e->tables.force_generic;
                (*companion)->tables.tile.any_hop = e->tables.tile.any_hop;
                (*companion)->tables.tile.balance.fixed = e->tables.tile.balance.fixed;
            }
        }
        if (e->analysis_engine[0] && e->analysis_engine[1] && e->frame_engine && e->measure_engine) {
            ea[0] = e->analysis_engine[0]; ea[1] = e->analysis_engine[1]; ef = e->frame_engine; em = e->measure_engine;
        }
    }
    StageTimes tail_times;                                   // the stages' own clocks on the groups' threads, beside this thread's
    TailPipeline tails;                                      // (after everything its threads refer to: it joins them when it goes)
    auto run_group = [&](const sono_segment *rois, size_t n_rois, size_t roi_base) -> int {
        if (n_rois == 0) return SONO_OK;
        std::unique_ptr<GroupTail> g(new GroupTail());
        g->rois.assign(rois, rois + n_rois);
        g->roi_base = roi_base;
        if (em) {
            tails.start(std::move(g),
                        [&, ea, d_rec](GroupTail &gt, unsigned int k) {
                            // (the sets of tables are the first analysing engine's)
                            return analyze_regions(ea[k % TailPipeline::kAnalysers], ea[0], hs, n, sample_rate, flow_khz, fhigh_khz, opt, d_rec,
                                                   k % TailPipeline::kSets, tail_times, gt);
                        },
                        [&, ef, d_rec](GroupTail &gt) { return frame_group(e, ef, hs, n, sample_rate, opt, gt, d_rec, tail_times); },
                        [&, em, d_rec](GroupTail &gt) {
                            return measure_group(e, em, hs, n, sample_rate, flow_khz, fhigh_khz, opt, gt, &calls_seen, d_rec, tail_times);
                        });
            return SONO_OK;
        }
        int r = analyze_regions(e, e, hs, n, sample_rate, flow_khz, fhigh_khz, opt, d_rec, 0u, times, *g);
        if (r != SONO_OK) return r;
        return finish_regions(e, e, hs, n, sample_rate, flow_khz, fhigh_khz, opt, *g, &calls_seen, d_rec, times);
    };
    PieceUpload up(e, hs, n, piece, d_rec);
    rc = up.start();
    if (rc) return rc;
    if (whole) {
        rc = up.wait_on(up.pieces() - 1, e->stream);
        if (rc) return rc;
        std::vector<sono_segment> &raw = e->scan_segments;  // kept by the engine: tens of MB of fresh pages per call otherwise
        size_t n_raw = 0;
        if (raw.size() < n / std::max<size_t>(hop, 1) + 2) raw.resize(n / std::max<size_t>(hop, 1) + 2);
        rc = sono::fast_scan_resident(e, d_rec, n, sample_rate, flow_khz, fhigh_khz, opt->threshold_db, raw.data(), raw.size(), &n_raw);
        if (rc) return rc;
        timer.lap("upload, fast scan");
        if (n_raw == 0) return SONO_OK;                                             // js:595-598
        std::vector<sono_segment> rois(n_raw);
        size_t n_rois = 0;
        rc = sono_merge_and_pad_segments(raw.data(), n_raw, n, sample_rate, opt->padding_ms, rois.data(), rois.size(), &n_rois);
        if (rc) return rc;
        rc = run_group(rois.data(), n_rois, 0);
        if (rc) return rc;
    } else {
        HIP_TRY(e->d_flags.reserve(F));
        std::vector<uint8_t> flags(F);
        const double pad_d = std::round((opt->padding_ms / 1000.0) * sample_rate);   // mergeAndPadSegments, js:1031
        const int64_t pad = (int64_t)pad_d;
        bool in_run = false, have_cur = false;
        uint64_t run_first = 0;
        int64_t cur_start = 0, cur_end = 0;
        size_t scanned = 0, roi_base = 0;                   // frames of the scan done; regions handed on
        std::vector<sono_segment> group;
        double t_wait = 0, t_scan = 0;
        auto add_segment = [&](uint64_t first, uint64_t last) {          // a run of active frames (js:930-945), merged as js:1028-1056 does
            const int64_t s0 = (int64_t)(first * hop), e0 = (int64_t)(last + 1 < F ? (last + 1) * hop + N : (uint64_t)n);
            const int64_t ns = s0 - pad > 0 ? s0 - pad : 0, ne = e0 + pad < (int64_t)n ? e0 + pad : (int64_t)n;
            if (have_cur && ns <= cur_end) { cur_end = ne > cur_end ? ne : cur_end; return; }
            if (have_cur) group.push_back({(uint64_t)cur_start, (uint64_t)cur_end});
            cur_start = ns; cur_end = ne; have_cur = true;
        };
        for (size_t p = 0; p < up.pieces(); ++p) {
            auto t0 = std::chrono::steady_clock::now();
            rc = up.wait_on(p, e->stream);
            if (rc) return rc;
            const size_t have = std::min(n, (p + 1) * piece);            // samples in HBM once the stream gets there
            size_t f_end = have >= N ? (have - N) / hop + 1 : 0;         // frames that lie in them
            if (f_end > F) f_end = F;
            if (p + 1 < up.pieces()) f_end &= ~(size_t)3;                 // the kernel takes the frames of a launch four to a wave, in pairs: launches start on a multiple of four
            if (f_end > scanned) {
                const size_t cnt = f_end - scanned, used = (cnt - 1) * (size_t)hop + N;
                sono_batch b{d_rec + scanned * hop, 1, used, used};
                rc = sono_stft_band_active_dev(e, &b, N - hop, min_bin, max_bin, target, (uint8_t *)e->d_flags.p + scanned, nullptr, nullptr);
                if (rc) return rc;
                HIP_TRY(hipMemcpyAsync(flags.data() + scanned, (const uint8_t *)e->d_flags.p + scanned, cnt, hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(hipStreamSynchronize(e->stream));
                t_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                t0 = std::chrono::steady_clock::now();
                for (size_t f = scanned; f < f_end; ++f) {
                    if (flags[f] && !in_run) { in_run = true; run_first = f; }
                    else if (!flags[f] && in_run) { in_run = false; add_segment(run_first, f - 1); }
                }
                scanned = f_end;
            }
            const bool last_piece = p + 1 == up.pieces();
            if (last_piece) {
                if (in_run) { in_run = false; add_segment(run_first, F - 1); }
                if (have_cur) { group.push_back({(uint64_t)cur_start, (uint64_t)cur_end}); have_cur = false; }
            } else if (have_cur) {
                // whatever comes next starts at the open run's first frame or at the first frame not scanned yet
                const int64_t next_start = (int64_t)((in_run ? run_first : scanned) * hop) - pad;
                if (next_start > cur_end) { group.push_back({(uint64_t)cur_start, (uint64_t)cur_end}); have_cur = false; }
            }
            t_scan += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            // Regions are handed on three thousand at a time.  Fewer make the measurement's launches too small (it is
            // one wave per call: fewer calls than the GPU has room for waves leave it idle for as long as one call
            // takes; 1 500 cost a sixth more), more leave a longer chain of analysis, framing and measurement behind
            // the recording's last piece (12 000: 100 ms per hour of audio against 84).
            if (!group.empty() && (last_piece || group.size() >= (e->small_upload_pieces ? (size_t)6 : (size_t)3000))) {
                rc = run_group(group.data(), group.size(), roi_base);
                if (rc) return rc;
                roi_base += group.size();
                group.clear();
            }
        }
        rc = tails.finish();
        if (rc) return rc;
        if (timer.on) {
            std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "waiting for pieces, fast scan", t_wait);
            std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "runs, regions", t_scan);
        }
    }
    if (timer.on) {
        std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "region analysis (GPU)", times.analysis);
        std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "framing (host)", times.framing);
        std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "measurement (GPU + threads)", times.measurement);
        std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "on threads of their own:", (double)tails.us.load() / 1000);
        std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "   region analysis", tail_times.analysis);
        std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "   framing", tail_times.framing);
        std::fprintf(stderr, "[sono_process_full_file] %-28s %8.2f ms\n", "   measurement", tail_times.measurement);
    }
    return sono_full_file_results(e, out, cap, count);
}

int sono_process_full_file(sono_engine *e, const float *x, size_t n, double sample_rate, double flow_khz, double fhigh_khz,
                           const sono_full_file_options *opt, sono_full_file_call *out, size_t cap, size_t *count)
{
    HostSamples hs;
    hs.x = x;
    return process_full_file(e, hs, n, sample_rate, flow_khz, fhigh_khz, opt, out, cap, count);
}

int sono_process_full_file_pcm16(sono_engine *e, const int16_t *pcm, size_t n, uint32_t channels, uint32_t channel, double sample_rate,
                                 double flow_khz, double fhigh_khz, const sono_full_file_options *opt, sono_full_file_call *out,
                                 size_t cap, size_t *count)
{
    HostSamples hs;
    hs.q = pcm; hs.channels = channels; hs.channel = channel;
    return process_full_file(e, hs, n, sample_rate, flow_khz, fhigh_khz, opt, out, cap, count);
}

int sono_full_file_results(sono_engine *e, sono_full_file_call *out, size_t cap, size_t *count)
{
    if (!e || (cap && !out)) return fail(SONO_EINVAL, "sono_full_file_results: NULL argument");
    const std::vector<sono_full_file_call> &kept = e->full_file_calls;
    if (count) *count = kept.size();
    const size_t take = std::min(cap, kept.size());
    if (take) std::memcpy(out, kept.data(), take * sizeof(sono_full_file_call));
    if (kept.size() > cap) return fail(SONO_ECAPACITY, "sono_full_file_results: output buffer too small (count holds the number of calls)");
    return SONO_OK;
}

int sono_full_file_call_spectrogram(sono_engine *e, size_t index, float *power_db, size_t cap_floats, double *time_frames,
                                    size_t cap_times, float *freq_bins, size_t cap_bins, size_t *frames, size_t *times,
                                    uint32_t *num_bins)
{
    if (!e) return fail(SONO_EINVAL, "sono_full_file_call_spectrogram: NULL engine");
    if (e->full_file_slices.size() != e->full_file_calls.size() || e->full_file_calls.empty())
        return fail(SONO_EINVAL, "sono_full_file_call_spectrogram: the engine's last sono_process_full_file kept no spectrograms "
                                 "(SONO_KEEP_CALL_SPECTROGRAMS was not set, or it found no calls)");
    if (index >= e->full_file_slices.size()) return fail(SONO_EINVAL, "sono_full_file_call_spectrogram: no such call");
    const sono::FullFileSlice &sl = e->full_file_slices[index];
    const sono::FullFileAxes &ax = e->full_file_axes;
    const size_t F = sl.db.size() / ax.num_bins;
    if (frames) *frames = F;
    if (times) *times = sl.times;
    if (num_bins) *num_bins = ax.num_bins;
    if ((power_db && cap_floats < sl.db.size()) || (time_frames && cap_times < sl.times) || (freq_bins && cap_bins < ax.num_bins))
        return fail(SONO_ECAPACITY, "sono_full_file_call_spectrogram: output buffer too small (frames, times and num_bins hold the sizes)");
    if (power_db) std::memcpy(power_db, sl.db.data(), sl.db.size() * sizeof(float));
    const double res = ax.sample_rate / (double)e->N;
    if (freq_bins)
        for (uint32_t i = 0; i < ax.num_bins; ++i) freq_bins[i] = (float)((double)(ax.min_bin + i) * res);          // js:1343-1345
    if (time_frames) {
        const uint64_t f0 = e->full_file_calls[index].start_frame;
        for (uint32_t f = 0; f < sl.times; ++f)
            time_frames[f] = ((double)(f0 + f) * (double)ax.hop + (double)e->N / 2) / ax.sample_rate;              // js:1350-1351
    }
    return SONO_OK;
}

}  // extern "C"
