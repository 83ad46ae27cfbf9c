// sono_measure.cpp -- see sono_measure.h.  Built with -ffp-contract=off.
//
// Both searches lower a threshold from 1 to 100 dB below the call's peak, one dB at
// a time, follow the call's lower (upper) edge while it moves smoothly, and stop
// where it jumps into the noise.  The reference keeps a log table only in its
// debug mode and consults it unconditionally in one branch of each search
// (batCallDetector.js:2815, 2261); outside debug mode that branch raises a
// TypeError.  reference_throws reports that the inputs lead there; with
// CallMatrix::debug_mode set the searches go on as the reference's debug mode does.
#include "sono_measure.h"

#include <cmath>
#include <vector>

#include "sono_framing.h"
#include "sono_measure_core.h"
#include "sono_rowscan.h"

namespace sono {

size_t host_zonal_floors(const float *db, size_t frames, uint32_t bins, const float *fb, int64_t start_frame, int64_t end_frame,
                         double *floors, size_t cap)
{
    std::vector<uint8_t> zone_of_bin;
    const size_t zones = host_zone_map(fb, bins, zone_of_bin);
    if (zones == 0 || zones > cap) return 0;
    std::vector<unsigned int> hist(zones * 141, 0u);
    for (int64_t f = start_frame < 0 ? 0 : start_frame; f <= end_frame; ++f) {
        if (f >= (int64_t)frames) break;                                         // js:1850
        const float *row = db + (size_t)f * bins;
        for (uint32_t b = 0; b < bins; ++b) {
            double v = (double)row[b];
            if (v != v || zone_of_bin[b] >= zones) continue;
            if (v < -100.0) v = -100.0;                                          // js:1857
            double fl = std::floor(v);                                           // js:1867
            fl = fl < -120 ? -120 : fl > 20 ? 20 : fl;                           // js:1870-1871
            hist[zone_of_bin[b] * 141 + (size_t)((int)fl + 120)]++;
        }
    }
    host_zone_modes(hist.data(), zones, floors);
    return zones;
}

namespace {

// The host's way through the shared searches: one thread per call, rows scanned four entries per comparison.
struct HostOps {
    double own_zonal[256];
    int64_t first_gt(const float *p, int64_t lo, int64_t hi, float t) const { return sono::first_gt(p, lo, hi, t); }
    int64_t last_gt(const float *p, int64_t lo, int64_t hi, float t) const { return sono::last_gt(p, lo, hi, t); }
    float row_max(const float *p, size_t n) const { return sono::row_max(p, n); }
    int64_t first_max(const float *p, size_t n, float *value) const { return sono::first_max(p, n, value); }
    template <class F> void for_each(int64_t n, const F &f) const { for (int64_t i = 0; i < n; ++i) f(i); }
    // one thread walks the frames itself
    void begin_low(const CallMatrix &, double, int64_t, int64_t) const {}
    void begin_high(const CallMatrix &, double, int64_t) const {}
    template <class BelowRef> bool low_walk(int, int64_t, int64_t, int64_t, const BelowRef &, int64_t &) const { return false; }
    template <class Edge, class Harmonic>
    bool high_walk(int, int64_t, int64_t, const Edge &, const Harmonic &, double &, int64_t &, int64_t &, bool &) const { return false; }
    // the floors of a call that comes without a zonal map, from the call's own frames f0..f1 (js:2437-2439, 1975-1977)
    const double *zonal_floors(const CallMatrix &m, int64_t f0, int64_t f1, size_t *zones)
    {
        *zones = host_zonal_floors(m.db, m.frames, m.bins, m.freq_bins, f0, f1, own_zonal, 256);
        return own_zonal;
    }
    // per bin: the largest entry of frames 0..last and the first frame that holds it (js:3160-3171)
    void max_spectrum(const CallMatrix &m, int64_t last, float *max_spec, uint16_t *frame_of) const
    {
        for (uint32_t b = 0; b < m.bins; ++b) { max_spec[b] = -INFINITY; frame_of[b] = 0; }
        for (int64_t f = 0; f <= last; ++f) {
            const float *row = m.db + (size_t)f * m.bins;
            for (uint32_t b = 0; b < m.bins; ++b)
                if (row[b] > max_spec[b]) { max_spec[b] = row[b]; frame_of[b] = (uint16_t)f; }
        }
    }
};

// The work area of one thread, grown to the largest call it has met.
struct HostWork {
    mcore::LowM low[mcore::kThresholds];
    mcore::HighM high[mcore::kThresholds];
    int valid[mcore::kThresholds];
    std::vector<double> per_frame;
    std::vector<int64_t> frames;
    std::vector<float> max_spec;
    std::vector<uint16_t> frame_of;
    mcore::Work sized(size_t F, size_t bins)
    {
        if (per_frame.size() < 5 * F) per_frame.resize(5 * F);
        if (frames.size() < F) frames.resize(F);
        if (max_spec.size() < bins) { max_spec.resize(bins); frame_of.resize(bins); }
        double *d = per_frame.data();
        return mcore::Work{low, high, valid, d, d + F, d + 2 * F, d + 3 * F, d + 4 * F, frames.data(), max_spec.data(), frame_of.data()};
    }
};
HostWork &host_work()
{
    static thread_local HostWork w;
    return w;
}

}  // namespace

LowResult find_optimal_low_threshold(const CallMatrix &m, double peak, int64_t peak_frame, int64_t limit_frame)
{
    HostOps ops;
    mcore::Work w = host_work().sized(m.frames, m.bins);
    const double *zonal = m.zonal;
    size_t zones = m.num_zones;
    const int64_t F = (int64_t)m.frames;
    if (!zonal && F > 0 && m.bins > 0) {                                         // js:2427-2439
        const int64_t search_end = (limit_frame >= 0 && limit_frame < F) ? limit_frame : F - 1;
        zonal = ops.zonal_floors(m, peak_frame < F - 1 ? peak_frame : F - 1, search_end, &zones);
    }
    return mcore::find_low(m, ops, w, zonal, zones, peak, peak_frame, limit_frame);
}

HighResult find_optimal_high_threshold(const CallMatrix &m, const double *tf, double peak, int64_t peak_frame)
{
    HostOps ops;
    mcore::Work w = host_work().sized(m.frames, m.bins);
    const double *zonal = m.zonal;
    size_t zones = m.num_zones;
    const int64_t F = (int64_t)m.frames;
    if (!zonal && F > 0 && m.bins > 0) zonal = ops.zonal_floors(m, 0, peak_frame < F - 1 ? peak_frame : F - 1, &zones);      // js:1950, 1975-1977
    return mcore::find_high(m, ops, w, zonal, zones, tf, peak, peak_frame);
}

void measure_frequency_parameters(const CallMatrix &m, const double *tf, size_t T, double fhigh_khz, double freq_res, CallMeasure &c)
{
    HostOps ops;
    mcore::Work w = host_work().sized(m.frames, m.bins);
    mcore::measure(m, ops, w, tf, T, fhigh_khz, freq_res, c);
}

void apply_time_expansion(CallMeasure &c, double factor) { mcore::time_expansion(c, factor); }

}  // namespace sono
