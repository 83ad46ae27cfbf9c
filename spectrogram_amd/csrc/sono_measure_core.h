// sono_measure_core.h -- the detector's measurement stage (the two threshold searches and measureFrequencyParameters
// of batCallDetector.js), written once for two executors:
//   * the host: one thread per call, rows scanned four entries per comparison (sono_rowscan.h);
//   * the GPU: one 64-lane wave per call; the lanes share out the bins of a row for every scan along a row and carry
//     out everything else (the search logic, all of it in doubles) in step on the same values.
// What differs between the two is the `Ops` object (row scans, the zonal noise floors of a call without a map);
// the searches themselves are the same source and so give the same numbers.  An executor may also take over one
// threshold's walk along the frames (`low_walk`, `high_walk`: the GPU answers it for 64 frames at a time from a table
// it builds in `begin_low` / `begin_high`); the tests a walk applies to a frame are handed to it as the searches' own
// expressions, and a walk that declines (returns false) leaves the frame-by-frame loop below to do it.  Arrays the reference grows as it goes
// (the per-threshold log, the contour) live in a caller-provided Work area of fixed capacity.
//
// Compile with -ffp-contract=off: the expressions below reproduce double expression sequences of the reference, and a
// fused multiply-add would change their rounding.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "sono_measure.h"

#if defined(__HIPCC__)
#define SONO_MC __host__ __device__
#else
#define SONO_MC
#endif

namespace sono {
namespace mcore {

constexpr int kThresholds = 100;             // both searches lower the threshold from -1 to -100 dB (js:2461, 1979)

struct LowM { double threshold; double low_hz; int64_t end_frame; int64_t bin; bool found; };
struct HighM { double threshold; double high_hz; int64_t bin, frame; double power; bool found; };

// Scratch of one measurement.  per_frame arrays hold one entry per frame of the call, per_bin arrays one per bin.
struct Work {
    LowM *low;                               // kThresholds
    HighM *high;                             // kThresholds
    int *valid;                              // kThresholds
    double *contour, *smooth, *d1, *d2, *curv;   // per frame
    int64_t *frames;                         // per frame
    float *max_spec;                         // per bin
    uint16_t *frame_of;                      // per bin (a Uint16Array in the reference)
};
SONO_MC inline size_t work_bytes_per_frame() { return 5 * sizeof(double) + sizeof(int64_t); }
SONO_MC inline size_t work_bytes_per_bin() { return sizeof(float) + sizeof(uint16_t); }

// The largest float <= thr: for a float x, (double)x > thr exactly when x > float_below(thr).
SONO_MC inline float float_below_thr(double thr)
{
    float t = (float)thr;
    if ((double)t > thr) t = nextafterf(t, -INFINITY);
    return t;
}

SONO_MC inline double zone_floor(const double *zonal, size_t zones, double freq_khz)
{
    const double z = floor(freq_khz / 10);                                       // js:2716, 2157: key = zone start in kHz
    if (z >= 0 && z < (double)zones) return zonal[(size_t)z];
    return -100;                                                                 // the map has no such key
}

// x^1.5, the power in the contour's curvature (Math.pow(1 + d1*d1, 1.5), js:3676): x * sqrt(x) carried in two doubles
// each, rounded once at the end.  Correctly rounded wherever a rounding boundary is not within 2^-100 of the exact
// value, on the host and on the GPU alike (both have an exact fma); the C library's pow is a unit in the last place
// off in about one case in 2 000, differently per library, and the curvature only ranks contour points.
SONO_MC inline double pow15(double x)
{
    if (!(x < 1e200)) return x * sqrt(x);                                        // overflows, infinities, NaN
    const double s = sqrt(x), r = fma(-s, s, x), e = r / (2 * s);               // sqrt(x) = s + e
    const double p = x * s, pe = fma(x, s, -p);                                  // x * s = p + pe
    return p + (pe + x * e);
}

// savitzkyGolay (batCallDetector.js:1610-1640), window 5, order 2; n < 5 copies.  One output entry: the entries are
// independent of one another, and an executor may work them out side by side (Ops::for_each).
SONO_MC inline double savitzky_golay_at(const double *in, int64_t n, int64_t i)
{
    if (n < 5) return in[i];                                                     // js:1611
    const double c[5] = {-3, 12, 17, 12, -3};                                    // js:1618
    double s = 0.0;
    for (int j = -2; j <= 2; ++j) {
        const int64_t idx = i + j;
        if (idx >= 0 && idx < n) s += in[idx] * c[j + 2];                        // js:1631
    }
    return s / 35;                                                               // js:1636, also at the edges
}

// findOptimalLowFrequencyThreshold, batCallDetector.js:2413-2909.  zonal: the noise floors to use (never null here).
template <class Ops>
SONO_MC LowResult find_low(const CallMatrix &m, Ops &ops, Work &w, const double *zonal, size_t zones, double peak, int64_t peak_frame,
                           int64_t limit_frame)
{
    LowResult r;
    const int64_t F = (int64_t)m.frames, nb = (int64_t)m.bins;
    if (F == 0 || nb == 0) return r;                                             // js:2414-2421
    auto row = [&](int64_t f) { return m.db + (size_t)f * m.bins; };
    auto at = [&](int64_t f, int64_t b) { return (double)m.db[(size_t)f * m.bins + (size_t)b]; };
    auto fb = [&](int64_t b) { return (double)m.freq_bins[b]; };
    const int64_t search_end = (limit_frame >= 0 && limit_frame < F) ? limit_frame : F - 1;       // js:2427-2429
    const int64_t valid_peak = peak_frame < F - 1 ? peak_frame : F - 1;                           // js:2431
    bool hit = false;
    double opt_threshold = -24;
    int opt = -1;                                                                // index into meas
    LowM *meas = w.low;
    int n_meas = 0;
    int64_t start_frame = valid_peak, max_bin = nb - 1;                          // js:2449, 2453
    auto last_valid = [&]() { for (int i = n_meas - 1; i >= 0; --i) if (meas[i].found) return i; return -1; };
    ops.begin_low(m, peak, valid_peak, search_end);

    for (int t = -1; t >= -kThresholds; --t) {                                   // js:2461
        const double thr = peak + (double)t;                                     // js:2475
        const float thr_f = float_below_thr(thr);                                // entry > thr  <=>  entry > thr_f
        const int ref_i = last_valid();
        const bool has_ref = ref_i >= 0;
        const double ref_khz = has_ref ? meas[ref_i].low_hz / 1000 : 0;          // js:2478-2484
        // forward scan along the call while frames keep a bin above the threshold (one silent frame is bridged)
        int64_t active_end = start_frame;
        auto below_ref = [&](int64_t b0) { return has_ref && fb(b0) < ref_khz * 1000; };           // js:2529-2537
        if (!ops.low_walk(-t, start_frame, search_end, max_bin, below_ref, active_end)) {
            int silence = 0;
            for (int64_t f = start_frame; f <= search_end; ++f) {                // js:2509-2546
                const int64_t b0 = ops.first_gt(row(f), 0, max_bin, thr_f);      // the frame's lowest bin above the threshold
                if (b0 >= 0) {
                    active_end = f;
                    silence = 0;
                    if (below_ref(b0)) break;
                } else if (++silence > 1) {
                    break;
                }
            }
        }
        start_frame = active_end;                                                // js:2548
        // lower edge of the frame where the scan ended
        bool found = false;
        double low_hz = 0, low_power = -INFINITY;
        int64_t found_bin = -1;
        const int64_t b = ops.first_gt(row(active_end), 0, max_bin, thr_f);      // js:2560-2612: the first bin above the threshold
        if (b >= 0) {
            double cand = fb(b);
            if (b > 0) {
                const double p1 = at(active_end, b), p0 = at(active_end, b - 1);
                if (p0 < thr && p1 > thr) cand = fb(b) - ((p1 - thr) / (p1 - p0)) * (fb(b) - fb(b - 1));      // js:2570-2574
            }
            if (has_ref && fabs(cand / 1000 - ref_khz) > 15.0) {                 // js:2582-2603: sub-harmonic, hard stop
                hit = true;
                const int lv = last_valid();
                if (lv >= 0) { opt = lv; opt_threshold = meas[lv].threshold; }
            } else {
                low_hz = cand; found = true; found_bin = b; low_power = at(active_end, b);
            }
        }
        if (found && low_hz <= 10000) {                                          // js:2624-2640: bottomed out, the call is noise
            r.threshold_db = (double)t; r.warning = true; r.discard = true;
            return r;
        }
        if (hit) break;                                                          // js:2654-2659
        if (found) {                                                             // js:2671-2741
            const int lv = last_valid();
            if (lv >= 0) {
                const double cur_khz = low_hz / 1000, jump = fabs(cur_khz - meas[lv].low_hz / 1000);
                if (jump > 8.0) { hit = true; opt = lv; opt_threshold = meas[lv].threshold; break; }          // js:2695-2710
                if (jump > 1.5 && !(low_power > zone_floor(zonal, zones, cur_khz))) {                          // js:2714-2738
                    hit = true; opt = lv; opt_threshold = meas[lv].threshold;
                    break;
                }
            }
        }
        meas[n_meas++] = LowM{(double)t, low_hz, active_end, found ? found_bin : -1, found};                   // js:2747-2757
        if (found && found_bin != -1 && found_bin < max_bin) max_bin = found_bin;                              // js:2760-2764
    }

    int *valid = w.valid;
    int n_valid = 0;
    for (int i = 0; i < n_meas; ++i) if (meas[i].found) valid[n_valid++] = i;
    if (opt < 0) {                                                               // js:2774-2793
        if (n_valid == 0) return r;
        opt = valid[0];
        opt_threshold = meas[opt].threshold;
    }
    if (!hit && n_valid > 0) {                                                   // js:2796-2845
        bool early = false;                                                      // recordedEarlyAnomaly / firstAnomalyIndex
        double early_thr = 0, last_thr = meas[valid[0]].threshold;
        int64_t first = -1;
        int last = valid[0];
        const int64_t n = (int64_t)n_valid;
        auto step = [&](int64_t i) { return fabs(meas[valid[i]].low_hz / 1000 - meas[valid[i - 1]].low_hz / 1000); };
        for (int64_t i = 1; i < n; ++i) {
            if (step(i) > 1.5) {
                if (!early && first == -1) { first = i; early = true; early_thr = meas[valid[i - 1]].threshold; last = valid[i - 1]; }
                if (!m.debug_mode) { r.reference_throws = true; return r; }      // js:2815: reads the log table it does not have
            } else {
                if (early && first != -1) {                                      // js:2821-2832: three smooth steps lift the mark
                    const int64_t cs = first + 1, ce = first + 3 < n - 1 ? first + 3 : n - 1;
                    bool stable = true;
                    for (int64_t k = cs; k <= ce; ++k) if (step(k) > 1.5) stable = false;
                    if (stable && ce - cs + 1 >= 3) { early = false; first = -1; }
                }
                last = valid[i];
                last_thr = meas[valid[i]].threshold;
            }
        }
        opt = last;                                                              // js:2838-2844
        opt_threshold = early ? early_thr : last_thr;
    }
    const double final_threshold = fmax(fmin(opt_threshold, -1), -100);          // js:2858
    const double safe = final_threshold <= -100 ? -30 : final_threshold;
    r.warning = final_threshold <= -100;
    r.has_freq = meas[opt].found;
    r.low_freq_hz = r.end_freq_hz = meas[opt].low_hz;
    if (safe != final_threshold) {                                               // js:2868-2897
        const double thr = peak + safe;
        const float thr_f = float_below_thr(thr);
        int64_t end = valid_peak;
        for (int64_t f = valid_peak; f <= search_end; ++f)                       // the last frame with any bin above
            if (ops.first_gt(row(f), 0, nb - 1, thr_f) >= 0) end = f;
        const int64_t b = ops.first_gt(row(end), 0, nb - 1, thr_f);
        if (b >= 0) {
            double lf = fb(b);
            if (b > 0) {
                const double p1 = at(end, b), p0 = at(end, b - 1);
                if (p0 < thr && p1 > thr) lf = fb(b) - ((p1 - thr) / (p1 - p0)) * (fb(b) - fb(b - 1));
            }
            r.low_freq_hz = r.end_freq_hz = lf;
            r.has_freq = true;
        }
    }
    r.threshold_db = safe;
    r.frame = meas[opt].end_frame; r.has_frame = true;                           // js:2905-2906
    r.bin = meas[opt].bin;
    return r;
}

// findOptimalHighFrequencyThreshold, batCallDetector.js:1931-2401.  tf: the call's centre times, seconds.
template <class Ops>
SONO_MC HighResult find_high(const CallMatrix &m, Ops &ops, Work &w, const double *zonal, size_t zones, const double *tf, double peak,
                             int64_t peak_frame)
{
    HighResult r;
    const int64_t F = (int64_t)m.frames, nb = (int64_t)m.bins;
    if (F == 0 || nb == 0) { r.threshold_db = -1; return r; }                   // js:1932-1941
    auto row = [&](int64_t f) { return m.db + (size_t)f * m.bins; };
    auto at = [&](int64_t f, int64_t b) { return (double)m.db[(size_t)f * m.bins + (size_t)b]; };
    auto fb = [&](int64_t b) { return (double)m.freq_bins[b]; };
    int64_t limit = peak_frame < F - 1 ? peak_frame : F - 1;                     // js:1950
    bool hit = false, is_cf = false, has_last = false;
    double opt_threshold = -1, last_khz = 0;
    int opt = -1, stable = 0;
    int64_t min_bin = 0;
    HighM *meas = w.high;
    int n_meas = 0;
    auto last_valid = [&]() { for (int i = n_meas - 1; i >= 0; --i) if (meas[i].found) return i; return -1; };
    // upper edge of one frame above a threshold, interpolated towards the bin above (js:2009-2020, 2340-2349)
    auto edge = [&](int64_t f, int64_t b, double thr) {
        double c = fb(b);
        if (b < nb - 1) {
            const double p1 = at(f, b), p2 = at(f, b + 1);
            if (p2 < thr && p1 > thr) c = fb(b) + ((p1 - thr) / (p1 - p2)) * (fb(b + 1) - fb(b));
        }
        return c;
    };
    ops.begin_high(m, peak, limit);

    for (int t = -1; t >= -kThresholds; --t) {                                   // js:1965, 1979
        const double thr = peak + (double)t;
        const float thr_f = float_below_thr(thr);
        const int ref_i = last_valid();
        const bool has_ref = ref_i >= 0;
        const double ref_khz = has_ref ? meas[ref_i].high_hz / 1000 : 0;
        double high_hz = 0;
        int64_t hbin = 0, hframe = 0;
        bool found = false;
        int silence = 0;
        auto edge_at = [&](int64_t f, int64_t b) { return edge(f, b, thr); };
        auto harmonic = [&](double cand) { return has_ref && cand / 1000 - ref_khz > 10.0; };      // js:2022-2026
        // A walk answers the threshold when no frame's topmost candidate is a harmonic: every frame then offers that
        // candidate, and what the loop below keeps is the largest of them, the first one met among equals.
        const bool walked = ops.high_walk(-t, limit, min_bin, edge_at, harmonic, high_hz, hbin, hframe, found);
        for (int64_t f = limit; !walked && f >= 0; --f) {                        // js:2002-2048: from the peak back to the call's start
            bool in_frame = false;
            // the frame's bins above the threshold from the top down (further ones only below a harmonic that is skipped)
            for (int64_t b = ops.last_gt(row(f), min_bin, nb - 1, thr_f); b >= min_bin; b = ops.last_gt(row(f), min_bin, b - 1, thr_f)) {
                // A frame whose edge lies a whole bin below the best one so far changes nothing (the frequency axis rises
                // with the bin index): its candidate is below fb(b + 1) <= fb(hbin) <= high_hz, so it neither is the new
                // maximum nor -- being lower than a candidate that was no harmonic -- a harmonic.  Its edge (two more
                // entries of the matrix and a division) is not worked out.
                if (found && b + 1 <= hbin) { in_frame = true; break; }
                const double cand = edge(f, b, thr);
                if (harmonic(cand)) continue;                                    // a harmonic, look further down
                if (!found || cand > high_hz) { high_hz = cand; hbin = b; hframe = f; found = true; }
                in_frame = true;
                break;
            }
            if (!in_frame) { if (++silence > 1) break; } else silence = 0;
        }
        if (found) {                                                             // js:2053-2071: constant-frequency pattern
            const double cur = high_hz / 1000;
            if (has_last) {
                const double d = fabs(cur - last_khz);
                if (d > 0 && d <= 0.05) ++stable; else if (d != 0) stable = 0;
                if (stable >= 10) is_cf = true;
            }
            last_khz = cur; has_last = true;
        } else {
            stable = 0;
        }
        if (found) {                                                             // js:2097-2177
            const double cur_khz = high_hz / 1000, cur_power = at(hframe, hbin);
            const int lv = last_valid();
            if (lv >= 0 && cur_power < -100) {                                   // js:2115-2134: weak and far away in time
                if (fabs(tf[hframe] - tf[meas[lv].frame]) * 1000 > 0.15) {
                    hit = true; opt = lv; opt_threshold = meas[lv].threshold;
                    break;
                }
            }
            if (lv >= 0) {
                const double jump = fabs(cur_khz - meas[lv].high_hz / 1000);
                if (is_cf && jump > 1.0) { hit = true; opt = lv; opt_threshold = meas[lv].threshold; break; }  // js:2145-2154
                if (!(is_cf && jump > 1.0) && jump > 1.5) {                      // js:2156-2175
                    const double noise = fmax(zone_floor(zonal, zones, cur_khz), -115);
                    if (!(cur_power > noise)) { hit = true; opt = lv; opt_threshold = meas[lv].threshold; break; }
                }
            }
        }
        meas[n_meas++] = HighM{(double)t, high_hz, hbin, hframe, found ? at(hframe, hbin) : 0.0, found};       // js:2183-2194
        if (found) {                                                             // js:2196-2204
            if (hframe < limit) limit = hframe;
            if (hbin > min_bin) min_bin = hbin;
        }
    }

    const int64_t final_limit = limit;                                           // js:2212
    int *valid = w.valid;
    int n_valid = 0;
    for (int i = 0; i < n_meas; ++i) if (meas[i].found) valid[n_valid++] = i;
    if (n_valid == 0) return r;                                                  // js:2215-2228 (threshold -24, nothing else)
    if (!hit) {                                                                  // js:2230-2306
        bool early = false;
        double early_thr = 0, last_thr = meas[valid[0]].threshold;
        int64_t first = -1;
        int last = valid[0];
        const int64_t n = (int64_t)n_valid;
        auto step = [&](int64_t i) { return fabs(meas[valid[i]].high_hz / 1000 - meas[valid[i - 1]].high_hz / 1000); };
        for (int64_t i = 1; i < n; ++i) {
            const double diff = step(i);
            if (diff > 4.0) break;                                               // js:2246-2250; the assignments there are overwritten below
            bool anomaly = false;
            if (diff > 2.5) {
                const double noise = fmax(zone_floor(zonal, zones, meas[valid[i]].high_hz / 1000), -115);
                if (meas[valid[i]].power <= noise) {
                    if (!m.debug_mode) { r.reference_throws = true; return r; }  // js:2260-2261: reads the log table
                    anomaly = true;
                }
            }
            if (anomaly) {                                                       // js:2271-2277
                if (!early && first == -1) {
                    first = i; early = true;
                    early_thr = last_thr = meas[valid[i - 1]].threshold;
                    last = valid[i - 1];
                }
            } else {
                if (early && first != -1) {                                      // js:2279-2293
                    const int64_t as = first + 1, ae = first + 3 < n - 1 ? first + 3 : n - 1;
                    bool ok = true;
                    for (int64_t k = as; k <= ae; ++k) if (step(k) > 2.5) { ok = false; break; }
                    if (ok && ae - as + 1 >= 3) { early = false; first = -1; }
                }
                last = valid[i];
                last_thr = meas[valid[i]].threshold;
            }
        }
        opt = last;                                                              // js:2299-2305
        opt_threshold = early ? early_thr : last_thr;
    }
    const double final_threshold = fmax(fmin(opt_threshold, -22), -100);         // js:2316
    const double safe = final_threshold <= -100 ? -30 : final_threshold;
    r.warning = final_threshold <= -100;
    r.high_freq_hz = meas[opt].high_hz; r.has_freq = meas[opt].found;
    r.bin = meas[opt].bin; r.frame = meas[opt].frame;
    if (safe != final_threshold) {                                               // js:2328-2387
        const double thr = peak + safe;
        const float thr_f = float_below_thr(thr);
        bool f_safe = false;
        double hz = 0;
        int64_t sb = 0, sf = 0;
        for (int64_t f = 0; f <= final_limit; ++f) {
            const int64_t b = ops.last_gt(row(f), 0, nb - 1, thr_f);             // the frame's highest bin above
            if (b < 0) continue;
            const double c = edge(f, b, thr);
            if (!f_safe || c > hz) { hz = c; sb = b; sf = f; f_safe = true; }
        }
        if (f_safe) {
            r.high_freq_hz = hz; r.has_freq = true; r.bin = sb; r.frame = sf;
            r.has_start = false;
            const int64_t b = ops.first_gt(row(0), 0, nb - 1, thr_f);            // js:2362-2376: lower edge of the first frame
            if (b >= 0) {
                double s = fb(b);
                if (b > 0) {
                    const double p1 = at(0, b), p0 = at(0, b - 1);
                    if (p0 < thr && p1 > thr) s = fb(b) - ((p1 - thr) / (p1 - p0)) * (fb(b) - fb(b - 1));
                }
                r.start_freq_hz = s; r.has_start = true;
            }
        }
    }
    r.threshold_db = safe;
    r.final_search_limit_frame = final_limit; r.has_limit = true;
    r.is_cf = is_cf;
    return r;
}

// BatCall.applyTimeExpansion (batCallDetector.js:185-278): frequencies times the factor, times and durations divided by
// it.  A NaN here is the reference's null, which it leaves alone; NaN * factor is NaN as well.  Flow and Fhigh are still
// null when the reference gets here (processFullFile sets them afterwards from the scaled fields, js:813-814), and the
// threshold and power fields are not touched.
SONO_MC inline void time_expansion(CallMeasure &c, double factor)
{
    if (!(factor > 1)) return;                                                   // js:186
    double *freqs[] = {&c.peak_freq_khz, &c.high_freq_khz, &c.start_freq_khz, &c.end_freq_khz, &c.low_freq_khz,
                       &c.characteristic_freq_khz, &c.knee_freq_khz, &c.heel_freq_khz, &c.bandwidth_khz};
    double *times[] = {&c.start_time_s, &c.end_time_s, &c.duration_ms, &c.peak_freq_ms, &c.high_freq_ms, &c.start_freq_ms,
                       &c.end_freq_ms, &c.low_freq_ms, &c.characteristic_freq_ms, &c.knee_freq_ms, &c.heel_freq_ms,
                       &c.start_freq_time_s, &c.end_freq_time_s};
    for (double *f : freqs) *f *= factor;
    for (double *t : times) *t /= factor;
}

// ---- the whole measurement of one call ---------------------------------------------------------------------------
//
// Steps, as the reference numbers them: 0 peak, 0.5 vertical-click filter, 1 low-frequency search, 2 high-frequency
// search (+ safety re-scan), 2.5 start frequency by backward tracing, 3 end frequency by forward tracing, 4
// characteristic frequency, 5 bandwidth, 6 knee / heel from the curvature of the smoothed contour, 7 times relative to
// the start frequency.  BatCall's constructor has no highFreqTime_ms, so step 5's second click test compares against
// NaN and never fires (js:3568-3570); likewise kneeFrameIdx starts undefined, which only matters through the nulls
// it leaves behind.
template <class Ops>
SONO_MC void measure(const CallMatrix &m, Ops &ops, Work &w, const double *tf, size_t T, double fhigh_khz, double freq_res, CallMeasure &c)
{
    const double nan = __builtin_nan("");
    c.peak_freq_khz = c.peak_freq_ms = c.peak_power_db = c.high_freq_khz = c.high_freq_ms = c.start_freq_khz = c.start_freq_ms =
        c.start_freq_time_s = c.end_freq_khz = c.end_freq_ms = c.end_freq_time_s = c.low_freq_khz = c.low_freq_ms =
        c.characteristic_freq_khz = c.characteristic_freq_ms = c.knee_freq_khz = c.knee_freq_ms = c.heel_freq_khz = c.heel_freq_ms =
        c.bandwidth_khz = c.high_threshold_db_used = c.low_threshold_db_used = nan;
    c.peak_frame = c.high_freq_frame = c.start_freq_frame = c.start_freq_bin = c.end_frame_for_low_freq = c.knee_frame = c.heel_frame = -1;
    c.discarded = c.reference_throws = false;
    const int64_t F = (int64_t)m.frames, nb = (int64_t)m.bins, NT = (int64_t)T;
    if (F == 0 || nb == 0) return;                                               // js:2999
    auto row = [&](int64_t f) { return m.db + (size_t)f * m.bins; };
    auto at = [&](int64_t f, int64_t b) { return (double)m.db[(size_t)f * m.bins + (size_t)b]; };
    auto fb = [&](int64_t b) { return (double)m.freq_bins[b]; };
    auto rel_ms = [&](int64_t f) { return (tf[f] - tf[0]) * 1000; };

    // STEP 0 (js:3004-3045)
    double peak = -INFINITY;
    int64_t pf = 0, pb = 0;
    for (int64_t f = 0; f < F; ++f) {                                            // the first maximum of the matrix
        if (!((double)ops.row_max(row(f), (size_t)nb) > peak)) continue;
        float v;
        pb = ops.first_max(row(f), (size_t)nb, &v);
        peak = (double)v; pf = f;
    }
    double peak_hz = fb(pb);
    if (pb > 0 && pb < nb - 1) {
        const double d0 = at(pf, pb - 1), d1 = at(pf, pb), d2 = at(pf, pb + 1), a = (d2 - 2 * d1 + d0) / 2;
        if (fabs(a) > 1e-10) peak_hz = fb(pb) + ((d0 - d2) / (4 * a)) * (fb(1) - fb(0));
    }
    const double peak_khz = peak_hz / 1000;
    c.peak_freq_khz = peak_khz;
    c.peak_power_db = peak;
    if (pf < NT) { c.peak_freq_ms = rel_ms(pf); c.peak_frame = pf; }

    // STEP 0.5 (js:3059-3089)
    {
        const float thr = float_below_thr(peak - 12);
        int64_t lo = ops.first_gt(row(pf), 0, nb - 1, thr), hi = ops.last_gt(row(pf), 0, nb - 1, thr);
        if (lo < 0) { lo = nb; hi = 0; }
        if (hi >= lo && (double)(hi - lo) * freq_res / 1000 > 20.0) { c.discarded = true; return; }
    }

    // STEP 1 (js:3096-3121)
    const double *zonal = m.zonal;
    size_t zones = m.num_zones;
    if (!zonal) zonal = ops.zonal_floors(m, pf < F - 1 ? pf : F - 1, F - 1, &zones);           // js:2437-2439
    const LowResult low = find_low(m, ops, w, zonal, zones, peak, pf, F - 1);
    if (low.reference_throws) { c.reference_throws = true; return; }
    if (low.discard) { c.discarded = true; return; }
    // STEP 2 (js:3128-3211)
    if (!m.zonal) zonal = ops.zonal_floors(m, 0, pf < F - 1 ? pf : F - 1, &zones);             // js:1975-1977
    const HighResult high = find_high(m, ops, w, zonal, zones, tf, peak, pf);
    if (high.reference_throws) { c.reference_throws = true; return; }
    // Both searches find the peak bin at their first threshold (1 dB below the peak), so what they return when nothing
    // was found cannot occur after step 0; the fields below would be null / undefined in the reference.
    if (!low.has_freq || !low.has_frame || !high.has_limit) { c.reference_throws = true; return; }
    const double low_khz = low.low_freq_hz / 1000;
    bool has_high = high.has_freq;
    double high_hz_safe = high.high_freq_hz, high_khz_safe = high.high_freq_hz / 1000, used_high = high.threshold_db;
    int64_t high_frame = high.frame, high_bin = high.bin;
    if (has_high && high_khz_safe < peak_khz) {                                  // js:3156-3188: safety re-scan on the max spectrum
        ops.max_spectrum(m, high.final_search_limit_frame, w.max_spec, w.frame_of);
        for (int t = -24; t >= -100; --t) {
            // _scanSpectrumForFrequency from the top (js:2936-2962): the highest bin above the threshold, moved towards
            // the bin above it where that one is below
            const double thr = peak + (double)t;
            const int64_t bin = ops.last_gt(w.max_spec, 0, nb - 1, float_below_thr(thr));
            if (bin < 0) continue;
            double hz = fb(bin);
            if (bin + 1 < nb && (double)w.max_spec[bin + 1] < thr) {
                const double ratio = ((double)w.max_spec[bin] - thr) / ((double)w.max_spec[bin] - (double)w.max_spec[bin + 1]);
                hz = fb(bin) + ratio * (fb(bin + 1) - fb(bin));                  // js:2955
            }
            if (hz / 1000 >= peak_khz) {
                high_hz_safe = hz; high_khz_safe = hz / 1000; high_bin = bin; high_frame = w.frame_of[bin]; used_high = t;
                break;
            }
        }
    }
    c.high_threshold_db_used = used_high;
    double high_hz = 0;
    if (has_high) {
        c.high_freq_khz = high_khz_safe; c.high_freq_frame = high_frame; high_hz = high_hz_safe;
        if (high_frame < NT) c.high_freq_ms = rel_ms(high_frame);
    } else {
        high_hz = fhigh_khz * 1000; c.high_freq_khz = fhigh_khz; c.high_freq_frame = 0; c.high_freq_ms = 0;
    }

    // STEP 2.5 (js:3218-3305)
    const double high_thr = peak + used_high;
    int64_t new_start = 0;
    {
        const float thr_f = float_below_thr(high_thr);
        for (int64_t f = 0; f < F; ++f)
            if (ops.first_gt(row(f), 0, nb - 1, thr_f) >= 0) { new_start = f; break; }
    }
    if (new_start < NT) c.start_time_s = tf[new_start];
    double start_hz = high_hz;
    int64_t start_bin = high_bin, start_frame = high_frame;
    bool trace_start = true;
    double start_thr = high_thr;
    if (high.is_cf) {
        start_thr = peak - 35;
    } else if (high_frame < F) {
        const double p = at(high_frame, high_bin);
        if (p < peak - 30 || p < -80) trace_start = false;
    }
    const int64_t jump_bins = (int64_t)ceil(2000 / freq_res);
    // strongest bin of frame f within jump_bins of bin `around`, and the frequency the tracing steps derive from it
    auto strongest = [&](int64_t f, int64_t around, double &power) {
        const int64_t b0 = around - jump_bins > 0 ? around - jump_bins : 0, b1 = around + jump_bins < nb - 1 ? around + jump_bins : nb - 1;
        int64_t best = -1;
        power = -INFINITY;
        for (int64_t b = b0; b <= b1; ++b) if (at(f, b) > power) { power = at(f, b); best = b; }
        return best;
    };
    auto traced_hz = [&](int64_t f, int64_t b, double power, double thr, double scale) {
        double hz = fb(b);
        if (b > 0 && b < nb - 1) {
            const double p0 = at(f, b - 1), p2 = at(f, b + 1);
            if (power > p0 && power > p2) {
                const double ratio = (power - thr) / (power - fmin(p0, p2)), diff = fb(b + 1) - fb(b);
                hz = fb(b) + ratio * diff * (p0 < p2 ? 1.0 : -1.0) * scale;
            }
        }
        return hz;
    };
    if (trace_start && high_frame > 0) {
        int64_t track = high_bin;
        for (int64_t f = high_frame - 1; f >= 0; --f) {
            double power;
            const int64_t best = strongest(f, track, power);
            if (!(power > start_thr)) break;
            track = start_bin = best;
            start_frame = f;
            start_hz = traced_hz(f, best, power, start_thr, 1.0);
        }
    }
    c.start_freq_khz = start_hz / 1000;
    c.start_freq_bin = start_bin;
    c.start_freq_frame = start_frame;
    if (start_frame < NT) { c.start_freq_time_s = tf[start_frame]; c.start_freq_ms = (c.start_freq_time_s - tf[0]) * 1000; }

    // STEP 3 (js:3313-3506)
    c.low_threshold_db_used = low.threshold_db;
    c.low_freq_khz = low_khz;
    double end_khz = low_khz;
    int64_t end_frame = low.frame;
    const int64_t anchor_frame = low.frame;
    int64_t anchor_bin = low.bin;
    if (anchor_bin == -1) {
        anchor_bin = (int64_t)floor(low_khz * 1000 / freq_res);
        anchor_bin = anchor_bin < 0 ? 0 : anchor_bin > nb - 1 ? nb - 1 : anchor_bin;
    }
    bool trace_end = anchor_frame < F;
    const double end_thr = peak + low.threshold_db;
    if (trace_end) {
        const double p = at(anchor_frame, anchor_bin);
        if (p < peak - 50 || p < -100) trace_end = false;
    }
    if (trace_end) {
        int64_t track = anchor_bin;
        for (int64_t f = anchor_frame + 1; f < F; ++f) {
            double power;
            const int64_t best = strongest(f, track, power);
            if (!(power > end_thr)) break;
            track = best;
            end_frame = f;
            end_khz = traced_hz(f, best, power, end_thr, 0.5) / 1000;
        }
    }
    c.end_freq_khz = end_khz;
    c.end_frame_for_low_freq = end_frame;
    if (end_frame < NT) {
        c.end_freq_time_s = tf[end_frame];
        c.end_time_s = tf[end_frame + 1 < NT - 1 ? end_frame + 1 : NT - 1];
        c.end_freq_ms = (c.end_freq_time_s - tf[0]) * 1000;
        if (low.frame < NT) c.low_freq_ms = rel_ms(low.frame);
    }
    if (c.start_freq_time_s == c.start_freq_time_s && c.end_freq_time_s == c.end_freq_time_s)
        c.duration_ms = (c.end_freq_time_s - c.start_freq_time_s) * 1000;
    if (c.start_freq_khz < c.low_freq_khz) c.low_freq_khz = c.start_freq_khz;
    if (c.end_freq_khz < c.low_freq_khz) c.low_freq_khz = c.end_freq_khz;

    // STEP 4 (js:3511-3560)
    const int64_t char_end = end_frame != 0 ? end_frame : F - 1;
    const int64_t last_start = (int64_t)floor((double)new_start + (double)(char_end - new_start) * (1 - 0.40));
    double char_hz = peak_hz;
    int64_t char_frame = 0;
    if (last_start < char_end) {
        const double dt_ms = NT > 1 ? (tf[1] - tf[0]) * 1000 : 0;
        double *freq = w.contour;                                                // the arrays of step 6, free until then
        int64_t *frame = w.frames;
        int64_t count = 0;
        for (int64_t f = last_start > 0 ? last_start : 0; f <= char_end; ++f) {
            float best;
            int64_t bin = ops.first_max(row(f), (size_t)nb, &best);
            if (bin < 0) bin = 0;
            freq[count] = fb(bin);
            frame[count++] = f;
        }
        double min_slope = INFINITY;
        int64_t pick = last_start;                                               // js:3538: a frame number, used below as a list index
        for (int64_t i = 0; i + 1 < count; ++i) {
            const double slope = dt_ms > 0 ? ((freq[i + 1] - freq[i]) / 1000) / dt_ms : 0;
            if (fabs(slope) < min_slope) { min_slope = fabs(slope); pick = i; }
        }
        if (pick >= 0 && pick < count) { char_hz = freq[pick]; char_frame = frame[pick]; }
    }
    c.characteristic_freq_khz = char_hz / 1000;
    if (char_frame < NT) c.characteristic_freq_ms = rel_ms(char_frame);
    if (c.characteristic_freq_khz < c.low_freq_khz) c.characteristic_freq_khz = c.low_freq_khz;
    else if (c.characteristic_freq_khz > c.peak_freq_khz) c.characteristic_freq_khz = c.peak_freq_khz;

    // STEP 5 (js:3565)
    c.bandwidth_khz = c.high_freq_khz - c.low_freq_khz;

    // STEP 6 (js:3584-3829)
    int64_t search_start = new_start > 0 ? new_start : 0;
    if (start_frame > search_start) search_start = start_frame;
    const int64_t search_end = F - 1 < char_end ? F - 1 : char_end;
    const double cmin = c.low_freq_khz * 1000 - 1000, cmax = c.high_freq_khz * 1000 + 1000;
    int64_t min_bin = 0, max_bin = nb - 1;
    for (int64_t b = 0; b < nb; ++b) if (fb(b) >= cmin) { min_bin = b; break; }
    for (int64_t b = nb - 1; b >= 0; --b) if (fb(b) <= cmax) { max_bin = b; break; }
    double *contour = w.contour;
    int64_t *frames = w.frames;
    int64_t n = 0;
    if (search_end - search_start + 1 > 2)
        for (int64_t f = search_start; f <= search_end; ++f) {
            float power;
            int64_t best = max_bin >= min_bin ? ops.first_max(row(f) + min_bin, (size_t)(max_bin - min_bin + 1), &power) : -1;
            if (best != -1) { best += min_bin; contour[n] = fb(best) / 1000; frames[n++] = f; }
        }
    if (n >= 5) {
        double *smooth = w.smooth, *d1 = w.d1, *d2 = w.d2, *curv = w.curv;
        // The derived arrays are maps over the contour's points (each entry from a few entries of the array before), so
        // they go through for_each: the host loops, the GPU gives every lane a point.
        ops.for_each(n, [&](int64_t i) { smooth[i] = savitzky_golay_at(contour, n, i); });
        // A frame number past the time axis reads undefined in the reference, the difference is NaN, and "NaN > 0" picks 0.001.
        auto t_at = [&](int64_t f) { return f < NT ? tf[f] : nan; };
        const int64_t n1 = n - 1, n2 = n - 2;
        ops.for_each(n1, [&](int64_t i) {                                        // js:3643-3653, kHz per ms
            const double dt = (t_at(frames[i + 1]) - t_at(frames[i])) * 1000;
            d1[i] = (smooth[i + 1] - smooth[i]) / (dt > 0 ? dt : 0.001);
        });
        ops.for_each(n2, [&](int64_t i) {                                        // js:3659-3668; entry i belongs to contour point i + 1
            const double dt = ((t_at(frames[i + 2]) - t_at(frames[i])) * 1000) / 2;
            d2[i] = (d1[i + 1] - d1[i]) / (dt > 0 ? dt : 0.001);
            // one value per contour point, asked for by the knee search and again by the heel search: worked out once
            curv[i] = fabs(d2[i]) / (pow15(1 + d1[i] * d1[i]) + 1e-10);
        });
        auto knee_shape = [&](int64_t k) {                                       // js:3671-3694: steep downward sweep that flattens
            if (k <= 0 || k >= n1) return false;
            const double in = d1[k - 1], out = d1[k];
            if (in > 0 || in > -0.5) return false;
            return !(fabs(out) >= fabs(in) * 0.8);
        };
        auto heel_shape = [&](int64_t k) {                                       // js:3757-3784: flat, then dropping
            if (k <= 0 || k >= n1) return false;
            const double in = d1[k - 1], out = d1[k];
            if (in < -0.5 || out > -0.5) return false;
            return !(fabs(out) < fabs(in) * 1.5);
        };
        int64_t knee = -1;
        double max_curv = -1;
        for (int64_t i = 0; i < n2; ++i) {                                       // js:3700-3713
            const double k = curv[i];
            if (k > max_curv && knee_shape(i)) { max_curv = k; knee = i + 1; }
        }
        if (knee < 0 || max_curv < 0.01) {                                       // js:3718-3729
            double max_change = -1;
            for (int64_t i = 0; i < n2; ++i)
                if (fabs(d2[i]) > max_change && d1[i] < -0.5) { max_change = fabs(d2[i]); knee = i + 1; }
        }
        const int64_t knee_frame = (knee >= 0 && knee < n) ? frames[knee] : -1;
        if (knee_frame >= 0 && knee_frame < NT) {                                // js:3738-3749
            c.knee_freq_khz = contour[knee]; c.knee_frame = knee_frame; c.knee_freq_ms = rel_ms(knee_frame);
        }
        int64_t heel = -1;
        double max_heel = -1;
        for (int64_t i = 0; i < n2; ++i) {                                       // js:3790-3809: after the knee
            if (knee != -1 && i + 1 <= knee + 1) continue;
            const double k = curv[i];
            if (k > max_heel && heel_shape(i)) { max_heel = k; heel = i + 1; }
        }
        const int64_t heel_frame = (heel >= 0 && heel < n) ? frames[heel] : -1;
        if (heel_frame >= 0 && heel_frame < NT) {
            c.heel_freq_khz = contour[heel]; c.heel_frame = heel_frame; c.heel_freq_ms = rel_ms(heel_frame);
        }
    }

    // STEP 7 (js:3835-3888)
    if (start_frame < NT) {
        const double t0 = tf[start_frame];
        auto norm = [&](int64_t f) { return (f < 0 || f >= NT) ? nan : (tf[f] - t0) * 1000; };
        c.start_freq_ms = 0;
        c.peak_freq_ms = norm(pf);
        c.high_freq_ms = norm(c.high_freq_frame);
        c.end_freq_ms = norm(end_frame);
        c.low_freq_ms = norm(low.frame);
        c.knee_freq_ms = norm(c.knee_frame);
        if (c.heel_frame >= 0) c.heel_freq_ms = norm(c.heel_frame);
        if (c.end_freq_ms == c.end_freq_ms) c.duration_ms = c.end_freq_ms - c.start_freq_ms;
    }
    if (c.start_freq_time_s == c.start_freq_time_s) c.start_time_s = c.start_freq_time_s;
    if (c.end_freq_time_s == c.end_freq_time_s) c.end_time_s = c.end_freq_time_s;
    if (m.time_expansion) time_expansion(c, kTimeExpansionFactor);               // js:3890-3893
}

}  // namespace mcore
}  // namespace sono
