// sono_measure.cpp -- see sono_measure.h.  Built with -ffp-contract=off.
//
// Both searches lower a threshold from 1 to 100 dB below the call's peak, one dB at
// a time, follow the call's lower (upper) edge while it moves smoothly, and stop
// where it jumps into the noise.  The reference keeps a log table only in its
// debug mode and consults it unconditionally in one branch of each search
// (batCallDetector.js:2815, 2261); outside debug mode that branch raises a
// TypeError.  reference_throws reports that the inputs lead there.
#include "sono_measure.h"

#include <cmath>
#include <vector>

#include "sono_framing.h"

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

double zone_floor(const double *zonal, size_t zones, double freq_khz)
{
    const double z = std::floor(freq_khz / 10);                                  // js:2716, 2157: key = zone start in kHz
    if (z >= 0 && z < (double)zones) return zonal[(size_t)z];
    return -100;                                                                 // the map has no such key
}

struct LowM { double threshold; double low_hz; bool found; int64_t end_frame; int64_t bin; };
struct HighM { double threshold; double high_hz; int64_t bin, frame; double power; bool found; };

}  // namespace

LowResult find_optimal_low_threshold(const CallMatrix &m, double peak, int64_t peak_frame, int64_t limit_frame)
{
    LowResult r;
    const int64_t F = (int64_t)m.frames, nb = (int64_t)m.bins;
    if (F == 0 || nb == 0) return r;                                             // js:2414-2421
    auto at = [&](int64_t f, int64_t b) { return (double)m.db[(size_t)f * m.bins + (size_t)b]; };
    auto fb = [&](int64_t b) { return (double)m.freq_bins[b]; };
    const int64_t search_end = (limit_frame >= 0 && limit_frame < F) ? limit_frame : F - 1;       // js:2427-2429
    const int64_t valid_peak = peak_frame < F - 1 ? peak_frame : F - 1;                           // js:2431
    double own_zonal[256];
    const double *zonal = m.zonal;
    size_t zones = m.num_zones;
    if (!zonal) {                                                                // js:2437-2439
        zones = host_zonal_floors(m.db, m.frames, m.bins, m.freq_bins, valid_peak, search_end, own_zonal, 256);
        zonal = own_zonal;
    }
    bool hit = false;
    double opt_threshold = -24;
    int opt = -1;                                                                // index into meas
    std::vector<LowM> meas;
    int64_t start_frame = valid_peak, max_bin = nb - 1;                          // js:2449, 2453
    auto last_valid = [&]() { for (int i = (int)meas.size() - 1; i >= 0; --i) if (meas[i].found) return i; return -1; };

    for (int t = -1; t >= -100; --t) {                                           // js:2461
        const double thr = peak + (double)t;                                     // js:2475
        const int ref_i = last_valid();
        const bool has_ref = ref_i >= 0;
        const double ref_khz = has_ref ? meas[ref_i].low_hz / 1000 : 0;          // js:2478-2484
        // forward scan along the call while frames keep a bin above the threshold (one silent frame is bridged)
        int64_t active_end = start_frame;
        int silence = 0;
        for (int64_t f = start_frame; f <= search_end; ++f) {                    // js:2509-2546
            bool has = false;
            double lowest = 0;
            for (int64_t b = 0; b <= max_bin; ++b)
                if (at(f, b) > thr) { has = true; lowest = fb(b); break; }
            if (has) {
                active_end = f;
                silence = 0;
                if (has_ref && lowest < ref_khz * 1000) break;                   // js:2529-2537
            } else if (++silence > 1) {
                break;
            }
        }
        start_frame = active_end;                                                // js:2548
        // lower edge of the frame where the scan ended
        bool found = false;
        double low_hz = 0, low_power = -INFINITY;
        int64_t found_bin = -1;
        for (int64_t b = 0; b <= max_bin; ++b) {                                 // js:2560-2612
            if (!(at(active_end, b) > thr)) continue;
            double cand = fb(b);
            if (b > 0) {
                const double p1 = at(active_end, b), p0 = at(active_end, b - 1);
                if (p0 < thr && p1 > thr) cand = fb(b) - ((p1 - thr) / (p1 - p0)) * (fb(b) - fb(b - 1));      // js:2570-2574
            }
            if (has_ref && std::fabs(cand / 1000 - ref_khz) > 15.0) {            // js:2582-2603: sub-harmonic, hard stop
                hit = true;
                const int lv = last_valid();
                if (lv >= 0) { opt = lv; opt_threshold = meas[lv].threshold; }
                break;
            }
            low_hz = cand; found = true; found_bin = b; low_power = at(active_end, b);
            break;
        }
        if (found && low_hz <= 10000) {                                          // js:2624-2640: bottomed out, the call is noise
            r.threshold_db = (double)t; r.warning = true; r.discard = true;
            return r;
        }
        if (hit) break;                                                          // js:2654-2659
        if (found) {                                                             // js:2671-2741
            const int lv = last_valid();
            if (lv >= 0) {
                const double cur_khz = low_hz / 1000, jump = std::fabs(cur_khz - meas[lv].low_hz / 1000);
                if (jump > 8.0) { hit = true; opt = lv; opt_threshold = meas[lv].threshold; break; }          // js:2695-2710
                if (jump > 1.5 && !(low_power > zone_floor(zonal, zones, cur_khz))) {                          // js:2714-2738
                    hit = true; opt = lv; opt_threshold = meas[lv].threshold;
                    break;
                }
            }
        }
        meas.push_back({(double)t, low_hz, found, active_end, found ? found_bin : -1});                        // js:2747-2757
        if (found && found_bin != -1 && found_bin < max_bin) max_bin = found_bin;                              // js:2760-2764
    }

    std::vector<int> valid;
    for (int i = 0; i < (int)meas.size(); ++i) if (meas[i].found) valid.push_back(i);
    if (opt < 0) {                                                               // js:2774-2793
        if (valid.empty()) return r;
        opt = valid[0];
        opt_threshold = meas[opt].threshold;
    }
    if (!hit && !valid.empty()) {                                                // js:2796-2845
        int last = valid[0];
        for (size_t i = 1; i < valid.size(); ++i) {
            const double diff = std::fabs(meas[valid[i]].low_hz / 1000 - meas[valid[i - 1]].low_hz / 1000);
            if (diff > 1.5) { r.reference_throws = true; return r; }             // js:2809-2815: reads the log table it does not have
            last = valid[i];
        }
        opt = last;
        opt_threshold = meas[last].threshold;
    }
    const double final_threshold = std::fmax(std::fmin(opt_threshold, -1), -100);                              // js:2858
    const double safe = final_threshold <= -100 ? -30 : final_threshold;
    r.warning = final_threshold <= -100;
    r.has_freq = meas[opt].found;
    r.low_freq_hz = r.end_freq_hz = meas[opt].low_hz;
    if (safe != final_threshold) {                                               // js:2868-2897
        const double thr = peak + safe;
        int64_t end = valid_peak;
        for (int64_t f = valid_peak; f <= search_end; ++f)
            for (int64_t b = 0; b < nb; ++b)
                if (at(f, b) > thr) { end = f; break; }
        for (int64_t b = 0; b < nb; ++b) {
            if (!(at(end, b) > thr)) continue;
            double lf = fb(b);
            if (b > 0) {
                const double p1 = at(end, b), p0 = at(end, b - 1);
                if (p0 < thr && p1 > thr) lf = fb(b) - ((p1 - thr) / (p1 - p0)) * (fb(b) - fb(b - 1));
            }
            r.low_freq_hz = r.end_freq_hz = lf;
            r.has_freq = true;
            break;
        }
    }
    r.threshold_db = safe;
    r.frame = meas[opt].end_frame; r.has_frame = true;                           // js:2905-2906
    r.bin = meas[opt].bin;
    return r;
}

HighResult find_optimal_high_threshold(const CallMatrix &m, const double *tf, double peak, int64_t peak_frame)
{
    HighResult r;
    const int64_t F = (int64_t)m.frames, nb = (int64_t)m.bins;
    if (F == 0 || nb == 0) { r.threshold_db = -1; return r; }                   // js:1932-1941
    auto at = [&](int64_t f, int64_t b) { return (double)m.db[(size_t)f * m.bins + (size_t)b]; };
    auto fb = [&](int64_t b) { return (double)m.freq_bins[b]; };
    int64_t limit = peak_frame < F - 1 ? peak_frame : F - 1;                     // js:1950
    double own_zonal[256];
    const double *zonal = m.zonal;
    size_t zones = m.num_zones;
    if (!zonal) {                                                                // js:1975-1977
        zones = host_zonal_floors(m.db, m.frames, m.bins, m.freq_bins, 0, limit, own_zonal, 256);
        zonal = own_zonal;
    }
    bool hit = false, is_cf = false, has_last = false;
    double opt_threshold = -1, last_khz = 0;
    int opt = -1, stable = 0;
    int64_t min_bin = 0;
    std::vector<HighM> meas;
    auto last_valid = [&]() { for (int i = (int)meas.size() - 1; i >= 0; --i) if (meas[i].found) return i; return -1; };
    // upper edge of one frame above a threshold, interpolated towards the bin above (js:2009-2020, 2340-2349)
    auto edge = [&](int64_t f, int64_t b, double thr) {
        double c = fb(b);
        if (b < nb - 1) {
            const double p1 = at(f, b), p2 = at(f, b + 1);
            if (p2 < thr && p1 > thr) c = fb(b) + ((p1 - thr) / (p1 - p2)) * (fb(b + 1) - fb(b));
        }
        return c;
    };

    for (int t = -1; t >= -100; --t) {                                           // js:1965, 1979
        const double thr = peak + (double)t;
        const int ref_i = last_valid();
        const bool has_ref = ref_i >= 0;
        const double ref_khz = has_ref ? meas[ref_i].high_hz / 1000 : 0;
        double high_hz = 0;
        int64_t hbin = 0, hframe = 0;
        bool found = false;
        int silence = 0;
        for (int64_t f = limit; f >= 0; --f) {                                   // js:2002-2048: from the peak back to the call's start
            bool in_frame = false;
            for (int64_t b = nb - 1; b >= min_bin; --b) {
                if (!(at(f, b) > thr)) continue;
                const double cand = edge(f, b, thr);
                if (has_ref && cand / 1000 - ref_khz > 10.0) continue;           // js:2022-2026: a harmonic, look further down
                if (!found || cand > high_hz) { high_hz = cand; hbin = b; hframe = f; found = true; }
                in_frame = true;
                break;
            }
            if (!in_frame) { if (++silence > 1) break; } else silence = 0;
        }
        if (found) {                                                             // js:2053-2071: constant-frequency pattern
            const double cur = high_hz / 1000;
            if (has_last) {
                const double d = std::fabs(cur - last_khz);
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
                if (std::fabs(tf[hframe] - tf[meas[lv].frame]) * 1000 > 0.15) {
                    hit = true; opt = lv; opt_threshold = meas[lv].threshold;
                    break;
                }
            }
            if (lv >= 0) {
                const double jump = std::fabs(cur_khz - meas[lv].high_hz / 1000);
                if (is_cf && jump > 1.0) { hit = true; opt = lv; opt_threshold = meas[lv].threshold; break; }  // js:2145-2154
                if (!(is_cf && jump > 1.0) && jump > 1.5) {                      // js:2156-2175
                    const double noise = std::fmax(zone_floor(zonal, zones, cur_khz), -115);
                    if (!(cur_power > noise)) { hit = true; opt = lv; opt_threshold = meas[lv].threshold; break; }
                }
            }
        }
        meas.push_back({(double)t, high_hz, hbin, hframe, found ? at(hframe, hbin) : 0.0, found});             // js:2183-2194
        if (found) {                                                             // js:2196-2204
            if (hframe < limit) limit = hframe;
            if (hbin > min_bin) min_bin = hbin;
        }
    }

    const int64_t final_limit = limit;                                           // js:2212
    std::vector<int> valid;
    for (int i = 0; i < (int)meas.size(); ++i) if (meas[i].found) valid.push_back(i);
    if (valid.empty()) return r;                                                 // js:2215-2228 (threshold -24, nothing else)
    if (!hit) {                                                                  // js:2230-2306
        opt = valid[0];
        int last = valid[0];
        double last_thr = meas[valid[0]].threshold;
        for (size_t i = 1; i < valid.size(); ++i) {
            const double cur = meas[valid[i]].high_hz / 1000, diff = std::fabs(cur - meas[valid[i - 1]].high_hz / 1000);
            if (diff > 4.0) break;                                               // js:2246-2250; the assignments there are overwritten below
            if (diff > 2.5) {
                const double noise = std::fmax(zone_floor(zonal, zones, cur), -115);
                if (meas[valid[i]].power <= noise) { r.reference_throws = true; return r; }   // js:2260-2261: reads the log table
            }
            last = valid[i];
            last_thr = meas[valid[i]].threshold;
        }
        opt = last;                                                              // js:2302-2305
        opt_threshold = last_thr;
    }
    const double final_threshold = std::fmax(std::fmin(opt_threshold, -22), -100);                             // js:2316
    const double safe = final_threshold <= -100 ? -30 : final_threshold;
    r.warning = final_threshold <= -100;
    r.high_freq_hz = meas[opt].high_hz; r.has_freq = meas[opt].found;
    r.bin = meas[opt].bin; r.frame = meas[opt].frame;
    if (safe != final_threshold) {                                               // js:2328-2387
        const double thr = peak + safe;
        bool f_safe = false;
        double hz = 0;
        int64_t sb = 0, sf = 0;
        for (int64_t f = 0; f <= final_limit; ++f)
            for (int64_t b = nb - 1; b >= 0; --b) {
                if (!(at(f, b) > thr)) continue;
                const double c = edge(f, b, thr);
                if (!f_safe || c > hz) { hz = c; sb = b; sf = f; f_safe = true; }
                break;
            }
        if (f_safe) {
            r.high_freq_hz = hz; r.has_freq = true; r.bin = sb; r.frame = sf;
            r.has_start = false;
            for (int64_t b = 0; b < nb; ++b) {                                   // js:2362-2376: lower edge of the first frame
                if (!(at(0, b) > thr)) continue;
                double s = fb(b);
                if (b > 0) {
                    const double p1 = at(0, b), p0 = at(0, b - 1);
                    if (p0 < thr && p1 > thr) s = fb(b) - ((p1 - thr) / (p1 - p0)) * (fb(b) - fb(b - 1));
                }
                r.start_freq_hz = s; r.has_start = true;
                break;
            }
        }
    }
    r.threshold_db = safe;
    r.final_search_limit_frame = final_limit; r.has_limit = true;
    r.is_cf = is_cf;
    return r;
}

}  // namespace sono
