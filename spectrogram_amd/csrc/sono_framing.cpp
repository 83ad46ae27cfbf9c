// sono_framing.cpp -- see sono_framing.h.  Built with -ffp-contract=off: the
// recurrences below reproduce double expression sequences of the reference, and
// a fused multiply-add would change their rounding.
#include "sono_framing.h"

#include <cmath>
#include <cstring>
#include <emmintrin.h>

namespace sono {

void host_savitzky_golay(const double *in, size_t n, double *out)
{
    if (n < 5) { if (n) std::memcpy(out, in, n * sizeof(double)); return; }     // js:1611
    static const double c[5] = {-3, 12, 17, 12, -3};                             // js:1618
    for (size_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = -2; j <= 2; ++j) {
            const long long idx = (long long)i + j;
            if (idx >= 0 && idx < (long long)n) s += in[idx] * c[j + 2];         // js:1631
        }
        out[i] = s / 35;                                                         // js:1636, also at the edges
    }
}

int host_refine_plan(size_t n, double sample_rate, int64_t start_sample, int64_t end_sample, RefinePlan &p)
{
    const int64_t safe_start = start_sample > 0 ? start_sample : 0;              // js:1650
    const int64_t safe_end = end_sample < (int64_t)n ? end_sample : (int64_t)n;
    if ((double)(safe_end - safe_start) < sample_rate * 0.0005) return 0;        // js:1653
    p.window = (int64_t)std::floor(sample_rate * (0.1 / 1000));                  // js:1660
    p.sustained = (int64_t)std::floor(sample_rate * (0.5 / 1000));               // js:1663
    p.hop = p.window / 2;                                                        // js:1664
    if (p.hop < 1) return -1;
    const int64_t span = safe_end - p.window - safe_start;
    p.steps = span > 0 ? (size_t)((span + p.hop - 1) / p.hop) : 0;
    p.safe_start = safe_start;
    return p.steps == 0 ? 0 : 1;
}

// The window energies (js:1676-1683).  A window's sum runs over its samples in order, one rounding per sample, so it
// cannot be split; but the windows are independent, and eight of them are summed side by side.
void host_window_sums(const float *audio, const RefinePlan &p, double *sums)
{
    const size_t steps = p.steps;
    const int64_t window = p.window, hop = p.hop;
    // Every sample lies in two windows: its square (exact in a double: 24-bit factors) is formed once, two per instruction.
    static thread_local std::vector<double> scratch;
    const size_t covered = (size_t)((int64_t)(steps - 1) * hop + window);
    if (scratch.size() < covered) scratch.resize(covered);
    double *sq = scratch.data();
    {
        const float *a = audio + p.safe_start;
        size_t i = 0;
        for (; i + 4 <= covered; i += 4) {
            const __m128 v = _mm_loadu_ps(a + i);
            const __m128d lo = _mm_cvtps_pd(v), hi = _mm_cvtps_pd(_mm_movehl_ps(v, v));
            _mm_storeu_pd(sq + i, _mm_mul_pd(lo, lo));
            _mm_storeu_pd(sq + i + 2, _mm_mul_pd(hi, hi));
        }
        for (; i < covered; ++i) { const double v = (double)a[i]; sq[i] = v * v; }
    }
    size_t k = 0;
    for (; k + 8 <= steps; k += 8) {
        const double *q0 = sq + k * (size_t)hop, *q1 = q0 + hop, *q2 = q1 + hop, *q3 = q2 + hop, *q4 = q3 + hop, *q5 = q4 + hop,
                     *q6 = q5 + hop, *q7 = q6 + hop;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
        for (int64_t j = 0; j < window; ++j) {
            a0 += q0[j]; a1 += q1[j]; a2 += q2[j]; a3 += q3[j]; a4 += q4[j]; a5 += q5[j]; a6 += q6[j]; a7 += q7[j];
        }
        sums[k] = a0; sums[k + 1] = a1; sums[k + 2] = a2; sums[k + 3] = a3;
        sums[k + 4] = a4; sums[k + 5] = a5; sums[k + 6] = a6; sums[k + 7] = a7;
    }
    for (; k < steps; ++k) {
        const double *q0 = sq + k * (size_t)hop;
        double a0 = 0.0;
        for (int64_t j = 0; j < window; ++j) a0 += q0[j];
        sums[k] = a0;
    }
}

int64_t host_refine_from_sums(const RefinePlan &p, const double *sums, int64_t end_sample)
{
    const size_t steps = p.steps;
    const int64_t window = p.window, hop = p.hop, sustained = p.sustained, safe_start = p.safe_start;
    // dB of one window (js:1685), worked out when it is first asked for: the walk below ends soon after the call does
    static thread_local std::vector<float> db_values;                            // Float32Array, js:1669
    static thread_local std::vector<char> have;
    if (db_values.size() < steps) db_values.resize(steps);
    have.assign(steps, 0);
    auto db_exact = [&](size_t i) { return 20 * std::log10(std::sqrt(sums[i] / (double)window) + 1e-9); };
    auto db_at = [&](size_t i) {
        if (!have[i]) { db_values[i] = (float)db_exact(i); have[i] = 1; }
        return (double)db_values[i];
    };
    auto centre = [&](size_t i) { return safe_start + (int64_t)i * hop + window / 2; };
    // The loudest window, first one on a tie, judged on the unrounded dB (js:1688).  The dB grows with the energy; when
    // the loudest window is above -60 dB (where the 1e-9 inside the logarithm no longer matters), a window a billionth
    // below the largest energy is 4e-9 dB lower, a million rounding steps of the dB: only the windows within that
    // margin can reach the largest dB, and only those are evaluated, in order.  Quieter spans, and energies that are
    // not finite numbers (NaN or infinite samples), go through every window as the reference does.
    double top = 0.0;
    bool finite = true;
    for (size_t i = 0; i < steps; ++i) {
        if (!(sums[i] <= 1.79769313486231570815e308)) finite = false;
        if (sums[i] > top) top = sums[i];
    }
    const bool margin = finite && top / (double)window > 1e-6;
    const double near_top = top * (1 - 1e-9);
    double peak = -INFINITY;
    size_t peak_index = 0;
    for (size_t i = 0; i < steps; ++i) {
        if (margin && sums[i] < near_top) continue;
        const double db = db_exact(i);
        if (db > peak) { peak = db; peak_index = i; }
    }
    double min_db = peak;                                                        // js:1699
    size_t min_index = peak_index;
    const size_t ahead = (size_t)std::ceil((double)sustained / (double)hop);
    for (size_t i = peak_index + 1; i < steps; ++i) {
        const double cur = db_at(i);
        if (cur < min_db) { min_db = cur; min_index = i; }
        if (min_db < -60 && cur < -60 + 2) return centre(min_index);             // js:1714
        if (cur - min_db > 0.5) {                                                // js:1722-1723
            if (min_db > -32) continue;                                          // js:1728: still inside the signal body
            const size_t limit = steps < i + ahead ? steps : i + ahead;          // js:1744
            bool is_sustained = true;
            for (size_t q = i + 1; q < limit; ++q)
                if (db_at(q) < min_db + 0.5) { is_sustained = false; break; }
            if (is_sustained) return centre(min_index);                          // js:1798
        }
    }
    return end_sample;
}

bool host_refine_end(const float *audio, size_t n, double sample_rate, int64_t start_sample, int64_t end_sample,
                     int64_t *refined)
{
    *refined = end_sample;
    RefinePlan p;
    const int planned = host_refine_plan(n, sample_rate, start_sample, end_sample, p);
    if (planned <= 0) return planned == 0;
    static thread_local std::vector<double> sums;
    if (sums.size() < p.steps) sums.resize(p.steps);
    host_window_sums(audio, p, sums.data());
    *refined = host_refine_from_sums(p, sums.data(), end_sample);
    return true;
}

double host_auto_highpass_khz(double peak_khz)
{
    if (peak_khz >= 40) return 30;
    if (peak_khz >= 35) return 25;
    if (peak_khz >= 30) return 20;
    return 0;
}

static void butterworth_stage(const float *x, size_t n, double wc, int order, float *y)
{
    const double wc2 = wc * wc;
    if (order == 1) {
        const double den = wc + 1, b0 = 1 / den, b1 = -1 / den, a1 = (wc - 1) / den;
        double x1 = 0, y1 = 0;
        for (size_t i = 0; i < n; ++i) {
            const double x0 = (double)x[i];
            const double y0 = b0 * x0 + b1 * x1 - a1 * y1;                       // js:4109
            y[i] = (float)y0;
            x1 = x0; y1 = y0;                                                    // the recurrence keeps the unrounded value
        }
        return;
    }
    const double s2 = std::sqrt(2.0) * wc, den = wc2 + s2 + 1;
    const double b0 = 1 / den, b1 = -2 / den, b2 = 1 / den, a1 = (2 * (wc2 - 1)) / den, a2 = (wc2 - s2 + 1) / den;
    double x1 = 0, x2 = 0, y1 = 0, y2 = 0;
    for (size_t i = 0; i < n; ++i) {
        const double x0 = (double)x[i];
        const double y0 = b0 * x0 + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;       // js:4131
        y[i] = (float)y0;
        x2 = x1; x1 = x0; y2 = y1; y1 = y0;
    }
}

void host_highpass(const float *in, size_t n, double filter_hz, double sample_rate, double order, float *out)
{
    if (n == 0) return;
    if (in != out) std::memcpy(out, in, n * sizeof(float));
    if (!(filter_hz > 0)) return;                                                // js:4046
    double o = std::floor(order + 0.5);                                          // Math.round, js:4051
    o = o < 1 ? 1 : o > 8 ? 8 : o;
    const double norm = filter_hz / (sample_rate / 2);
    if (norm >= 1) return;                                                       // js:4058
    const double wc = std::tan(M_PI * norm / 2);                                 // js:4063
    std::vector<float> tmp(n);
    const int stages2 = (int)o / 2;
    for (int s = 0; s < stages2; ++s) {
        butterworth_stage(out, n, wc, 2, tmp.data());
        std::memcpy(out, tmp.data(), n * sizeof(float));
    }
    if ((int)o % 2) {
        butterworth_stage(out, n, wc, 1, tmp.data());
        std::memcpy(out, tmp.data(), n * sizeof(float));
    }
}

void host_scan_spectrum(const float *row, const float *fb, size_t n, double thr, bool high_to_low, double *freq_hz,
                        int64_t *bin)
{
    *freq_hz = -1.0;
    *bin = -1;
    for (size_t s = 0; s < n; ++s) {
        const size_t b = high_to_low ? n - 1 - s : s;
        if (!((double)row[b] > thr)) continue;
        double f = (double)fb[b];
        if (high_to_low ? b + 1 < n : b > 0) {
            const size_t nb = high_to_low ? b + 1 : b - 1;
            if ((double)row[nb] < thr) {
                const double ratio = ((double)row[b] - thr) / ((double)row[b] - (double)row[nb]);
                f = high_to_low ? (double)fb[b] + ratio * ((double)fb[nb] - (double)fb[b])       // js:2955
                                : (double)fb[b] - ratio * ((double)fb[b] - (double)fb[nb]);      // js:2975
            }
        }
        *freq_hz = f;
        *bin = (int64_t)b;
        return;
    }
}

bool host_frame_calls(const FrameRun *segs, size_t count, const double *tf, size_t frames, const float *audio, size_t n,
                      double sample_rate, uint32_t fft_size, double padding_ms, double min_ms, bool full_file_rules,
                      std::vector<CallFrame> &out, const RefineFn *refine)
{
    out.clear();
    if (frames < 2) return true;                 // the reference has no defined behaviour for a one-frame matrix
    const double frame_sec = 1 / (sample_rate / (double)fft_size);               // js:1130
    const double per_frame = tf[1] - tf[0];                                      // js:1147
    const double pad_d = std::ceil((padding_ms / 1000) / per_frame);             // js:1146-1148, 743-745
    const int64_t pad = pad_d > 0 ? (pad_d < 9e15 ? (int64_t)pad_d : (int64_t)9e15) : 0;
    for (size_t k = 0; k < count; ++k) {
        const int64_t s = (int64_t)segs[k].start, e = (int64_t)segs[k].end;
        if (s > e || e >= (int64_t)frames) continue;
        if (!full_file_rules && (double)(e - s + 1) * frame_sec * 1000 < min_ms) continue;   // js:1131-1133
        const int64_t safe_start = s - pad > 0 ? s - pad : 0;                    // js:1151
        int64_t safe_end = e + pad < (int64_t)frames - 1 ? e + pad : (int64_t)frames - 1;
        const int64_t start_sample = (int64_t)std::floor(tf[safe_start] * sample_rate);       // js:1158-1159
        const int64_t end_sample = (int64_t)std::floor(tf[safe_end] * sample_rate);
        int64_t refined = end_sample;
        if (!(refine ? (*refine)(start_sample, end_sample, &refined) : host_refine_end(audio, n, sample_rate, start_sample, end_sample, &refined)))
            return false;
        if (refined < end_sample) {
            const double t_end = (double)refined / sample_rate;                  // js:1170
            int64_t new_end = safe_end;
            while (new_end > safe_start && tf[new_end] > t_end) --new_end;       // js:1172
            safe_end = new_end + 1 < (int64_t)frames - 1 ? new_end + 1 : (int64_t)frames - 1;   // js:1188
        }
        CallFrame c;
        c.start_frame = (uint64_t)safe_start;
        c.end_frame = (uint64_t)safe_end;
        c.start_time_s = tf[safe_start];                                         // js:1196
        const int64_t last = safe_end + 1 < (int64_t)frames - 1 ? safe_end + 1 : (int64_t)frames - 1;
        c.end_time_s = tf[last];                                                 // js:1197
        c.duration_ms = (c.end_time_s - c.start_time_s) * 1000;                  // BatCall.calculateDuration, js:161
        if (c.duration_ms <= 1.0 || (!full_file_rules && c.duration_ms < min_ms)) continue;    // js:1206, 807
        out.push_back(c);
    }
    return true;
}

double host_peak_freq_khz(float left_db, float peak_db, float right_db, bool interior, float bin_freq_hz, double bin_width_hz)
{
    double f = (double)bin_freq_hz;                                              // js:3022
    if (interior) {
        const double db0 = (double)left_db, db1 = (double)peak_db, db2 = (double)right_db;
        const double a = (db2 - 2 * db1 + db0) / 2;                              // js:3028
        if (std::fabs(a) > 1e-10) f = (double)bin_freq_hz + ((db0 - db2) / (4 * a)) * bin_width_hz;    // js:3029-3032
    }
    return f / 1000;                                                             // js:3036
}

double host_instant_bandwidth_khz(int32_t lo_bin, int32_t hi_bin, double freq_resolution_hz)
{
    if (lo_bin < 0 || hi_bin < lo_bin) return 0.0;                               // js:3075
    return ((double)(hi_bin - lo_bin) * freq_resolution_hz) / 1000;              // js:3076-3077
}

size_t host_zone_map(const float *fb, size_t n, std::vector<uint8_t> &zone_of_bin)
{
    zone_of_bin.assign(n, 255);
    if (n == 0) return 0;
    const double zmax = std::floor((double)fb[n - 1] / 10000);                   // js:1837-1838
    if (!(zmax >= 0) || zmax > 253) return 0;
    const size_t zones = (size_t)zmax + 1;
    for (size_t b = 0; b < n; ++b) {
        const double z = std::floor((double)fb[b] / 10000);                      // js:1864
        zone_of_bin[b] = (z >= 0 && z < (double)zones) ? (uint8_t)z : 255;      // js:1874: other bins are not counted
    }
    return zones;
}

void host_zone_modes(const unsigned int *hist, size_t zones, double *floors)
{
    for (size_t z = 0; z < zones; ++z) {
        unsigned int best = 0;
        int mode = -100;
        bool any = false;
        for (int i = 0; i < 141; ++i) {
            const unsigned int c = hist[z * 141 + i];
            if (c > 0) any = true;
            if (c > best) { best = c; mode = i - 120; }                          // strict >: lowest dB wins a tie, js:1907
        }
        floors[z] = any ? (double)mode - 2.0 : -102.0;                           // js:1916-1920
    }
}

}  // namespace sono
