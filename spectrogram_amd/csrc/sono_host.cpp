// sono_host.cpp -- host-only parts of libsonostft.  Built with -ffp-contract=off:
// these routines reproduce f32 expression sequences of the reference exactly,
// and a fused multiply-add would change their rounding.
#include "sono_host.h"

#include <cmath>
#include <cstring>

#include "sono_hostmath.h"

namespace sono {

void host_make_window(const char *name, unsigned int size, float alpha, float *out)
{
    make_window(name, size, alpha, out);
}

// Rust `f32 as usize`: saturating, NaN -> 0.
static size_t f32_as_usize(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 1.8446744e19f) return (size_t)-1;
    return (size_t)v;
}

size_t host_power_hop(size_t fft_size, bool has_overlap, float overlap_percent)
{
    const float overlap = has_overlap ? overlap_percent : 0.0f;
    float h;
    if (overlap <= 0.0f || overlap >= 100.0f) h = (float)fft_size * 0.25f;
    else h = (float)fft_size * (1.0f - overlap / 100.0f);
    const size_t hop = f32_as_usize(h);
    return hop < 1 ? 1 : hop;
}

float host_find_peak_frequency(const float *spectrum, size_t len, uint32_t sample_rate,
                               size_t fft_size, float flow_hz, float fhigh_hz)
{
    if (len == 0) return 0.0f;
    const float res = (float)sample_rate / (float)fft_size;
    const size_t min_bin = f32_as_usize(flow_hz / res);
    size_t max_bin = f32_as_usize(fhigh_hz / res);
    if (max_bin > len - 1) max_bin = len - 1;
    if (min_bin >= max_bin) return 0.0f;
    size_t peak_bin = min_bin;
    float peak_db = spectrum[min_bin];
    for (size_t i = min_bin + 1; i <= max_bin; ++i)
        if (spectrum[i] > peak_db) { peak_db = spectrum[i]; peak_bin = i; }
    if (peak_bin > min_bin && peak_bin < max_bin) {
        const float db0 = spectrum[peak_bin - 1], db1 = spectrum[peak_bin], db2 = spectrum[peak_bin + 1];
        const float a = (db2 - 2.0f * db1 + db0) / 2.0f;
        if (std::fabs(a) > 1e-10f) {
            const float corr = (db0 - db2) / (4.0f * a);
            return ((float)peak_bin + corr) * res;
        }
    }
    return (float)peak_bin * res;
}

static float max_abs(const float *x, size_t b, size_t e)
{
    float m = 0.0f;
    for (size_t i = b; i < e; ++i) {
        const float a = std::fabs(x[i]);
        if (a > m) m = a;            // f32::max fold from 0.0: NaN samples are skipped
    }
    return m;
}

size_t host_wave_peaks(const float *x, size_t n, size_t num_peaks, float *out)
{
    if (num_peaks == 0 || n == 0) return 0;
    const float step = (float)n / (float)num_peaks;
    for (size_t k = 0; k < num_peaks; ++k) {
        const size_t start = f32_as_usize((float)k * step);
        size_t end = f32_as_usize(std::ceil((float)(k + 1) * step));
        if (end > n) end = n;
        out[k] = start < end ? max_abs(x, start, end) : 0.0f;
    }
    return num_peaks;
}

float host_find_global_max(const float *x, size_t n) { return max_abs(x, 0, n); }

size_t host_peaks_in_range(const float *x, size_t data_len, size_t start_sample, size_t end_sample,
                           size_t target_width, float *out)
{
    if (target_width == 0) { out[0] = 0.0f; return 1; }
    for (size_t i = 0; i < target_width; ++i) out[i] = 0.0f;
    if (end_sample > data_len) end_sample = data_len;
    const size_t range = end_sample > start_sample ? end_sample - start_sample : 0;
    if (range == 0) return target_width;
    const float step = (float)range / (float)target_width;
    for (size_t px = 0; px < target_width; ++px) {
        const size_t ps = f32_as_usize((float)px * step);
        const size_t pe = f32_as_usize(std::ceil(((float)px + 1.0f) * step));
        size_t cs = start_sample + ps, ce = start_sample + pe;
        if (cs > data_len) cs = data_len;
        if (ce > data_len) ce = data_len;
        if (cs < ce && cs < data_len) out[px] = max_abs(x, cs, ce);
    }
    return target_width;
}

// u8 code of compute_frame_spectrum (lib.rs:659-672)
static uint8_t image_code(float magnitude, float gain_db, float range_db)
{
    const float db = magnitude > 0.0f ? 20.0f * std::log10(magnitude) : -80.0f;
    float normalized = (db + range_db / 2.0f + gain_db) / range_db;
    if (!(normalized > 0.0f)) normalized = 0.0f;     // clamp(0,1); NaN -> 0 after the cast
    if (normalized > 1.0f) normalized = 1.0f;
    return (uint8_t)(normalized * 255.0f);
}

void host_render_image(const float *mags, size_t frames, size_t bins, const float *bank, size_t bank_len,
                       size_t num_filters, bool use_bank, const uint32_t *color_map, size_t width,
                       size_t height, float gain_db, float range_db, uint8_t *out)
{
    std::memset(out, 0, width * height * 4);
    if (frames == 0) return;
    const size_t spec_height = use_bank ? num_filters : bins;      // lib.rs:528-532
    if (spec_height == 0) return;
    // per-frame u8 spectra (what compute_frame_spectrum returns for each frame)
    const bool filt = use_bank && bank_len != 0;
    const size_t cols = filt ? num_filters : bins;
    std::vector<uint8_t> spec(frames * cols);
    const size_t filter_len = bins + 1;
    for (size_t f = 0; f < frames; ++f) {
        const float *m = mags + f * bins;
        for (size_t i = 0; i < cols; ++i) {
            float v;
            if (filt) {
                v = 0.0f;
                const size_t r0 = i * filter_len;
                for (size_t j = 0; j < bins && r0 + j < bank_len; ++j) v += m[j] * bank[r0 + j];
            } else {
                v = m[i];
            }
            spec[f * cols + i] = image_code(v, gain_db, range_db);
        }
    }
    const float time_step = (float)frames / (float)width;
    const float freq_step = (float)spec_height / (float)height;
    for (size_t y = 0; y < height; ++y) {
        const float sf = (float)(height - 1 - y) * freq_step;
        const size_t fi = (size_t)std::floor(sf);
        const float ff = sf - (float)fi;
        const size_t f0 = fi < spec_height - 1 ? fi : spec_height - 1;
        const size_t f1 = fi + 1 < spec_height - 1 ? fi + 1 : spec_height - 1;
        for (size_t x = 0; x < width; ++x) {
            const float st = (float)x * time_step;
            const size_t ti = (size_t)std::floor(st);
            const float tf = st - (float)ti;
            const size_t t0 = ti < frames - 1 ? ti : frames - 1;
            const size_t t1 = ti + 1 < frames - 1 ? ti + 1 : frames - 1;
            float magnitude = 0.0f;
            const size_t ts[2] = {t0, t1}, fs[2] = {f0, f1};
            for (size_t a = 0; a < 2; ++a)
                for (size_t b = 0; b < 2; ++b) {
                    const size_t t = ts[a], fq = fs[b];
                    if (fq < cols) {
                        const float val = (float)spec[t * cols + fq] / 255.0f;
                        const float tw = t == t0 ? 1.0f - tf : tf;      // lib.rs:586-595
                        const float fw = fq == f0 ? 1.0f - ff : ff;
                        magnitude += val * tw * fw;
                    }
                }
            float c = magnitude * 255.0f;
            if (!(c > 0.0f)) c = 0.0f;
            if (c > 255.0f) c = 255.0f;
            const uint32_t rgba = color_map[(size_t)c];
            uint8_t *px = out + (y * width + x) * 4;
            px[0] = (uint8_t)(rgba >> 24); px[1] = (uint8_t)(rgba >> 16);
            px[2] = (uint8_t)(rgba >> 8);  px[3] = (uint8_t)rgba;
        }
    }
}

}  // namespace sono
