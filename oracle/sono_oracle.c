/*
 * sono_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the spectrogram hot path of
 * the reference (spectrogram-wasm/src/lib.rs).  It exists so that the HIP
 * kernels can be checked on machines where the reference cannot run.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the shipped library never links or calls anything in this directory.
 *
 * Parity status: PINNED against golden vectors captured from the reference's
 * own wasm binary (tests/golden/, frozen fixtures; case definitions in oracle/make_golden.py).  The one
 * part that cannot be restated bit-for-bit is the FFT itself: the reference
 * calls the third-party crate rustfft 6.4.1 (spectrogram-wasm/Cargo.lock:113-116),
 * whose source is not in the reference tree.  Here the DFT is an iterative
 * radix-2 decimation-in-time FFT in f32 with f64-derived twiddles (the same
 * precision class as rustfft's scalar path: f32 data, twiddles rounded from
 * f64), so spectra agree with the reference to f32 FFT rounding (~1e-7 of the
 * frame's largest bin), not bitwise.  Everything around the FFT (framing,
 * window formulas, magnitude scaling, dB law, u8 truncation, peak rules,
 * power-spectrum averaging) follows lib.rs operation by operation in f32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "msun_math.h"

#define SO_PI_F 3.14159274101257324219f /* std::f32::consts::PI */

/* ------------------------------------------------------------------ */
/* window table: lib.rs:698-776 (create_window)                        */
/* ------------------------------------------------------------------ */
void so_create_window(const char *name, size_t size, float alpha, float *w)
{
    const float fs = (float)size;
    size_t i;
    if (strcmp(name, "bartlett") == 0) {                    /* lib.rs:702-707 */
        for (i = 0; i < size; i++)
            w[i] = 2.0f / (fs - 1.0f) *
                   ((fs - 1.0f) / 2.0f - fabsf((float)i - (fs - 1.0f) / 2.0f));
    } else if (strcmp(name, "bartlettHann") == 0) {         /* lib.rs:708-715 */
        for (i = 0; i < size; i++) {
            float ni = (float)i / (fs - 1.0f);
            w[i] = 0.62f - 0.48f * fabsf(ni - 0.5f) - 0.38f * so_cosf(2.0f * SO_PI_F * ni);
        }
    } else if (strcmp(name, "blackman") == 0) {             /* lib.rs:716-722 */
        for (i = 0; i < size; i++)
            w[i] = (1.0f - alpha) / 2.0f
                 - 0.5f * so_cosf(2.0f * SO_PI_F * (float)i / (fs - 1.0f))
                 + alpha / 2.0f * so_cosf(4.0f * SO_PI_F * (float)i / (fs - 1.0f));
    } else if (strcmp(name, "cosine") == 0) {               /* lib.rs:723-727 */
        for (i = 0; i < size; i++)
            w[i] = so_cosf(SO_PI_F * (float)i / (fs - 1.0f) - SO_PI_F / 2.0f);
    } else if (strcmp(name, "gauss") == 0) {                /* lib.rs:728-734 */
        float sigma = 0.25f * (fs - 1.0f) / 2.0f;
        for (i = 0; i < size; i++) {
            float x = ((float)i - (fs - 1.0f) / 2.0f) / sigma;
            w[i] = so_expf(-0.5f * x * x);
        }
    } else if (strcmp(name, "hamming") == 0) {              /* lib.rs:735-739 */
        for (i = 0; i < size; i++)
            w[i] = 0.54f - 0.46f * so_cosf(2.0f * SO_PI_F * (float)i / (fs - 1.0f));
    } else if (strcmp(name, "lanczos") == 0) {              /* lib.rs:745-755 */
        for (i = 0; i < size; i++) {
            float x = 2.0f * (float)i / (fs - 1.0f) - 1.0f;
            float pix = SO_PI_F * x;
            w[i] = fabsf(pix) < 1e-6f ? 1.0f : so_sinf(pix) / pix;
        }
    } else if (strcmp(name, "rectangular") == 0) {          /* lib.rs:756-760 */
        for (i = 0; i < size; i++) w[i] = 1.0f;
    } else if (strcmp(name, "triangular") == 0) {           /* lib.rs:761-766 */
        for (i = 0; i < size; i++)
            w[i] = 2.0f / fs * (fs / 2.0f - fabsf((float)i - (fs - 1.0f) / 2.0f));
    } else {                                /* "hann" and any unknown name: lib.rs:740-744,767-772 */
        for (i = 0; i < size; i++)
            w[i] = 0.5f * (1.0f - so_cosf(2.0f * SO_PI_F * (float)i / (fs - 1.0f)));
    }
}

/* frame count: lib.rs:118-123 (usize arithmetic; noverlap < fft_size required) */
size_t so_num_frames(size_t n, size_t fft_size, size_t noverlap)
{
    size_t step = fft_size - noverlap;
    return n >= fft_size ? (n - fft_size) / step + 1 : 0;
}

/* ------------------------------------------------------------------ */
/* forward complex DFT, f32, in place (stands in for rustfft's          */
/* Fft::process, call sites lib.rs:147,250,1085)                        */
/* ------------------------------------------------------------------ */
typedef struct {
    size_t n;
    float *twr, *twi;     /* n/2 twiddles exp(-2*pi*i*k/n), rounded from f64 */
    uint32_t *rev;        /* bit-reversal permutation */
} so_plan;

so_plan *so_plan_create(size_t n)
{
    size_t i, bits = 0;
    so_plan *p;
    if (n < 2 || (n & (n - 1)) != 0) return NULL;
    while (((size_t)1 << bits) < n) bits++;
    p = (so_plan *)malloc(sizeof(*p));
    p->n = n;
    p->twr = (float *)malloc(sizeof(float) * (n / 2));
    p->twi = (float *)malloc(sizeof(float) * (n / 2));
    p->rev = (uint32_t *)malloc(sizeof(uint32_t) * n);
    for (i = 0; i < n / 2; i++) {
        double a = -2.0 * 3.14159265358979323846 * (double)i / (double)n;
        p->twr[i] = (float)cos(a);
        p->twi[i] = (float)sin(a);
    }
    for (i = 0; i < n; i++) {
        size_t r = 0, b;
        for (b = 0; b < bits; b++) if (i & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
        p->rev[i] = (uint32_t)r;
    }
    return p;
}

void so_plan_destroy(so_plan *p)
{
    if (!p) return;
    free(p->twr); free(p->twi); free(p->rev); free(p);
}

void so_fft_forward(const so_plan *p, float *re, float *im)
{
    const size_t n = p->n;
    size_t i, half, k, j;
    for (i = 0; i < n; i++) {
        size_t r = p->rev[i];
        if (r > i) {
            float t = re[i]; re[i] = re[r]; re[r] = t;
            t = im[i]; im[i] = im[r]; im[r] = t;
        }
    }
    for (half = 1; half < n; half <<= 1) {
        const size_t stride = n / (2 * half);
        for (k = 0; k < n; k += 2 * half) {
            for (j = 0; j < half; j++) {
                const float wr = p->twr[j * stride], wi = p->twi[j * stride];
                const size_t a = k + j, b = a + half;
                const float tr = re[b] * wr - im[b] * wi;
                const float ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] = re[a] + tr; im[a] = im[a] + ti;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* compute_spectrogram: lib.rs:113-161                                  */
/* out: frames x (N/2) f32, frame-major.  Returns the frame count.      */
/* ------------------------------------------------------------------ */
size_t so_compute_spectrogram(const so_plan *p, const float *window,
                              const float *x, size_t n, size_t noverlap, float *out)
{
    const size_t N = p->n, B = N / 2, step = N - noverlap;
    const size_t F = so_num_frames(n, N, noverlap);
    const float scale = 2.0f / (float)N;                   /* lib.rs:150 */
    float *re = (float *)malloc(sizeof(float) * N);
    float *im = (float *)malloc(sizeof(float) * N);
    size_t f, i, pos = 0;
    for (f = 0; f < F; f++) {
        for (i = 0; i < N; i++) { re[i] = x[pos + i] * window[i]; im[i] = 0.0f; }  /* :138-144 */
        so_fft_forward(p, re, im);
        for (i = 0; i < B; i++)                                                     /* :151-155 */
            out[f * B + i] = sqrtf(re[i] * re[i] + im[i] * im[i]) * scale;
        pos += step;
    }
    free(re); free(im);
    return F;
}

/* apply_filter_bank: lib.rs:310-334.  Row stride is the number of magnitude
 * bins (N/2), NOT the N/2+1 the JS caller lays rows out with; kept as is. */
static void so_apply_filter_bank(const float *bank, size_t bank_len, size_t num_filters,
                                 const float *mag, size_t bins, float *out)
{
    size_t i, j;
    for (i = 0; i < num_filters; i++) {
        float sum = 0.0f;
        size_t row = i * bins;
        if (bank_len != 0)
            for (j = 0; j < bins; j++)
                if (row + j < bank_len) sum += mag[j] * bank[row + j];
        out[i] = sum;
    }
}

/* the dB -> u8 law: lib.rs:277-293 (Rust `as u8` = saturating truncation) */
static uint8_t so_quantise(float mag, float gain_db_neg, float range_db, float range_db_reciprocal)
{
    float safe = mag > 1e-10f ? mag : 1e-10f;
    float db = 20.0f * so_log10f(safe);
    float v;
    if (db < gain_db_neg - range_db) return 0;
    if (db > gain_db_neg) return 255;
    v = (db - (gain_db_neg - range_db)) * range_db_reciprocal;
    if (!(v > 0.0f)) return 0;          /* also NaN -> 0 */
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

/* ------------------------------------------------------------------ */
/* compute_spectrogram_u8: lib.rs:199-304                               */
/* out_u8: frames x out_bins; mags (may be NULL): frames x N/2 linear   */
/* magnitudes (the engine's last_magnitude_buffer); *gmax: running max. */
/* bank == NULL or num_filters == 0 disables the filter bank.           */
/* ------------------------------------------------------------------ */
size_t so_compute_spectrogram_u8(const so_plan *p, const float *window,
                                 const float *x, size_t n, size_t noverlap,
                                 float gain_db, float range_db,
                                 const float *bank, size_t bank_len, size_t num_filters,
                                 uint8_t *out_u8, float *mags, float *gmax)
{
    const size_t N = p->n, B = N / 2, step = N - noverlap;
    const size_t F = so_num_frames(n, N, noverlap);
    const int use_bank = bank != NULL && num_filters > 0;
    const size_t out_bins = use_bank ? num_filters : B;     /* :217-221 */
    const float gain_db_neg = -gain_db;                      /* :232 */
    const float range_db_reciprocal = 255.0f / range_db;     /* :233 */
    const float scale = 2.0f / (float)N;                     /* :253 */
    float global_max = 0.0f;
    float *re = (float *)malloc(sizeof(float) * N);
    float *im = (float *)malloc(sizeof(float) * N);
    float *mag = (float *)malloc(sizeof(float) * B);
    float *filt = (float *)malloc(sizeof(float) * (out_bins ? out_bins : 1));
    size_t f, i, pos = 0;
    for (f = 0; f < F; f++) {
        const float *src;
        for (i = 0; i < N; i++) { re[i] = x[pos + i] * window[i]; im[i] = 0.0f; }
        so_fft_forward(p, re, im);
        for (i = 0; i < B; i++) {
            float m = sqrtf(re[i] * re[i] + im[i] * im[i]) * scale;
            mag[i] = m;
            if (m > global_max) global_max = m;              /* :261-263 */
            if (mags) mags[f * B + i] = m;                   /* :266 */
        }
        if (use_bank) { so_apply_filter_bank(bank, bank_len, num_filters, mag, B, filt); src = filt; }
        else src = mag;
        for (i = 0; i < out_bins; i++)
            out_u8[f * out_bins + i] = so_quantise(src[i], gain_db_neg, range_db, range_db_reciprocal);
        pos += step;
    }
    if (gmax) *gmax = global_max;
    free(re); free(im); free(mag); free(filt);
    return F;
}

/* get_peaks: lib.rs:349-387.  Returns the element count written (0 = the
 * reference's "empty vector" case). */
size_t so_get_peaks(const float *mags, size_t frames, size_t bins, float gmax,
                    float threshold_ratio, uint16_t *out)
{
    size_t f, i;
    float threshold;
    if (frames * bins == 0 || !(gmax > 0.0f)) return 0;      /* :350-352 */
    threshold = gmax * threshold_ratio;
    for (f = 0; f < frames; f++) {
        size_t max_idx = 0; float max_val = 0.0f;            /* fold init (0, 0.0): :372 */
        for (i = 0; i < bins; i++) {
            float v = mags[f * bins + i];
            if (v > max_val) { max_val = v; max_idx = i; }   /* strict >, first wins */
        }
        out[f] = max_val >= threshold ? (uint16_t)max_idx : (uint16_t)0xFFFF;
    }
    return frames;
}

/* get_peak_magnitudes: lib.rs:398-431 */
size_t so_get_peak_magnitudes(const float *mags, size_t frames, size_t bins, float gmax,
                              float threshold_ratio, float *out)
{
    size_t f, i;
    float threshold;
    if (frames * bins == 0 || !(gmax > 0.0f)) return 0;
    threshold = gmax * threshold_ratio;
    for (f = 0; f < frames; f++) {
        float max_val = 0.0f;
        for (i = 0; i < bins; i++) {
            float v = mags[f * bins + i];
            if (v > max_val) max_val = v;
        }
        out[f] = max_val >= threshold ? max_val : 0.0f;
    }
    return frames;
}

/* ------------------------------------------------------------------ */
/* compute_power_spectrum: lib.rs:1019-1118.  has_overlap == 0 encodes   */
/* Option::None.  out must hold fft_size/2+1 floats.  Returns the bin    */
/* count, or 0 for the reference's empty result.                         */
/* ------------------------------------------------------------------ */
size_t so_power_spectrum_hop(size_t fft_size, int has_overlap, float overlap_percent)
{
    float overlap = has_overlap ? overlap_percent : 0.0f;    /* :1031 */
    float h;
    size_t hop;
    if (overlap <= 0.0f || overlap >= 100.0f) h = (float)fft_size * 0.25f;       /* :1032-1034 */
    else h = (float)fft_size * (1.0f - overlap / 100.0f);                        /* :1036 */
    hop = h > 0.0f ? (size_t)h : 0;                          /* `as usize` saturates at 0 */
    return hop < 1 ? 1 : hop;                                /* :1038 */
}

size_t so_compute_power_spectrum(const float *x, size_t n, uint32_t sample_rate,
                                 size_t fft_size, const char *window_type,
                                 int has_overlap, float overlap_percent, float *out)
{
    so_plan *p;
    float *window, *re, *im, *spectrum;
    size_t hop, num_bins, offset = 0, frame_count = 0, i;
    float freq_resolution, max_freq, frame_count_f;
    if (n == 0) return 0;                                    /* :1026-1028 */
    p = so_plan_create(fft_size);
    if (!p) return 0;
    hop = so_power_spectrum_hop(fft_size, has_overlap, overlap_percent);
    window = (float *)malloc(sizeof(float) * fft_size);
    so_create_window(window_type, fft_size, 0.16f, window);  /* :1041 */
    freq_resolution = (float)sample_rate / (float)fft_size;  /* :1044 */
    max_freq = (float)sample_rate / 2.0f;
    num_bins = (size_t)(max_freq / freq_resolution) + 1;     /* :1046 */
    if (num_bins > fft_size / 2 + 1) num_bins = fft_size / 2 + 1;
    spectrum = (float *)calloc(num_bins, sizeof(float));
    re = (float *)malloc(sizeof(float) * fft_size);
    im = (float *)malloc(sizeof(float) * fft_size);
    while (offset + fft_size <= n) {                         /* :1058 */
        float dc_sum = 0.0f, dc_offset;
        for (i = 0; i < fft_size; i++) re[i] = x[offset + i] * window[i];        /* :1064-1066 */
        for (i = 0; i < fft_size; i++) dc_sum += re[i];      /* sequential f32 sum :1069-1072 */
        dc_offset = dc_sum / (float)fft_size;
        for (i = 0; i < fft_size; i++) { re[i] -= dc_offset; im[i] = 0.0f; }
        so_fft_forward(p, re, im);
        for (i = 0; i < num_bins; i++) {
            /* Complex::norm() is hypot(re, im) (num-complex 0.4.6); :1090-1092 */
            float magnitude = hypotf(re[i], im[i]);
            spectrum[i] += magnitude * magnitude;
        }
        frame_count++;
        offset += hop;
    }
    if (frame_count == 0) {                                  /* :1101-1103 */
        num_bins = 0;
    } else {
        frame_count_f = (float)frame_count;
        for (i = 0; i < num_bins; i++) {                     /* :1107-1115 */
            float avg_power = spectrum[i] / frame_count_f;
            float rms = sqrtf(avg_power);
            float psd = (rms * rms) / (float)fft_size;
            out[i] = 10.0f * so_log10f(psd > 1e-16f ? psd : 1e-16f);
        }
    }
    free(window); free(re); free(im); free(spectrum);
    so_plan_destroy(p);
    return num_bins;
}

/* `x as usize` for f32 in Rust: saturating, NaN -> 0 */
static size_t so_f32_as_usize(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 1.8446744e19f) return (size_t)-1;
    return (size_t)v;
}

/* find_peak_frequency_from_spectrum: lib.rs:1132-1179 */
float so_find_peak_frequency(const float *spectrum, size_t len, uint32_t sample_rate,
                             size_t fft_size, float flow_hz, float fhigh_hz)
{
    float freq_resolution, peak_db;
    size_t min_bin, max_bin, peak_bin, i;
    if (len == 0) return 0.0f;
    freq_resolution = (float)sample_rate / (float)fft_size;
    min_bin = so_f32_as_usize(flow_hz / freq_resolution);
    max_bin = so_f32_as_usize(fhigh_hz / freq_resolution);
    if (max_bin > len - 1) max_bin = len - 1;
    if (min_bin >= max_bin) return 0.0f;
    peak_bin = min_bin; peak_db = spectrum[min_bin];
    for (i = min_bin + 1; i <= max_bin; i++)
        if (spectrum[i] > peak_db) { peak_db = spectrum[i]; peak_bin = i; }
    if (peak_bin > min_bin && peak_bin < max_bin) {
        float db0 = spectrum[peak_bin - 1], db1 = spectrum[peak_bin], db2 = spectrum[peak_bin + 1];
        float a = (db2 - 2.0f * db1 + db0) / 2.0f;
        if (fabsf(a) > 1e-10f) {
            float bin_correction = (db0 - db2) / (4.0f * a);
            float refined_bin = (float)peak_bin + bin_correction;
            return refined_bin * freq_resolution;
        }
    }
    return (float)peak_bin * freq_resolution;
}

/* compute_wave_peaks: lib.rs:794-824 */
size_t so_compute_wave_peaks(const float *x, size_t n, size_t num_peaks, float *out)
{
    float step_size;
    size_t k, i;
    if (num_peaks == 0 || n == 0) return 0;
    step_size = (float)n / (float)num_peaks;
    for (k = 0; k < num_peaks; k++) {
        size_t start = so_f32_as_usize((float)k * step_size);
        size_t end = so_f32_as_usize(ceilf((float)(k + 1) * step_size));
        float m = 0.0f;
        if (end > n) end = n;
        for (i = start; i < end; i++) { float a = fabsf(x[i]); if (a > m) m = a; }   /* f32::max fold */
        out[k] = m;
    }
    return num_peaks;
}

/* find_global_max: lib.rs:834-840 */
float so_find_global_max(const float *x, size_t n)
{
    float m = 0.0f; size_t i;
    for (i = 0; i < n; i++) { float a = fabsf(x[i]); if (a > m) m = a; }
    return m;
}

/* WaveformEngine::get_peaks_in_range for one loaded channel: lib.rs:904-955.
 * Returns the number of floats written (target_width, or 1 when target_width == 0). */
size_t so_wave_peaks_in_range(const float *x, size_t data_len, size_t start_sample,
                              size_t end_sample, size_t target_width, float *out)
{
    size_t sample_range, px, i;
    float step;
    if (target_width == 0) { out[0] = 0.0f; return 1; }      /* vec![0.0; target_width.max(1)] :913 */
    for (px = 0; px < target_width; px++) out[px] = 0.0f;
    if (end_sample > data_len) end_sample = data_len;
    sample_range = end_sample > start_sample ? end_sample - start_sample : 0;
    if (sample_range == 0) return target_width;
    step = (float)sample_range / (float)target_width;
    for (px = 0; px < target_width; px++) {
        size_t pixel_start = so_f32_as_usize((float)px * step);
        size_t pixel_end = so_f32_as_usize(ceilf(((float)px + 1.0f) * step));
        size_t cs = start_sample + pixel_start, ce = start_sample + pixel_end;
        if (cs > data_len) cs = data_len;
        if (ce > data_len) ce = data_len;
        if (cs < ce && cs < data_len) {
            float m = 0.0f;
            for (i = cs; i < ce; i++) { float a = fabsf(x[i]); if (a > m) m = a; }
            out[px] = m;
        }
    }
    return target_width;
}

/* Vector probe of the msun log10f restatement, for the oracle's own unit test. */
void so_probe_log10f(const float *x, size_t n, float *out)
{
    size_t i;
    for (i = 0; i < n; i++) out[i] = so_log10f(x[i]);
}
