/*
 * sonostft_napi.c -- Node.js N-API addon over the C ABI of libsonostft
 * (include/sonostft.h).  It exports the classes and functions of the
 * reference's wasm-bindgen module (modules/spectrogram_wasm.d.ts:4-271) under
 * the same names, with the same argument order and the same typed-array result
 * types, so that modules/spectrogram.esm.js, batCallDetector.js, powerSpectrum.js
 * and wavesurfer.esm.js can run against it unchanged.  napi/sonoradar_gpu.js adds
 * the ES-module shell (default init / initSync) around this addon.
 *
 * Error behaviour: a non-zero libsonostft status becomes a thrown JS Error whose
 * message is sono_last_error(); a call on a freed object throws
 * "null pointer passed to rust", the message the wasm glue gives for that case.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sonostft.h"

#define NAPI_OK(call)                                                       \
    do {                                                                    \
        if ((call) != napi_ok) {                                            \
            napi_throw_error(env, NULL, "sonostft: N-API call failed: " #call); \
            return NULL;                                                    \
        }                                                                   \
    } while (0)

static napi_value throw_status(napi_env env, int status)
{
    (void)status;
    const char *msg = sono_last_error();
    napi_throw_error(env, NULL, (msg && msg[0]) ? msg : "sonostft: call failed");
    return NULL;
}

static napi_value undefined(napi_env env)
{
    napi_value v;
    napi_get_undefined(env, &v);
    return v;
}

/* ---- argument helpers ------------------------------------------------------ */

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv, napi_value *self)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, self, NULL) != napi_ok) return 0;
    for (size_t i = argc; i < want; ++i) napi_get_undefined(env, &argv[i]);
    return 1;
}

static int get_typed(napi_env env, napi_value v, napi_typedarray_type want, void **data, size_t *len)
{
    bool is = false;
    napi_typedarray_type t;
    *data = NULL; *len = 0;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return 0;
    if (napi_get_typedarray_info(env, v, &t, len, data, NULL, NULL) != napi_ok) return 0;
    return t == want;
}

static int get_u32(napi_env env, napi_value v, uint32_t *out)
{
    double d = 0;
    if (napi_get_value_double(env, v, &d) != napi_ok) return 0;
    *out = d > 0 ? (d >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)d) : 0;      /* wasm-bindgen: `>>> 0` of the number */
    return 1;
}

static int get_f64(napi_env env, napi_value v, double *out) { return napi_get_value_double(env, v, out) == napi_ok; }

static int is_nullish(napi_env env, napi_value v)
{
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 1;
    return t == napi_undefined || t == napi_null;
}

static char *get_string(napi_env env, napi_value v)
{
    size_t n = 0;
    if (napi_get_value_string_utf8(env, v, NULL, 0, &n) != napi_ok) return NULL;
    char *s = (char *)malloc(n + 1);
    if (!s) return NULL;
    if (napi_get_value_string_utf8(env, v, s, n + 1, &n) != napi_ok) { free(s); return NULL; }
    return s;
}

static napi_value make_typed(napi_env env, napi_typedarray_type t, size_t elem, size_t count, void **data)
{
    napi_value ab, ta;
    void *p = NULL;
    if (napi_create_arraybuffer(env, count * elem, &p, &ab) != napi_ok ||
        napi_create_typedarray(env, t, count, ab, 0, &ta) != napi_ok) {
        napi_throw_error(env, NULL, "sonostft: cannot allocate the result array");
        return NULL;
    }
    *data = p;
    return ta;
}

/* The result arrays are sized exactly: computed into a scratch buffer of the
 * upper bound, then copied into a fresh typed array of the returned length
 * (the wasm glue does the same `.slice()` out of linear memory). */
static napi_value typed_from(napi_env env, napi_typedarray_type t, size_t elem, const void *src, size_t count)
{
    void *dst = NULL;
    napi_value v = make_typed(env, t, elem, count, &dst);
    if (v && count) memcpy(dst, src, count * elem);
    return v;
}

/* ---- SpectrogramEngine ------------------------------------------------------- */

typedef struct { sono_engine *e; size_t last_frames; } engine_box;

static void engine_finalize(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    engine_box *b = (engine_box *)data;
    if (b->e) sono_engine_destroy(b->e);       /* FinalizationRegistry of the wasm glue (spectrogram_wasm.js:131-137) */
    free(b);
}

static engine_box *engine_of(napi_env env, napi_value self)
{
    engine_box *b = NULL;
    if (napi_unwrap(env, self, (void **)&b) != napi_ok || !b || !b->e) {
        napi_throw_error(env, NULL, "null pointer passed to rust");
        return NULL;
    }
    return b;
}

static napi_value engine_new(napi_env env, napi_callback_info info)
{
    napi_value argv[3], self;
    if (!get_args(env, info, 3, argv, &self)) return NULL;
    uint32_t n = 0;
    if (!get_u32(env, argv[0], &n)) { napi_throw_type_error(env, NULL, "fft_size must be a number"); return NULL; }
    char *win = get_string(env, argv[1]);
    if (!win) { napi_throw_type_error(env, NULL, "window_func must be a string"); return NULL; }
    double alpha = 0;
    const int has_alpha = !is_nullish(env, argv[2]) && get_f64(env, argv[2], &alpha);
    engine_box *b = (engine_box *)calloc(1, sizeof(*b));
    int rc = b ? sono_engine_create(n, win, has_alpha, (float)alpha, -1, &b->e) : SONO_ENOMEM;
    free(win);
    if (rc) { free(b); return throw_status(env, rc); }
    NAPI_OK(napi_wrap(env, self, b, engine_finalize, NULL, NULL));
    return self;
}

static napi_value engine_free(napi_env env, napi_callback_info info)
{
    napi_value self;
    size_t argc = 0;
    engine_box *b = NULL;
    NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));
    if (napi_unwrap(env, self, (void **)&b) == napi_ok && b && b->e) { sono_engine_destroy(b->e); b->e = NULL; }
    return undefined(env);
}

#define ENGINE_GETTER_U32(fn_name, call)                                     \
    static napi_value fn_name(napi_env env, napi_callback_info info)         \
    {                                                                        \
        napi_value self, out;                                                \
        size_t argc = 0;                                                     \
        NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));      \
        engine_box *b = engine_of(env, self);                                \
        if (!b) return NULL;                                                 \
        NAPI_OK(napi_create_uint32(env, call(b->e), &out));                  \
        return out;                                                          \
    }
ENGINE_GETTER_U32(engine_get_fft_size, sono_engine_fft_size)
ENGINE_GETTER_U32(engine_get_freq_bins, sono_engine_freq_bins)
ENGINE_GETTER_U32(engine_get_num_filters, sono_engine_num_filters)

static napi_value engine_get_global_max(napi_env env, napi_callback_info info)
{
    napi_value self, out;
    size_t argc = 0;
    NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    NAPI_OK(napi_create_double(env, (double)sono_global_max(b->e), &out));
    return out;
}

static napi_value engine_get_window_values(napi_env env, napi_callback_info info)
{
    napi_value self;
    size_t argc = 0;
    NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    const size_t n = sono_engine_fft_size(b->e);
    void *dst = NULL;
    napi_value out = make_typed(env, napi_float32_array, 4, n, &dst);
    if (!out) return NULL;
    int rc = sono_engine_window_values(b->e, (float *)dst, n);
    return rc ? throw_status(env, rc) : out;
}

static napi_value engine_load_filter_bank(napi_env env, napi_callback_info info)
{
    napi_value argv[2], self;
    if (!get_args(env, info, 2, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    void *w; size_t len; uint32_t nf = 0;
    if (!get_typed(env, argv[0], napi_float32_array, &w, &len) || !get_u32(env, argv[1], &nf)) {
        napi_throw_type_error(env, NULL, "load_filter_bank(Float32Array, number)");
        return NULL;
    }
    int rc = sono_engine_load_filter_bank(b->e, (const float *)w, len, nf);
    return rc ? throw_status(env, rc) : undefined(env);
}

#define ENGINE_VOID(fn_name, call)                                           \
    static napi_value fn_name(napi_env env, napi_callback_info info)         \
    {                                                                        \
        napi_value self;                                                     \
        size_t argc = 0;                                                     \
        NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));      \
        engine_box *b = engine_of(env, self);                                \
        if (!b) return NULL;                                                 \
        int rc = call(b->e);                                                 \
        return rc ? throw_status(env, rc) : undefined(env);                  \
    }
ENGINE_VOID(engine_clear_filter_bank, sono_engine_clear_filter_bank)
ENGINE_VOID(engine_release_memory, sono_engine_release_memory)

static napi_value engine_compute_spectrogram(napi_env env, napi_callback_info info)
{
    napi_value argv[2], self;
    if (!get_args(env, info, 2, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    void *x; size_t n; uint32_t v = 0;
    if (!get_typed(env, argv[0], napi_float32_array, &x, &n) || !get_u32(env, argv[1], &v)) {
        napi_throw_type_error(env, NULL, "compute_spectrogram(Float32Array, number)");
        return NULL;
    }
    const size_t bins = sono_engine_freq_bins(b->e);
    const size_t cap = sono_frames_for(n, sono_engine_fft_size(b->e), v) * bins;
    void *dst = NULL;
    napi_value out = make_typed(env, napi_float32_array, 4, cap, &dst);
    if (!out) return NULL;
    size_t frames = 0;
    int rc = sono_stft_mag_f32(b->e, (const float *)x, n, v, (float *)dst, cap, &frames);
    return rc ? throw_status(env, rc) : out;
}

static napi_value engine_compute_spectrogram_u8(napi_env env, napi_callback_info info)
{
    napi_value argv[4], self;
    if (!get_args(env, info, 4, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    void *x; size_t n; uint32_t v = 0; double gain = 0, range = 0;
    if (!get_typed(env, argv[0], napi_float32_array, &x, &n) || !get_u32(env, argv[1], &v) ||
        !get_f64(env, argv[2], &gain) || !get_f64(env, argv[3], &range)) {
        napi_throw_type_error(env, NULL, "compute_spectrogram_u8(Float32Array, number, number, number)");
        return NULL;
    }
    const size_t nf = sono_engine_num_filters(b->e);
    const size_t bins = nf ? nf : sono_engine_freq_bins(b->e);
    const size_t cap = sono_frames_for(n, sono_engine_fft_size(b->e), v) * bins;
    void *dst = NULL;
    napi_value out = make_typed(env, napi_uint8_array, 1, cap, &dst);
    if (!out) return NULL;
    size_t frames = 0;
    int rc = sono_stft_u8(b->e, (const float *)x, n, v, (float)gain, (float)range, (uint8_t *)dst, cap, &frames);
    if (rc) return throw_status(env, rc);
    b->last_frames = frames;
    return out;
}

static napi_value engine_get_peaks(napi_env env, napi_callback_info info)
{
    napi_value argv[1], self;
    if (!get_args(env, info, 1, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    double ratio = 0;
    if (!get_f64(env, argv[0], &ratio)) { napi_throw_type_error(env, NULL, "get_peaks(number)"); return NULL; }
    const size_t cap = b->last_frames ? b->last_frames : 1;
    uint16_t *tmp = (uint16_t *)malloc(cap * sizeof(uint16_t));
    size_t got = 0;
    int rc = tmp ? sono_peaks_u16(b->e, (float)ratio, tmp, cap, &got) : SONO_ENOMEM;
    napi_value out = rc ? throw_status(env, rc) : typed_from(env, napi_uint16_array, 2, tmp, got);
    free(tmp);
    return out;
}

static napi_value engine_get_peak_magnitudes(napi_env env, napi_callback_info info)
{
    napi_value argv[1], self;
    if (!get_args(env, info, 1, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    double ratio = 0;
    if (!get_f64(env, argv[0], &ratio)) { napi_throw_type_error(env, NULL, "get_peak_magnitudes(number)"); return NULL; }
    const size_t cap = b->last_frames ? b->last_frames : 1;
    float *tmp = (float *)malloc(cap * sizeof(float));
    size_t got = 0;
    int rc = tmp ? sono_peak_mags_f32(b->e, (float)ratio, tmp, cap, &got) : SONO_ENOMEM;
    napi_value out = rc ? throw_status(env, rc) : typed_from(env, napi_float32_array, 4, tmp, got);
    free(tmp);
    return out;
}

static napi_value engine_set_color_map(napi_env env, napi_callback_info info)
{
    napi_value argv[1], self;
    if (!get_args(env, info, 1, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    void *c; size_t len;
    if (!get_typed(env, argv[0], napi_uint8_array, &c, &len)) { napi_throw_type_error(env, NULL, "set_color_map(Uint8Array)"); return NULL; }
    int rc = sono_engine_set_color_map(b->e, (const uint8_t *)c, len);
    return rc ? throw_status(env, rc) : undefined(env);
}

static napi_value engine_set_spectrum_config(napi_env env, napi_callback_info info)
{
    napi_value argv[3], self;
    if (!get_args(env, info, 3, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    char *scale = get_string(env, argv[0]);
    double fmin = 0, fmax = 0;
    if (!scale || !get_f64(env, argv[1], &fmin) || !get_f64(env, argv[2], &fmax)) {
        free(scale);
        napi_throw_type_error(env, NULL, "set_spectrum_config(string, number, number)");
        return NULL;
    }
    int rc = sono_engine_set_spectrum_config(b->e, scale, (float)fmin, (float)fmax);
    free(scale);
    return rc ? throw_status(env, rc) : undefined(env);
}

static napi_value engine_compute_spectrogram_image(napi_env env, napi_callback_info info)
{
    napi_value argv[6], self;
    if (!get_args(env, info, 6, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    void *x; size_t n; uint32_t w = 0, h = 0, v = 0; double gain = 0, range = 0;
    if (!get_typed(env, argv[0], napi_float32_array, &x, &n) || !get_u32(env, argv[1], &w) || !get_u32(env, argv[2], &h) ||
        !get_u32(env, argv[3], &v) || !get_f64(env, argv[4], &gain) || !get_f64(env, argv[5], &range)) {
        napi_throw_type_error(env, NULL, "compute_spectrogram_image(Float32Array, width, height, noverlap, gain_db, range_db)");
        return NULL;
    }
    const size_t bytes = (size_t)w * h * 4;
    void *dst = NULL;
    napi_value out = make_typed(env, napi_uint8_array, 1, bytes, &dst);
    if (!out) return NULL;
    int rc = sono_spectrogram_image(b->e, (const float *)x, n, w, h, v, (float)gain, (float)range, (uint8_t *)dst, bytes);
    return rc ? throw_status(env, rc) : out;
}

/* ---- WaveformEngine ------------------------------------------------------------ */

typedef struct { sono_waveform *w; } wave_box;

static void wave_finalize(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    wave_box *b = (wave_box *)data;
    if (b->w) sono_waveform_destroy(b->w);
    free(b);
}

static wave_box *wave_of(napi_env env, napi_value self)
{
    wave_box *b = NULL;
    if (napi_unwrap(env, self, (void **)&b) != napi_ok || !b || !b->w) {
        napi_throw_error(env, NULL, "null pointer passed to rust");
        return NULL;
    }
    return b;
}

static napi_value wave_new(napi_env env, napi_callback_info info)
{
    napi_value self;
    size_t argc = 0;
    NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));
    wave_box *b = (wave_box *)calloc(1, sizeof(*b));
    int rc = b ? sono_waveform_create(&b->w) : SONO_ENOMEM;
    if (rc) { free(b); return throw_status(env, rc); }
    NAPI_OK(napi_wrap(env, self, b, wave_finalize, NULL, NULL));
    return self;
}

static napi_value wave_free(napi_env env, napi_callback_info info)
{
    napi_value self;
    size_t argc = 0;
    wave_box *b = NULL;
    NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));
    if (napi_unwrap(env, self, (void **)&b) == napi_ok && b && b->w) { sono_waveform_destroy(b->w); b->w = NULL; }
    return undefined(env);
}

static napi_value wave_resize(napi_env env, napi_callback_info info)
{
    napi_value argv[1], self;
    if (!get_args(env, info, 1, argv, &self)) return NULL;
    wave_box *b = wave_of(env, self);
    if (!b) return NULL;
    uint32_t n = 0;
    if (!get_u32(env, argv[0], &n)) { napi_throw_type_error(env, NULL, "resize(number)"); return NULL; }
    int rc = sono_waveform_resize(b->w, n);
    return rc ? throw_status(env, rc) : undefined(env);
}

static napi_value wave_load_channel(napi_env env, napi_callback_info info)
{
    napi_value argv[2], self;
    if (!get_args(env, info, 2, argv, &self)) return NULL;
    wave_box *b = wave_of(env, self);
    if (!b) return NULL;
    uint32_t idx = 0; void *x; size_t n;
    if (!get_u32(env, argv[0], &idx) || !get_typed(env, argv[1], napi_float32_array, &x, &n)) {
        napi_throw_type_error(env, NULL, "load_channel(number, Float32Array)");
        return NULL;
    }
    int rc = sono_waveform_load_channel(b->w, idx, (const float *)x, n);
    return rc ? throw_status(env, rc) : undefined(env);
}

static napi_value wave_get_peaks_in_range(napi_env env, napi_callback_info info)
{
    napi_value argv[4], self;
    if (!get_args(env, info, 4, argv, &self)) return NULL;
    wave_box *b = wave_of(env, self);
    if (!b) return NULL;
    uint32_t idx = 0, s0 = 0, s1 = 0, width = 0;
    if (!get_u32(env, argv[0], &idx) || !get_u32(env, argv[1], &s0) || !get_u32(env, argv[2], &s1) || !get_u32(env, argv[3], &width)) {
        napi_throw_type_error(env, NULL, "get_peaks_in_range(channel, start, end, width)");
        return NULL;
    }
    const size_t cap = width ? width : 1;
    void *dst = NULL;
    napi_value out = make_typed(env, napi_float32_array, 4, cap, &dst);
    if (!out) return NULL;
    size_t got = 0;
    int rc = sono_waveform_peaks_in_range(b->w, idx, s0, s1, width, (float *)dst, cap, &got);
    return rc ? throw_status(env, rc) : out;
}

static napi_value wave_get_channel_length(napi_env env, napi_callback_info info)
{
    napi_value argv[1], self, out;
    if (!get_args(env, info, 1, argv, &self)) return NULL;
    wave_box *b = wave_of(env, self);
    if (!b) return NULL;
    uint32_t idx = 0;
    if (!get_u32(env, argv[0], &idx)) { napi_throw_type_error(env, NULL, "get_channel_length(number)"); return NULL; }
    NAPI_OK(napi_create_double(env, (double)sono_waveform_channel_length(b->w, idx), &out));
    return out;
}

static napi_value wave_get_num_channels(napi_env env, napi_callback_info info)
{
    napi_value self, out;
    size_t argc = 0;
    NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));
    wave_box *b = wave_of(env, self);
    if (!b) return NULL;
    NAPI_OK(napi_create_double(env, (double)sono_waveform_num_channels(b->w), &out));
    return out;
}

#define WAVE_VOID(fn_name, call)                                             \
    static napi_value fn_name(napi_env env, napi_callback_info info)         \
    {                                                                        \
        napi_value self;                                                     \
        size_t argc = 0;                                                     \
        NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));      \
        wave_box *b = wave_of(env, self);                                    \
        if (!b) return NULL;                                                 \
        int rc = call(b->w);                                                 \
        return rc ? throw_status(env, rc) : undefined(env);                  \
    }
WAVE_VOID(wave_clear, sono_waveform_clear)
WAVE_VOID(wave_release_memory, sono_waveform_release_memory)

/* ---- free functions ------------------------------------------------------------ */

static napi_value fn_compute_power_spectrum(napi_env env, napi_callback_info info)
{
    napi_value argv[5], self;
    if (!get_args(env, info, 5, argv, &self)) return NULL;
    void *x; size_t n; uint32_t sr = 0, fft = 0; double ov = 0;
    char *win = NULL;
    if (!get_typed(env, argv[0], napi_float32_array, &x, &n) || !get_u32(env, argv[1], &sr) || !get_u32(env, argv[2], &fft) ||
        !(win = get_string(env, argv[3]))) {
        napi_throw_type_error(env, NULL, "compute_power_spectrum(Float32Array, sample_rate, fft_size, window_type, overlap_percent?)");
        return NULL;
    }
    const int has_ov = !is_nullish(env, argv[4]) && get_f64(env, argv[4], &ov);
    const size_t cap = (size_t)fft / 2 + 1;
    float *tmp = (float *)malloc(cap * sizeof(float));
    size_t bins = 0;
    int rc = tmp ? sono_power_spectrum((const float *)x, n, sr, fft, win, has_ov, (float)ov, -1, tmp, cap, &bins) : SONO_ENOMEM;
    free(win);
    napi_value out = rc ? throw_status(env, rc) : typed_from(env, napi_float32_array, 4, tmp, bins);
    free(tmp);
    return out;
}

static napi_value fn_find_peak_frequency_from_spectrum(napi_env env, napi_callback_info info)
{
    napi_value argv[5], self, out;
    if (!get_args(env, info, 5, argv, &self)) return NULL;
    void *s; size_t n; uint32_t sr = 0, fft = 0; double lo = 0, hi = 0;
    if (!get_typed(env, argv[0], napi_float32_array, &s, &n) || !get_u32(env, argv[1], &sr) || !get_u32(env, argv[2], &fft) ||
        !get_f64(env, argv[3], &lo) || !get_f64(env, argv[4], &hi)) {
        napi_throw_type_error(env, NULL, "find_peak_frequency_from_spectrum(Float32Array, sample_rate, fft_size, flow_hz, fhigh_hz)");
        return NULL;
    }
    NAPI_OK(napi_create_double(env, (double)sono_find_peak_frequency((const float *)s, n, sr, fft, (float)lo, (float)hi), &out));
    return out;
}

static napi_value fn_compute_wave_peaks(napi_env env, napi_callback_info info)
{
    napi_value argv[2], self;
    if (!get_args(env, info, 2, argv, &self)) return NULL;
    void *x; size_t n; uint32_t k = 0;
    if (!get_typed(env, argv[0], napi_float32_array, &x, &n) || !get_u32(env, argv[1], &k)) {
        napi_throw_type_error(env, NULL, "compute_wave_peaks(Float32Array, number)");
        return NULL;
    }
    float *tmp = (float *)malloc((k ? k : 1) * sizeof(float));
    size_t got = 0;
    int rc = tmp ? sono_wave_peaks((const float *)x, n, k, tmp, k ? k : 1, &got) : SONO_ENOMEM;
    napi_value out = rc ? throw_status(env, rc) : typed_from(env, napi_float32_array, 4, tmp, got);
    free(tmp);
    return out;
}

static napi_value fn_find_global_max(napi_env env, napi_callback_info info)
{
    napi_value argv[1], self, out;
    if (!get_args(env, info, 1, argv, &self)) return NULL;
    void *x; size_t n;
    if (!get_typed(env, argv[0], napi_float32_array, &x, &n)) { napi_throw_type_error(env, NULL, "find_global_max(Float32Array)"); return NULL; }
    NAPI_OK(napi_create_double(env, (double)sono_find_global_max((const float *)x, n), &out));
    return out;
}

static napi_value fn_device_count(napi_env env, napi_callback_info info)
{
    (void)info;
    napi_value out;
    NAPI_OK(napi_create_int32(env, sono_device_count(), &out));
    return out;
}

/* ---- module -------------------------------------------------------------------- */

#define METHOD(name, fn) { name, NULL, fn, NULL, NULL, NULL, napi_default, NULL }


/* ---- not in the wasm surface: BatCallDetector.processFullFile (batCallDetector.js:584-840) as one native call ----
 * engine.process_full_file(audio: Float32Array, sampleRate, flowKHz, fhighKHz, options, config) -> Array of objects
 * carrying BatCall's numeric fields under BatCall's own names (null where the reference leaves null).
 * options: {threshold_dB = -60, padding_ms = 5}; config: {hopPercent = 3.125, callThreshold_dB = -24,
 * minCallDuration_ms = 1, debugMode = false, timeExpansion = false,
 * keepSpectrograms = false (true: every call also carries spectrogram, timeFrames and freqBins, js:785-787)}.  Throws where the reference's promise rejects. */
static double prop_f64(napi_env env, napi_value obj, const char *name, double dflt)
{
    napi_value v;
    napi_valuetype t;
    double d;
    if (is_nullish(env, obj) || napi_get_named_property(env, obj, name, &v) != napi_ok) return dflt;
    if (napi_typeof(env, v, &t) != napi_ok) return dflt;
    if (t == napi_boolean) { bool b = false; napi_get_value_bool(env, v, &b); return b ? 1.0 : 0.0; }
    if (t != napi_number || napi_get_value_double(env, v, &d) != napi_ok) return dflt;
    return d;
}

static int set_f64(napi_env env, napi_value obj, const char *name, double v)
{
    napi_value x;
    if (v != v) { if (napi_get_null(env, &x) != napi_ok) return 0; }
    else if (napi_create_double(env, v, &x) != napi_ok) return 0;
    return napi_set_named_property(env, obj, name, x) == napi_ok;
}

static int set_idx(napi_env env, napi_value obj, const char *name, int64_t v)
{
    return set_f64(env, obj, name, v < 0 ? 0.0 / 0.0 : (double)v);
}

static int set_str(napi_env env, napi_value obj, const char *name, const char *v)
{
    napi_value x;
    return napi_create_string_utf8(env, v, NAPI_AUTO_LENGTH, &x) == napi_ok && napi_set_named_property(env, obj, name, x) == napi_ok;
}

/* call.spectrogram, call.timeFrames, call.freqBins of result `i` (batCallDetector.js:785-787): an Array of Float32Array
 * rows (views of one buffer per call), an Array of numbers, and the Float32Array `*freq_bins` shared by all calls of the
 * run as in the reference (created with the first call). */
static int set_spectrogram(napi_env env, napi_value o, sono_engine *e, size_t i, napi_value *freq_bins)
{
    size_t frames = 0, times = 0;
    uint32_t nb = 0;
    if (sono_full_file_call_spectrogram(e, i, NULL, 0, NULL, 0, NULL, 0, &frames, &times, &nb) != SONO_OK) return 0;
    void *db = NULL, *fb = NULL;
    napi_value all = make_typed(env, napi_float32_array, sizeof(float), frames * nb, &db);
    if (!all) return 0;
    if (!*freq_bins && !(*freq_bins = make_typed(env, napi_float32_array, sizeof(float), nb, &fb))) return 0;
    double *tf = (double *)malloc((times ? times : 1) * sizeof(double));
    if (!tf) return 0;
    int ok = sono_full_file_call_spectrogram(e, i, (float *)db, frames * nb, tf, times, (float *)fb, fb ? nb : 0, NULL, NULL, NULL) == SONO_OK;
    napi_value rows, tarr, ab;
    ok = ok && napi_get_typedarray_info(env, all, NULL, NULL, NULL, &ab, NULL) == napi_ok &&
         napi_create_array_with_length(env, frames, &rows) == napi_ok && napi_create_array_with_length(env, times, &tarr) == napi_ok;
    for (size_t f = 0; ok && f < frames; ++f) {
        napi_value row;
        ok = napi_create_typedarray(env, napi_float32_array, nb, ab, f * nb * sizeof(float), &row) == napi_ok &&
             napi_set_element(env, rows, (uint32_t)f, row) == napi_ok;
    }
    for (size_t f = 0; ok && f < times; ++f) {
        napi_value v;
        ok = napi_create_double(env, tf[f], &v) == napi_ok && napi_set_element(env, tarr, (uint32_t)f, v) == napi_ok;
    }
    free(tf);
    return ok && napi_set_named_property(env, o, "spectrogram", rows) == napi_ok &&
           napi_set_named_property(env, o, "timeFrames", tarr) == napi_ok &&
           napi_set_named_property(env, o, "freqBins", *freq_bins) == napi_ok;
}

/* engine.set_option(name: string, on: boolean): the engine's switches (sono_engine_set_option).  Not part of the wasm
 * surface. */
static napi_value engine_set_option(napi_env env, napi_callback_info info)
{
    napi_value argv[2], self;
    if (!get_args(env, info, 2, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    char name[64];
    size_t len = 0;
    bool on = false;
    if (napi_get_value_string_utf8(env, argv[0], name, sizeof name, &len) != napi_ok || napi_get_value_bool(env, argv[1], &on) != napi_ok) {
        napi_throw_type_error(env, NULL, "set_option(string, boolean)");
        return NULL;
    }
    const int rc = sono_engine_set_option(b->e, name, on ? 1 : 0);
    return rc ? throw_status(env, rc) : undefined(env);
}

static napi_value engine_process_full_file(napi_env env, napi_callback_info info)
{
    static const char *const types[] = {"CF", "FM", "CF-FM"};
    static const char *const qualities[] = {"Very Poor", "Poor", "Normal", "Good", "Excellent"};
    napi_value argv[6], self, arr;
    if (!get_args(env, info, 6, argv, &self)) return NULL;
    engine_box *b = engine_of(env, self);
    if (!b) return NULL;
    void *x = NULL;
    size_t n = 0;
    double sr = 0, flow = 0, fhigh = 0;
    /* the recording: a Float32Array as the reference's processFullFile takes it, or the Int16Array of a decoded 16-bit
     * WAV channel (mono), which goes to the GPU as it is: half the bytes over the link, no float copy on the host */
    const int is_f32 = get_typed(env, argv[0], napi_float32_array, &x, &n);
    const int pcm16 = !is_f32 && get_typed(env, argv[0], napi_int16_array, &x, &n);
    if ((!is_f32 && !pcm16) || !get_f64(env, argv[1], &sr) || !get_f64(env, argv[2], &flow) || !get_f64(env, argv[3], &fhigh)) {
        napi_throw_type_error(env, NULL, "process_full_file(Float32Array | Int16Array, sampleRate, flowKHz, fhighKHz, options?, config?)");
        return NULL;
    }
    sono_full_file_options opt;
    opt.threshold_db = prop_f64(env, argv[4], "threshold_dB", -60);
    if (opt.threshold_db == 0) opt.threshold_db = -60;                 /* `options.threshold_dB || -60`, js:585 */
    opt.padding_ms = prop_f64(env, argv[4], "padding_ms", 5);
    if (opt.padding_ms == 0) opt.padding_ms = 5;                       /* js:586 */
    opt.hop_percent = prop_f64(env, argv[5], "hopPercent", 3.125);
    opt.call_threshold_db = prop_f64(env, argv[5], "callThreshold_dB", -24);
    opt.min_call_duration_ms = prop_f64(env, argv[5], "minCallDuration_ms", 1);
    opt.debug_mode = (prop_f64(env, argv[5], "debugMode", 0) != 0 ? SONO_DEBUG_MODE : 0) |
                     (prop_f64(env, argv[5], "timeExpansion", 0) != 0 ? SONO_TIME_EXPANSION : 0);
    const int keep = prop_f64(env, argv[5], "keepSpectrograms", 0) != 0;
    if (keep) opt.debug_mode |= SONO_KEEP_CALL_SPECTROGRAMS;
    napi_value freq_bins = NULL;
    opt.threads = 0;
    size_t count = 0;
    int rc = pcm16 ? sono_process_full_file_pcm16(b->e, (const int16_t *)x, n, 1, 0, sr, flow, fhigh, &opt, NULL, 0, &count)
                   : sono_process_full_file(b->e, (const float *)x, n, sr, flow, fhigh, &opt, NULL, 0, &count);
    if (rc != SONO_OK && rc != SONO_ECAPACITY) return throw_status(env, rc);
    sono_full_file_call *calls = (sono_full_file_call *)malloc((count ? count : 1) * sizeof(*calls));
    if (!calls) { napi_throw_error(env, NULL, "sonostft: out of memory"); return NULL; }
    rc = sono_full_file_results(b->e, calls, count, &count);
    if (rc != SONO_OK) { free(calls); return throw_status(env, rc); }
    if (napi_create_array_with_length(env, count, &arr) != napi_ok) { free(calls); return NULL; }
    for (size_t i = 0; i < count; ++i) {
        const sono_full_file_call *k = &calls[i];
        const sono_call_measurement *m = &k->m;
        napi_value o;
        int ok = napi_create_object(env, &o) == napi_ok;
        ok = ok && set_f64(env, o, "startTime_s", m->start_time_s) && set_f64(env, o, "endTime_s", m->end_time_s) &&
             set_f64(env, o, "duration_ms", m->duration_ms) && set_f64(env, o, "peakFreq_kHz", m->peak_freq_khz) &&
             set_f64(env, o, "peakFreq_ms", m->peak_freq_ms) && set_f64(env, o, "peakPower_dB", m->peak_power_db) &&
             set_f64(env, o, "highFreq_kHz", m->high_freq_khz) && set_f64(env, o, "highFreq_ms", m->high_freq_ms) &&
             set_f64(env, o, "startFreq_kHz", m->start_freq_khz) && set_f64(env, o, "startFreq_ms", m->start_freq_ms) &&
             set_f64(env, o, "startFreqTime_s", m->start_freq_time_s) && set_f64(env, o, "endFreq_kHz", m->end_freq_khz) &&
             set_f64(env, o, "endFreq_ms", m->end_freq_ms) && set_f64(env, o, "endFreqTime_s", m->end_freq_time_s) &&
             set_f64(env, o, "lowFreq_kHz", m->low_freq_khz) && set_f64(env, o, "lowFreq_ms", m->low_freq_ms) &&
             set_f64(env, o, "characteristicFreq_kHz", m->characteristic_freq_khz) &&
             set_f64(env, o, "characteristicFreq_ms", m->characteristic_freq_ms) && set_f64(env, o, "kneeFreq_kHz", m->knee_freq_khz) &&
             set_f64(env, o, "kneeFreq_ms", m->knee_freq_ms) && set_f64(env, o, "kneeTime_ms", m->knee_freq_ms) &&
             set_f64(env, o, "heelFreq_kHz", m->heel_freq_khz) && set_f64(env, o, "heelFreq_ms", m->heel_freq_ms) &&
             set_f64(env, o, "bandwidth_kHz", m->bandwidth_khz) &&
             set_f64(env, o, "highFreqThreshold_dB_used", m->high_threshold_db_used) &&
             set_f64(env, o, "lowFreqThreshold_dB_used", m->low_threshold_db_used) && set_idx(env, o, "peakFrameIdx", m->peak_frame) &&
             set_idx(env, o, "highFreqFrameIdx", m->high_freq_frame) && set_idx(env, o, "startFreqFrameIdx", m->start_freq_frame) &&
             set_idx(env, o, "startFreqBinIdx", m->start_freq_bin) && set_idx(env, o, "endFrameIdx_forLowFreq", m->end_frame_for_low_freq) &&
             set_idx(env, o, "kneeFrameIdx", m->knee_frame) && set_idx(env, o, "heelFrameIdx", m->heel_frame) &&
             set_f64(env, o, "Flow", k->flow_hz) && set_f64(env, o, "Fhigh", k->fhigh_khz) && set_f64(env, o, "snr_dB", k->snr_db) &&
             set_str(env, o, "callType", types[k->call_type]) && set_str(env, o, "quality", qualities[k->quality]) &&
             set_f64(env, o, "roi", (double)k->roi) && set_f64(env, o, "startFrame", (double)k->start_frame) &&
             set_f64(env, o, "endFrame", (double)k->end_frame) && set_f64(env, o, "highpass_kHz", k->highpass_khz) &&
             (!keep || set_spectrogram(env, o, b->e, i, &freq_bins)) && napi_set_element(env, arr, (uint32_t)i, o) == napi_ok;
        if (!ok) { free(calls); napi_throw_error(env, NULL, "sonostft: cannot build the result array"); return NULL; }
    }
    free(calls);
    return arr;
}

static napi_value init(napi_env env, napi_value exports)
{
    const napi_property_descriptor engine_methods[] = {
        METHOD("free", engine_free),
        METHOD("get_fft_size", engine_get_fft_size),
        METHOD("get_freq_bins", engine_get_freq_bins),
        METHOD("get_num_filters", engine_get_num_filters),
        METHOD("get_window_values", engine_get_window_values),
        METHOD("load_filter_bank", engine_load_filter_bank),
        METHOD("clear_filter_bank", engine_clear_filter_bank),
        METHOD("compute_spectrogram", engine_compute_spectrogram),
        METHOD("compute_spectrogram_u8", engine_compute_spectrogram_u8),
        METHOD("get_peaks", engine_get_peaks),
        METHOD("get_peak_magnitudes", engine_get_peak_magnitudes),
        METHOD("get_global_max", engine_get_global_max),
        METHOD("set_color_map", engine_set_color_map),
        METHOD("set_spectrum_config", engine_set_spectrum_config),
        METHOD("compute_spectrogram_image", engine_compute_spectrogram_image),
        METHOD("release_memory", engine_release_memory),
        METHOD("process_full_file", engine_process_full_file),
        METHOD("set_option", engine_set_option),
    };
    const napi_property_descriptor wave_methods[] = {
        METHOD("free", wave_free),
        METHOD("resize", wave_resize),
        METHOD("load_channel", wave_load_channel),
        METHOD("get_peaks_in_range", wave_get_peaks_in_range),
        METHOD("get_channel_length", wave_get_channel_length),
        METHOD("get_num_channels", wave_get_num_channels),
        METHOD("clear", wave_clear),
        METHOD("release_memory", wave_release_memory),
    };
    napi_value engine_class, wave_class;
    NAPI_OK(napi_define_class(env, "SpectrogramEngine", NAPI_AUTO_LENGTH, engine_new, NULL,
                              sizeof(engine_methods) / sizeof(engine_methods[0]), engine_methods, &engine_class));
    NAPI_OK(napi_define_class(env, "WaveformEngine", NAPI_AUTO_LENGTH, wave_new, NULL,
                              sizeof(wave_methods) / sizeof(wave_methods[0]), wave_methods, &wave_class));
    const napi_property_descriptor props[] = {
        { "SpectrogramEngine", NULL, NULL, NULL, NULL, engine_class, napi_default, NULL },
        { "WaveformEngine", NULL, NULL, NULL, NULL, wave_class, napi_default, NULL },
        METHOD("compute_power_spectrum", fn_compute_power_spectrum),
        METHOD("find_peak_frequency_from_spectrum", fn_find_peak_frequency_from_spectrum),
        METHOD("compute_wave_peaks", fn_compute_wave_peaks),
        METHOD("find_global_max", fn_find_global_max),
        METHOD("device_count", fn_device_count),
    };
    NAPI_OK(napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props));
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
