// sono_capi.hip -- the C ABI of libsonostft (include/sonostft.h): engine objects,
// device buffers, host<->device staging and kernel dispatch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <algorithm>
#include <cstring>
#include <thread>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/sonostft.h"
#include "sono_core.h"
#include "sono_engine.h"
#include "sono_host.h"
#include "sono_kernels.h"
#include "sono_tables.h"
#include "sono_tile.h"

namespace sono {

int noalloc_init_device(int device);      // sono_capi_noalloc.hip

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    (void)hipGetLastError();   // clear the sticky error so that later calls report their own
    return fail(e == hipErrorOutOfMemory ? SONO_ENOMEM : SONO_EDEVICE,
                std::string(what) + ": " + hipGetErrorString(e));
}

namespace {
std::atomic<unsigned int> g_host_threads{0};
}

unsigned int host_threads()
{
    const unsigned int set = g_host_threads.load(std::memory_order_relaxed);
    if (set) return set;
    static const unsigned int dflt = [] {
        if (const char *v = std::getenv("SONO_HOST_THREADS")) return (unsigned int)std::max(1, atoi(v));
        if (const char *v = std::getenv("SONO_COPY_THREADS")) return (unsigned int)std::max(1, atoi(v));      // older name
        unsigned int local = 1;
        if (const char *v = std::getenv("LOCAL_WORLD_SIZE")) local = (unsigned int)std::max(1, atoi(v));
        const unsigned int hw = std::max(1u, std::thread::hardware_concurrency());
        return std::min(16u, std::max(1u, hw / local));
    }();
    return dflt;
}

bool stage_timing()
{
    static const bool on = std::getenv("SONO_TIMING") != nullptr;
    return on;
}

}  // namespace sono

using sono::DevBuf;
using sono::DeviceGuard;
using sono::Tables;
using sono::fail;
using sono::hip_fail;

namespace {

int build_tables(unsigned int N, const float *window, Tables &t, bool with_tile = true)
{
    t.N = N;
    std::vector<float2> twh = sono::make_twiddles(N / 2, N / 2);
    std::vector<float2> twf = sono::make_twiddles(N, N / 2 + 1);
    HIP_TRY(hipMalloc((void **)&t.d_window, N * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&t.d_tw_half, twh.size() * sizeof(float2)));
    HIP_TRY(hipMalloc((void **)&t.d_tw_full, twf.size() * sizeof(float2)));
    HIP_TRY(hipMemcpy(t.d_window, window, N * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_tw_half, twh.data(), twh.size() * sizeof(float2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_tw_full, twf.data(), twf.size() * sizeof(float2), hipMemcpyHostToDevice));
    if (!with_tile) return SONO_OK;
    hipError_t e = sono::tile_tables_build(N, t.tile);
    if (e != hipSuccess) return hip_fail(e, "tile_tables_build");
    return SONO_OK;
}

// compute_power_spectrum is a free function in the reference (lib.rs:1019): no engine keeps its window and twiddle
// tables between calls, and the application calls it on a few thousand samples at a time (powerSpectrum.js:49),
// where building and freeing tables and staging buffers per call costs more than the transform.  The last few
// (device, fft size, window) combinations therefore keep theirs, with staging buffers up to kPowerKeepBytes; a call
// takes the cache's lock for its whole duration (the reference is single-threaded; concurrent callers take turns).
// The entries are never freed: they are small, and a static destructor must not call into a HIP runtime that may
// already be gone.
constexpr size_t kPowerKeepBytes = (size_t)64 << 20;
struct PowerCtx {
    int device = -1;
    unsigned int N = 0;
    std::string window;
    Tables t;
    DevBuf d_x, d_pow, d_db;
    unsigned long long used = 0;
};
std::mutex g_power_mu;
PowerCtx *g_power_ctx[4] = {nullptr, nullptr, nullptr, nullptr};
unsigned long long g_power_clock = 0;

// -> the context for (device, N, window), tables built; nullptr with *rc set on failure.  Caller holds g_power_mu.
PowerCtx *power_ctx(int device, unsigned int N, const char *window, int *rc)
{
    const std::string w = window ? window : "";
    PowerCtx *lru = nullptr;
    for (PowerCtx *&c : g_power_ctx) {
        if (c && c->device == device && c->N == N && c->window == w) { c->used = ++g_power_clock; return c; }
        if (!c) { c = new PowerCtx(); c->used = 0; }
        if (!lru || c->used < lru->used) lru = c;
    }
    lru->t.release();
    lru->d_x.release(); lru->d_pow.release(); lru->d_db.release();
    lru->device = -1;
    std::vector<float> win(N);
    sono::host_make_window(w.c_str(), N, 0.16f, win.data());                  // lib.rs:1041
    *rc = build_tables(N, win.data(), lru->t, false);
    if (*rc) { lru->t.release(); return nullptr; }
    lru->device = device; lru->N = N; lru->window = w; lru->used = ++g_power_clock;
    return lru;
}

// The switches an engine starts with: the environment is read here, once per engine, and never on a launch
// (sono_engine_set_option changes them afterwards).
bool env_flag(const char *name)
{
    const char *v = std::getenv(name);
    return v && v[0] && v[0] != '0';
}

// Picks the kernel for one launch.  The register-tiled kernel covers the common
// sizes; everything else (and the engine option "force_generic", used by the tests to
// compare the two) goes through the generic LDS-staged kernel.
hipError_t dispatch_stft(int mode, const sono::StftArgs &a, const Tables &t, hipStream_t stream)
{
    if (!t.force_generic && sono::tile_supports(mode, a, t.tile))
        return sono::launch_stft_tile(mode, a, t.tile, stream);
    return sono::launch_stft_generic(mode, a, stream);
}

}  // namespace

struct sono_waveform {
    std::vector<std::vector<float>> channels;
};

extern "C" {

const char *sono_last_error(void) { return sono::g_err.c_str(); }

uint32_t sono_abi_version(void) { return 1; }

int sono_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

size_t sono_frames_for(size_t n, uint32_t fft_size, uint32_t noverlap)
{
    if (fft_size == 0 || noverlap >= fft_size) return 0;
    const size_t step = fft_size - noverlap;
    return n >= fft_size ? (n - fft_size) / step + 1 : 0;
}

int sono_engine_create(uint32_t fft_size, const char *window, int has_alpha, float alpha, int device,
                       sono_engine **out)
{
    if (!out) return fail(SONO_EINVAL, "sono_engine_create: out is NULL");
    *out = nullptr;
    if (!sono::supported_fft_size(fft_size))
        return fail(SONO_EINVAL, "sono_engine_create: fft_size must be a power of two in [256, 4096]");
    if (sono_device_count() <= 0) return fail(SONO_EDEVICE, "sono_engine_create: no HIP device available");
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError(), "hipGetDevice");
    DeviceGuard guard(dev);
    if (!guard.ok()) return fail(SONO_EDEVICE, "sono_engine_create: cannot select the device");
    sono_engine *e = new (std::nothrow) sono_engine();
    if (!e) return fail(SONO_ENOMEM, "sono_engine_create: out of host memory");
    e->N = fft_size;
    e->device = dev;
    e->alpha = has_alpha ? alpha : 0.16f;                  // lib.rs:47
    e->window_name = window ? window : "";
    e->window.resize(fft_size);
    sono::host_make_window(e->window_name.c_str(), fft_size, e->alpha, e->window.data());
    int rc = build_tables(fft_size, e->window.data(), e->tables);
    e->tables.force_generic = env_flag("SONO_FORCE_GENERIC");
    e->tables.tile.any_hop = env_flag("SONO_FORCE_ANY_HOP");
    e->tables.tile.balance.fixed = env_flag("SONO_FIXED_TILES");
    e->tables.tile.balance.debug = env_flag("SONO_BALANCE_DEBUG");
    e->no_streaming = env_flag("SONO_NO_STREAMING");
    e->host_measure = env_flag("SONO_HOST_MEASURE");
    e->no_upload_overlap = env_flag("SONO_NO_UPLOAD_OVERLAP");
    e->no_measure_overlap = env_flag("SONO_NO_MEASURE_OVERLAP");
    if (rc == SONO_OK && hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
        rc = hip_fail(hipGetLastError(), "hipStreamCreate");
    if (rc == SONO_OK) rc = sono::noalloc_init_device(dev);      // this GPU's fixed scratch of sono_detect_runs_static_dev
    if (rc != SONO_OK) { sono_engine_destroy(e); return rc; }
    *out = e;
    return SONO_OK;
}

void sono_engine_destroy(sono_engine *e)
{
    if (!e) return;
    if (e->measure_engine) { sono_engine_destroy(e->measure_engine); e->measure_engine = nullptr; }
    if (e->frame_engine) { sono_engine_destroy(e->frame_engine); e->frame_engine = nullptr; }
    for (sono_engine *&a : e->analysis_engine) if (a) { sono_engine_destroy(a); a = nullptr; }
    DeviceGuard guard(e->device);
    if (e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
    e->tables.release();
    e->d_in.release(); e->d_out.release(); e->d_peak_bin.release(); e->d_peak_mag.release();
    e->d_gmax.release(); e->d_mags.release(); e->d_bank.release(); e->d_rec.release();
    e->d_db.release(); e->d_flags.release(); e->d_edges.release(); e->d_runs.release();
    e->d_count.release(); e->d_select_tmp.release();
    e->h_in.release(); e->h_out.release();
    e->d_refine.release(); e->h_refine.release();
    e->d_frame_table.release();
    for (auto &tables : e->frame_tables) tables.release();
    for (auto &slot : e->measure_slot) slot.release();
    e->d_measure_tables.release(); e->d_measure_scratch.release();
    for (auto &slot : e->upload_slot) slot.release();
    for (auto &slot : e->ragged) {
        if (slot.done) { (void)hipEventSynchronize(slot.done); (void)hipEventDestroy(slot.done); }
        slot.buf.release();
    }
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->pipe) sono::stream_pipe_destroy(e->pipe);
    delete e;
}

int sono_set_host_threads(uint32_t n)
{
    sono::g_host_threads.store(n > 256 ? 256u : n, std::memory_order_relaxed);
    return SONO_OK;
}

uint32_t sono_host_threads(void) { return sono::host_threads(); }

int sono_engine_set_option(sono_engine *e, const char *name, int value)
{
    if (!e || !name) return fail(SONO_EINVAL, "sono_engine_set_option: NULL argument");
    const std::string n(name);
    const bool on = value != 0;
    if (n == "force_generic") e->tables.force_generic = on;
    else if (n == "force_any_hop") e->tables.tile.any_hop = on;
    else if (n == "fixed_tiles") e->tables.tile.balance.fixed = on;
    else if (n == "balance_debug") e->tables.tile.balance.debug = on;
    else if (n == "no_streaming") e->no_streaming = on;
    else if (n == "bank_two_kernels") e->bank_two_kernels = on;
    else if (n == "host_measure") e->host_measure = on;
    else if (n == "no_upload_overlap") e->no_upload_overlap = on;
    else if (n == "no_measure_overlap") e->no_measure_overlap = on;
    else if (n == "small_upload_pieces") e->small_upload_pieces = on;
    else return fail(SONO_EINVAL, "sono_engine_set_option: unknown option '" + n + "'");
    return SONO_OK;
}

int sono_engine_get_option(const sono_engine *e, const char *name, int *value)
{
    if (!e || !name || !value) return fail(SONO_EINVAL, "sono_engine_get_option: NULL argument");
    const std::string n(name);
    if (n == "force_generic") *value = e->tables.force_generic;
    else if (n == "force_any_hop") *value = e->tables.tile.any_hop;
    else if (n == "fixed_tiles") *value = e->tables.tile.balance.fixed;
    else if (n == "balance_debug") *value = e->tables.tile.balance.debug;
    else if (n == "no_streaming") *value = e->no_streaming;
    else if (n == "bank_two_kernels") *value = e->bank_two_kernels;
    else if (n == "host_measure") *value = e->host_measure;
    else if (n == "no_upload_overlap") *value = e->no_upload_overlap;
    else if (n == "no_measure_overlap") *value = e->no_measure_overlap;
    else if (n == "small_upload_pieces") *value = e->small_upload_pieces;
    else return fail(SONO_EINVAL, "sono_engine_get_option: unknown option '" + n + "'");
    return SONO_OK;
}

uint32_t sono_engine_fft_size(const sono_engine *e) { return e ? e->N : 0; }
uint32_t sono_engine_freq_bins(const sono_engine *e) { return e ? e->N / 2 : 0; }
uint32_t sono_engine_num_filters(const sono_engine *e) { return e ? e->num_filters : 0; }
int sono_engine_device(const sono_engine *e) { return e ? e->device : -1; }

int sono_engine_window_values(const sono_engine *e, float *out, size_t cap)
{
    if (!e || !out) return fail(SONO_EINVAL, "sono_engine_window_values: NULL argument");
    if (cap < e->N) return fail(SONO_ECAPACITY, "sono_engine_window_values: buffer too small");
    std::memcpy(out, e->window.data(), e->N * sizeof(float));
    return SONO_OK;
}

int sono_engine_load_filter_bank(sono_engine *e, const float *w, size_t len, uint32_t num_filters)
{
    if (!e || (!w && len)) return fail(SONO_EINVAL, "sono_engine_load_filter_bank: NULL argument");
    e->bank.assign(w, w + len);                            // lib.rs:91-93
    e->num_filters = num_filters;
    e->use_bank = true;
    e->bank_taps_ready = false;
    return SONO_OK;
}

int sono_engine_clear_filter_bank(sono_engine *e)
{
    if (!e) return fail(SONO_EINVAL, "sono_engine_clear_filter_bank: NULL engine");
    e->bank.clear(); e->num_filters = 0; e->use_bank = false; e->bank_taps_ready = false;
    return SONO_OK;
}

int sono_engine_release_memory(sono_engine *e)
{
    if (!e) return fail(SONO_EINVAL, "sono_engine_release_memory: NULL engine");
    if (e->measure_engine) (void)sono_engine_release_memory(e->measure_engine);
    if (e->frame_engine) (void)sono_engine_release_memory(e->frame_engine);
    for (sono_engine *a : e->analysis_engine) if (a) (void)sono_engine_release_memory(a);
    DeviceGuard guard(e->device);
    (void)hipStreamSynchronize(e->stream);
    e->bank.clear(); e->num_filters = 0; e->use_bank = false; e->bank_taps_ready = false;       // lib.rs:683-692
    e->color_map.clear();
    e->has_last = false; e->last_peak_bin.clear(); e->last_peak_mag.clear(); e->last_gmax = 0.0f;
    e->d_in.release(); e->d_out.release(); e->d_peak_bin.release(); e->d_peak_mag.release();
    e->d_mags.release(); e->d_bank.release(); e->d_rec.release();
    e->d_db.release(); e->d_flags.release(); e->d_edges.release(); e->d_runs.release(); e->d_select_tmp.release();
    e->h_in.release(); e->h_out.release(); e->d_refine.release(); e->h_refine.release();
    e->full_file_calls.clear(); e->full_file_calls.shrink_to_fit();
    for (auto &slot : e->measure_slot) slot.release();
    e->d_measure_tables.release(); e->d_measure_scratch.release();
    for (auto &slot : e->upload_slot) slot.release();
    for (auto &tables : e->frame_tables) tables.release();
    e->d_frame_table.release();
    e->scan_segments.clear(); e->scan_segments.shrink_to_fit();
    e->full_file_slices.clear(); e->full_file_slices.shrink_to_fit();
    if (e->pipe) { sono::stream_pipe_destroy(e->pipe); e->pipe = nullptr; }
    return SONO_OK;
}

int sono_engine_synchronize(sono_engine *e)
{
    if (!e) return fail(SONO_EINVAL, "sono_engine_synchronize: NULL engine");
    DeviceGuard guard(e->device);
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SONO_OK;
}

}  // extern "C"

namespace {

void fill_args(const sono_engine *e, const sono_batch &in, uint32_t noverlap, sono::StftArgs &a)
{
    std::memset(&a, 0, sizeof(a));
    a.x = in.d_samples;
    a.n_streams = in.n_streams;
    a.stream_stride = in.stream_stride;
    a.frames_per_stream = sono_frames_for(in.samples_per_stream, e->N, noverlap);
    a.hop = e->N - noverlap;
    a.N = e->N;
    a.window = e->tables.d_window;
    a.tw_half = e->tables.d_tw_half;
    a.tw_full = e->tables.d_tw_full;
    a.inv_n = 1.0f / (float)e->N;
}

int check_batch(const sono_engine *e, const sono_batch *in, uint32_t noverlap, const char *who)
{
    if (!e || !in) return fail(SONO_EINVAL, std::string(who) + ": NULL argument");
    if (noverlap >= e->N) return fail(SONO_EINVAL, std::string(who) + ": noverlap must be < fft_size");
    if (in->n_streams && in->samples_per_stream && !in->d_samples)
        return fail(SONO_EINVAL, std::string(who) + ": d_samples is NULL");
    if (in->n_streams > 1 && in->stream_stride < in->samples_per_stream)
        return fail(SONO_EINVAL, std::string(who) + ": stream_stride < samples_per_stream");
    return SONO_OK;
}

// Every kernel writes its rows with 16-byte (magnitudes) or 4-byte (u8 codes) stores.
int check_aligned(const void *p, unsigned int bytes, const char *who, const char *what)
{
    if (reinterpret_cast<uintptr_t>(p) & (bytes - 1))
        return fail(SONO_EINVAL, std::string(who) + ": " + what + " must be " + std::to_string(bytes) + "-byte aligned");
    return SONO_OK;
}

// Uploads the samples the frames actually touch.
int stage_input(sono_engine *e, const float *x, size_t n, uint32_t noverlap, size_t frames)
{
    (void)n;
    const size_t used = (frames - 1) * (size_t)(e->N - noverlap) + e->N;
    HIP_TRY(e->d_in.reserve(used * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(e->d_in.p, x, used * sizeof(float), hipMemcpyHostToDevice, e->stream));
    return SONO_OK;
}

}  // namespace

extern "C" {

int sono_stft_mag_f32_dev(sono_engine *e, const sono_batch *in, uint32_t noverlap, float *d_out,
                          void *hip_stream)
{
    int rc = check_batch(e, in, noverlap, "sono_stft_mag_f32_dev");
    if (rc) return rc;
    sono::StftArgs a;
    fill_args(e, *in, noverlap, a);
    if (a.n_streams * a.frames_per_stream == 0) return SONO_OK;
    if (!d_out) return fail(SONO_EINVAL, "sono_stft_mag_f32_dev: d_out is NULL");
    if ((rc = check_aligned(d_out, 16, "sono_stft_mag_f32_dev", "d_out")) != SONO_OK) return rc;
    a.out_mag = d_out;
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    HIP_TRY(dispatch_stft(sono::MODE_MAG, a, e->tables, s));
    return SONO_OK;
}

int sono_stft_u8_dev(sono_engine *e, const sono_batch *in, uint32_t noverlap, float gain_db,
                     float range_db, uint8_t *d_out, uint16_t *d_peak_bin, float *d_peak_mag,
                     uint32_t *d_gmax_bits, void *hip_stream)
{
    int rc = check_batch(e, in, noverlap, "sono_stft_u8_dev");
    if (rc) return rc;
    sono::StftArgs a;
    fill_args(e, *in, noverlap, a);
    if (a.n_streams * a.frames_per_stream == 0) return SONO_OK;
    if (!d_out) return fail(SONO_EINVAL, "sono_stft_u8_dev: d_out is NULL");
    if ((rc = check_aligned(d_out, 4, "sono_stft_u8_dev", "d_out")) != SONO_OK) return rc;
    const sono::Quant q = sono::make_quant(e->N, gain_db, range_db);
    a.out_u8 = d_out; a.peak_bin = d_peak_bin; a.peak_mag = d_peak_mag; a.gmax_bits = d_gmax_bits;
    a.q_a = q.a; a.q_c = q.c; a.q_floor = q.floor;
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    HIP_TRY(dispatch_stft(sono::MODE_U8, a, e->tables, s));
    return SONO_OK;
}

int sono_stft_u8_pcm16_dev(sono_engine *e, const int16_t *d_pcm, size_t n_streams, size_t stream_stride,
                           size_t samples_per_stream, uint32_t channels, uint32_t channel, uint32_t noverlap,
                           float gain_db, float range_db, uint8_t *d_out, uint16_t *d_peak_bin,
                           float *d_peak_mag, uint32_t *d_gmax_bits, void *hip_stream)
{
    const char *who = "sono_stft_u8_pcm16_dev";
    if (!e) return fail(SONO_EINVAL, std::string(who) + ": NULL argument");
    if (noverlap >= e->N) return fail(SONO_EINVAL, std::string(who) + ": noverlap must be < fft_size");
    if (channels == 0 || channel >= channels) return fail(SONO_EINVAL, std::string(who) + ": channel out of range");
    if (n_streams && samples_per_stream && !d_pcm) return fail(SONO_EINVAL, std::string(who) + ": d_pcm is NULL");
    if (n_streams > 1 && stream_stride < samples_per_stream)
        return fail(SONO_EINVAL, std::string(who) + ": stream_stride < samples_per_stream");
    if (!d_out && !d_peak_bin && !d_peak_mag && !d_gmax_bits)
        return fail(SONO_EINVAL, std::string(who) + ": every output is NULL");
    sono_batch in;
    in.d_samples = nullptr; in.n_streams = n_streams; in.stream_stride = stream_stride; in.samples_per_stream = samples_per_stream;
    sono::StftArgs a;
    fill_args(e, in, noverlap, a);
    if (a.n_streams * a.frames_per_stream == 0) return SONO_OK;
    a.x16 = d_pcm; a.x16_channels = channels; a.x16_channel = channel;
    const sono::Quant q = sono::make_quant(e->N, gain_db, range_db);
    a.out_u8 = d_out; a.peak_bin = d_peak_bin; a.peak_mag = d_peak_mag; a.gmax_bits = d_gmax_bits;
    a.q_a = q.a; a.q_c = q.c; a.q_floor = q.floor;
    if (!sono::tile_supports(sono::MODE_U8, a, e->tables.tile))
        return fail(SONO_EINVAL, std::string(who) + ": needs fft_size 256, 512, 1024, 2048 or 4096 and a 4-byte aligned d_out");
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    HIP_TRY(sono::launch_stft_tile(sono::MODE_U8, a, e->tables.tile, s));
    return SONO_OK;
}

int sono_stft_features_dev(sono_engine *e, const sono_batch *in, uint32_t noverlap, uint16_t *d_peak_bin,
                           float *d_peak_mag, uint32_t *d_gmax_bits, void *hip_stream)
{
    int rc = check_batch(e, in, noverlap, "sono_stft_features_dev");
    if (rc) return rc;
    sono::StftArgs a;
    fill_args(e, *in, noverlap, a);
    if (a.n_streams * a.frames_per_stream == 0) return SONO_OK;
    if (!d_peak_bin && !d_peak_mag && !d_gmax_bits)
        return fail(SONO_EINVAL, "sono_stft_features_dev: every output is NULL");
    const sono::Quant q = sono::make_quant(e->N, 20.0f, 80.0f);      // unused without rows; keeps the arithmetic defined
    a.out_u8 = nullptr; a.peak_bin = d_peak_bin; a.peak_mag = d_peak_mag; a.gmax_bits = d_gmax_bits;
    a.q_a = q.a; a.q_c = q.c; a.q_floor = q.floor;
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    HIP_TRY(dispatch_stft(sono::MODE_U8, a, e->tables, s));
    return SONO_OK;
}

int sono_ragged_rows(const uint64_t *lengths, size_t n_streams, uint32_t fft_size, uint32_t noverlap, uint64_t *row0)
{
    if (!row0 || (n_streams && !lengths)) return fail(SONO_EINVAL, "sono_ragged_rows: NULL argument");
    if (fft_size == 0 || noverlap >= fft_size) return fail(SONO_EINVAL, "sono_ragged_rows: noverlap must be < fft_size");
    uint64_t rows = 0;
    for (size_t s = 0; s < n_streams; ++s) {
        row0[s] = rows;
        rows += sono_frames_for((size_t)lengths[s], fft_size, noverlap);
    }
    row0[n_streams] = rows;
    return SONO_OK;
}

namespace {

// Builds and uploads the per-stream / per-tile tables of a ragged launch and runs it.  Where the tiled
// kernel cannot take the launch (forced generic kernel, unaligned output), every stream gets a launch
// of its own through the usual dispatch.
int run_ragged(sono_engine *e, const sono_ragged_batch *in, uint32_t noverlap, int mode, sono::StftArgs &a, void *hip_stream,
               const char *who)
{
    if (!e || !in) return fail(SONO_EINVAL, std::string(who) + ": NULL argument");
    if (noverlap >= e->N) return fail(SONO_EINVAL, std::string(who) + ": noverlap must be < fft_size");
    if (in->n_streams && (!in->offsets || !in->lengths || !in->d_samples))
        return fail(SONO_EINVAL, std::string(who) + ": NULL stream table or samples");
    const uint32_t N = e->N, B = N / 2;
    const unsigned int fpw = sono::tile_frames_per_wave(N);
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    a.x = in->d_samples;
    a.n_streams = 0; a.stream_stride = 0; a.frames_per_stream = 0;
    const bool tiled = fpw && !e->tables.force_generic &&
                       [&] { sono::StftArgs p = a; p.frames_per_stream = 2; return sono::tile_supports(mode, p, e->tables.tile); }();
    std::vector<sono::RaggedStreamEntry> streams;
    std::vector<unsigned int> tile_stream;
    unsigned long long rows = 0, tiles = 0;
    for (size_t i = 0; i < in->n_streams; ++i) {
        const unsigned long long F = sono_frames_for((size_t)in->lengths[i], N, noverlap);
        if (tiled) {
            const unsigned long long t = (F + fpw - 1) / fpw;
            if (tiles + t >= 0xFFFFFFFFull || in->n_streams >= 0xFFFFFFFFull)
                return fail(SONO_EINVAL, std::string(who) + ": more than 2^32 - 1 tiles or streams in one launch");
            streams.push_back({in->offsets[i], F, rows, tiles});
            tile_stream.insert(tile_stream.end(), (size_t)t, (unsigned int)i);
            tiles += t;
        } else if (F) {
            sono::StftArgs one = a;
            one.x = in->d_samples + in->offsets[i];
            one.n_streams = 1; one.stream_stride = in->lengths[i]; one.frames_per_stream = F;
            if (one.out_mag) one.out_mag += rows * B;
            if (one.out_u8) one.out_u8 += rows * B;
            if (one.peak_bin) one.peak_bin += rows;
            if (one.peak_mag) one.peak_mag += rows;
            HIP_TRY(dispatch_stft(mode, one, e->tables, s));
        }
        rows += F;
    }
    if (!tiled || tiles == 0) return SONO_OK;
    sono_engine::RaggedSlot &slot = e->ragged[e->ragged_next++ & 1u];
    if (slot.used) HIP_TRY(hipEventSynchronize(slot.done));          // the launch that last read this slot has finished
    if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    const size_t sb = streams.size() * sizeof(sono::RaggedStreamEntry);
    const size_t tb = tile_stream.size() * sizeof(unsigned int);
    HIP_TRY(slot.buf.reserve(sb + tb));
    HIP_TRY(hipMemcpyAsync(slot.buf.p, streams.data(), sb, hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipMemcpyAsync((char *)slot.buf.p + sb, tile_stream.data(), tb, hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipStreamSynchronize(e->copy_stream));                   // the tables are in HBM before the kernel is queued
    sono::RaggedPlan plan;
    plan.n_tiles = tiles;
    plan.d_streams = (const sono::RaggedStreamEntry *)slot.buf.p;
    plan.d_tile_stream = (const unsigned int *)((const char *)slot.buf.p + sb);
    HIP_TRY(sono::launch_stft_tile_ragged(mode, a, e->tables.tile, plan, s));
    HIP_TRY(hipEventRecord(slot.done, s));
    slot.used = true;
    return SONO_OK;
}

void fill_common(const sono_engine *e, uint32_t noverlap, sono::StftArgs &a)
{
    std::memset(&a, 0, sizeof(a));
    a.hop = e->N - noverlap;
    a.N = e->N;
    a.window = e->tables.d_window;
    a.tw_half = e->tables.d_tw_half;
    a.tw_full = e->tables.d_tw_full;
    a.inv_n = 1.0f / (float)e->N;
}

}  // namespace

int sono_stft_u8_ragged_dev(sono_engine *e, const sono_ragged_batch *in, uint32_t noverlap, float gain_db, float range_db,
                            uint8_t *d_out, uint16_t *d_peak_bin, float *d_peak_mag, uint32_t *d_gmax_bits, void *hip_stream)
{
    if (!e) return fail(SONO_EINVAL, "sono_stft_u8_ragged_dev: NULL engine");
    if (noverlap >= e->N) return fail(SONO_EINVAL, "sono_stft_u8_ragged_dev: noverlap must be < fft_size");
    if (!d_out && !d_peak_bin && !d_peak_mag && !d_gmax_bits)
        return fail(SONO_EINVAL, "sono_stft_u8_ragged_dev: every output is NULL");
    sono::StftArgs a;
    fill_common(e, noverlap, a);
    const sono::Quant q = sono::make_quant(e->N, gain_db, range_db);
    a.out_u8 = d_out; a.peak_bin = d_peak_bin; a.peak_mag = d_peak_mag; a.gmax_bits = d_gmax_bits;
    a.q_a = q.a; a.q_c = q.c; a.q_floor = q.floor;
    if (const int rc = check_aligned(d_out, 4, "sono_stft_u8_ragged_dev", "d_out")) return rc;
    return run_ragged(e, in, noverlap, sono::MODE_U8, a, hip_stream, "sono_stft_u8_ragged_dev");
}

int sono_stft_mag_f32_ragged_dev(sono_engine *e, const sono_ragged_batch *in, uint32_t noverlap, float *d_out, void *hip_stream)
{
    if (!e) return fail(SONO_EINVAL, "sono_stft_mag_f32_ragged_dev: NULL engine");
    if (noverlap >= e->N) return fail(SONO_EINVAL, "sono_stft_mag_f32_ragged_dev: noverlap must be < fft_size");
    if (!d_out) return fail(SONO_EINVAL, "sono_stft_mag_f32_ragged_dev: d_out is NULL");
    sono::StftArgs a;
    fill_common(e, noverlap, a);
    a.out_mag = d_out;
    if (const int rc = check_aligned(d_out, 16, "sono_stft_mag_f32_ragged_dev", "d_out")) return rc;
    return run_ragged(e, in, noverlap, sono::MODE_MAG, a, hip_stream, "sono_stft_mag_f32_ragged_dev");
}

}  // extern "C"

// sono_stft_mag_f32_ragged_dev for a caller that reads bins lo..hi of the rows only (the detector's region analysis and
// measurement: 316 of 512 bins at the usual band): the other bins need not be written, and at the analysis hop the
// kernel's time is the rows' way to HBM.
int sono::stft_mag_ragged_band(sono_engine *e, const sono_ragged_batch *in, uint32_t noverlap, uint32_t lo, uint32_t hi, float *d_out)
{
    if (!e || !d_out || noverlap >= e->N) return fail(SONO_EINVAL, "stft_mag_ragged_band: bad argument");
    sono::StftArgs a;
    fill_common(e, noverlap, a);
    a.out_mag = d_out;
    if (hi >= lo && hi < e->N / 2) { a.band_lo = lo; a.band_hi = hi; }
    if (const int rc = check_aligned(d_out, 16, "stft_mag_ragged_band", "d_out")) return rc;
    return run_ragged(e, in, noverlap, sono::MODE_MAG, a, nullptr, "stft_mag_ragged_band");
}

extern "C" {

int sono_stft_band_active_dev(sono_engine *e, const sono_batch *in, uint32_t noverlap, uint32_t min_bin, uint32_t max_bin,
                              double target_energy_sum, uint8_t *d_active, double *d_band_energy, void *hip_stream)
{
    int rc = check_batch(e, in, noverlap, "sono_stft_band_active_dev");
    if (rc) return rc;
    sono::StftArgs a;
    fill_args(e, *in, noverlap, a);
    if (a.n_streams * a.frames_per_stream == 0) return SONO_OK;
    if (!d_active) return fail(SONO_EINVAL, "sono_stft_band_active_dev: d_active is NULL");
    if (max_bin >= e->N / 2 || min_bin > max_bin)
        return fail(SONO_EINVAL, "sono_stft_band_active_dev: need min_bin <= max_bin < fft_size/2");
    a.band_lo = min_bin; a.band_hi = max_bin; a.band_target = target_energy_sum;
    a.band_active = d_active; a.band_energy = d_band_energy;
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    if (e->tables.force_generic || !sono::tile_supports(sono::MODE_BAND, a, e->tables.tile)) {
        // the generic kernel has no fused form: magnitudes to a scratch matrix, then the band pass
        const size_t F = (size_t)(a.n_streams * a.frames_per_stream), B = e->N / 2;
        HIP_TRY(e->d_mags.reserve(F * B * sizeof(float)));
        rc = sono_stft_mag_f32_dev(e, in, noverlap, (float *)e->d_mags.p, hip_stream);
        if (rc) return rc;
        return sono_detect_band_active_dev(e, (const float *)e->d_mags.p, F, (uint32_t)B, min_bin, max_bin,
                                           target_energy_sum, d_active, d_band_energy, hip_stream);
    }
    HIP_TRY(sono::launch_stft_tile(sono::MODE_BAND, a, e->tables.tile, s));
    return SONO_OK;
}

// The survey pass (a night-long channel: per-frame peak features at 75 % overlap, and the detector's fast scan, whose
// 50 % overlap makes its frames the even frames of the feature pass): one read of the samples where the tiled kernel
// has the fused form (fft 1024), else the two passes one after the other.  d_pcm != NULL: 16-bit sample frames.
static int survey_pass(sono_engine *e, const float *d_x, const int16_t *d_pcm, uint32_t channels, uint32_t channel,
                       size_t n_streams, size_t stream_stride, size_t samples_per_stream, uint32_t min_bin, uint32_t max_bin,
                       double target, uint16_t *d_peak_bin, float *d_peak_mag, uint32_t *d_gmax_bits, uint8_t *d_active,
                       double *d_band_energy, void *hip_stream, const char *who)
{
    if (!e) return fail(SONO_EINVAL, std::string(who) + ": NULL argument");
    const uint32_t N = e->N, V = N - N / 4;
    if (d_pcm && (channels == 0 || channel >= channels)) return fail(SONO_EINVAL, std::string(who) + ": channel out of range");
    if (n_streams && samples_per_stream && !d_x && !d_pcm) return fail(SONO_EINVAL, std::string(who) + ": the samples are NULL");
    if (n_streams > 1 && stream_stride < samples_per_stream)
        return fail(SONO_EINVAL, std::string(who) + ": stream_stride < samples_per_stream");
    if (max_bin >= N / 2 || min_bin > max_bin) return fail(SONO_EINVAL, std::string(who) + ": need min_bin <= max_bin < fft_size/2");
    sono_batch in;
    in.d_samples = d_x; in.n_streams = n_streams; in.stream_stride = stream_stride; in.samples_per_stream = samples_per_stream;
    sono::StftArgs a;
    fill_args(e, in, V, a);
    if (a.n_streams * a.frames_per_stream == 0) return SONO_OK;
    if (!d_active) return fail(SONO_EINVAL, std::string(who) + ": d_active is NULL");
    a.x16 = d_pcm; a.x16_channels = d_pcm ? channels : 0; a.x16_channel = d_pcm ? channel : 0;
    a.peak_bin = d_peak_bin; a.peak_mag = d_peak_mag; a.gmax_bits = d_gmax_bits;
    a.band_lo = min_bin; a.band_hi = max_bin; a.band_target = target;
    a.band_active = d_active; a.band_energy = d_band_energy;
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    if (!e->tables.force_generic && sono::tile_supports(sono::MODE_SURVEY, a, e->tables.tile)) {
        HIP_TRY(sono::launch_stft_tile(sono::MODE_SURVEY, a, e->tables.tile, s));
        return SONO_OK;
    }
    // two passes: the features at hop N/4, then the fast scan at hop N/2
    int rc;
    if (d_peak_bin || d_peak_mag || d_gmax_bits) {
        rc = d_pcm ? sono_stft_u8_pcm16_dev(e, d_pcm, n_streams, stream_stride, samples_per_stream, channels, channel, V, 20.0f, 80.0f,
                                            nullptr, d_peak_bin, d_peak_mag, d_gmax_bits, hip_stream)
                   : sono_stft_features_dev(e, &in, V, d_peak_bin, d_peak_mag, d_gmax_bits, hip_stream);
        if (rc) return rc;
    }
    return d_pcm ? sono_stft_band_active_pcm16_dev(e, d_pcm, n_streams, stream_stride, samples_per_stream, channels, channel, N / 2,
                                                   min_bin, max_bin, target, d_active, d_band_energy, hip_stream)
                 : sono_stft_band_active_dev(e, &in, N / 2, min_bin, max_bin, target, d_active, d_band_energy, hip_stream);
}

int sono_stft_survey_dev(sono_engine *e, const sono_batch *in, uint32_t min_bin, uint32_t max_bin, double target_energy_sum,
                         uint16_t *d_peak_bin, float *d_peak_mag, uint32_t *d_gmax_bits, uint8_t *d_active, double *d_band_energy,
                         void *hip_stream)
{
    if (!in) return fail(SONO_EINVAL, "sono_stft_survey_dev: NULL argument");
    return survey_pass(e, in->d_samples, nullptr, 0, 0, in->n_streams, in->stream_stride, in->samples_per_stream, min_bin, max_bin,
                       target_energy_sum, d_peak_bin, d_peak_mag, d_gmax_bits, d_active, d_band_energy, hip_stream, "sono_stft_survey_dev");
}

int sono_stft_survey_pcm16_dev(sono_engine *e, const int16_t *d_pcm, size_t n_streams, size_t stream_stride, size_t samples_per_stream,
                               uint32_t channels, uint32_t channel, uint32_t min_bin, uint32_t max_bin, double target_energy_sum,
                               uint16_t *d_peak_bin, float *d_peak_mag, uint32_t *d_gmax_bits, uint8_t *d_active, double *d_band_energy,
                               void *hip_stream)
{
    if (!d_pcm && n_streams && samples_per_stream) return fail(SONO_EINVAL, "sono_stft_survey_pcm16_dev: d_pcm is NULL");
    return survey_pass(e, nullptr, d_pcm, channels, channel, n_streams, stream_stride, samples_per_stream, min_bin, max_bin,
                       target_energy_sum, d_peak_bin, d_peak_mag, d_gmax_bits, d_active, d_band_energy, hip_stream,
                       "sono_stft_survey_pcm16_dev");
}

int sono_stft_band_active_pcm16_dev(sono_engine *e, const int16_t *d_pcm, size_t n_streams, size_t stream_stride,
                                    size_t samples_per_stream, uint32_t channels, uint32_t channel, uint32_t noverlap,
                                    uint32_t min_bin, uint32_t max_bin, double target_energy_sum, uint8_t *d_active,
                                    double *d_band_energy, void *hip_stream)
{
    const char *who = "sono_stft_band_active_pcm16_dev";
    if (!e) return fail(SONO_EINVAL, std::string(who) + ": NULL argument");
    if (noverlap >= e->N) return fail(SONO_EINVAL, std::string(who) + ": noverlap must be < fft_size");
    if (channels == 0 || channel >= channels) return fail(SONO_EINVAL, std::string(who) + ": channel out of range");
    if (n_streams && samples_per_stream && !d_pcm) return fail(SONO_EINVAL, std::string(who) + ": d_pcm is NULL");
    if (n_streams > 1 && stream_stride < samples_per_stream)
        return fail(SONO_EINVAL, std::string(who) + ": stream_stride < samples_per_stream");
    sono_batch in;
    in.d_samples = nullptr; in.n_streams = n_streams; in.stream_stride = stream_stride; in.samples_per_stream = samples_per_stream;
    sono::StftArgs a;
    fill_args(e, in, noverlap, a);
    if (a.n_streams * a.frames_per_stream == 0) return SONO_OK;
    if (!d_active) return fail(SONO_EINVAL, std::string(who) + ": d_active is NULL");
    if (max_bin >= e->N / 2 || min_bin > max_bin)
        return fail(SONO_EINVAL, std::string(who) + ": need min_bin <= max_bin < fft_size/2");
    a.x16 = d_pcm; a.x16_channels = channels; a.x16_channel = channel;
    a.band_lo = min_bin; a.band_hi = max_bin; a.band_target = target_energy_sum;
    a.band_active = d_active; a.band_energy = d_band_energy;
    if (!sono::tile_supports(sono::MODE_BAND, a, e->tables.tile))
        return fail(SONO_EINVAL, std::string(who) + ": needs fft_size 256, 512, 1024, 2048 or 4096");
    DeviceGuard guard(e->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    HIP_TRY(sono::launch_stft_tile(sono::MODE_BAND, a, e->tables.tile, s));
    return SONO_OK;
}

int sono_stft_mag_f32(sono_engine *e, const float *x, size_t n, uint32_t noverlap, float *out,
                      size_t cap, size_t *frames)
{
    if (frames) *frames = 0;
    if (!e || (!x && n)) return fail(SONO_EINVAL, "sono_stft_mag_f32: NULL argument");
    if (noverlap >= e->N) return fail(SONO_EINVAL, "sono_stft_mag_f32: noverlap must be < fft_size");
    const size_t F = sono_frames_for(n, e->N, noverlap), B = e->N / 2;
    if (F == 0) return SONO_OK;                           // n < N: empty result, not an error
    if (!out || cap < F * B) return fail(SONO_ECAPACITY, "sono_stft_mag_f32: output buffer too small");
    DeviceGuard guard(e->device);
    int rc = stage_input(e, x, n, noverlap, F);
    if (rc) return rc;
    HIP_TRY(e->d_out.reserve(F * B * sizeof(float)));
    sono_batch b{(const float *)e->d_in.p, 1, n, n};
    rc = sono_stft_mag_f32_dev(e, &b, noverlap, (float *)e->d_out.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, e->d_out.p, F * B * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (frames) *frames = F;
    return SONO_OK;
}

int sono_stft_u8(sono_engine *e, const float *x, size_t n, uint32_t noverlap, float gain_db,
                 float range_db, uint8_t *out, size_t cap, size_t *frames)
{
    if (frames) *frames = 0;
    if (!e || (!x && n)) return fail(SONO_EINVAL, "sono_stft_u8: NULL argument");
    if (noverlap >= e->N) return fail(SONO_EINVAL, "sono_stft_u8: noverlap must be < fft_size");
    const size_t F = sono_frames_for(n, e->N, noverlap), B = e->N / 2;
    const bool bank = e->use_bank && e->num_filters > 0;            // lib.rs:217-221
    const size_t out_bins = bank ? e->num_filters : B;
    // the call replaces the engine's peak state even when it yields no frames (lib.rs:299-301)
    e->has_last = true; e->last_peak_bin.clear(); e->last_peak_mag.clear(); e->last_gmax = 0.0f;
    if (F == 0) return SONO_OK;
    if (!out || cap < F * out_bins) return fail(SONO_ECAPACITY, "sono_stft_u8: output buffer too small");
    DeviceGuard guard(e->device);
    if (!bank && F > 2 * sono::stream_chunk_frames(e->N - noverlap) && !e->no_streaming) {
        // a long recording: chunks of frames through pinned rings, copies and kernels overlapped (sono_stream.hip)
        e->last_peak_bin.resize(F);
        e->last_peak_mag.resize(F);
        uint32_t gbits = 0;
        const int rc = sono::stream_stft_u8(e, &e->pipe, x, F, noverlap, gain_db, range_db, out, e->last_peak_bin.data(),
                                            e->last_peak_mag.data(), &gbits);
        if (rc) { e->last_peak_bin.clear(); e->last_peak_mag.clear(); return rc; }
        std::memcpy(&e->last_gmax, &gbits, 4);
        if (frames) *frames = F;
        return SONO_OK;
    }
    int rc = stage_input(e, x, n, noverlap, F);
    if (rc) return rc;
    HIP_TRY(e->d_out.reserve(F * out_bins));
    HIP_TRY(e->d_peak_bin.reserve(F * sizeof(uint16_t)));
    HIP_TRY(e->d_peak_mag.reserve(F * sizeof(float)));
    HIP_TRY(e->d_gmax.reserve(sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(e->d_gmax.p, 0, sizeof(uint32_t), e->stream));
    sono_batch b{(const float *)e->d_in.p, 1, n, n};
    if (!bank) {
        rc = sono_stft_u8_dev(e, &b, noverlap, gain_db, range_db, (uint8_t *)e->d_out.p,
                              (uint16_t *)e->d_peak_bin.p, (float *)e->d_peak_mag.p,
                              (uint32_t *)e->d_gmax.p, nullptr);
        if (rc) return rc;
    } else {
        if (!e->bank_taps_ready) {
            // the bank's non-zero taps, listed once per loaded bank and kept on the device
            std::vector<unsigned int> start, bin;
            std::vector<float> w;
            sono::bank_taps(e->bank.data(), e->bank.size(), e->num_filters, B, start, bin, w);
            const size_t sb = start.size() * sizeof(unsigned int), tb = bin.size() * sizeof(unsigned int);
            HIP_TRY(e->d_bank.reserve(sb + 2 * tb + 16));
            HIP_TRY(hipStreamSynchronize(e->stream));         // an earlier launch may still be reading the old list
            HIP_TRY(hipMemcpy(e->d_bank.p, start.data(), sb, hipMemcpyHostToDevice));
            if (tb) {
                HIP_TRY(hipMemcpy((char *)e->d_bank.p + sb, bin.data(), tb, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy((char *)e->d_bank.p + sb + tb, w.data(), tb, hipMemcpyHostToDevice));
            }
            e->bank_tap_count = bin.size();
            e->bank_taps_ready = true;
        }
        const size_t sb = e->num_filters * sizeof(unsigned int), tb = e->bank_tap_count * sizeof(unsigned int);
        // A sparse bank on the default display size runs inside the transform's epilogue, straight from the power image
        // in LDS: the samples are read once, the codes written once, no magnitude ever reaches HBM.
        {
            sono::StftArgs a;
            fill_args(e, b, noverlap, a);
            a.out_u8 = (uint8_t *)e->d_out.p; a.peak_bin = (uint16_t *)e->d_peak_bin.p; a.peak_mag = (float *)e->d_peak_mag.p;
            a.gmax_bits = (uint32_t *)e->d_gmax.p;
            a.bank_tap_bin = (const unsigned int *)((char *)e->d_bank.p + sb);
            a.bank_tap_w = (const float *)((char *)e->d_bank.p + sb + tb);
            a.bank_filters = e->num_filters;
            a.bank_max_taps = e->num_filters ? (unsigned int)(e->bank_tap_count / e->num_filters) : 0u;
            const sono::BankQuant bq = sono::make_bank_quant(gain_db, range_db);
            a.bank_q_a = bq.a; a.bank_q_c = bq.c;
            if (!e->tables.force_generic && !e->bank_two_kernels && sono::tile_supports(sono::MODE_BANK, a, e->tables.tile)) {
                HIP_TRY(sono::launch_stft_tile(sono::MODE_BANK, a, e->tables.tile, e->stream));
                goto copy_back;
            }
        }
        // The magnitudes only exist between the two kernels: they are formed for one piece of the recording at a time
        // (pieces start on multiples of 16 frames, so every frame meets the partner it has in the whole transform) in a
        // scratch of at most 2 GiB, whatever the recording's length.
        const size_t piece = std::max<size_t>(16, (((size_t)2 << 30) / (B * sizeof(float))) / 16 * 16);
        HIP_TRY(e->d_mags.reserve(std::min(F, piece) * B * sizeof(float)));
        const size_t hop = e->N - noverlap;
        for (size_t f0 = 0; f0 < F; f0 += piece) {
            const size_t cnt = std::min(piece, F - f0), span = (cnt - 1) * hop + e->N;
            sono_batch part{(const float *)e->d_in.p + f0 * hop, 1, span, span};
            rc = sono_stft_mag_f32_dev(e, &part, noverlap, (float *)e->d_mags.p, nullptr);
            if (rc) return rc;
            HIP_TRY(sono::launch_filterbank_u8((const float *)e->d_mags.p, cnt, (int)B, (const unsigned int *)e->d_bank.p,
                                               (const unsigned int *)((char *)e->d_bank.p + sb),
                                               (const float *)((char *)e->d_bank.p + sb + tb), (unsigned int)e->bank_tap_count,
                                               (int)e->num_filters, gain_db, range_db,
                                               (uint8_t *)e->d_out.p + f0 * e->num_filters, (uint16_t *)e->d_peak_bin.p + f0,
                                               (float *)e->d_peak_mag.p + f0, (uint32_t *)e->d_gmax.p, e->stream));
        }
    }
copy_back:
    e->last_peak_bin.resize(F);
    e->last_peak_mag.resize(F);
    uint32_t gbits = 0;
    HIP_TRY(hipMemcpyAsync(out, e->d_out.p, F * out_bins, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->last_peak_bin.data(), e->d_peak_bin.p, F * sizeof(uint16_t),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->last_peak_mag.data(), e->d_peak_mag.p, F * sizeof(float),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(&gbits, e->d_gmax.p, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::memcpy(&e->last_gmax, &gbits, 4);
    if (frames) *frames = F;
    return SONO_OK;
}

int sono_peaks_u16(const sono_engine *e, float ratio, uint16_t *out, size_t cap, size_t *count)
{
    if (count) *count = 0;
    if (!e) return fail(SONO_EINVAL, "sono_peaks_u16: NULL engine");
    const size_t F = e->last_peak_bin.size();
    if (!e->has_last || F == 0 || !(e->last_gmax > 0.0f)) return SONO_OK;   // lib.rs:350-352
    if (!out || cap < F) return fail(SONO_ECAPACITY, "sono_peaks_u16: output buffer too small");
    const float threshold = e->last_gmax * ratio;                            // lib.rs:355
    for (size_t f = 0; f < F; ++f)
        out[f] = e->last_peak_mag[f] >= threshold ? e->last_peak_bin[f] : (uint16_t)0xFFFF;
    if (count) *count = F;
    return SONO_OK;
}

int sono_peak_mags_f32(const sono_engine *e, float ratio, float *out, size_t cap, size_t *count)
{
    if (count) *count = 0;
    if (!e) return fail(SONO_EINVAL, "sono_peak_mags_f32: NULL engine");
    const size_t F = e->last_peak_mag.size();
    if (!e->has_last || F == 0 || !(e->last_gmax > 0.0f)) return SONO_OK;   // lib.rs:399-401
    if (!out || cap < F) return fail(SONO_ECAPACITY, "sono_peak_mags_f32: output buffer too small");
    const float threshold = e->last_gmax * ratio;
    for (size_t f = 0; f < F; ++f)
        out[f] = e->last_peak_mag[f] >= threshold ? e->last_peak_mag[f] : 0.0f;
    if (count) *count = F;
    return SONO_OK;
}

float sono_global_max(const sono_engine *e) { return e ? e->last_gmax : 0.0f; }

int sono_engine_set_color_map(sono_engine *e, const uint8_t *rgba, size_t len)
{
    if (!e) return fail(SONO_EINVAL, "sono_engine_set_color_map: NULL engine");
    if (len != 1024 || !rgba) return SONO_OK;                               // lib.rs:452-454
    e->color_map.resize(256);
    for (size_t i = 0; i < 256; ++i)
        e->color_map[i] = ((uint32_t)rgba[4 * i] << 24) | ((uint32_t)rgba[4 * i + 1] << 16) |
                          ((uint32_t)rgba[4 * i + 2] << 8) | (uint32_t)rgba[4 * i + 3];
    return SONO_OK;
}

int sono_engine_set_spectrum_config(sono_engine *e, const char *scale, float fmin, float fmax)
{
    if (!e) return fail(SONO_EINVAL, "sono_engine_set_spectrum_config: NULL engine");
    e->scale = scale ? scale : "";
    e->freq_min = fmin; e->freq_max = fmax;
    return SONO_OK;
}

int sono_spectrogram_image(sono_engine *e, const float *x, size_t n, uint32_t width, uint32_t height,
                           uint32_t noverlap, float gain_db, float range_db, uint8_t *out, size_t cap)
{
    if (!e || (!x && n)) return fail(SONO_EINVAL, "sono_spectrogram_image: NULL argument");
    const size_t bytes = (size_t)width * height * 4;
    if (bytes && (!out || cap < bytes)) return fail(SONO_ECAPACITY, "sono_spectrogram_image: output buffer too small");
    if (bytes) std::memset(out, 0, bytes);
    if (width == 0 || height == 0 || e->color_map.empty()) return SONO_OK;  // lib.rs:504-506
    if (noverlap >= e->N) return fail(SONO_EINVAL, "sono_spectrogram_image: noverlap must be < fft_size");
    // (n - noverlap) / step of lib.rs:517-521 equals the usual frame count for n >= N
    const size_t F = sono_frames_for(n, e->N, noverlap), B = e->N / 2;
    if (F == 0) return SONO_OK;
    std::vector<float> mags(F * B);
    size_t got = 0;
    int rc = sono_stft_mag_f32(e, x, n, noverlap, mags.data(), mags.size(), &got);
    if (rc) return rc;
    sono::host_render_image(mags.data(), F, B, e->bank.data(), e->bank.size(), e->num_filters, e->use_bank,
                            e->color_map.data(), width, height, gain_db, range_db, out);
    return SONO_OK;
}

int sono_power_spectrum(const float *x, size_t n, uint32_t sample_rate, uint32_t fft_size,
                        const char *window, int has_overlap, float overlap_percent, int device,
                        float *out, size_t cap, size_t *bins)
{
    (void)sample_rate;   // only sets the bin count, which is N/2+1 for every power-of-two N (lib.rs:1044-1046)
    if (bins) *bins = 0;
    if (n == 0) return SONO_OK;                                              // lib.rs:1026-1028
    if (!x) return fail(SONO_EINVAL, "sono_power_spectrum: x is NULL");
    if (!sono::supported_fft_size(fft_size))
        return fail(SONO_EINVAL, "sono_power_spectrum: fft_size must be a power of two in [256, 4096]");
    const size_t nb = fft_size / 2 + 1;
    if (n < fft_size) return SONO_OK;                                        // no frame: empty (lib.rs:1101-1103)
    if (!out || cap < nb) return fail(SONO_ECAPACITY, "sono_power_spectrum: output buffer too small");
    if (sono_device_count() <= 0) return fail(SONO_EDEVICE, "sono_power_spectrum: no HIP device available");
    DeviceGuard guard(device);
    if (!guard.ok()) return fail(SONO_EDEVICE, "sono_power_spectrum: cannot select the device");
    const size_t hop = sono::host_power_hop(fft_size, has_overlap != 0, overlap_percent);
    const size_t F = (n - fft_size) / hop + 1;
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError(), "hipGetDevice");
    std::lock_guard<std::mutex> lock(g_power_mu);
    int rc = SONO_OK;
    PowerCtx *c = power_ctx(dev, fft_size, window, &rc);
    if (!c) return rc;
    Tables &t = c->t;
    DevBuf &d_x = c->d_x, &d_pow = c->d_pow, &d_db = c->d_db;
    // large staging buffers do not outlive the call
    auto cleanup = [&]() { if (d_x.cap > kPowerKeepBytes) d_x.release(); if (d_pow.cap > kPowerKeepBytes) d_pow.release(); };
    const size_t used = (F - 1) * hop + fft_size;
    hipError_t he;
    if ((he = d_x.reserve(used * sizeof(float))) != hipSuccess ||
        (he = d_pow.reserve(F * nb * sizeof(float))) != hipSuccess ||
        (he = d_db.reserve(nb * sizeof(float))) != hipSuccess ||
        (he = hipMemcpy(d_x.p, x, used * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) {
        cleanup();
        return hip_fail(he, "sono_power_spectrum: staging");
    }
    sono::StftArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = (const float *)d_x.p; a.n_streams = 1; a.stream_stride = used; a.frames_per_stream = F;
    a.hop = (unsigned int)hop; a.N = fft_size; a.window = t.d_window;
    a.tw_half = t.d_tw_half; a.tw_full = t.d_tw_full; a.inv_n = 1.0f / (float)fft_size;
    a.out_power = (float *)d_pow.p;
    if ((he = sono::launch_stft_generic(sono::MODE_POWER, a, nullptr)) != hipSuccess ||
        (he = sono::launch_power_reduce((const float *)d_pow.p, F, (int)nb, fft_size, (float *)d_db.p, nullptr)) != hipSuccess ||
        (he = hipMemcpy(out, d_db.p, nb * sizeof(float), hipMemcpyDeviceToHost)) != hipSuccess) {
        cleanup();
        return hip_fail(he, "sono_power_spectrum: compute");
    }
    cleanup();
    if (bins) *bins = nb;
    return SONO_OK;
}

float sono_find_peak_frequency(const float *spectrum, size_t len, uint32_t sample_rate, uint32_t fft_size,
                               float flow_hz, float fhigh_hz)
{
    if (!spectrum || fft_size == 0) return 0.0f;
    return sono::host_find_peak_frequency(spectrum, len, sample_rate, fft_size, flow_hz, fhigh_hz);
}

int sono_wave_peaks(const float *x, size_t n, size_t num_peaks, float *out, size_t cap, size_t *count)
{
    if (count) *count = 0;
    if (num_peaks == 0 || n == 0) return SONO_OK;                           // lib.rs:795-797
    if (!x || !out) return fail(SONO_EINVAL, "sono_wave_peaks: NULL argument");
    if (cap < num_peaks) return fail(SONO_ECAPACITY, "sono_wave_peaks: output buffer too small");
    const size_t k = sono::host_wave_peaks(x, n, num_peaks, out);
    if (count) *count = k;
    return SONO_OK;
}

float sono_find_global_max(const float *x, size_t n) { return x ? sono::host_find_global_max(x, n) : 0.0f; }

int sono_waveform_create(sono_waveform **out)
{
    if (!out) return fail(SONO_EINVAL, "sono_waveform_create: out is NULL");
    *out = new (std::nothrow) sono_waveform();
    return *out ? SONO_OK : fail(SONO_ENOMEM, "sono_waveform_create: out of host memory");
}

void sono_waveform_destroy(sono_waveform *w) { delete w; }

int sono_waveform_resize(sono_waveform *w, size_t num_channels)
{
    if (!w) return fail(SONO_EINVAL, "sono_waveform_resize: NULL handle");
    w->channels.clear();                                                    // lib.rs:866-869
    w->channels.resize(num_channels);
    return SONO_OK;
}

int sono_waveform_load_channel(sono_waveform *w, size_t idx, const float *x, size_t n)
{
    if (!w || (!x && n)) return fail(SONO_EINVAL, "sono_waveform_load_channel: NULL argument");
    if (idx >= w->channels.size()) return SONO_OK;                          // silently ignored, lib.rs:880-882
    w->channels[idx].assign(x, x + n);
    return SONO_OK;
}

size_t sono_waveform_channel_length(const sono_waveform *w, size_t idx)
{
    return (w && idx < w->channels.size()) ? w->channels[idx].size() : 0;
}

size_t sono_waveform_num_channels(const sono_waveform *w) { return w ? w->channels.size() : 0; }

int sono_waveform_clear(sono_waveform *w)
{
    if (!w) return fail(SONO_EINVAL, "sono_waveform_clear: NULL handle");
    w->channels.clear();
    return SONO_OK;
}

int sono_waveform_release_memory(sono_waveform *w) { return sono_waveform_clear(w); }

int sono_waveform_peaks_in_range(const sono_waveform *w, size_t idx, size_t start_sample, size_t end_sample,
                                 size_t target_width, float *out, size_t cap, size_t *count)
{
    if (count) *count = 0;
    if (!w || !out) return fail(SONO_EINVAL, "sono_waveform_peaks_in_range: NULL argument");
    const size_t need = target_width ? target_width : 1;
    if (cap < need) return fail(SONO_ECAPACITY, "sono_waveform_peaks_in_range: output buffer too small");
    if (idx >= w->channels.size() || target_width == 0) {                   // lib.rs:912-914
        for (size_t i = 0; i < need; ++i) out[i] = 0.0f;
        if (count) *count = need;
        return SONO_OK;
    }
    const std::vector<float> &c = w->channels[idx];
    const size_t k = sono::host_peaks_in_range(c.data(), c.size(), start_sample, end_sample, target_width, out);
    if (count) *count = k;
    return SONO_OK;
}

}  // extern "C"
