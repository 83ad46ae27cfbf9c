/*
 * sonostft.h -- C ABI of libsonostft, the MI355X (gfx950) implementation of the
 * SonoRadar spectrogram hot path.
 *
 * Every entry point is the C-level counterpart of one export of the reference's
 * wasm module, whose raw signatures are declared in
 *   modules/spectrogram_wasm_bg.wasm.d.ts:4-38
 * and whose semantics are in spectrogram-wasm/src/lib.rs.  Where the wasm ABI
 * passes (ptr, len) into wasm linear memory and returns a freshly allocated
 * (ptr, len) pair, this ABI takes (ptr, len) host pointers in and writes into a
 * caller-owned buffer of stated capacity; `sono_frames_for` sizes that buffer.
 *
 * Conventions
 *   - Plain C types only; no exceptions cross the boundary.  Every function that
 *     can fail returns an int status (SONO_OK == 0); `sono_last_error()` gives a
 *     message for the calling thread's most recent failure.
 *   - One engine = one mutable object bound to one GPU and one HIP stream, not
 *     re-entrant (same as one wasm SpectrogramEngine, whose borrow flag throws
 *     on re-entry).  Different engines may be used from different threads.
 *   - Entry points ending in `_dev` take DEVICE pointers (hipMalloc memory on the
 *     engine's GPU), launch asynchronously on the engine's stream and do not
 *     synchronise; the others take host pointers and return when results are in
 *     the caller's buffer.
 *   - The library never computes on the CPU: with no usable gfx950 device every
 *     compute entry point fails with SONO_EDEVICE.
 */
#ifndef SONOSTFT_H
#define SONOSTFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONO_OK        0
#define SONO_EINVAL    1   /* bad argument (null handle, noverlap >= fft_size, unsupported fft_size ...) */
#define SONO_ENOMEM    2   /* host or device allocation failed */
#define SONO_EDEVICE   3   /* no usable GPU / HIP runtime error */
#define SONO_ECAPACITY 4   /* caller's output buffer is too small */
#define SONO_EREFERENCE 5  /* the reference itself raises an exception on these inputs (see sono_process_full_file) */

typedef struct sono_engine sono_engine;       /* SpectrogramEngine, lib.rs:8-35 */
typedef struct sono_waveform sono_waveform;   /* WaveformEngine,    lib.rs:844-849 */

/* ---- library ------------------------------------------------------------ */

/* Message for the calling thread's last failed call ("" if none). */
const char *sono_last_error(void);

/* Number of visible GPUs (0 if none / no HIP runtime). */
int sono_device_count(void);

/* ABI version of this header (bumped on incompatible change). */
uint32_t sono_abi_version(void);

/* Frame count rule of lib.rs:118-123: n >= N ? (n - N) / (N - noverlap) + 1 : 0.
 * Returns 0 when noverlap >= fft_size (the reference panics there). */
size_t sono_frames_for(size_t n_samples, uint32_t fft_size, uint32_t noverlap);

/* ---- SpectrogramEngine -------------------------------------------------- */

/* spectrogramengine_new (wasm.d.ts:22; lib.rs:45-79).  fft_size: power of two in
 * [256, 4096].  window: one of the ten names of lib.rs:698-776; any other string
 * selects Hann, as the reference does.  has_alpha == 0 encodes Option::None
 * (alpha = 0.16).  device < 0 selects the current HIP device. */
int sono_engine_create(uint32_t fft_size, const char *window, int has_alpha, float alpha,
                       int device, sono_engine **out);

/* __wbg_spectrogramengine_free (wasm.d.ts:4).  NULL is ignored. */
void sono_engine_destroy(sono_engine *e);

uint32_t sono_engine_fft_size(const sono_engine *e);      /* spectrogramengine_get_fft_size   :14 */
uint32_t sono_engine_freq_bins(const sono_engine *e);     /* spectrogramengine_get_freq_bins  :15, = N/2 */
uint32_t sono_engine_num_filters(const sono_engine *e);   /* spectrogramengine_get_num_filters:17 */
int sono_engine_device(const sono_engine *e);

/* Host threads ONE library call may use for its copies, call framing and measurement (no counterpart in the
 * reference, which is single-threaded).  Process-wide.  Default, fixed the first time it is needed: the environment's
 * SONO_HOST_THREADS if set, else min(16, cores / LOCAL_WORLD_SIZE) -- LOCAL_WORLD_SIZE is what torchrun exports to
 * the ranks it starts on one host, which share that host's cores.  n = 0 returns to the default. */
int sono_set_host_threads(uint32_t n);
uint32_t sono_host_threads(void);

/* Switches of one engine (no counterpart in the reference: the wasm has one code path).  Every option is 0 or 1
 * and starts at 0 unless the environment variable named beside it was set to something other than "0" when the
 * engine was CREATED -- the environment is never read on a launch.
 *   "force_generic"  (SONO_FORCE_GENERIC)  every transform through the generic LDS-staged kernel
 *   "force_any_hop"  (SONO_FORCE_ANY_HOP)  the tiled kernel loads both frames of a pair whole at every hop
 *   "fixed_tiles"    (SONO_FIXED_TILES)    equal tile ranges per XCD, no adapting to measured speeds
 *   "balance_debug"  (SONO_BALANCE_DEBUG)  print the XCDs' times and the next ranges to stderr
 *   "no_streaming"   (SONO_NO_STREAMING)   long host-pointer calls as one copy each way instead of the chunk pipeline
 *   "bank_two_kernels" (no variable)       filter banks as magnitude matrix + bank kernel even where the fused form exists
 *   "host_measure"   (SONO_HOST_MEASURE)   sono_process_full_file / sono_measure_roi_calls copy the calls' dB matrices to the
 *                                          host and measure them on host threads instead of on the GPU
 *   "no_upload_overlap" (SONO_NO_UPLOAD_OVERLAP) sono_process_full_file uploads the whole recording before it scans and
 *                                          analyses it, instead of working on the pieces that have arrived while the next
 *                                          ones cross the link
 *   "no_measure_overlap" (SONO_NO_MEASURE_OVERLAP) sono_process_full_file frames and measures a group of regions' calls
 *                                          in line, instead of on threads of their own beside the scan of the next
 *                                          pieces (analysis, framing and measurement each on companion engines --
 *                                          streams and buffers of their own, created on first use)
 *   "small_upload_pieces" (no variable)    those pieces are 64 fft sizes long instead of 2^26 samples (tests: many pieces
 *                                          and several groups of regions from a few seconds of audio)
 * None of them changes a result bit (the tests compare the forms).  SONO_EINVAL for an unknown name.
 * Like every call on an engine: not while another thread is inside a call on the same engine. */
int sono_engine_set_option(sono_engine *e, const char *name, int value);
int sono_engine_get_option(const sono_engine *e, const char *name, int *value);

/* spectrogramengine_get_window_values (wasm.d.ts:20; lib.rs:165): copies N floats. */
int sono_engine_window_values(const sono_engine *e, float *out, size_t cap);

/* spectrogramengine_load_filter_bank / _clear_filter_bank (wasm.d.ts:21,10;
 * lib.rs:90-102).  The weights are used with row stride N/2 (lib.rs:317-326). */
int sono_engine_load_filter_bank(sono_engine *e, const float *flat_weights, size_t len,
                                 uint32_t num_filters);
int sono_engine_clear_filter_bank(sono_engine *e);

/* spectrogramengine_release_memory (wasm.d.ts:23; lib.rs:681-693): drops the
 * filter bank, the colour map, the last call's peak state and device scratch. */
int sono_engine_release_memory(sono_engine *e);

/* spectrogramengine_compute_spectrogram (wasm.d.ts:11; lib.rs:113-161).
 * out: frames x N/2 linear magnitudes, frame-major; cap in floats. */
int sono_stft_mag_f32(sono_engine *e, const float *x, size_t n, uint32_t noverlap,
                      float *out, size_t cap, size_t *frames);

/* spectrogramengine_compute_spectrogram_u8 (wasm.d.ts:13; lib.rs:199-304).
 * out: frames x out_bins bytes (out_bins = num_filters if a bank is loaded, else
 * N/2); cap in bytes.  Also records per-frame peak bin / peak magnitude and the
 * global maximum for the three getters below. */
int sono_stft_u8(sono_engine *e, const float *x, size_t n, uint32_t noverlap,
                 float gain_db, float range_db, uint8_t *out, size_t cap, size_t *frames);

/* spectrogramengine_get_peaks (wasm.d.ts:19; lib.rs:349-387).  *count = frames of
 * the last u8 call, or 0 (empty result) before any call / when the global
 * maximum is <= 0.  Bins below threshold are 0xFFFF. */
int sono_peaks_u16(const sono_engine *e, float threshold_ratio, uint16_t *out, size_t cap,
                   size_t *count);

/* spectrogramengine_get_peak_magnitudes (wasm.d.ts:18; lib.rs:398-431). */
int sono_peak_mags_f32(const sono_engine *e, float threshold_ratio, float *out, size_t cap,
                       size_t *count);

/* spectrogramengine_get_global_max (wasm.d.ts:16; lib.rs:441-443). */
float sono_global_max(const sono_engine *e);

/* spectrogramengine_set_color_map (wasm.d.ts:24; lib.rs:450-471): len != 1024 is
 * silently ignored, as in the reference. */
int sono_engine_set_color_map(sono_engine *e, const uint8_t *rgba, size_t len);

/* spectrogramengine_set_spectrum_config (wasm.d.ts:25; lib.rs:475-479): stored only. */
int sono_engine_set_spectrum_config(sono_engine *e, const char *scale, float freq_min,
                                    float freq_max);

/* spectrogramengine_compute_spectrogram_image (wasm.d.ts:12; lib.rs:494-616).
 * out: width*height*4 RGBA bytes; all zero when no colour map is set. */
int sono_spectrogram_image(sono_engine *e, const float *x, size_t n, uint32_t width,
                           uint32_t height, uint32_t noverlap, float gain_db, float range_db,
                           uint8_t *out, size_t cap);

/* ---- device-resident / batched forms (no wasm counterpart: the wasm call takes
 * one channel per call; these run the same computation over data that already
 * lives in HBM, many recordings per launch) ---------------------------------- */

typedef struct sono_batch {
    const float *d_samples;     /* device: stream s starts at d_samples + s*stream_stride */
    size_t n_streams;           /* independent recordings / channels / time chunks */
    size_t stream_stride;       /* samples between stream starts (>= samples_per_stream) */
    size_t samples_per_stream;  /* every stream has this many samples */
} sono_batch;

/* Fused STFT -> dB -> u8 over a batch.  Outputs are device pointers; stream s,
 * frame f lands at row s*frames_per_stream + f.  d_peak_bin / d_peak_mag /
 * d_gmax_bits may be NULL.  d_gmax_bits is ONE uint32 holding the f32 bit pattern
 * of the running maximum; the caller zeroes it (it is only ever raised).
 * hip_stream: a hipStream_t, or NULL for the engine's stream.
 * Alignment (all _dev forms, uniform and ragged): d_out of the u8 forms on 4 bytes, d_out of the
 * magnitude forms on 16 bytes -- the rows are written with dword / 4-dword stores; anything else is
 * SONO_EINVAL.  (hipMalloc returns 256-byte aligned memory; only offset pointers can fail this.)
 * The samples need no alignment beyond that of their type. */
int sono_stft_u8_dev(sono_engine *e, const sono_batch *in, uint32_t noverlap,
                     float gain_db, float range_db,
                     uint8_t *d_out, uint16_t *d_peak_bin, float *d_peak_mag,
                     uint32_t *d_gmax_bits, void *hip_stream);

/* sono_stft_u8_dev on 16-bit PCM as it lies in a WAV data chunk (fileLoader.js:7-63 walks the
 * chunks; the browser's decoder then turns the integers into value / 32768): d_pcm holds interleaved
 * sample frames of `channels` int16 values; stream s starts at sample frame s*stream_stride and has
 * samples_per_stream sample frames; `channel` is transformed.  The conversion happens in the STFT
 * kernel's loads, so no f32 copy of the recording exists (2 B instead of 4 B read per sample, half
 * the HBM footprint); every output equals, bit for bit, sono_pcm_to_f32_dev followed by
 * sono_stft_u8_dev.  d_out may be NULL (features only).  No wasm counterpart. */
int sono_stft_u8_pcm16_dev(sono_engine *e, const int16_t *d_pcm, size_t n_streams, size_t stream_stride,
                           size_t samples_per_stream, uint32_t channels, uint32_t channel, uint32_t noverlap,
                           float gain_db, float range_db, uint8_t *d_out, uint16_t *d_peak_bin,
                           float *d_peak_mag, uint32_t *d_gmax_bits, void *hip_stream);

/* The per-frame features of sono_stft_u8_dev without the u8 rows: peak bin
 * (lib.rs:370-378: first maximum), peak magnitude and the running global maximum
 * (lib.rs:261-266) of every frame of a batch.  For survey-scale passes (a whole
 * night per channel) whose rows nobody looks at: batCallDetector only needs the
 * per-frame peaks to decide where to look closer.  At least one output non-NULL. */
int sono_stft_features_dev(sono_engine *e, const sono_batch *in, uint32_t noverlap,
                           uint16_t *d_peak_bin, float *d_peak_mag, uint32_t *d_gmax_bits,
                           void *hip_stream);

/* Linear-magnitude form of the same (lib.rs:113-161 over a batch). */
int sono_stft_mag_f32_dev(sono_engine *e, const sono_batch *in, uint32_t noverlap,
                          float *d_out, void *hip_stream);

/* Blocks until the engine's stream is idle. */
int sono_engine_synchronize(sono_engine *e);

/* ---- detector front end ----------------------------------------------------
 * The array passes of modules/batCallDetector.js that sit directly on the STFT
 * output.  The wasm module has no export for them (the reference runs them as
 * JavaScript loops over what compute_spectrogram returned); here they run on
 * the GPU so that only segment lists and per-frame features leave HBM.  Numbers
 * that the JavaScript holds as doubles are doubles here. */

typedef struct sono_segment {      /* {start, end} sample range (fastScanSegmentsWasm) or */
    uint64_t start, end;           /* {startFrame, endFrame} inclusive frames (detectCallSegments) */
} sono_segment;

typedef struct sono_frame_features {
    int32_t peak_bin;   /* first bin of the largest dB value in [peak_b0, peak_b1]; -1 if the range is empty */
    float peak_db;      /* that value (-inf if none) */
    int32_t low_bin;    /* lowest bin <= low_max_bin with dB > low_threshold_db; -1 if none */
    int32_t high_bin;   /* highest bin >= high_min_bin with dB > high_threshold_db; -1 if none */
} sono_frame_features;

/* fastScanSegmentsWasm (batCallDetector.js:878-949): STFT with 50 % overlap, a
 * frame is active when the band energy sum of mag^2 over [minBin, maxBin] exceeds
 * 10^(threshold_db/10) * fft_size * 0.25; maximal active runs become sample
 * ranges {f0*hop, f1*hop + fft_size} (f1 = first inactive frame after the run),
 * a run that reaches the last frame ends at n. */
int sono_fast_scan_segments(sono_engine *e, const float *x, size_t n, double sample_rate, double flow_khz,
                            double fhigh_khz, double threshold_db, sono_segment *out, size_t cap, size_t *count);

/* generateSpectrogramWasm (batCallDetector.js:1296-1377): STFT with
 * hop = floor(fft_size * hop_percent / 100), power in dB cropped to the bins
 * [*min_bin, *min_bin + *num_bins).  out: frames x num_bins f32, frame-major.
 * *frames == 0 stands for the reference's `null` (hop < 1, n < fft_size, empty band). */
int sono_generate_spectrogram(sono_engine *e, const float *x, size_t n, double sample_rate, double flow_khz,
                              double fhigh_khz, double hop_percent, float *out, size_t cap, size_t *frames,
                              uint32_t *min_bin, uint32_t *num_bins);

/* detectCallSegments (batCallDetector.js:1540-1598) over a host dB matrix. */
int sono_detect_call_segments(sono_engine *e, const float *power_db, size_t frames, uint32_t num_bins,
                              double call_threshold_db, sono_segment *out, size_t cap, size_t *count);

/* Per-frame peak bin / first bin above / last bin above (batCallDetector.js:3520-3527,
 * 3612-3630, 2509-2522, 2002-2038) over a host dB matrix.  out: frames records. */
int sono_frame_features_db(sono_engine *e, const float *power_db, size_t frames, uint32_t num_bins,
                           uint32_t peak_b0, uint32_t peak_b1, uint32_t low_max_bin, double low_threshold_db,
                           uint32_t high_min_bin, double high_threshold_db, sono_frame_features *out, size_t cap);

/* Device-resident building blocks of the four calls above (device pointers,
 * asynchronous on hip_stream or the engine's stream). */
/* ---- ragged batches: recordings of different lengths in one launch ------------ */

typedef struct sono_ragged_batch {
    const float *d_samples;     /* device: stream s starts at d_samples + offsets[s] */
    size_t n_streams;
    const uint64_t *offsets;    /* host, n_streams entries (samples) */
    const uint64_t *lengths;    /* host, n_streams entries (samples); shorter than fft_size = no frames */
} sono_ragged_batch;

/* Row layout of a ragged batch: row0[s] = first output row of stream s (rows of
 * all streams are contiguous, in stream order), row0[n_streams] = total frames.
 * row0: n_streams + 1 entries.  Host only. */
int sono_ragged_rows(const uint64_t *lengths, size_t n_streams, uint32_t fft_size, uint32_t noverlap, uint64_t *row0);

/* sono_stft_u8_dev / sono_stft_mag_f32_dev over a ragged batch.  Output rows as
 * sono_ragged_rows lays them out; everything else as in the uniform forms. */
int sono_stft_u8_ragged_dev(sono_engine *e, const sono_ragged_batch *in, uint32_t noverlap, float gain_db,
                            float range_db, uint8_t *d_out /* may be NULL: features only */, uint16_t *d_peak_bin,
                            float *d_peak_mag, uint32_t *d_gmax_bits, void *hip_stream);
int sono_stft_mag_f32_ragged_dev(sono_engine *e, const sono_ragged_batch *in, uint32_t noverlap, float *d_out,
                                 void *hip_stream);

/* The fast scan's two passes in one launch (batCallDetector.js:894-927): STFT at
 * `noverlap`, then per frame the sum over bins min_bin..max_bin of the squared
 * f32 magnitude (accumulated in double), compared with target_energy_sum --
 * without ever writing the magnitudes.  d_active: one flag per frame (stream s,
 * frame f at s*frames_per_stream + f); d_band_energy may be NULL.  Requires
 * max_bin < fft_size/2 (a band that reaches past the last bin makes the
 * reference read into the next frame's row: use sono_stft_mag_f32_dev +
 * sono_detect_band_active_dev for that case). */
int sono_stft_band_active_dev(sono_engine *e, const sono_batch *in, uint32_t noverlap, uint32_t min_bin,
                              uint32_t max_bin, double target_energy_sum, uint8_t *d_active,
                              double *d_band_energy /* may be NULL */, void *hip_stream);

/* sono_stft_band_active_dev on 16-bit PCM sample frames in HBM (layout and conversion as in
 * sono_stft_u8_pcm16_dev): the fast scan of batCallDetector.js:878-949 over a WAV data chunk as it
 * lies, flags and sums bit-identical to converting first.  No wasm counterpart. */
int sono_stft_band_active_pcm16_dev(sono_engine *e, const int16_t *d_pcm, size_t n_streams, size_t stream_stride,
                                    size_t samples_per_stream, uint32_t channels, uint32_t channel, uint32_t noverlap,
                                    uint32_t min_bin, uint32_t max_bin, double target_energy_sum, uint8_t *d_active,
                                    double *d_band_energy, void *hip_stream);

/* The survey pass over long channels in ONE read of the samples: the per-frame peak features of
 * sono_stft_features_dev at 75 % overlap (hop N/4: d_peak_bin / d_peak_mag, one per frame, and the global maximum;
 * lib.rs:349-443) AND the fast scan's band test of sono_stft_band_active_dev at 50 % overlap
 * (batCallDetector.js:878-949: d_active / d_band_energy, one per scan frame = ceil(frames / 2) per stream, stream
 * after stream) -- the scan's frames are the even frames of the feature pass.  The peak outputs equal those of the
 * separate calls bit for bit; the band sums add the same terms in another order (flags equal unless a sum lies
 * within rounding of the threshold).  Fused at fft 1024 (the detector's analysis engine, wsManager.js:350-361); other
 * sizes run the two passes one after the other.  The _pcm16 form reads 16-bit sample frames like
 * sono_stft_u8_pcm16_dev. */
int sono_stft_survey_dev(sono_engine *e, const sono_batch *in, uint32_t min_bin, uint32_t max_bin, double target_energy_sum,
                         uint16_t *d_peak_bin, float *d_peak_mag, uint32_t *d_gmax_bits, uint8_t *d_active, double *d_band_energy,
                         void *hip_stream);
int sono_stft_survey_pcm16_dev(sono_engine *e, const int16_t *d_pcm, size_t n_streams, size_t stream_stride, size_t samples_per_stream,
                               uint32_t channels, uint32_t channel, uint32_t min_bin, uint32_t max_bin, double target_energy_sum,
                               uint16_t *d_peak_bin, float *d_peak_mag, uint32_t *d_gmax_bits, uint8_t *d_active, double *d_band_energy,
                               void *hip_stream);
int sono_detect_band_active_dev(sono_engine *e, const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin,
                                uint32_t max_bin, double target_energy_sum, uint8_t *d_active,
                                double *d_band_energy /* may be NULL */, void *hip_stream);
int sono_detect_power_db_dev(sono_engine *e, const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin,
                             uint32_t num_bins, float *d_db, void *hip_stream);
/* d_max: one f32, the largest finite value of d_db (-inf if there is none). */
int sono_detect_db_max_dev(sono_engine *e, const float *d_db, size_t count, float *d_max, void *hip_stream);
int sono_detect_db_active_dev(sono_engine *e, const float *d_db, size_t frames, uint32_t num_bins,
                              double threshold_db, uint8_t *d_active, void *hip_stream);
int sono_detect_frame_features_dev(sono_engine *e, const float *d_db, size_t frames, uint32_t num_bins,
                                   uint32_t peak_b0, uint32_t peak_b1, uint32_t low_max_bin, double low_threshold_db,
                                   uint32_t high_min_bin, double high_threshold_db, sono_frame_features *d_out,
                                   void *hip_stream);
/* Per row of a magnitude matrix (frames x bins, device memory), over bins min_bin .. min_bin + num_bins - 1 converted to
 * dB as sono_detect_power_db_dev does: the first bin (band-relative) of the largest dB value, that value and its two
 * neighbours (NaN at the ends of the band) -- the inputs of the parabolic peak interpolation, batCallDetector.js:3010-3034
 * -- and the lowest and highest bin above (peak - 12 dB), the vertical-click filter's "instantaneous bandwidth"
 * (js:3059-3088).  What the batched region analysis (sono_analyze_rois) keeps of every analysis frame.  form 0: a dB
 * matrix in engine scratch, then a pass over it; 1: one pass, the dB values staying in registers (num_bins <= 512; the
 * form the region analysis uses).  Same records from both. */
typedef struct sono_frame_peak {
    int32_t bin;                    /* -1: the row holds no value above -infinity */
    float db, left_db, right_db;
    int32_t lo12_bin, hi12_bin;     /* -1: none */
} sono_frame_peak;
int sono_detect_frame_peaks_dev(sono_engine *e, const float *d_mags, size_t frames, uint32_t bins, uint32_t min_bin,
                                uint32_t num_bins, sono_frame_peak *d_out, int form, void *hip_stream);
/* Maximal runs of non-zero flags, in increasing order: run i covers frames
 * d_first[i] .. d_last[i] (both arrays need `frames/2 + 1` entries), *d_count runs.
 * Synchronises the stream (it sizes a temporary from the result). */
int sono_detect_runs_dev(sono_engine *e, const uint8_t *d_active, size_t frames, uint64_t *d_first,
                         uint64_t *d_last, uint64_t *d_count, void *hip_stream);

/* sono_detect_runs_dev for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same outputs as
 * sono_detect_runs_dev, bit for bit; calls no hipMalloc / hipFree and no allocating library routine,
 * and does not synchronise (the three kernels are queued on the stream).  Its only device memory
 * beside the caller's buffers is a fixed 128 KiB array that every GPU the library runs on holds until
 * the process ends; that bounds the input: frames <= SONO_RUNS_STATIC_MAX_FRAMES, else SONO_EINVAL.
 * All calls on one GPU share that array: calls that overlap on one GPU (different streams or
 * threads) race; calls queued one after another on one stream are fine. */
#define SONO_RUNS_STATIC_MAX_FRAMES 67108864ull
int sono_detect_runs_static_dev(sono_engine *e, const uint8_t *d_active, size_t frames, uint64_t *d_first,
                                uint64_t *d_last, uint64_t *d_count, void *hip_stream);

/* ---- call framing: what detectCalls (batCallDetector.js:1070-1289) does between
 * detectCallSegments and measureFrequencyParameters -------------------------- */

/* calculateZonalNoiseFloors (batCallDetector.js:1818-1924) over frames
 * start_frame..end_frame (inclusive, clipped to the matrix) of a host dB matrix:
 * floors[z] = noise floor of the zone [10z, 10z+10) kHz; *num_zones zones
 * (floor(freq_bins[num_bins-1] / 10 kHz) + 1).  freq_bins: host, num_bins values.
 * The _dev form takes the matrix in HBM; freq_bins and floors stay on the host. */
int sono_zonal_noise_floors(sono_engine *e, const float *power_db, size_t frames, uint32_t num_bins,
                            const float *freq_bins, size_t start_frame, size_t end_frame, double *floors,
                            size_t cap, size_t *num_zones);
int sono_zonal_noise_floors_dev(sono_engine *e, const float *d_power_db, size_t frames, uint32_t num_bins,
                                const float *freq_bins, size_t start_frame, size_t end_frame, double *floors,
                                size_t cap, size_t *num_zones, void *hip_stream);

/* The order statistic detectCalls takes as noise floor (batCallDetector.js:1223-1233):
 * the value at index floor(count * fraction) of the ascending sort of all `count`
 * dB values (fraction 0.25 in the reference; its -80 dB floor is the caller's). */
int sono_db_percentile(sono_engine *e, const float *power_db, size_t count, double fraction, float *value);
int sono_db_percentile_dev(sono_engine *e, const float *d_power_db, size_t count, double fraction, float *value,
                           void *hip_stream);

/* refineEndUsingOscillogram (batCallDetector.js:1646-1803).  Host only. */
int sono_refine_end_oscillogram(const float *audio, size_t n, double sample_rate, int64_t start_sample,
                                int64_t end_sample, int64_t *refined_end_sample);

/* One framed call: frames start_frame..end_frame of the region's dB matrix
 * (padded by 3 ms, end refined on the oscillogram), its times and duration
 * (batCallDetector.js:1196-1204). */
typedef struct sono_call_frame {
    uint64_t start_frame, end_frame;
    double start_time_s, end_time_s, duration_ms;
} sono_call_frame;

/* The per-segment part of detectCalls (batCallDetector.js:1128-1208): duration
 * filter on the frame count, padding (3 ms there), oscillogram refinement of
 * the end, time stamps, second duration filter (<= 1 ms or < min).  With
 * full_file_rules != 0 the per-segment part of processFullFile
 * (batCallDetector.js:739-810; padding 5 ms there): no filter on the frame
 * count, only calls of <= 1 ms are dropped.  segs: runs from
 * sono_detect_call_segments; time_frames: the `frames` centre times of
 * sono_generate_spectrogram's rule ((f*hop + fft/2) / sample_rate); audio: the
 * region's samples.  Host only. */
int sono_frame_calls(const sono_segment *segs, size_t count, const double *time_frames, size_t frames,
                     const float *audio, size_t n, double sample_rate, uint32_t fft_size, double padding_ms,
                     double min_call_duration_ms, int full_file_rules, sono_call_frame *out, size_t cap,
                     size_t *out_count);

/* _scanSpectrumForFrequency (batCallDetector.js:2936-2987): first bin above the
 * threshold from the top (high_to_low != 0) or from the bottom, with the
 * reference's linear interpolation towards the neighbour below the threshold.
 * *bin = -1 and *freq_hz = -1 if no bin qualifies.  Host only. */
int sono_scan_spectrum_for_frequency(const float *row_db, const float *freq_bins, size_t num_bins,
                                     double threshold_db, int high_to_low, double *freq_hz, int64_t *bin);

/* First step of measureFrequencyParameters (batCallDetector.js:3021-3036): the
 * peak frequency in kHz of the row's bin `bin`, moved by a parabola through the
 * bin and its two neighbours where it has both.  Host only. */
int sono_peak_frequency_khz(const float *row_db, const float *freq_bins, size_t num_bins, int64_t bin, double *khz);

/* The vertical-click filter of measureFrequencyParameters (batCallDetector.js:3059-3088):
 * instantaneous bandwidth in kHz of a dB row = (highest - lowest bin above
 * peak_db - 12) * freq_resolution_hz / 1000; the reference discards a call whose
 * peak frame exceeds 20 kHz.  Host only. */
int sono_instant_bandwidth_khz(const float *row_db, size_t num_bins, double peak_db, double freq_resolution_hz,
                               double *khz);

/* calculateAutoHighpassFilterFreq (batCallDetector.js:4028-4035): corner in kHz. */
double sono_auto_highpass_khz(double peak_freq_khz);

/* applyHighpassFilter (batCallDetector.js:4045-4140): cascaded Butterworth
 * sections, order rounded and clamped to 1..8; filter_hz <= 0 or >= Nyquist
 * copies.  out: n floats (may be `in`).  Host only. */
int sono_highpass_filter(const float *in, size_t n, double filter_hz, double sample_rate, double order, float *out);

/* savitzkyGolay (batCallDetector.js:1610-1640), window 5, order 2.  Host only. */
int sono_savitzky_golay(const double *in, size_t n, double *out);

/* calculateRMSbasedSNR (batCallDetector.js:402-571): signal = the bins of the box
 * frames start_idx..end_idx x centre frequencies in [low_hz, high_hz] that lie
 * more than a quarter of the box's dynamic range above its minimum; noise =
 * everything outside the box, or (noise_db != NULL) the columns of an external
 * noise spectrogram whose frequencies lie in [sel_low_hz, sel_high_hz].  Means
 * are taken on linear power.  status: 0 = snr_db valid (+inf if there is no
 * noise at all), 1 = no bin in the box, 2 = no bin above the threshold (the
 * reference returns null for 1 and 2). */
typedef struct sono_snr_result {
    double snr_db, signal_mean_db, noise_mean_db;
    double threshold_db, signal_max_db, signal_min_db;
    uint64_t signal_count, noise_count;
    int32_t status, pad;
} sono_snr_result;
int sono_rms_snr(sono_engine *e, const float *power_db, size_t frames, uint32_t num_bins, const float *freq_bins,
                 int64_t start_idx, int64_t end_idx, double low_hz, double high_hz, const float *noise_db,
                 size_t noise_frames, uint32_t noise_bins, const float *noise_freq_bins, double sel_low_hz,
                 double sel_high_hz, sono_snr_result *out);

/* ---- measurement stage, threshold searches (host only, serial per call) ------- */

/* The two global switches the measurement stage reads, as bits of every `debug_mode` field and argument below.
 * SONO_DEBUG_MODE: the detector's debugMode (setDebugMode, batCallDetector.js:359-361): the reference keeps a log
 *   table only then and raises a TypeError without it on some inputs.
 * SONO_TIME_EXPANSION: getTimeExpansionMode() (fileState.js:158-166), for recordings played back 10 x slowed down:
 *   a call that reaches the end of measureFrequencyParameters has its frequencies (peak, high, start, end, low,
 *   characteristic, knee, heel, bandwidth) multiplied by 10 and its times and durations (start / end time, duration,
 *   every *_ms field, start / end frequency times) divided by 10 (BatCall.applyTimeExpansion, js:185-278, 3890-3893).
 *   What follows sees the scaled values, as in the reference: the <= 1 ms filter, Flow / Fhigh, the bandwidth classes,
 *   and the region offset, which is added unscaled (js:802-825).
 * SONO_KEEP_CALL_SPECTROGRAMS (sono_full_file_options only): keep every returned call's slice of its region's
 *   analysis spectrogram for sono_full_file_call_spectrogram, as the reference leaves it on the BatCall (js:785-787). */
enum { SONO_DEBUG_MODE = 1, SONO_TIME_EXPANSION = 2, SONO_KEEP_CALL_SPECTROGRAMS = 4 };

/* The call's slice of the analysis spectrogram and what the searches need beside it. */
typedef struct sono_call_matrix {
    const float *power_db;     /* frames x num_bins, row-major */
    size_t frames;
    uint32_t num_bins;
    const float *freq_bins;    /* num_bins centre frequencies, Hz */
    const double *time_frames; /* frames centre times, s (the high-frequency search reads them) */
    const double *zonal;       /* floors of the 10 kHz zones (sono_zonal_noise_floors), or NULL: the searches
                                  compute them from the slice over the frames the reference uses */
    size_t num_zones;
    int32_t debug_mode;        /* SONO_DEBUG_MODE and SONO_TIME_EXPANSION or-ed together (0 and 1 mean what they always did) */
    int32_t pad;
} sono_call_matrix;

/* findOptimalLowFrequencyThreshold (batCallDetector.js:2413-2909).  has_freq = 0
 * where the reference returns null frequencies; discard = the call bottoms out
 * at or below 10 kHz (noise); reference_throws = the reference, outside its debug
 * mode, raises a TypeError on these inputs (js:2815) and the other fields are unset. */
typedef struct sono_low_threshold {
    double threshold_db, low_freq_hz, end_freq_hz;
    int64_t low_freq_frame, low_freq_bin;
    int32_t has_freq, has_frame, warning, discard, reference_throws, pad;
} sono_low_threshold;
int sono_find_optimal_low_frequency_threshold(const sono_call_matrix *m, double call_peak_power_db, int64_t peak_frame,
                                              int64_t limit_frame /* < 0: none */, sono_low_threshold *out);

/* findOptimalHighFrequencyThreshold (batCallDetector.js:1931-2401).  has_limit = 0
 * in the reference's "nothing found" return, which carries neither
 * finalSearchLimitFrame nor isCFStablePattern; reference_throws as above (js:2261). */
typedef struct sono_high_threshold {
    double threshold_db, high_freq_hz, start_freq_hz;
    int64_t high_freq_bin, high_freq_frame, final_search_limit_frame;
    int32_t has_freq, has_start, has_limit, warning, is_cf_stable_pattern, reference_throws;
} sono_high_threshold;
int sono_find_optimal_high_frequency_threshold(const sono_call_matrix *m, double call_peak_power_db, int64_t peak_frame,
                                               sono_high_threshold *out);

/* measureFrequencyParameters (batCallDetector.js:2994-3894), the whole measurement of one call: peak, vertical-click
 * filter, both threshold searches, start / end frequency tracing, characteristic frequency, bandwidth, knee and heel,
 * times relative to the start frequency.  Time expansion (fileState.js) is taken as off.  m->time_frames holds
 * num_times entries: the call's frames plus, unless the call ends the analysis matrix, one more (js:787, 1201).
 * In: start_time_s, end_time_s, duration_ms as the framing stage set them.  Out: every field; doubles are NaN and
 * frame numbers -1 where the reference leaves null.  discarded = call.isDiscarded; reference_throws as for the
 * searches (then, and when discarded, only the fields of the steps already done are set). */
typedef struct sono_call_measurement {
    double start_time_s, end_time_s, duration_ms;
    double peak_freq_khz, peak_freq_ms, peak_power_db;
    double high_freq_khz, high_freq_ms;
    double start_freq_khz, start_freq_ms, start_freq_time_s;
    double end_freq_khz, end_freq_ms, end_freq_time_s;
    double low_freq_khz, low_freq_ms;
    double characteristic_freq_khz, characteristic_freq_ms;
    double knee_freq_khz, knee_freq_ms, heel_freq_khz, heel_freq_ms;
    double bandwidth_khz, high_threshold_db_used, low_threshold_db_used;
    int64_t peak_frame, high_freq_frame, start_freq_frame, start_freq_bin, end_frame_for_low_freq, knee_frame, heel_frame;
    int32_t discarded, reference_throws;
} sono_call_measurement;
int sono_measure_frequency_parameters(const sono_call_matrix *m, size_t num_times, double fhigh_khz,
                                      double freq_resolution_hz, sono_call_measurement *io);

/* ---- batched region analysis: processFullFile's per-region work
 * (batCallDetector.js:609-711) for all regions of a recording at once ---------- */

#define SONO_MAX_ZONES 32

typedef struct sono_roi_table {
    /* arrays the caller provides */
    uint64_t *row0;        /* n_rois + 1: first record of every region, then the total */
    float *peak_db;        /* one per analysis frame: largest dB value of the frame inside the band
                              (after the automatic high-pass where it applies) */
    int32_t *peak_bin;     /* its bin, counted from the band's first bin */
    float *peak_left_db;   /* one per frame: the dB values next to the peak bin (NaN where the peak sits on the */
    float *peak_right_db;  /* band's edge): inputs of the parabolic peak interpolation (js:3021-3034) */
    int32_t *inst_lo_bin;  /* one per frame: lowest / highest bin above (the frame's peak - 12 dB), */
    int32_t *inst_hi_bin;  /* the extent the vertical-click filter measures (js:3059-3088) */
    double *zonal;         /* n_rois * SONO_MAX_ZONES: calculateZonalNoiseFloors of the region's first six frames */
    float *highpass_khz;   /* n_rois: corner of the automatic high-pass, 0 = none */
    /* filled in by sono_analyze_rois */
    uint32_t num_zones, min_bin, num_bins, hop;
} sono_roi_table;

/* Records per region at fft_size and the analysis hop floor(fft_size * hop_percent / 100)
 * (batCallDetector.js:1306): regions are [start, end) sample ranges.  row0: n_rois + 1. */
int sono_roi_rows(const sono_segment *rois, size_t n_rois, uint32_t fft_size, double hop_percent, uint64_t *row0);

/* For every region (sorted, disjoint [start, end) sample ranges of x, as
 * sono_merge_and_pad_segments returns them): the analysis spectrogram
 * (generateSpectrogramWasm, js:1296-1377), the zonal noise floors of its first
 * frames (js:622-627), the region's peak and, if auto_highpass, the automatic
 * Butterworth high-pass and a second spectrogram of the filtered samples
 * (js:631-671) -- thousands of regions per launch, the spectrograms never leave
 * the GPU.  x_filtered (n floats, may be NULL, may be a copy of x): the samples
 * of high-passed regions are overwritten with their filtered values; nothing
 * else is touched. */
int sono_analyze_rois(sono_engine *e, const float *x, size_t n, const sono_segment *rois, size_t n_rois,
                      double sample_rate, double flow_khz, double fhigh_khz, double hop_percent, int auto_highpass,
                      sono_roi_table *out, size_t frames_cap, float *x_filtered);

typedef struct sono_roi_call {
    uint64_t roi, start_frame, end_frame, peak_frame;           /* frames inside the region */
    double start_time_s, end_time_s, duration_ms, peak_time_s;  /* times from the start of the recording */
    double peak_freq_khz;   /* measureFrequencyParameters' first step (js:3009-3036): parabolic peak frequency ... */
    double peak_freq_ms;    /* ... and the peak's time after the call's first frame (js:3039-3045) */
    double inst_bandwidth_khz;  /* of the peak frame (js:3061-3077) */
    float peak_power_db;
    int32_t peak_bin;
    int32_t discarded;      /* 1 = instantaneous bandwidth above 20 kHz: a vertical click, the reference drops it (js:3080-3087, 802) */
    int32_t pad;
} sono_roi_call;

/* The rest of processFullFile's per-region work (js:677-828, without the
 * measurement stage) from sono_analyze_rois's table: call segments, duration
 * filter, strongest-first 30 ms echo filter, 5 ms padding, oscillogram
 * refinement, time stamps.  x: the recording; x_filtered: what sono_analyze_rois
 * wrote (read only inside the regions that were high-passed; NULL = read x
 * everywhere).  Calls come out region by region, strongest first inside a
 * region, like the reference's.  Host only (regions shared out over threads). */
int sono_frame_roi_calls(const float *x, const float *x_filtered, size_t n, const sono_segment *rois, size_t n_rois,
                         const sono_roi_table *table,
                         double sample_rate, uint32_t fft_size, double call_threshold_db, double min_call_duration_ms,
                         sono_roi_call *out, size_t cap, size_t *count);

/* The measurement stage (sono_measure_frequency_parameters) for calls sono_frame_roi_calls returned, all at once:
 * the calls' frames are transformed again from the recording (x, x_filtered as for sono_frame_roi_calls) in one ragged
 * launch -- bit for bit the dB values of the region's own transform -- and the calls are measured where those
 * matrices lie, on the GPU, one wave per call (the same source as sono_measure_frequency_parameters, same numbers);
 * with the engine option "host_measure", or when a call has more than 4096 frames, the matrices are copied to the
 * host and measured on `threads` host threads (0 = the process's budget, sono_host_threads).  out[i] belongs to calls[i]; its times are relative to the call's region, as
 * processFullFile has them at that point (batCallDetector.js:786-797; add rois[roi].start / sample_rate for file
 * time, js:822-825).  Table, band and hop must be those of the sono_analyze_rois call. */
int sono_measure_roi_calls(sono_engine *e, const float *x, const float *x_filtered, size_t n, const sono_segment *rois, size_t n_rois,
                           const sono_roi_table *t, const sono_roi_call *calls, size_t n_calls, double sample_rate,
                           double flow_khz, double fhigh_khz, double hop_percent, int debug_mode, unsigned int threads,
                           sono_call_measurement *out);

/* processFullFile (batCallDetector.js:584-840) in one call: fast scan, merge & pad, batched region analysis, framing,
 * measurement, and the follow-up of js:802-827 (discarded and <= 1 ms calls dropped, classification, placeholder SNR
 * = peak + 80 dB, quality, region time -> file time).  Calls come in the reference's order (per region, strongest
 * first).  SONO_EREFERENCE: SONO_DEBUG_MODE is not set and the reference would raise on one of the calls (its promise rejects).
 * The recording is read from the caller's memory by a second thread for the length of the call (it crosses the link
 * in pieces while the pieces that have arrived are scanned); x must stay unchanged until the call returns.  Further
 * threads, one per group of regions (3 000 at a time), analyse a group, frame its calls and measure them, each stage in
 * the groups' order and beside the other stages of the neighbouring groups (on four companion engines the engine
 * creates on first use and keeps -- two that analyse, one that frames, one that measures: engine option
 * "no_measure_overlap"); all of them have ended when the call returns.  No sample returns
 * to the host: the automatic high-pass filters the resident recording in place, and what the framing stage needs of
 * the samples (the energies of 0.1 ms windows, js:1676-1683) is summed on the GPU. */
typedef struct sono_full_file_options {
    double threshold_db;            /* options.threshold_dB of the fast scan, reference default -60 */
    double padding_ms;              /* options.padding_ms of mergeAndPadSegments, default 5 */
    double hop_percent;             /* config.hopPercent, default 3.125 */
    double call_threshold_db;       /* config.callThreshold_dB, default -24 */
    double min_call_duration_ms;    /* config.minCallDuration_ms, default 1 */
    int32_t debug_mode;             /* SONO_DEBUG_MODE | SONO_TIME_EXPANSION | SONO_KEEP_CALL_SPECTROGRAMS */
    uint32_t threads;               /* host threads of the measurement stage where it runs on the host, 0 = the process's budget */
} sono_full_file_options;
enum { SONO_CALL_CF = 0, SONO_CALL_FM = 1, SONO_CALL_CF_FM = 2 };
typedef struct sono_full_file_call {
    sono_call_measurement m;        /* times in seconds are file time */
    uint64_t roi, start_frame, end_frame;       /* region, and the call's frames within the region's analysis spectrogram */
    double highpass_khz;            /* automatic high-pass applied to the region, 0 = none */
    double flow_hz, fhigh_khz;      /* call.Flow, call.Fhigh (js:813-814) */
    double snr_db;
    int32_t call_type;              /* SONO_CALL_* */
    int32_t quality;                /* 0 'Very Poor', 1 'Poor', 2 'Normal', 3 'Good', 4 'Excellent' */
} sono_full_file_call;
int sono_process_full_file(sono_engine *e, const float *x, size_t n, double sample_rate, double flow_khz, double fhigh_khz,
                           const sono_full_file_options *opt, sono_full_file_call *out, size_t cap, size_t *count);
/* The same on a recording held as a WAV file holds it: `n` interleaved 16-bit sample frames of `channels` channels in
 * host memory, of which `channel` is analysed (fileLoader.js:7-63 reads such files; the browser's decoder turns the
 * integers into value / 32768, which is exact, so the calls equal those of sono_process_full_file on the converted
 * floats value for value).  The recording crosses the link as 16-bit data -- half the bytes -- and no float copy of
 * it exists on the host. */
int sono_process_full_file_pcm16(sono_engine *e, const int16_t *pcm, size_t n, uint32_t channels, uint32_t channel, double sample_rate,
                                 double flow_khz, double fhigh_khz, const sono_full_file_options *opt, sono_full_file_call *out,
                                 size_t cap, size_t *count);
/* The engine keeps the calls of its last sono_process_full_file.  When that returned SONO_ECAPACITY (*count = the
 * number of calls, the first cap of them written), fetch them all here with a larger buffer instead of running the
 * recording again.  cap = 0 only asks for the count. */
int sono_full_file_results(sono_engine *e, sono_full_file_call *out, size_t cap, size_t *count);
/* Call `index` (position in the results) of the engine's last sono_process_full_file run with SONO_KEEP_CALL_SPECTROGRAMS:
 * what processFullFile leaves on the BatCall beside the numbers (batCallDetector.js:785-787): call.spectrogram = the
 * call's rows of its region's dB matrix (frames x num_bins, row-major, the very values the measurement read),
 * call.timeFrames = the rows' centre times in seconds from the region's start and one more (frames + 1 values; frames
 * where the call ends its region's matrix) -- neither shifted to file time nor rescaled by time expansion, as in the
 * reference -- and call.freqBins = the num_bins centre frequencies in Hz.  Any buffer may be NULL; frames, times and
 * num_bins always receive the sizes, so a first call with NULL buffers sizes the second.  SONO_ECAPACITY: a buffer is
 * too small.  Each kept call holds frames x num_bins floats in host memory until the engine's next run or its end. */
int sono_full_file_call_spectrogram(sono_engine *e, size_t index, float *power_db, size_t cap_floats, double *time_frames,
                                    size_t cap_times, float *freq_bins, size_t cap_bins, size_t *frames, size_t *times,
                                    uint32_t *num_bins);

/* mergeAndPadSegments (batCallDetector.js:1028-1056).  Host only; sorts `segs` in place. */
int sono_merge_and_pad_segments(sono_segment *segs, size_t count, uint64_t total_samples, double sample_rate,
                                double padding_ms, sono_segment *out, size_t cap, size_t *out_count);

/* ---- WAV / PCM ingest -------------------------------------------------------
 * The application gets its samples from the browser (WebAudio decodeAudioData,
 * wavesurfer.esm.js:67-70) after sniffing the RIFF header itself
 * (modules/fileLoader.js:7-63).  A batch driver has no browser: these calls walk
 * the chunks the same way and convert interleaved PCM to one f32 channel on the
 * GPU (integer PCM is divided by 2^(bits-1), as WebAudio does). */

typedef struct sono_wav_info {
    uint32_t format_tag;        /* 1 = integer PCM, 3 = IEEE float, 0xFFFE = extensible (sub-format in the first two GUID bytes) */
    uint32_t channels;
    uint32_t sample_rate;       /* fileLoader.js:20 */
    uint32_t bits_per_sample;   /* fileLoader.js:47 */
    uint64_t data_offset;       /* byte offset of the first sample */
    uint64_t data_bytes;        /* size of the data chunk, clipped to the buffer */
    uint64_t frames;            /* samples per channel = data_bytes / (channels * bits/8) (fileLoader.js:55-57) */
    uint64_t guano_offset;      /* 'guan' metadata chunk (modules/guanoReader.js), 0 if absent */
    uint64_t guano_bytes;
} sono_wav_info;

/* Host only.  SONO_EINVAL if the buffer is not RIFF/WAVE or has no fmt/data chunk. */
int sono_wav_parse(const uint8_t *bytes, size_t len, sono_wav_info *info);

/* d_pcm: `frames` interleaved sample frames of `channels` channels in HBM, 8/16/24/32-bit integer
 * (is_float == 0) or 32-bit float (is_float != 0); d_out: `frames` f32 samples of `channel`. */
int sono_pcm_to_f32_dev(sono_engine *e, const void *d_pcm, size_t frames, uint32_t channels, uint32_t channel,
                        uint32_t bits_per_sample, int is_float, float *d_out, void *hip_stream);

/* Host convenience: parse, upload the data chunk, convert one channel, copy it back.
 * *count = frames; cap in floats. */
int sono_wav_decode_channel(sono_engine *e, const uint8_t *bytes, size_t len, uint32_t channel, float *out,
                            size_t cap, size_t *count, sono_wav_info *info /* may be NULL */);

/* ---- free functions ----------------------------------------------------- */

/* compute_power_spectrum (wasm.d.ts:6; lib.rs:1019-1118).  has_overlap == 0
 * encodes Option::None.  out: N/2+1 dB values; *bins = 0 for the reference's
 * empty result (n == 0 or n < fft_size).  device < 0: current device. */
int sono_power_spectrum(const float *x, size_t n, uint32_t sample_rate, uint32_t fft_size,
                        const char *window, int has_overlap, float overlap_percent,
                        int device, float *out, size_t cap, size_t *bins);

/* find_peak_frequency_from_spectrum (wasm.d.ts:9; lib.rs:1132-1179).  Host only. */
float sono_find_peak_frequency(const float *spectrum, size_t len, uint32_t sample_rate,
                               uint32_t fft_size, float flow_hz, float fhigh_hz);

/* compute_wave_peaks (wasm.d.ts:7; lib.rs:794-824) and find_global_max
 * (wasm.d.ts:8; lib.rs:834-840).  Host only (UI helpers, O(n) scans). */
int sono_wave_peaks(const float *x, size_t n, size_t num_peaks, float *out, size_t cap,
                    size_t *count);
float sono_find_global_max(const float *x, size_t n);

/* ---- WaveformEngine (wasm.d.ts:5,26-33; lib.rs:851-1000).  Host only. ------ */
int sono_waveform_create(sono_waveform **out);                                   /* waveformengine_new */
void sono_waveform_destroy(sono_waveform *w);                                    /* __wbg_waveformengine_free */
int sono_waveform_resize(sono_waveform *w, size_t num_channels);                 /* waveformengine_resize */
int sono_waveform_load_channel(sono_waveform *w, size_t idx, const float *x, size_t n); /* _load_channel */
size_t sono_waveform_channel_length(const sono_waveform *w, size_t idx);         /* _get_channel_length */
size_t sono_waveform_num_channels(const sono_waveform *w);                       /* _get_num_channels */
int sono_waveform_clear(sono_waveform *w);                                       /* waveformengine_clear */
int sono_waveform_release_memory(sono_waveform *w);                              /* _release_memory */
/* waveformengine_get_peaks_in_range: writes max(target_width, 1) floats. */
int sono_waveform_peaks_in_range(const sono_waveform *w, size_t idx, size_t start_sample,
                                 size_t end_sample, size_t target_width, float *out,
                                 size_t cap, size_t *count);

#ifdef __cplusplus
}
#endif
#endif /* SONOSTFT_H */
