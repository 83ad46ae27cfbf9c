// sono_engine.h -- internals shared by the translation units that implement the
// C ABI (sono_capi.hip, sono_capi_detect.hip): error reporting, device buffers
// and the engine object behind the opaque sono_engine handle.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/sonostft.h"
#include "sono_core.h"
#include "sono_tile.h"

struct sono_engine;

namespace sono {

// Records the calling thread's error message and returns `code`.
int fail(int code, const std::string &msg);
// Maps a HIP error to a status code, clears HIP's sticky error and records the message.
int hip_fail(hipError_t e, const char *what);

#define HIP_TRY(expr)                                                    \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) return ::sono::hip_fail(_e, #expr);        \
    } while (0)

// A device allocation that only ever grows.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Host staging memory that grows and is kept between calls: fresh pages cost a fault each on first touch, so the
// large staging areas of the whole-file passes are reused instead of allocated per call.  Pinned (hipHostMalloc): a copy between it and the device runs
// at the link's rate, about four times that of pageable memory, and the stages that use it are copy-bound.  Falls back
// to malloc when pinning fails (a host without a GPU never gets here: the callers need one).
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    bool reserve(size_t bytes)
    {
        if (bytes <= cap) return true;
        release();
        bytes += bytes / 8;                      // a little headroom: the next call is rarely exactly as large
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) pinned = true;
        else { (void)hipGetLastError(); p = std::malloc(bytes); pinned = false; }
        cap = p ? bytes : 0;
        return p != nullptr;
    }
    void release()
    {
        if (p) { if (pinned) (void)hipHostFree(p); else std::free(p); }
        p = nullptr; cap = 0; pinned = false;
    }
};

struct Tables {
    unsigned int N = 0;
    float *d_window = nullptr;
    float2 *d_tw_half = nullptr;   // N/2 entries
    float2 *d_tw_full = nullptr;   // N/2 + 1 entries
    TileTables tile;               // inter-pass twiddles of the register-tiled kernel (may be empty), its launch switches
    bool force_generic = false;    // engine option "force_generic": every launch through the generic LDS-staged kernel
    void release()
    {
        if (d_window) (void)hipFree(d_window);
        if (d_tw_half) (void)hipFree(d_tw_half);
        if (d_tw_full) (void)hipFree(d_tw_full);
        tile_tables_release(tile);
        d_window = nullptr; d_tw_half = nullptr; d_tw_full = nullptr;
    }
};

// The chunked host-pointer pipeline of sono_stft_u8 (sono_stream.hip): pinned and device rings, copy threads.
struct StreamPipe;

// What sono_process_full_file keeps of a call's analysis spectrogram when asked to (SONO_KEEP_CALL_SPECTROGRAMS):
// the call's rows of the region's dB matrix (batCallDetector.js:785) and how many centre times go with them (js:786).
struct FullFileSlice {
    std::vector<float> db;
    uint32_t times = 0;
};
struct FullFileAxes {
    uint32_t hop, min_bin, num_bins;
    double sample_rate;
};
void stream_pipe_destroy(StreamPipe *p);
size_t stream_chunk_frames(uint32_t hop);

// Host threads one library call may use for copies, call framing and measurement: sono_set_host_threads' value, else
// (read once per process) SONO_HOST_THREADS, else min(16, cores / LOCAL_WORLD_SIZE) -- the ranks that a launcher
// such as torchrun starts on one host share its cores, and eight ranks that each took sixteen threads would oversubscribe them.
unsigned int host_threads();
// True when SONO_TIMING was set when the process first asked (stage times of the whole-file entry points on stderr).
bool stage_timing();

// Makes `dev` (if >= 0) the current HIP device for the lifetime of the guard.
class DeviceGuard {
public:
    explicit DeviceGuard(int dev) : ok_(false), prev_(-1)
    {
        if (hipGetDevice(&prev_) != hipSuccess) { (void)hipGetLastError(); return; }
        if (dev >= 0 && dev != prev_) {
            if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return; }
            changed_ = true;
        }
        ok_ = true;
    }
    ~DeviceGuard() { if (changed_) (void)hipSetDevice(prev_); }
    bool ok() const { return ok_; }
private:
    bool ok_, changed_ = false;
    int prev_;
};

// hipMemcpyAsync(d_dst, h_src, bytes, host to device, s) for LARGE arrays in pageable host memory: pieces go through
// three pinned buffers the engine keeps, copied in by several host threads while the previous piece is on the bus
// (a plain copy from pageable memory runs at less than half the link's rate).  Returns once the last piece is queued;
// h_src may be changed after the call returns.  Small arrays take the plain copy.
int upload_host_array(sono_engine *e, void *d_dst, const void *h_src, size_t bytes, hipStream_t s, unsigned int threads = 0);

// sono_stft_mag_f32_ragged_dev on the engine's stream, for a caller that reads bins lo..hi of the rows only (sono_capi.hip).
int stft_mag_ragged_band(sono_engine *e, const sono_ragged_batch *in, uint32_t noverlap, uint32_t lo, uint32_t hi, float *d_out);

// fastScanSegmentsWasm (batCallDetector.js:878-949) on a recording that already lies in HBM (sono_capi_detect.hip).
int fast_scan_resident(sono_engine *e, const float *d_x, size_t n, double sample_rate, double flow_khz, double fhigh_khz,
                       double threshold_db, sono_segment *out, size_t cap, size_t *count);

int stream_stft_u8(sono_engine *e, StreamPipe **pipe, const float *x, size_t frames, uint32_t noverlap, float gain_db,
                   float range_db, uint8_t *out, uint16_t *peak_bin, float *peak_mag, uint32_t *gmax_bits);

}  // namespace sono

struct sono_engine {
    unsigned int N = 0;
    int device = 0;
    float alpha = 0.16f;
    std::string window_name;
    std::vector<float> window;
    sono::Tables tables;
    hipStream_t stream = nullptr;
    // device scratch of the host-pointer entry points
    sono::DevBuf d_in, d_out, d_peak_bin, d_peak_mag, d_gmax, d_mags, d_bank;
    // device scratch of the detector front end (sono_capi_detect.hip)
    sono::DevBuf d_db, d_flags, d_edges, d_runs, d_count, d_select_tmp;
    sono::DevBuf d_rec;                     // sono_process_full_file: the whole recording, resident from the fast scan on
    // ... and, per group of regions, the per-frame records of its regions (six 4-byte arrays): four sets: one for the
    // group whose calls are being measured, one for the group being framed, two for the groups being analysed meanwhile
    sono::HostBuf frame_tables[4];          // (pinned: the finished arrays arrive from the GPU in plain copies)
    sono::DevBuf d_frame_table;             // region analysis: the six per-frame arrays of the regions being analysed, filled by the GPU
    std::vector<sono_segment> scan_segments;   // ... and the fast scan's segments (one per half frame at most)
    sono::HostBuf h_in, h_out;              // staging of the whole-file passes (packed samples; records / dB matrices)
    sono::DevBuf d_refine;                  // call framing on a resident recording: the spans whose windows are summed, and the sums
    sono::HostBuf h_refine;                 // ... and their host side (pinned, kept)
    // The measurement stage's two slots: one batch is measured on host threads while the next is on the GPU.
    struct MeasureSlot {
        sono::HostBuf h_in, h_out;          // packed samples of the batch's calls; their dB matrices
        sono::DevBuf d_in, d_db;
        sono::HostBuf h_jobs;               // measurement on the GPU: the batch's calls ...
        sono::DevBuf d_jobs, d_results;     // ... on the device, and what the kernel measured (copied to h_out instead of the dB)
        sono::HostBuf h_spans;              // region analysis: the batch's row spans (sono::PeakSpan) ...
        sono::DevBuf d_spans;               // ... on the device, followed by what each span found (sono::SpanBest)
        hipEvent_t done = nullptr;          // recorded behind the copy of the dB matrices (or of the results) to h_out
        void release()
        {
            h_in.release(); h_out.release(); d_in.release(); d_db.release();
            h_jobs.release(); d_jobs.release(); d_results.release(); h_spans.release(); d_spans.release();
            if (done) { (void)hipEventDestroy(done); done = nullptr; }
        }
    };
    MeasureSlot measure_slot[2];
    sono::DevBuf d_measure_tables, d_measure_scratch;   // measurement on the GPU: frequency axis + zonal floors; the waves' work areas
    bool no_upload_overlap = false;         // engine option "no_upload_overlap": whole-file detection uploads the recording in one piece, then scans and analyses it
    bool small_upload_pieces = false;       // engine option "small_upload_pieces": pieces of 64 fft sizes instead of 2^26 samples (tests: many pieces from a short recording)
    bool host_measure = false;              // engine option "host_measure": the calls' dB matrices go to the host and are measured there
    bool no_measure_overlap = false;        // engine option "no_measure_overlap": whole-file detection frames and measures a group's calls before it analyses the next group
    sono_engine *analysis_engine[2] = {nullptr, nullptr};   // whole-file detection, piece by piece: the companions that analyse the groups of regions in turn while this engine scans the next pieces,
    sono_engine *frame_engine = nullptr;    // ... and the one on whose stream the next group's framing has its windows summed meanwhile
    sono_engine *measure_engine = nullptr;  // whole-file detection: the companion (same fft size and window) on whose stream and buffers a group's calls are measured beside the next group's analysis; created on first use
    // pinned pieces of upload_host_array, each guarded by an event recorded behind its copy to the device
    struct UploadSlot {
        sono::HostBuf buf;
        hipEvent_t done = nullptr;
        bool used = false;
        void release()
        {
            buf.release();
            if (done) { (void)hipEventDestroy(done); done = nullptr; }
            used = false;
        }
    };
    UploadSlot upload_slot[3];
    // Tables of ragged-batch launches.  A launch returns before its kernel has run, so its tables must stay
    // untouched until then: two slots are used in turn, each guarded by an event recorded behind its kernel.
    struct RaggedSlot {
        sono::DevBuf buf;
        hipEvent_t done = nullptr;
        bool used = false;
    };
    RaggedSlot ragged[2];
    unsigned int ragged_next = 0;
    hipStream_t copy_stream = nullptr;      // uploads the tables without waiting for the caller's stream
    sono::StreamPipe *pipe = nullptr;       // rings of the chunked host-pointer path, created on first use
    // calls of the last sono_process_full_file, kept for sono_full_file_results
    std::vector<sono_full_file_call> full_file_calls;
    std::vector<sono::FullFileSlice> full_file_slices;      // parallel to full_file_calls when the run kept them, else empty
    sono::FullFileAxes full_file_axes{};
    // filter bank (lib.rs:20-22)
    std::vector<float> bank;
    unsigned int num_filters = 0;
    bool use_bank = false;
    bool bank_taps_ready = false;       // d_bank holds the current bank's non-zero taps (sono::bank_taps)
    size_t bank_tap_count = 0;
    // state left behind by the last u8 call (lib.rs:24-26)
    bool has_last = false;
    std::vector<uint16_t> last_peak_bin;
    std::vector<float> last_peak_mag;
    float last_gmax = 0.0f;
    bool bank_two_kernels = false;      // engine option "bank_two_kernels": filter banks through the magnitude matrix even where the fused form exists
    bool no_streaming = false;          // engine option "no_streaming": long host-pointer calls as one copy each way
    // colour map / config (lib.rs:28-32)
    std::vector<uint32_t> color_map;
    std::string scale = "linear";
    float freq_min = 0.0f, freq_max = 0.0f;
};
