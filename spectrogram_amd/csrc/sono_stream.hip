// sono_stream.hip -- the host-pointer form of the fused STFT -> u8 path for long recordings
// (sono_stft_u8 on more than kStreamMinFrames frames): the recording is cut into chunks of
// frames with their fft_size - hop halo, and host copy-in, upload, kernel, download and host
// copy-out of successive chunks overlap:
//
//   caller's samples --(copy threads)--> pinned ring --(DMA, stream `up`)--> HBM ring
//        --(kernel, the engine's stream)--> HBM ring --(DMA, stream `down`)--> pinned ring
//        --(copy threads)--> caller's rows
//
// The wasm module copies its inputs in and its outputs out as well
// (spectrogram_wasm.js:58-63, 313-320); here the copies are what the call costs, so they are
// what is organised.  Chunks start on multiples of 16 frames, so every frame meets the same
// partner in the kernel's frame pairs as in one launch over the whole recording and the rows
// are bit-identical to sono_stft_u8_dev's.  Device memory: three chunks in flight, whatever
// the recording's length.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/sonostft.h"
#include "sono_engine.h"

namespace sono {

namespace {

constexpr int kSlots = 3;
constexpr size_t kChunkInputBytes = 32u << 20;       // samples of one chunk, about
constexpr size_t kFrameAlign = 16;

// A few threads that copy slices of one large block each: one thread moves about 10 GB/s,
// the PCIe link 50 and more.
class CopyPool {
public:
    explicit CopyPool(unsigned int n) : stop_(false), pending_(0)
    {
        for (unsigned int i = 0; i < n; ++i) threads_.emplace_back([this] { run(); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // memcpy(dst, src, bytes), cut into one slice per thread; returns when all of it is done
    void copy(void *dst, const void *src, size_t bytes)
    {
        const size_t parts = std::max<size_t>(1, std::min<size_t>(threads_.size(), bytes >> 20));
        if (parts == 1) { std::memcpy(dst, src, bytes); return; }
        const size_t slice = ((bytes + parts - 1) / parts + 63) & ~(size_t)63;
        {
            std::lock_guard<std::mutex> l(m_);
            for (size_t p = 0; p < parts; ++p) {
                const size_t o = p * slice;
                if (o >= bytes) break;
                jobs_.push_back(Job{(char *)dst + o, (const char *)src + o, std::min(slice, bytes - o)});
                ++pending_;
            }
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }

private:
    struct Job { char *dst; const char *src; size_t bytes; };
    void run()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                j = jobs_.back();
                jobs_.pop_back();
            }
            std::memcpy(j.dst, j.src, j.bytes);
            {
                std::lock_guard<std::mutex> l(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::vector<Job> jobs_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_;
    size_t pending_;
};

}  // namespace

bool no_pinned_upload()
{
    static const bool off = [] { const char *v = std::getenv("SONO_NO_PINNED_UPLOAD"); return v && *v && *v != '0'; }();      // read once per process
    return off;
}

struct StreamPipe {
    struct Slot {
        float *h_in = nullptr, *d_in = nullptr;
        uint8_t *h_out = nullptr, *d_out = nullptr;
        uint16_t *h_pb = nullptr, *d_pb = nullptr;
        float *h_pm = nullptr, *d_pm = nullptr;
        hipEvent_t up_done = nullptr, kernel_done = nullptr, down_done = nullptr;
    };
    Slot slot[kSlots];
    size_t cap_in = 0, cap_frames = 0, cap_bins = 0;      // what the slots hold: samples, frames, bytes per row
    hipStream_t up = nullptr, down = nullptr;
    CopyPool in_pool, out_pool;
    StreamPipe(unsigned int threads) : in_pool(threads), out_pool(threads) {}

    void release_slots()
    {
        for (auto &s : slot) {
            if (s.h_in) (void)hipHostFree(s.h_in);
            if (s.h_out) (void)hipHostFree(s.h_out);
            if (s.h_pb) (void)hipHostFree(s.h_pb);
            if (s.h_pm) (void)hipHostFree(s.h_pm);
            if (s.d_in) (void)hipFree(s.d_in);
            if (s.d_out) (void)hipFree(s.d_out);
            if (s.d_pb) (void)hipFree(s.d_pb);
            if (s.d_pm) (void)hipFree(s.d_pm);
            s.h_in = s.d_in = nullptr; s.h_out = s.d_out = nullptr; s.h_pb = s.d_pb = nullptr; s.h_pm = s.d_pm = nullptr;
        }
        cap_in = cap_frames = cap_bins = 0;
    }
    ~StreamPipe()
    {
        release_slots();
        for (auto &s : slot) {
            if (s.up_done) (void)hipEventDestroy(s.up_done);
            if (s.kernel_done) (void)hipEventDestroy(s.kernel_done);
            if (s.down_done) (void)hipEventDestroy(s.down_done);
        }
        if (up) (void)hipStreamDestroy(up);
        if (down) (void)hipStreamDestroy(down);
    }
    int reserve(size_t samples, size_t frames, size_t bins)
    {
        if (!up) HIP_TRY(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        if (!down) HIP_TRY(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
        for (auto &s : slot) {
            if (!s.up_done) HIP_TRY(hipEventCreateWithFlags(&s.up_done, hipEventDisableTiming));
            if (!s.kernel_done) HIP_TRY(hipEventCreateWithFlags(&s.kernel_done, hipEventDisableTiming));
            if (!s.down_done) HIP_TRY(hipEventCreateWithFlags(&s.down_done, hipEventDisableTiming));
        }
        if (samples <= cap_in && frames <= cap_frames && bins == cap_bins) return SONO_OK;
        release_slots();
        for (auto &s : slot) {
            HIP_TRY(hipHostMalloc((void **)&s.h_in, samples * sizeof(float), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&s.h_out, frames * bins, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&s.h_pb, frames * sizeof(uint16_t), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&s.h_pm, frames * sizeof(float), hipHostMallocDefault));
            HIP_TRY(hipMalloc((void **)&s.d_in, samples * sizeof(float)));
            HIP_TRY(hipMalloc((void **)&s.d_out, frames * bins));
            HIP_TRY(hipMalloc((void **)&s.d_pb, frames * sizeof(uint16_t)));
            HIP_TRY(hipMalloc((void **)&s.d_pm, frames * sizeof(float)));
        }
        cap_in = samples; cap_frames = frames; cap_bins = bins;
        return SONO_OK;
    }
};

void stream_pipe_destroy(StreamPipe *p) { delete p; }

int upload_host_array(sono_engine *e, void *d_dst, const void *h_src, size_t bytes, hipStream_t s, unsigned int threads)
{
    constexpr size_t kPiece = (size_t)64 << 20;
    if (bytes < 2 * kPiece || no_pinned_upload()) {
        HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
        if (bytes >= 2 * kPiece) HIP_TRY(hipStreamSynchronize(s));
        return SONO_OK;
    }
    CopyPool pool(threads ? threads : host_threads());
    size_t piece = 0;
    for (size_t off = 0; off < bytes; off += kPiece, ++piece) {
        sono_engine::UploadSlot &slot = e->upload_slot[piece % 3];
        if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
        if (slot.used) HIP_TRY(hipEventSynchronize(slot.done));          // its last piece has left the buffer
        if (!slot.buf.reserve(kPiece)) return fail(SONO_ENOMEM, "upload_host_array: cannot allocate pinned staging memory");
        const size_t count = std::min(kPiece, bytes - off);
        pool.copy(slot.buf.p, (const char *)h_src + off, count);
        HIP_TRY(hipMemcpyAsync((char *)d_dst + off, slot.buf.p, count, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(slot.done, s));
        slot.used = true;
    }
    return SONO_OK;
}

// Frames per chunk at this hop: a multiple of 16, about kChunkInputBytes of samples.
size_t stream_chunk_frames(uint32_t hop)
{
    size_t c = kChunkInputBytes / (sizeof(float) * (size_t)hop);
    c = std::max<size_t>(c, 4096);
    return c / kFrameAlign * kFrameAlign;
}

// The fused u8 path over host memory, chunk by chunk.  x: the recording; out: F rows of N/2 bytes;
// peak_bin / peak_mag: F entries each (host); *gmax_bits: the f32 bit pattern of the global maximum.
int stream_stft_u8(sono_engine *e, StreamPipe **pipe, const float *x, size_t F, uint32_t noverlap, float gain_db,
                   float range_db, uint8_t *out, uint16_t *peak_bin, float *peak_mag, uint32_t *gmax_bits)
{
    const uint32_t N = e->N, hop = N - noverlap;
    const size_t B = N / 2, C = stream_chunk_frames(hop);
    const size_t chunks = (F + C - 1) / C;
    if (!*pipe) {
        const unsigned int hw = std::max(1u, host_threads() / 2);        // per direction
        *pipe = new (std::nothrow) StreamPipe(hw);
        if (!*pipe) return fail(SONO_ENOMEM, "sono_stft_u8: out of host memory");
    }
    StreamPipe &p = **pipe;
    int rc = p.reserve((C - 1) * (size_t)hop + N, C, B);
    if (rc) return rc;
    HIP_TRY(e->d_gmax.reserve(sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(e->d_gmax.p, 0, sizeof(uint32_t), e->stream));

    // The rows of finished chunks leave the pinned ring on a thread of their own, so that the copy-in of
    // the next chunk (this thread) and the copy-out of an earlier one run side by side.
    std::atomic<size_t> drained{0};
    std::atomic<int> drain_error{0};
    std::mutex m;
    std::condition_variable cv;
    size_t queued = 0;                                   // chunks whose download has been queued
    const int device = e->device;
    std::thread drainer([&] {
        (void)hipSetDevice(device);
        for (size_t j = 0; j < chunks; ++j) {
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return queued > j || drain_error.load() != 0; });
                if (queued <= j) return;
            }
            StreamPipe::Slot &s = p.slot[j % kSlots];
            if (hipEventSynchronize(s.down_done) != hipSuccess) { drain_error = 1; drained = chunks; return; }
            const size_t f0 = j * C, nf = std::min(C, F - f0);
            p.out_pool.copy(out + f0 * B, s.h_out, nf * B);
            std::memcpy(peak_bin + f0, s.h_pb, nf * sizeof(uint16_t));
            std::memcpy(peak_mag + f0, s.h_pm, nf * sizeof(float));
            drained = j + 1;
        }
    });
    auto finish = [&](int code) {
        { std::lock_guard<std::mutex> l(m); if (code) drain_error = 1; }
        cv.notify_all();
        if (drainer.joinable()) drainer.join();
        if (code) {
            // Work of up to three ring slots may still be queued; the next call starts by writing into slot 0's pinned
            // buffer from host threads, so a failed call only returns once the rings are idle.
            (void)hipStreamSynchronize(p.up);
            (void)hipStreamSynchronize(e->stream);
            (void)hipStreamSynchronize(p.down);
            (void)hipGetLastError();
        }
        return code;
    };

    for (size_t i = 0; i < chunks; ++i) {
        StreamPipe::Slot &s = p.slot[i % kSlots];
        // the slot's previous chunk (i - kSlots) must have left the pinned ring
        while (i >= kSlots && drained.load() < i - kSlots + 1) {
            if (drain_error.load()) return finish(fail(SONO_EDEVICE, "sono_stft_u8: a download failed"));
            std::this_thread::yield();
        }
        const size_t f0 = i * C, nf = std::min(C, F - f0), ns = (nf - 1) * (size_t)hop + N;
        p.in_pool.copy(s.h_in, x + f0 * (size_t)hop, ns * sizeof(float));
        hipError_t err = hipMemcpyAsync(s.d_in, s.h_in, ns * sizeof(float), hipMemcpyHostToDevice, p.up);
        if (err == hipSuccess) err = hipEventRecord(s.up_done, p.up);
        if (err == hipSuccess) err = hipStreamWaitEvent(e->stream, s.up_done, 0);
        if (err != hipSuccess) return finish(hip_fail(err, "sono_stft_u8: upload"));
        sono_batch b{s.d_in, 1, ns, ns};
        rc = sono_stft_u8_dev(e, &b, noverlap, gain_db, range_db, s.d_out, s.d_pb, s.d_pm, (uint32_t *)e->d_gmax.p, nullptr);
        if (rc) return finish(rc);
        err = hipEventRecord(s.kernel_done, e->stream);
        if (err == hipSuccess) err = hipStreamWaitEvent(p.down, s.kernel_done, 0);
        if (err == hipSuccess) err = hipMemcpyAsync(s.h_out, s.d_out, nf * B, hipMemcpyDeviceToHost, p.down);
        if (err == hipSuccess) err = hipMemcpyAsync(s.h_pb, s.d_pb, nf * sizeof(uint16_t), hipMemcpyDeviceToHost, p.down);
        if (err == hipSuccess) err = hipMemcpyAsync(s.h_pm, s.d_pm, nf * sizeof(float), hipMemcpyDeviceToHost, p.down);
        if (err == hipSuccess) err = hipEventRecord(s.down_done, p.down);
        if (err != hipSuccess) return finish(hip_fail(err, "sono_stft_u8: download"));
        { std::lock_guard<std::mutex> l(m); queued = i + 1; }
        cv.notify_all();
    }
    drainer.join();
    if (drain_error.load()) return finish(fail(SONO_EDEVICE, "sono_stft_u8: a download failed"));
    HIP_TRY(hipMemcpyAsync(gmax_bits, e->d_gmax.p, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SONO_OK;
}

}  // namespace sono
