// sono_capi_noalloc.hip -- sono_detect_runs_static_dev: the run extraction of
// sono_detect_runs_dev for callers that cannot allocate (HIP graph capture, tight
// serving loops).  It calls no hipMalloc / hipFree and no library routine that
// might: its only device memory beside the caller's buffers is `scratch`, a
// file-scope device array of fixed size, and its three kernels are this file's own.
//
//   pass 1  runs_count_kernel   per block of kBlockFrames flags: how many runs start and how
//                               many end in the block                    -> scratch
//   pass 2  runs_scan_kernel    one workgroup: exclusive prefix sums of both count rows in
//                               place, the total                         -> *d_count
//   pass 3  runs_emit_kernel    per block: the flags again, a block-wide prefix sum, and every
//                               start / end frame written at its rank    -> d_first, d_last
//
// Every GPU holds its own copy of `scratch` at its own address; the address is
// looked up once per device (noalloc_init_device) and handed to the kernels as an
// argument.  The name `scratch` itself is never used as a pointer in host code.
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/sonostft.h"
#include "sono_engine.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPerThread = 16;
constexpr unsigned int kBlockFrames = kThreads * kPerThread;      // 4096 flags per block
constexpr unsigned int kMaxBlocks = SONO_RUNS_STATIC_MAX_FRAMES / kBlockFrames;
static_assert((unsigned long long)kMaxBlocks * kBlockFrames == SONO_RUNS_STATIC_MAX_FRAMES, "whole blocks");
constexpr int kMaxDevices = 64;

// [0, kMaxBlocks): runs starting in each block, then their exclusive prefix sums;
// [kMaxBlocks, 2 kMaxBlocks): the same for runs ending in each block.  128 KiB per GPU.
static __device__ __align__(256) unsigned int scratch[2 * kMaxBlocks];

unsigned int *g_scratch_ptr[kMaxDevices];      // device address of `scratch` on every GPU seen so far
std::mutex g_scratch_mutex;

// is_start / is_end of frame f (f < frames): a run starts where an active frame follows an
// inactive one (or the beginning), and ends where an inactive one (or the end) follows.
__device__ __forceinline__ void edge_flags(const uint8_t *active, unsigned long long frames, unsigned long long f0,
                                           bool (&st)[kPerThread], bool (&en)[kPerThread])
{
    bool prev = f0 > 0 && f0 - 1 < frames && active[f0 - 1] != 0;
    bool cur = f0 < frames && active[f0] != 0;
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
        const unsigned long long f = f0 + i;
        const bool next = f + 1 < frames && active[f + 1] != 0;
        const bool in = f < frames;
        st[i] = in && cur && !prev;
        en[i] = in && cur && !next;
        prev = cur;
        cur = next;
    }
}

// Sum over the workgroup in every thread (kThreads = 4 waves).
__device__ __forceinline__ unsigned int block_sum(unsigned int v, unsigned int *s_part)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    const unsigned int t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(kThreads) void runs_count_kernel(const uint8_t *active, unsigned long long frames,
                                                              unsigned int *counts)
{
    __shared__ unsigned int s_part[kThreads / 64];
    bool st[kPerThread], en[kPerThread];
    edge_flags(active, frames, (unsigned long long)blockIdx.x * kBlockFrames + threadIdx.x * kPerThread, st, en);
    unsigned int ns = 0, ne = 0;
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) { ns += st[i]; ne += en[i]; }
    ns = block_sum(ns, s_part);
    ne = block_sum(ne, s_part);
    if (threadIdx.x == 0) { counts[blockIdx.x] = ns; counts[kMaxBlocks + blockIdx.x] = ne; }
}

// One workgroup of 1024 threads; thread t owns the 16 consecutive entries 16 t .. 16 t + 15 of each row.
__global__ __launch_bounds__(1024) void runs_scan_kernel(unsigned int *counts, unsigned int blocks,
                                                         unsigned long long *d_count)
{
    static_assert(kMaxBlocks <= 1024 * 16, "one workgroup scans a whole row");
    __shared__ unsigned int s_wave[16];
    for (int row = 0; row < 2; ++row) {
        unsigned int *c = counts + row * kMaxBlocks;
        unsigned int v[16], sum = 0;
        for (int i = 0; i < 16; ++i) {
            const unsigned int j = threadIdx.x * 16 + i;
            v[i] = j < blocks ? c[j] : 0u;
            sum += v[i];
        }
        // inclusive scan of `sum` over the wave, then over the 16 waves
        unsigned int inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int up = __shfl_up(inc, o, 64);
            if ((int)(threadIdx.x & 63) >= o) inc += up;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = inc;
        __syncthreads();
        unsigned int base = 0, total = 0;
        for (unsigned int w = 0; w < 16; ++w) {
            if (w < (threadIdx.x >> 6)) base += s_wave[w];
            total += s_wave[w];
        }
        unsigned int run = base + inc - sum;
        for (int i = 0; i < 16; ++i) {
            const unsigned int j = threadIdx.x * 16 + i;
            if (j < blocks) c[j] = run;
            run += v[i];
        }
        if (row == 0 && threadIdx.x == 0) *d_count = total;      // starts and ends are equally many
        __syncthreads();
    }
}

__global__ __launch_bounds__(kThreads) void runs_emit_kernel(const uint8_t *active, unsigned long long frames,
                                                             const unsigned int *offsets, unsigned long long *d_first,
                                                             unsigned long long *d_last)
{
    __shared__ unsigned int s_wave[2][kThreads / 64];
    bool st[kPerThread], en[kPerThread];
    const unsigned long long f0 = (unsigned long long)blockIdx.x * kBlockFrames + threadIdx.x * kPerThread;
    edge_flags(active, frames, f0, st, en);
    unsigned int ns = 0, ne = 0;
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) { ns += st[i]; ne += en[i]; }
    unsigned int is = ns, ie = ne;                           // inclusive scans over the wave
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int us = __shfl_up(is, o, 64), ue = __shfl_up(ie, o, 64);
        if ((int)(threadIdx.x & 63) >= o) { is += us; ie += ue; }
    }
    if ((threadIdx.x & 63) == 63) { s_wave[0][threadIdx.x >> 6] = is; s_wave[1][threadIdx.x >> 6] = ie; }
    __syncthreads();
    unsigned int rs = offsets[blockIdx.x] + is - ns, re = offsets[kMaxBlocks + blockIdx.x] + ie - ne;
    for (unsigned int w = 0; w < (threadIdx.x >> 6); ++w) { rs += s_wave[0][w]; re += s_wave[1][w]; }
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
        if (st[i]) d_first[rs++] = f0 + i;
        if (en[i]) d_last[re++] = f0 + i;
    }
}

}  // namespace

namespace sono {

// Looks up this GPU's copy of `scratch`.  Called with `device` current, from
// sono_engine_create (so that the entry point below finds the address ready) and,
// for engines of a library state that skipped it, from the entry point itself.
int noalloc_init_device(int device)
{
    if (device < 0 || device >= kMaxDevices) return fail(SONO_EINVAL, "noalloc_init_device: device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return SONO_OK;
    unsigned int *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(SONO_EDEVICE, "noalloc_init_device: hipGetSymbolAddress returned NULL");
    g_scratch_ptr[device] = ptr;
    return SONO_OK;
}

}  // namespace sono

extern "C" int sono_detect_runs_static_dev(sono_engine *e, const uint8_t *d_active, size_t frames, uint64_t *d_first,
                                           uint64_t *d_last, uint64_t *d_count, void *hip_stream)
{
    using sono::fail;
    if (!e || !d_count || (frames && (!d_active || !d_first || !d_last)))
        return fail(SONO_EINVAL, "sono_detect_runs_static_dev: NULL argument");
    if (frames > SONO_RUNS_STATIC_MAX_FRAMES)
        return fail(SONO_EINVAL, "sono_detect_runs_static_dev: more than SONO_RUNS_STATIC_MAX_FRAMES (67108864) frames; "
                                 "use sono_detect_runs_dev");
    sono::DeviceGuard guard(e->device);
    if (!guard.ok()) return fail(SONO_EDEVICE, "sono_detect_runs_static_dev: cannot select the engine's device");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    if (frames == 0) {
        HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint64_t), s));
        return SONO_OK;
    }
    const int rc = sono::noalloc_init_device(e->device);
    if (rc != SONO_OK) return rc;
    unsigned int *ptr = g_scratch_ptr[e->device];
    const unsigned int blocks = (unsigned int)((frames + kBlockFrames - 1) / kBlockFrames);
    // Nothing in the scratch has to start at zero: pass 1 overwrites the entries of all `blocks`
    // blocks of both rows, and passes 2 and 3 read no others.
    hipLaunchKernelGGL(runs_count_kernel, dim3(blocks), dim3(kThreads), 0, s, d_active, (unsigned long long)frames, ptr);
    hipLaunchKernelGGL(runs_scan_kernel, dim3(1), dim3(1024), 0, s, ptr, blocks, (unsigned long long *)d_count);
    hipLaunchKernelGGL(runs_emit_kernel, dim3(blocks), dim3(kThreads), 0, s, d_active, (unsigned long long)frames,
                       (const unsigned int *)ptr, (unsigned long long *)d_first, (unsigned long long *)d_last);
    HIP_TRY(hipGetLastError());
    return SONO_OK;
}
