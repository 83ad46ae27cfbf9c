// sono_tile.hip -- register-tiled fused STFT kernel for fft sizes 256/512/1024.
//
// Persistent waves: a workgroup is four independent waves, each owning a private
// 17 KB LDS image; a wave walks tiles wave_id, wave_id + n_waves, ... so that the
// waves in flight at any moment read one contiguous stretch of samples (75 % of a
// tile's samples are shared with its neighbours and are served by L2).  The only
// data shared inside a workgroup is the read-only W_N^k table.  Loads of the next
// tile are issued right after phase 1 has consumed the current one.
#include "sono_tile.h"

#include "sono_tile_core.h"

namespace sono {

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlocksPerCU = 2;

struct LdsMax64 {
    __device__ void operator()(unsigned long long *p, unsigned long long v) const { atomicMax(p, v); }
};

// Orders this wave's LDS writes before its later LDS reads of other lanes' data.
// LDS operations of one wave execute in issue order, so no instruction is
// needed; this only stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LOG_M1, int MODE>
__global__ __launch_bounds__(64 * kWavesPerBlock, kBlocksPerCU) void stft_tile_kernel(const StftArgs a,
                                                                                     const tile::TileIter it)
{
    using G = tile::Geo<LOG_M1>;
    __shared__ __align__(16) float2 s_img[kWavesPerBlock * tile::kWaveLdsComplex];
    __shared__ __align__(16) float2 s_tw[G::M];
    __shared__ unsigned long long s_peak[kWavesPerBlock * 16];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < G::M; i += blockDim.x) s_tw[i] = a.tw_full[i];
    if (threadIdx.x < kWavesPerBlock * 16) s_peak[threadIdx.x] = peak_key(0.0f, 0u);
    __syncthreads();

    float2 *img = s_img + wave * tile::kWaveLdsComplex;
    unsigned long long *peak_slot = s_peak + wave * 16 + (lane >> LOG_M1);
    const unsigned long long stride = (unsigned long long)gridDim.x * kWavesPerBlock;
    unsigned long long t = (unsigned long long)blockIdx.x * kWavesPerBlock + wave;

    tile::TileLane<LOG_M1, MODE> L(a, lane);
    if (t < it.n_tiles) L.load(t, it);
    for (; t < it.n_tiles; t += stride) {
        L.run_phase1(img);
        if (t + stride < it.n_tiles) L.load(t + stride, it);      // prefetch: in flight during phases 2-3
        wave_lds_fence();
        float re[32], im[32];
        L.run_phase2_load(img, re, im);
        wave_lds_fence();
        L.run_phase2_fft_store(re, im, img);
        wave_lds_fence();
        L.run_phase3(t, it, img, s_tw, peak_slot, LdsMax64());
        if (MODE == MODE_U8) {
            wave_lds_fence();
            L.run_phase4(t, it, peak_slot);
        }
        wave_lds_fence();
    }
    if (MODE == MODE_U8 && a.gmax_bits) {
        // wave-wide maximum of |2X|^2, then one atomic per wave on the magnitude's bit pattern
        float m = L.run_max;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0 && m > 0.0f) atomicMax(a.gmax_bits, __float_as_uint(fast_sqrt(m) * a.inv_n));
    }
}

template <int LOG_M1, int MODE>
hipError_t launch_one(const StftArgs &a, hipStream_t stream)
{
    using G = tile::Geo<LOG_M1>;
    tile::TileIter it;
    const unsigned long long tps = (a.frames_per_stream + G::FPW - 1) / G::FPW;
    it.tiles_per_stream = (unsigned int)tps;
    it.n_tiles = a.n_streams * tps;
    if (it.n_tiles == 0) return hipSuccess;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    unsigned long long blocks = (it.n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned long long resident = (unsigned long long)cus * kBlocksPerCU;
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL((stft_tile_kernel<LOG_M1, MODE>), dim3((unsigned int)blocks), dim3(64 * kWavesPerBlock), 0,
                       stream, a, it);
    return hipGetLastError();
}

template <int MODE>
hipError_t launch_mode(const StftArgs &a, hipStream_t stream)
{
    switch (a.N) {
    case 1024: return launch_one<4, MODE>(a, stream);
    case 512: return launch_one<3, MODE>(a, stream);
    case 256: return launch_one<2, MODE>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t tile_tables_build(unsigned int N, const float *, TileTables &t) { t.N = N; return hipSuccess; }
void tile_tables_release(TileTables &t) { if (t.d_blob) (void)hipFree(t.d_blob); t.d_blob = nullptr; }

bool tile_supports(int mode, const StftArgs &a, const TileTables &)
{
    if (mode != MODE_MAG && mode != MODE_U8) return false;
    if (a.N != 1024 && a.N != 512 && a.N != 256) return false;
    if ((a.hop & 1u) || (reinterpret_cast<uintptr_t>(a.x) & 7u)) return false;
    if (a.n_streams > 1 && (a.stream_stride & 1ull)) return false;
    const unsigned long long tps = (a.frames_per_stream + 3) / 4;
    if (tps >= 0xFFFFFFFFull) return false;
    if (mode == MODE_MAG && (reinterpret_cast<uintptr_t>(a.out_mag) & 15u)) return false;
    if (mode == MODE_U8 && (reinterpret_cast<uintptr_t>(a.out_u8) & 3u)) return false;
    return true;
}

hipError_t launch_stft_tile(int mode, const StftArgs &a, const TileTables &, hipStream_t stream)
{
    return mode == MODE_MAG ? launch_mode<MODE_MAG>(a, stream) : launch_mode<MODE_U8>(a, stream);
}

}  // namespace sono
