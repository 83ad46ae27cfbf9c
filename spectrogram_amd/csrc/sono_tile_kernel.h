// sono_tile_kernel.h -- register-tiled fused STFT kernel for fft sizes 256 ... 4096 and its launcher,
// as templates.  Included by one translation unit per mode (sono_tile_u8.hip, sono_tile_mag.hip,
// sono_tile_band.hip: the sixty-odd instantiations compile side by side instead of one after another).
//
// Persistent waves.  A workgroup is a set of independent waves, each owning a
// private LDS region that is first the pass-1 -> pass-2 exchange image and then
// the bin-major power image; the only data shared inside a workgroup are the
// read-only window and inter-pass twiddle tables and the tile counter.  One
// workgroup per CU.  N <= 1024 (32 values per lane and pass): sixteen waves, four
// per SIMD and at most 128 VGPRs per lane, 8.25 KB regions.  N >= 2048 (64 values
// per lane and pass): eight waves, two per SIMD and up to 256 VGPRs, 16 KB regions.
// Enough waves to keep the vector ALU issuing while others sit in LDS or memory waits.
//
// Tile walk: the tiles of a launch are cut into one contiguous range per XCD
// (workgroups are dealt round-robin to the eight XCDs), and inside a range the
// resident waves sweep forward together, so the 75 % of a tile's samples that it
// shares with its neighbours is served by that XCD's L2 instead of being fetched
// from HBM once per XCD.
#pragma once
#include "sono_tile.h"
#include "sono_tile_core.h"

namespace sono {

namespace {

// Overridable so that tools/expbuild.sh can sweep occupancy.
#ifndef SONO_WPB
#define SONO_WPB 16
#endif
#ifndef SONO_WPS
#define SONO_WPS 4
#endif
#ifndef SONO_WIDE_WPB
#define SONO_WIDE_WPB 8
#endif
#ifndef SONO_WIDE_WPS
#define SONO_WIDE_WPS 2
#endif
#ifndef SONO_EPI_PRIO
#define SONO_EPI_PRIO 2
#endif
// The 32-column geometries fit 128 VGPRs (the any-hop forms with a handful of spilled
// dwords) and gain 2-4 % from a fourth wave per SIMD over three.
template <int LOG_R1, int LOG_R2>
struct Occupancy {
    static constexpr int kWavesPerBlock = LOG_R2 == 6 ? SONO_WIDE_WPB : SONO_WPB;
    static constexpr int kWavesPerSimd = LOG_R2 == 6 ? SONO_WIDE_WPS : SONO_WPS;
    static constexpr int kBlocksPerCU = 4 * kWavesPerSimd / kWavesPerBlock;
};


// Diagnostic build (-DSONO_EXP_STAMPS, tools/expbuild.sh): every wave sums the
// shader-clock cycles it spends in each phase; tools/kbench prints the totals.
#ifdef SONO_EXP_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_wg_ends[4 * 2048];     // per workgroup: first start, last end (100 MHz clock), XCC id, tiles done
#define STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc[i] += now_ - last; last = now_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// One step of a reduction over neighbouring lanes in a single instruction: v_max_f32 /
// v_min_u32 with a data-parallel-primitive operand (the lane's partner is named by the
// DPP control).  The compiler writes such a step as a move plus the operation, and a move
// with a DPP operand costs as much as the fused instruction (3.2 cycles of a SIMD).
// The s_nop is the hardware's rule, not padding: a DPP operand must not be read within two
// wait states of the vector instruction that wrote it, and the compiler, which keeps that
// distance for its own DPP instructions, does not look inside an asm statement.
#define SONO_DPP_OP(name, type, op, ctrl)                                                                  \
    __device__ __forceinline__ type name(type v)                                                           \
    {                                                                                                      \
        type r;                                                                                            \
        asm volatile("s_nop 1\n\t" op " %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); \
        return r;                                                                                          \
    }
SONO_DPP_OP(max_quad_1, float, "v_max_f32_dpp", "quad_perm:[1,0,3,2]")
SONO_DPP_OP(max_quad_2, float, "v_max_f32_dpp", "quad_perm:[2,3,0,1]")
SONO_DPP_OP(max_half_mirror, float, "v_max_f32_dpp", "row_half_mirror")
SONO_DPP_OP(max_mirror, float, "v_max_f32_dpp", "row_mirror")
SONO_DPP_OP(max_ror_8, float, "v_max_f32_dpp", "row_ror:8")
SONO_DPP_OP(max_ror_4, float, "v_max_f32_dpp", "row_ror:4")
SONO_DPP_OP(max_ror_2, float, "v_max_f32_dpp", "row_ror:2")
SONO_DPP_OP(max_ror_1, float, "v_max_f32_dpp", "row_ror:1")
SONO_DPP_OP(min_quad_1, unsigned int, "v_min_u32_dpp", "quad_perm:[1,0,3,2]")
SONO_DPP_OP(min_quad_2, unsigned int, "v_min_u32_dpp", "quad_perm:[2,3,0,1]")
SONO_DPP_OP(min_half_mirror, unsigned int, "v_min_u32_dpp", "row_half_mirror")
SONO_DPP_OP(min_mirror, unsigned int, "v_min_u32_dpp", "row_mirror")
#undef SONO_DPP_OP

// Maximum of a non-negative float / minimum of an unsigned integer over each group of
// 2^LOG_LANES consecutive lanes (4 ... 32), in every lane of the group: no LDS round trip.
template <int LOG_LANES>
__device__ __forceinline__ float group_max_f32(float v)
{
    static_assert(LOG_LANES >= 2 && LOG_LANES <= 5, "groups of 4 to 32 lanes");
    v = max_quad_1(v);
    v = max_quad_2(v);
    if constexpr (LOG_LANES >= 3) v = max_half_mirror(v);     // lane i <-> 7 - i inside each half row
    if constexpr (LOG_LANES >= 4) v = max_mirror(v);          // lane i <-> 15 - i
    if constexpr (LOG_LANES == 5) {
        // rows 0/1 and 2/3 trade places
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}
template <int LOG_LANES>
__device__ __forceinline__ unsigned int group_min_u32(unsigned int v)
{
    static_assert(LOG_LANES >= 2 && LOG_LANES <= 5, "groups of 4 to 32 lanes");
    v = min_quad_1(v);
    v = min_quad_2(v);
    if constexpr (LOG_LANES >= 3) v = min_half_mirror(v);
    if constexpr (LOG_LANES >= 4) v = min_mirror(v);
    if constexpr (LOG_LANES == 5) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = r[0] < r[1] ? r[0] : r[1];
    }
    return v;
}

// Orders this wave's LDS writes before its later LDS reads of other lanes' data.
// LDS operations of one wave execute in issue order, so no instruction is
// needed; this only stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);      // and keeps one phase's arithmetic (and registers) out of the next
}

// Maximum over the 32 lanes of this lane's half of the wave (R2 = 32) or over the
// whole wave (R2 = 64), in every lane.
template <int LOG_R2>
__device__ __forceinline__ float column_max(float v)
{
    v = max_ror_8(v);
    v = max_ror_4(v);
    v = max_ror_2(v);
    v = max_ror_1(v);
    // rows 0/1 and 2/3 trade places: every lane then holds both rows' maxima of its half
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    if constexpr (LOG_R2 == 6) {
        // the two halves trade places
        const auto w = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(w[0]), __uint_as_float(w[1]));
    }
    return v;
}

// {lower half's value of lane % 32, upper half's value of lane % 32}, in every lane.
__device__ __forceinline__ float2 both_halves(float v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float2 o;
    o.x = __uint_as_float(r[0]);
    o.y = __uint_as_float(r[1]);
    return o;
}

// RAGGED: streams of different lengths, located through TileIter's tables.
// HOPQ: 1 / 2 = the launch's hop is N/4 / N/2 (frame B reuses frame A's loads), 0 = any hop.
// PCM: 16-bit integer samples (StftArgs::x16) instead of f32.
template <int LOG_R1, int LOG_R2, int MODE, bool RAGGED, int HOPQ, bool PCM = false>
__global__ __launch_bounds__((64 * Occupancy<LOG_R1, LOG_R2>::kWavesPerBlock), (Occupancy<LOG_R1, LOG_R2>::kWavesPerSimd))
void stft_tile_kernel(const StftArgs a, const tile::TileIter it, const float2 *tw_rows)
{
    using G = tile::Geo<LOG_R1, LOG_R2>;
    constexpr int R2 = G::R2;
    constexpr int kWavesPerBlock = Occupancy<LOG_R1, LOG_R2>::kWavesPerBlock;
    __shared__ __align__(16) float s_img[kWavesPerBlock * G::REGION];
    __shared__ __align__(16) float2 s_tw[G::TW_ROWS * R2];
    __shared__ __align__(16) float s_win[G::N];
    __shared__ unsigned int s_next, s_done;
    __shared__ float s_meta[kWavesPerBlock * tile::kMetaFloats];
    // MODE_BANK: the bank's tap list (12 KB; the other modes' workgroups do not have this array)
    constexpr bool kBank = MODE == MODE_BANK;
    __shared__ BankTap s_taps[kBank ? kBankLdsTaps : 1];
    if constexpr (kBank) {
        const unsigned int taps = a.bank_max_taps * a.bank_filters;
        for (unsigned int t = threadIdx.x; t < taps; t += blockDim.x) s_taps[t] = BankTap{a.bank_tap_w[t], a.bank_tap_bin[t]};
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < G::TW_ROWS * R2; i += blockDim.x) s_tw[i] = tw_rows[i];
    for (int i = threadIdx.x; i < G::N; i += blockDim.x) s_win[i] = PCM ? a.window[i] * (1.0f / 32768.0f) : a.window[i];
    if (threadIdx.x == 0) { s_next = 0u; s_done = 0u; }
    const unsigned long long wg_start = __builtin_amdgcn_s_memrealtime();
    __syncthreads();

    float *img = s_img + wave * G::REGION;
    float *meta = s_meta + wave * tile::kMetaFloats;

    // This workgroup's share of the tiles.  The launch is cut into one contiguous
    // range per XCD; inside a range, groups of kWavesPerBlock consecutive tiles are
    // dealt round-robin to the range's workgroups, so the resident workgroups sweep
    // forward together.  Inside a workgroup (= all the waves of one CU) every wave
    // claims the next unclaimed tile from an LDS counter: the SIMD favours its
    // oldest wave, and with a fixed split the youngest waves would still be
    // working long after the oldest had left.
    const unsigned int parts = gridDim.x < tile::kTileParts ? gridDim.x : tile::kTileParts;
    const unsigned int part = blockIdx.x % parts, lb = blockIdx.x / parts;
    const unsigned int blocks_in_part = (gridDim.x - part + parts - 1) / parts;
    // The ranges are equal unless the launcher has cut them by what the parts managed last time (TileIter::part_begin).
    const unsigned long long t_first = it.parts == parts ? (unsigned long long)it.part_begin[part] : it.n_tiles * part / parts;
    const unsigned long long t_count = (it.parts == parts ? (unsigned long long)it.part_begin[part + 1] : it.n_tiles * (part + 1) / parts) - t_first;
    tile::TileLane<LOG_R1, LOG_R2, MODE, HOPQ, PCM> L(a, lane);
    auto claim = [&]() -> unsigned long long {          // -> index of the claimed tile inside the range
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(&s_next, 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        const unsigned long long group = (unsigned long long)(c / kWavesPerBlock) * blocks_in_part + lb;
        return group * kWavesPerBlock + c % kWavesPerBlock;
    };
    unsigned long long cur = claim();

#ifdef SONO_EXP_STAMPS
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime(), rt_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long last = t_begin;
    unsigned int tiles_done = 0;
#endif
    while (cur < t_count) {
        const unsigned long long t = t_first + cur;
        const unsigned long long nxt = claim();           // returns while this tile is being worked on
        const tile::TileRef tr = tile::locate<RAGGED, G::FPW>(a, it, t);
        float re[R2], im[R2];
        {
            float rr[R2], ri[R2];
            L.load(tr);
            STAMP(0);
            if constexpr (G::SPLIT) {
                float y[G::R1], p;
                L.pass1_dft_split(s_win, y, p);
                __builtin_amdgcn_sched_barrier(0);
                // Frame A's lanes are the wave's lower half, frame B's the upper: each value of a column's
                // two lanes that the other one needs comes over with one v_permlane32_swap ([0] = A's, [1] = B's).
                const auto pm = both_halves(column_max<LOG_R2>(p));
                const auto y0 = both_halves(y[0]), yh = both_halves(y[G::HALF]);
                L.pass1_rows_split(s_tw, y, y0.x, y0.y, yh.x, yh.y, pm.x, pm.y, rr, ri, meta);
            } else {
                float ya[R2], yb[R2], pa[G::JPL], pb[G::JPL];
                L.pass1_dft(s_win, ya, yb, pa, pb);
                __builtin_amdgcn_sched_barrier(0);     // keeps the twiddle reads (60 registers) below the transforms
                for (int jj = 0; jj < G::JPL; ++jj) { pa[jj] = column_max<LOG_R2>(pa[jj]); pb[jj] = column_max<LOG_R2>(pb[jj]); }
                L.pass1_rows(s_tw, ya, yb, pa, pb, rr, ri, meta);
            }
            __builtin_amdgcn_sched_barrier(0);
            STAMP(1);
            L.exchange_store(rr, img);
            wave_lds_fence();
            L.exchange_load(img, re);
            wave_lds_fence();
            L.exchange_store(ri, img);
            wave_lds_fence();
            L.exchange_load(img, im);
            wave_lds_fence();
            STAMP(2);
        }
        L.pass2_power(re, im, img);
        wave_lds_fence();
        // The epilogue is a few instructions between LDS reads and stores: with issue
        // priority over the waves that are in their transforms it gets through sooner
        // and is back to arithmetic itself (2 % of the launch).
        __builtin_amdgcn_s_setprio(SONO_EPI_PRIO);
        STAMP(3);
        if constexpr (MODE == MODE_BAND) {
            double e = L.band_sum(img, meta);
            for (int o = G::HALF / 2; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
            L.band_emit(tr, e);
        } else {
            float best_p;
            unsigned int best_k;
            if (MODE == MODE_U8 && a.q_floor > 0.0f) L.template epilogue<true>(tr, img, meta, best_p, best_k);
            else L.template epilogue<false>(tr, img, meta, best_p, best_k);
            if (MODE == MODE_U8 || MODE == MODE_SURVEY || MODE == MODE_BANK) {
                // the frame's largest power, then the lowest bin among the lanes that hold it (lib.rs:370-378)
                const float pmax = group_max_f32<LOG_R1 - 1>(best_p);
                const unsigned int bin = group_min_u32<LOG_R1 - 1>(best_p == pmax ? best_k : 0xFFFFFFFFu);
                L.features(tr, pmax, bin);
            }
            if constexpr (MODE == MODE_BANK) {
                L.bank_magnitudes(img, meta);
                wave_lds_fence();
                L.bank_rows(tr, img, s_taps);
            }
            if constexpr (MODE == MODE_SURVEY) {
                // the fast scan's band test on the tile's even frames, from the power image that is still in place
                double e = L.survey_band_sum(img, meta);
                for (int o = G::HALF; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
                L.survey_band_emit(tr, e);
            }
        }
        wave_lds_fence();
        __builtin_amdgcn_s_setprio(0);
        STAMP(4);
#ifdef SONO_EXP_STAMPS
        ++tiles_done;
#endif
        cur = nxt;
    }
#ifdef SONO_EXP_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&g_stamps[i], acc[i]);
        atomicAdd(&g_stamps[8], __builtin_amdgcn_s_memtime() - t_begin);
        atomicAdd(&g_stamps[9], __builtin_amdgcn_s_memrealtime() - rt_begin);
        atomicAdd(&g_stamps[10], 1ull);
        atomicMax(&g_stamps[11], __builtin_amdgcn_s_memtime() - t_begin);
        atomicMax(&g_stamps[12], __builtin_amdgcn_s_memrealtime());
        atomicMin(&g_stamps[13], rt_begin);
        atomicMax(&g_stamps[14], rt_begin);
        atomicMin(&g_stamps[15], __builtin_amdgcn_s_memrealtime());
        if (blockIdx.x < 2048) {
            unsigned int xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            atomicMin(&g_wg_ends[4 * blockIdx.x], rt_begin);
            atomicMax(&g_wg_ends[4 * blockIdx.x + 1], __builtin_amdgcn_s_memrealtime());
            g_wg_ends[4 * blockIdx.x + 2] = xcc & 15;
            atomicAdd(&g_wg_ends[4 * blockIdx.x + 3], (unsigned long long)tiles_done);
        }
    }
#endif
    if (it.feedback) {
        // The workgroup's last wave reports when the workgroup began and ended (TileIter::feedback).
        unsigned int done = 0;
        if (lane == 0) done = atomicAdd(&s_done, 1u);
        if (lane == 0 && done == (unsigned int)kWavesPerBlock - 1u) {
            unsigned long long *fb = it.feedback + (size_t)blockIdx.x * tile::kFeedbackWords;
            // the old record is marked void before its fields change, so that a reader between the stores cannot take
            // a mix of two launches for a whole record (it reads stamp, fields, stamp)
            __hip_atomic_store(fb + 3, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(fb + 0, wg_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(fb + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(fb + 2, t_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(fb + 3, it.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // last: a record with this launch's stamp is whole
        }
    }
    if ((MODE == MODE_U8 || MODE == MODE_SURVEY || MODE == MODE_BANK) && a.gmax_bits) {
        // wave-wide maximum of |2X|^2, then one atomic per wave on the magnitude's bit pattern
        float m = L.run_max;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0 && m > 0.0f) atomicMax(a.gmax_bits, __float_as_uint(fast_sqrt(m) * a.inv_n));
    }
}

template <int LOG_R1, int LOG_R2, int MODE>
hipError_t launch_one(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    const float2 *tw_rows = (const float2 *)t.d_blob;
    TileBalance *balance = t.balance.feedback ? &t.balance : nullptr;
    using G = tile::Geo<LOG_R1, LOG_R2>;
    constexpr int kWavesPerBlock = Occupancy<LOG_R1, LOG_R2>::kWavesPerBlock;
    constexpr int kBlocksPerCU = Occupancy<LOG_R1, LOG_R2>::kBlocksPerCU;
    tile::TileIter it;
    if (plan) {
        it.tiles_per_stream = 0;
        it.n_tiles = plan->n_tiles;
        it.streams = (const tile::RaggedStream *)plan->d_streams;
        it.tile_stream = plan->d_tile_stream;
    } else {
        const unsigned long long tps = (a.frames_per_stream + G::FPW - 1) / G::FPW;
        it.tiles_per_stream = (unsigned int)tps;
        it.n_tiles = a.n_streams * tps;
        it.streams = nullptr;
        it.tile_stream = nullptr;
    }
    if (it.n_tiles == 0) return hipSuccess;
    const int cus = t.cus;
    if (cus <= 0) return hipErrorInvalidDevice;
    unsigned long long blocks = (it.n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned long long resident = (unsigned long long)cus * kBlocksPerCU;
    if (blocks > resident) blocks = resident;
    const dim3 grid((unsigned int)blocks), block(64 * kWavesPerBlock);
    tile_balance_plan(balance, blocks, resident, kWavesPerBlock, it);
    // The hops that let frame B reuse frame A's loads: a quarter frame (75 % overlap) for
    // the row and magnitude modes, half a frame (the fast scan's 50 %) for the band mode.
    // The engine option "force_any_hop" takes the any-hop form regardless: the tests compare the two.
    constexpr int kHopQ = MODE == MODE_BAND ? 2 : 1;
    // (a geometry that keeps one frame per lane in pass 1 loads every frame whole: it has the any-hop form only)
    const bool q = !G::SPLIT && a.hop * (kHopQ == 2 ? 2u : 4u) == (unsigned int)G::N && !t.any_hop;
    constexpr int kQ = G::SPLIT ? 0 : kHopQ;
    if (a.x16 && (plan || MODE == MODE_MAG)) return hipErrorInvalidValue;
    if constexpr (MODE == MODE_BANK) {          // f32 samples, uniform batches, any hop
        if (a.x16 || plan) return hipErrorInvalidValue;
        if (q) hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, false, kQ, false>), grid, block, 0, stream, a, it, tw_rows);
        else hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, false, 0, false>), grid, block, 0, stream, a, it, tw_rows);
        return hipGetLastError();
    } else if constexpr (MODE == MODE_SURVEY) {        // hop N/4, uniform batches: the two forms this mode is built in
        if (!q || plan) return hipErrorInvalidValue;
        if (a.x16) hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, false, kQ, true>), grid, block, 0, stream, a, it, tw_rows);
        else hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, false, kQ, false>), grid, block, 0, stream, a, it, tw_rows);
        return hipGetLastError();
    } else {
    if constexpr (MODE != MODE_MAG) {
        if (a.x16) {        // 16-bit PCM input: uniform batches of the row and band modes only (tile_supports)
            if (q) hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, false, kQ, true>), grid, block, 0, stream, a, it, tw_rows);
            else hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, false, 0, true>), grid, block, 0, stream, a, it, tw_rows);
            return hipGetLastError();
        }
    }
    if (plan) {
        if (q) hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, true, kQ>), grid, block, 0, stream, a, it, tw_rows);
        else hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, true, 0>), grid, block, 0, stream, a, it, tw_rows);
    } else {
        if (q) hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, false, kQ>), grid, block, 0, stream, a, it, tw_rows);
        else hipLaunchKernelGGL((stft_tile_kernel<LOG_R1, LOG_R2, MODE, false, 0>), grid, block, 0, stream, a, it, tw_rows);
    }
    return hipGetLastError();
    }
}

template <int MODE>
hipError_t launch_mode(const StftArgs &a, const TileTables &t, const RaggedPlan *plan, hipStream_t stream)
{
    if constexpr (MODE == MODE_SURVEY) {
        // one geometry only: the detector's analysis engine is fft 1024 (wsManager.js:350-361); other sizes run the
        // two passes one after the other (sono_capi.hip)
        if (a.N != 1024 || plan || a.hop * 4u != a.N) return hipErrorInvalidValue;
        return launch_one<10 - tile_log_r2(1024), tile_log_r2(1024), MODE>(a, t, plan, stream);
    } else if constexpr (MODE == MODE_BANK) {
        // fft 1024 (the application's default display size): the one geometry whose workgroup has LDS left for the
        // tap list; other sizes and longer lists go through the magnitude matrix (sono_capi.hip)
        if (a.N != 1024 || plan) return hipErrorInvalidValue;
        return launch_one<10 - tile_log_r2(1024), tile_log_r2(1024), MODE>(a, t, plan, stream);
    } else {
    switch (a.N) {
#ifndef SONO_EXP_ONLY_1024      // tools/expbuild.sh: quicker experimental builds
    case 4096: return launch_one<6, 6, MODE>(a, t, plan, stream);
    case 2048: return launch_one<11 - tile_log_r2(2048), tile_log_r2(2048), MODE>(a, t, plan, stream);
    case 512: return launch_one<9 - tile_log_r2(512), tile_log_r2(512), MODE>(a, t, plan, stream);
    case 256: return launch_one<3, 5, MODE>(a, t, plan, stream);
#endif
    case 1024: return launch_one<10 - tile_log_r2(1024), tile_log_r2(1024), MODE>(a, t, plan, stream);
    default: return hipErrorInvalidValue;
    }
    }
}

}  // namespace

}  // namespace sono
