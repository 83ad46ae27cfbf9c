// sono_tile.h -- the register-tiled STFT kernel (sono_tile.hip): interface.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "sono_core.h"

namespace sono {

// Values per lane and pass (log2) of the geometry that serves fft size N: 32 below SONO_WIDE_FROM (four waves per
// SIMD), 64 from there on (two).  fft 2048 is 32 x 64 by default.  -DSONO_WIDE_FROM=4096 builds it as 64 x 32 (one
// frame per lane in pass 1, see "split" in sono_tile_core.h): measured 2.5 % faster at hops that are not a quarter
// frame and 5 % slower at 75 % overlap, where the 32 x 64 form shares loads between the frames of a pair and the
// 64 x 32 form cannot (tools/kbench: 3.35 against 3.18 ms per hour of audio; 2.52 against 2.59 ms at 95 %).
#ifndef SONO_WIDE_FROM
#define SONO_WIDE_FROM 2048
#endif
constexpr int tile_log_r2(unsigned int N) { return N >= SONO_WIDE_FROM ? 6 : 5; }

// Device table private to the tiled kernel: the inter-pass twiddle rows of
// make_tile_twiddles(N) (empty for fft sizes the kernel does not cover).
// What an engine knows about how fast the chip's XCDs get through their ranges of a launch.  The tiled kernel cuts
// a launch into one contiguous range of tiles per XCD; the XCDs of one chip do not run at one speed (with equal ranges
// they end up to 9 % apart, the same ones ahead launch after launch), and a launch is only over when the slowest is.
// Every workgroup of a large launch therefore reports when it began and ended, into host memory that the device
// writes directly, and the next large launch's ranges are cut in proportion to what each XCD managed.  Results do
// not depend on the cut: a tile's output is the same whoever transforms it.
struct TileBalance {
    static constexpr unsigned int kMaxGroups = 1024;      // workgroups of a launch (one or two per CU)
    unsigned long long *feedback = nullptr;               // kMaxGroups x tile::kFeedbackWords, hipHostMalloc
    double share[8] = {1, 1, 1, 1, 1, 1, 1, 1};            // relative size of each part's range (mean 1)
    unsigned long long stamp = 0;                         // stamps handed to launches so far
    unsigned long long seen = 0;                          // stamp of the reports taken up last
    bool fixed = false;                                   // keep the ranges equal (engine option "fixed_tiles")
    bool debug = false;                                   // print times and shares to stderr (engine option "balance_debug")
    std::mutex mu;                                        // launches of one engine from several threads plan in turn
};

// What the launcher needs beside the kernel arguments.  Filled once per engine (tile_tables_build); the switches
// are the engine's options (sono_engine_set_option), so that no launch reads the environment or asks the runtime
// about the device.
struct TileTables {
    unsigned int N = 0;
    void *d_blob = nullptr;
    int cus = 0;                    // compute units of the engine's device
    bool any_hop = false;           // take the any-hop form even where frame B could reuse frame A's loads (option "force_any_hop")
    mutable TileBalance balance;
};

namespace tile { struct TileIter; }
// Fills it.parts / part_begin / feedback / stamp for a launch of it.n_tiles tiles by `blocks` workgroups of `waves_per_block`
// waves, `resident` of which the chip holds at once; takes up the reports of the last large launch first.  Small
// launches, launches on a device that is not one whole chip of eight XCDs (blocks must be 256 or 512: a partition or
// another part deals its workgroups differently), and all launches of an engine with b->fixed set get equal ranges
// and report nothing.
void tile_balance_plan(TileBalance *b, unsigned long long blocks, unsigned long long resident, unsigned int waves_per_block, tile::TileIter &it);

hipError_t tile_tables_build(unsigned int N, TileTables &t);
void tile_tables_release(TileTables &t);

// True when the tiled kernel can run this launch: fft size 256 ... 4096, mode MAG, U8
// or BAND, output rows aligned for its 16-byte (MAG) / 4-byte (U8) stores.
// Samples are read one float per lane, so any hop and sample alignment works.
bool tile_supports(int mode, const StftArgs &a, const TileTables &t);

hipError_t launch_stft_tile(int mode, const StftArgs &a, const TileTables &t, hipStream_t stream);

// ---- ragged batches: streams of different lengths in one launch -----------------
// Frames one wave of the tiled kernel handles at fft size N (0 = size not covered): the unit in
// which a stream's tiles are counted (tiles never straddle streams).
unsigned int tile_frames_per_wave(unsigned int N);

// One entry per stream, in device memory (same layout as the kernel's own view).
struct RaggedStreamEntry {
    unsigned long long sample_offset;  // first sample of the stream, relative to StftArgs::x
    unsigned long long frames;         // frames of the stream
    unsigned long long row0;           // output row of its frame 0
    unsigned long long tile0;          // index of its first tile = sum of ceil(frames / frames_per_wave) of the streams before
};
struct RaggedPlan {
    unsigned long long n_tiles;
    const RaggedStreamEntry *d_streams;     // device, one per stream
    const unsigned int *d_tile_stream;      // device, n_tiles entries: the stream of every tile
};
// StftArgs::n_streams / stream_stride / frames_per_stream are not used; the alignment rules of
// tile_supports apply to the output base pointers.
hipError_t launch_stft_tile_ragged(int mode, const StftArgs &a, const TileTables &t, const RaggedPlan &plan,
                                   hipStream_t stream);

}  // namespace sono
