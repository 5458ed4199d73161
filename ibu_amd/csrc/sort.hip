// sort.hip — device sort of 24-byte records by (barcode, umi, index): the order `derive(Ord)` gives
// Record (src/constructs/record.rs:58-66) and the header's sorted flag promises (header.rs:111-113).
//
// Stable LSD radix sort, 8-bit digits, over the 192-bit key — but only over the digits that actually
// vary: a census kernel ORs and ANDs each field over all records, and a digit whose bits are equal in
// OR and AND is constant, so its pass would be the identity and is skipped.  16-base barcodes, 12-base
// UMIs and indices below 2^32 need 4 + 3 + 4 passes instead of 24 — and 4 + 3 when the input already runs in index
// order (the sort is stable and the index is the least significant field), which the same census detects.
//
// Structure: census (OR / AND per field, "already sorted", "already in index order": one streaming read) -> per
// varying digit one pass of count / scan / scatter, where only the FIRST count reads the records: later passes count from a
// 1-byte-per-record digit side stream the previous scatter left behind (details above the pass kernels).
//
// Since round 3 large inputs do not run a pass per varying digit any more.  PREFIX + FINISH: passes over the most significant
// P varying bytes only (P = 4 at 1e9 records), after which everything left to decide lies inside runs of equal prefix — a
// handful of records each when the keys are well spread — and ONE finishing kernel ranks every record inside its run in LDS
// and writes the final records (ibu_k_sort_finish / ibu_k_sort_finish_elems).  P comes from a pair count over sample ranges
// (ibu_k_sort_sample_pairs*), long runs that are already in order pass through, other long runs escalate (one retry with a
// longer prefix, then all passes).  Map of this file: census | 24-byte passes | compact keys (compress, element passes,
// expand) | finishing kernels + the sample estimate | host side (layout, variants, launch_compact_passes,
// launch_sort_records) | splitter search | per-barcode runs.
#include <stdio.h>
#include <stdlib.h>

#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {

typedef u32 u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
static constexpr int kSortThreads = 256;
static constexpr int kSortWaves = kSortThreads / kWave;       // 4
static constexpr int kBins = 256;
static constexpr int kDigits = 24;                            // 3 fields x 8 bytes

__device__ __forceinline__ u64 shfl_xor64(u64 v, int m) {
  u32 lo = __shfl_xor((u32)v, m), hi = __shfl_xor((u32)(v >> 32), m);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_up64(u64 v, int d) {
  u32 lo = __shfl_up((u32)v, d), hi = __shfl_up((u32)(v >> 32), d);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ bool rec_less(u64 b, u64 u, u64 x, u64 pb, u64 pu, u64 px) {  // (b,u,x) < (pb,pu,px): record.rs:58
  return b != pb ? b < pb : (u != pu ? u < pu : x < px);
}

// =====================================================================================================
// Census: OR / AND of each field, "some index is smaller than its predecessor's", "some record is smaller than its
// predecessor" — and, with the same loop, ibu_is_sorted.  Tiled like every streaming kernel here: a wave stages 128
// records in its LDS slice with three coalesced dwordx4 loads, lane L then owns records 2L and 2L+1 and reads record
// 2L-1 from the slice as well (lane 0: one 24-byte global load of the record before the tile).
// =====================================================================================================
static constexpr int kCensusSlots = 64;                       // power of two
static constexpr size_t kCensusBytes = (size_t)kCensusSlots * 8 * sizeof(u64);   // 4 KiB at the head of the sort scratch
// Register diet (round 3): the accumulator keeps ONE word per field, d |= x ^ ref, where `ref` is a wave-uniform record that
// is itself part of the launch's rows (the first one: scalar loads, SGPRs).  OR = ref | d and AND = ref & ~d hold exactly for
// any set of rows that contains ref, and the per-wave words (ref | d_w, ref & ~d_w) merge to exactly that under the atomics
// below — so three u64 per lane do the work of six.  (With six, ibu_k_sort_compress<true, W> spilled 36 / 72 bytes per lane
// under its 64-VGPR budget and wrote 20.3 B/record instead of 13: profiles/r02_ar_pmc_WRITE_SIZE_sort_1e9.csv.)
struct CensusAcc {
  u64 d[3] = {0, 0, 0};
  bool index_drops = false, order_drops = false;
  __device__ __forceinline__ void rec(u64 b, u64 u, u64 x, const u64 (&ref)[3]) { d[0] |= b ^ ref[0]; d[1] |= u ^ ref[1]; d[2] |= x ^ ref[2]; }
  __device__ __forceinline__ void pair(u64 pb, u64 pu, u64 px, u64 b, u64 u, u64 x) {
    if (x < px) index_drops = true;                       // input not in index order: the index passes are needed
    if (rec_less(b, u, x, pb, pu, px)) order_drops = true;  // not already sorted
  }
  // c == nullptr: only the order flag is wanted (ibu_is_sorted); flag32 != nullptr receives it.
  // c: kCensusSlots x 8 words; a workgroup adds into slot blockIdx % kCensusSlots and ibu_k_sort_census_fold folds the slots
  // into slot 0 afterwards.  (With ONE slot the ~43 000 same-address atomics of a resident grid's waves took 0.5 ms — more
  // than the census of a million records itself.)
  // any_rows: wave-uniform, false for a wave that saw no row (its ref is not part of anything: it must add nothing).
  __device__ __forceinline__ void flush(u64* c, u32* flag32, const u64 (&ref)[3], bool any_rows) {
    const u32 lane = threadIdx.x & (kWave - 1);
    // (the flag only ever goes 0 -> 1: a wave that already sees it set has nothing to add — on unsorted input that spares
    // thousands of same-address atomics, ~80 us of a resident grid's tail)
    if (flag32 && __ballot(order_drops) && lane == 0 && *reinterpret_cast<volatile u32*>(flag32) == 0) atomicOr(flag32, 1u);
    if (!c || !any_rows) return;
    c += 8 * (blockIdx.x & (kCensusSlots - 1));
    if (__ballot(index_drops) && lane == 0) atomicOr(&c[6], 1ull);
    if (__ballot(order_drops) && lane == 0) atomicOr(&c[7], 1ull);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
      for (int f = 0; f < 3; ++f) d[f] |= shfl_xor64(d[f], m);
    if (lane == 0)
#pragma unroll
      for (int f = 0; f < 3; ++f) { atomicOr(&c[f], ref[f] | d[f]); atomicAnd(&c[3 + f], ref[f] & ~d[f]); }
  }
};

extern "C" __global__ void ibu_k_sort_census_init(u64* c) {   // one block of kCensusSlots * 8 threads
  const u32 w = threadIdx.x & 7u;
  c[threadIdx.x] = (w >= 3 && w < 6) ? ~0ull : 0;            // [0..2] OR, [3..5] AND, [6]: some index smaller than its predecessor's; [7]: some record smaller
}
// slots -> slot 0 (one wave: lane = slot)
extern "C" __global__ void ibu_k_sort_census_fold(u64* c) {
  const u32 lane = threadIdx.x;
  u64 v[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) v[w] = c[8 * lane + w];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const u64 o = shfl_xor64(v[w], m);
      v[w] = (w >= 3 && w < 6) ? (v[w] & o) : (v[w] | o);
    }
  if (lane == 0)
#pragma unroll
    for (int w = 0; w < 8; ++w) c[w] = v[w];
}
// recs0: row 0 of the caller's array (8-B aligned); the tiles start at row `row0` (16-B aligned there).
// The record IN FRONT of a tile (the partner of the tile's first record in the order checks) travels with the tile: its 24
// bytes are the tail of the 32 bytes in front of the tile, which every lane loads as one more dwordx4 of the prefetch (two
// distinct chunks, one cache line) and lanes 0 / 1 stage right in front of the tile in LDS — so record 2L-1 is `r[-3 .. -1]`
// for lane 0 too.  (Round 2 had lane 0 fetch it with a separate 24-byte global load INSIDE the iteration that used it: the
// wait for that load was a vmcnt(0), which also waited for the next tile's prefetch — every iteration paid a full memory
// latency; ibu_k_sort_compress<true> likewise: 8.5 ms against 6.7 without the census.)
static constexpr int kPrevBytes = 32;                         // staged in front of each wave's tile
static constexpr int kSliceBytes = kTileBytes + kPrevBytes;
__device__ __forceinline__ const uint8_t* prev_chunk(const uint8_t* tile_src, bool has_prev, u32 lane) {
  return (has_prev ? tile_src - kPrevBytes : tile_src) + 16 * (lane & 1u);   // no record in front: any valid bytes (ignored)
}
struct CensusRegs { u32x4 v[4]; };                            // a tile (three dwordx4 per lane) + the 32 bytes in front of it
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_sort_census(const u64* __restrict__ recs0, u64 row0, u32 ntiles, u64* __restrict__ c, u32* __restrict__ flag32) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kSliceBytes];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kSliceBytes + kPrevBytes;
  const uint8_t* base = reinterpret_cast<const uint8_t*>(recs0 + 3 * row0);
  const u64* rp = recs0 + 3 * row0;                         // ntiles >= 1: the first tiled row is a row of this launch
  const u64 ref[3] = {rp[0], rp[1], rp[2]};                 // uniform address: scalar loads
  CensusAcc acc;
  const TileRange tr = {logical_block() * (u32)kWavesPerBlock + wib, gridDim.x * (u32)kWavesPerBlock, ntiles};
  sweep_tiles<CensusRegs>(
      tr,
      [&](CensusRegs& g, u32 t) {
        const uint8_t* src = base + (size_t)t * kTileBytes;
#pragma unroll
        for (int k = 0; k < 3; ++k) g.v[k] = ld16(src + 1024 * k + 16 * lane);
        g.v[3] = ld16(prev_chunk(src, row0 + (u64)t * kTileRecs > 0, lane));
      },
      [&](const CensusRegs& g, u32 t) {
        const bool has_prev = lane > 0 || row0 + (u64)t * kTileRecs > 0;   // global row of the tile's first record > 0
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(tile + 1024 * k + 16 * lane) = g.v[k];
        if (lane < 2) *reinterpret_cast<u32x4*>(tile - kPrevBytes + 16 * lane) = g.v[3];
        wave_lds_fence();
        const u64* r = reinterpret_cast<const u64*>(tile + (2 * lane) * 24);  // records 2L, 2L+1 (and 2L-1 just below)
        const u64 p0 = r[-3], p1 = r[-2], p2 = r[-1];
        const u64 x0 = r[0], x1 = r[1], x2 = r[2], y0 = r[3], y1 = r[4], y2 = r[5];
        acc.rec(x0, x1, x2, ref);
        acc.rec(y0, y1, y2, ref);
        if (has_prev) acc.pair(p0, p1, p2, x0, x1, x2);
        acc.pair(x0, x1, x2, y0, y1, y2);
      });
  acc.flush(c, flag32, ref, tr.t < tr.end);
}
// rows [row0, n), one thread per row (the n % 128 rest, a peeled first row); compares with row - 1 as well
extern "C" __global__ void ibu_k_sort_census_tail(const u64* __restrict__ recs, u64 row0, u64 n, u64* __restrict__ c,
                                                  u32* __restrict__ flag32) {
  CensusAcc acc;
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 w0 = row0 + ((u64)blockIdx.x * blockDim.x + (threadIdx.x & ~(u32)(kWave - 1)));   // the wave's first row
  const bool any_rows = w0 < n;                              // wave-uniform
  const u64* rp = recs + 3 * (any_rows ? w0 : row0);
  const u64 ref[3] = {rp[0], rp[1], rp[2]};
  if (i < n) {
    const u64 b = recs[3 * i], u = recs[3 * i + 1], x = recs[3 * i + 2];
    acc.rec(b, u, x, ref);
    if (i > 0) acc.pair(recs[3 * i - 3], recs[3 * i - 2], recs[3 * i - 1], b, u, x);
  }
  acc.flush(c, flag32, ref, any_rows);
}
static void launch_census(const LaunchCfg& cfg, const void* recs, size_t n, u64* census, u32* flag32, hipStream_t st) {
  const Span sp[1] = {{recs, 24}};
  const RowSplit rs = split_rows(sp, 1, n, kTileRecs);   // an 8-B aligned base peels exactly one record
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_sort_census_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, (const u64*)recs, (u64)0, (u64)rs.head,
                       census, flag32);
  if (rs.main) {
    const u32 ntiles = (u32)(rs.main / kTileRecs);
    static std::atomic<int> occ;
    hipLaunchKernelGGL(ibu_k_sort_census, dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_sort_census, 0, &occ))),
                       dim3(kBlock), 0, st, (const u64*)recs, (u64)rs.head, ntiles, census, flag32);
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_sort_census_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, (const u64*)recs,
                       (u64)(rs.head + rs.main), (u64)n, census, flag32);
}
hipError_t launch_sorted_check(const LaunchCfg& cfg, const void* recs, size_t n, uint32_t* flag, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n < 2) return hipSuccess;
  launch_census(cfg, recs, n, nullptr, flag, st);
  return hipGetLastError();
}

// =====================================================================================================
// One radix pass = count, scan, scatter — WITHOUT reading the records twice:
//
//   tile counts   256 bin counts (u16) of every T-record tile of the pass's input.  For the FIRST pass they come from
//                 one streaming read of the records (24 B/record, once per sort).  For every later pass they come from
//                 the DIGIT SIDE STREAM: while a pass scatters record r to position g it also stores the digit the
//                 NEXT pass will sort by at byte g of a side buffer, so the next pass counts by reading 1 byte per
//                 record instead of 24.
//   scan          three small kernels turn the [tile][bin] counts into each tile's first output position per bin
//                 (sums per block of 1024 tiles -> running sums over blocks and bin bases -> running sums inside a block).
//   scatter       a workgroup stages its tile in LDS, ranks it with wave-level match-any (8 ballots per record), permutes
//                 it into digit order inside LDS and writes the runs out as consecutive 8-byte words, starting at the
//                 positions the scan left for it.  No workgroup ever waits for another one.
// HBM traffic per record: 24 B (census) + 24 B (first counts) once, then per pass 24 B read + 24 B written + 1 B side
// stream written + 1 B read back + 6.5 B of tables per 2048-record tile... (0.5 KiB counts written, read twice; 1 KiB positions
// written and read) = about 50.2 B.
//
// Why not a single-kernel "onesweep" with decoupled look-back?  It was built first (profiles/experiments/r02_sort_onesweep_*):
// correct, but on this part a status poll from a CU that is streaming takes ~2 us, a tile had to walk ~20-30 predecessor
// rows (the walk must cross every tile that started during one walk), and the look-back cost HALF of each pass
// (5e8 records, 11 passes: 101 ms with look-back, 52 ms with the look-back compiled out, 6.0 TB/s).  Windowed polls,
// a dedicated scan workgroup handing prefixes out, and three tiles per CU did not change that.  Precomputed positions
// cost 2.2 B/record/pass of extra traffic and no waiting at all.
// =====================================================================================================
#ifndef IBU_TILES_PER_BLOCK
#define IBU_TILES_PER_BLOCK 256   // 1024: the position walk of a block (ibu_k_sort_tilepos) took 0.30 ms per pass at 1e9 records; 256: 3 ms less per sort
#endif
static constexpr int kTilesPerBlock = IBU_TILES_PER_BLOCK;                   // tiles per scan block
// One count into an LDS histogram; when all the wave's active lanes hold the same digit (runs of equal keys), one lane adds for
// all of them: 42 lanes adding to one word would take 42 turns.
__device__ __forceinline__ void hist_add(u32* h, u32 d, bool active) {
  if (active) {
    const u32 f = (u32)__builtin_amdgcn_readfirstlane((int)d);
    if (__ballot(d != f) == 0) {                               // wave-uniform
      const u64 act = __ballot(true);
      if ((threadIdx.x & (kWave - 1)) == (u32)__ffsll((long long)act) - 1u) atomicAdd(&h[f], (u32)__popcll(act));
    } else {
      atomicAdd(&h[d], 1u);
    }
  }
}

// ---- tile counts from the records (first pass): chunk-field trick of ibu_k_reduce, no LDS staging ------------------
// One workgroup per tile.  The wave stride (3072 B = 384 u64) is a multiple of 3, so the u64 in slot (k, h) of a lane's
// three dwordx4 loads always belongs to field (2 (64 k + lane) + h) % 3; only the slots of the pass's field count.
template <int T>
__global__ void __launch_bounds__(kSortThreads, 8)
ibu_k_sort_tilecounts_recs(const uint8_t* __restrict__ recs, u32 nfull, u32 field, u32 shift, uint16_t* __restrict__ counts,
                           uint8_t* __restrict__ copy_dst) {   // copy_dst != nullptr: the records are copied there on the way
  __shared__ u32 h[kBins];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  constexpr int kSub = T / kTileRecs;                         // 128-record sub-tiles per tile
  static_assert(T % kTileRecs == 0 && kSub % kSortWaves == 0, "tile must be a whole number of sub-tiles per wave");
  bool mine[3][2];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) mine[k][hh] = (2 * (64 * k + lane) + hh) % 3 == field;
  for (u32 tile = blockIdx.x; tile < nfull; tile += gridDim.x) {
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t* base = recs + (size_t)tile * T * 24 + 16 * lane;
#pragma unroll
    for (int i = 0; i < kSub / kSortWaves; ++i) {
      const uint8_t* p = base + (size_t)(wib + kSortWaves * i) * kTileBytes;
      u32x4 a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) a[k] = ld16(p + 1024 * k);
      if (copy_dst) {                                          // block-uniform
        uint8_t* q = copy_dst + (p - recs);
#pragma unroll
        for (int k = 0; k < 3; ++k) st16(q + 1024 * k, a[k]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const u64 v0 = ((u64)a[k].y << 32) | a[k].x, v1 = ((u64)a[k].w << 32) | a[k].z;
        hist_add(h, (u32)(v0 >> shift) & 255u, mine[k][0]);
        hist_add(h, (u32)(v1 >> shift) & 255u, mine[k][1]);
      }
    }
    __syncthreads();
    counts[(size_t)tile * kBins + threadIdx.x] = (uint16_t)h[threadIdx.x];
    __syncthreads();
  }
}
// any alignment, any tile length: one thread per record of tile `tile0 + blockIdx.x` (the ragged last tile, or every
// tile of an input that is only 8-byte aligned)
template <int T>
__global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_tilecounts_recs_tail(const u64* __restrict__ recs, u64 n, u32 tile0, u32 field, u32 shift, uint16_t* __restrict__ counts,
                                u64* __restrict__ copy_dst) {
  __shared__ u32 h[kBins];
  const u32 tile = tile0 + blockIdx.x;
  h[threadIdx.x] = 0;
  __syncthreads();
  const u64 tbase = (u64)tile * T;
  const u32 cnt = n - tbase < (u64)T ? (u32)(n - tbase) : (u32)T;
  for (u32 i = threadIdx.x; i < cnt; i += kSortThreads) {
    const u64* r = recs + 3 * (tbase + i);
    atomicAdd(&h[(u32)(r[field] >> shift) & 255u], 1u);
    if (copy_dst) { u64* w = copy_dst + 3 * (tbase + i); w[0] = r[0]; w[1] = r[1]; w[2] = r[2]; }
  }
  __syncthreads();
  counts[(size_t)tile * kBins + threadIdx.x] = (uint16_t)h[threadIdx.x];
}
// ---- tile counts from the digit side stream: one WAVE per tile, wave-private LDS histogram --------------------------
template <int T>
__global__ void __launch_bounds__(kSortThreads, 8)
ibu_k_sort_tilecounts_bytes(const uint8_t* __restrict__ digits, u64 n, u32 ntiles, uint16_t* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) u32 hist[kSortWaves][kBins];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  u32* h = hist[wib];
  constexpr int kLoads = (T + 16 * kWave - 1) / (16 * kWave);  // dwordx4 per lane per tile (the last may hang over: ignored)
  const u32 nwaves = gridDim.x * kSortWaves;
  for (u32 tile = blockIdx.x * kSortWaves + wib; tile < ntiles; tile += nwaves) {
    const u64 tbase = (u64)tile * T;
    const u32 cnt = n - tbase < (u64)T ? (u32)(n - tbase) : (u32)T;
    u32x4 v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) v[k] = ld16(digits + tbase + 16 * (lane + kWave * k));   // the buffer is padded to whole tiles + 1 KiB
    wave_lds_fence();
    *reinterpret_cast<u32x4*>(&h[4 * lane]) = u32x4{0, 0, 0, 0};
    wave_lds_fence();
    // Equal digits next to each other are the rule in the later passes of grouped input (barcodes from a whitelist: once the
    // low barcode bytes are sorted, the high ones come in runs of hundreds to millions), and 64 lanes adding to ONE LDS word
    // take 64 turns: a pass over such a stream took 3.4 ms instead of 0.27 (profiles/README.md r03_wl).  So a lane adds a run
    // of equal bytes once, and lanes whose 16 bytes are one value hand them to the first lane of their stretch.
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const u32 b0 = 16 * (lane + kWave * k);                 // tile-relative byte of this chunk
      const u32 w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      const u32 val = w[0] & 255u;
      const bool flat = w[0] == val * 0x01010101u && w[1] == w[0] && w[2] == w[0] && w[3] == w[0] && b0 + 16 <= cnt;
      const u64 flat_m = __ballot(flat);
      const u32 left = (u32)__shfl_up((int)val, 1);
      const bool follows = flat && lane > 0 && ((flat_m >> (lane - 1)) & 1ull) && left == val;   // the lane before holds the same 16 bytes
      const u64 follow_m = __ballot(follows);
      if (flat) {
        if (!follows) {                                        // first of its stretch: add for the lanes that follow it
          const u64 behind = lane < kWave - 1 ? follow_m >> (lane + 1) : 0ull;
          atomicAdd(&h[val], 16u * (1u + (u32)__builtin_ctzll(~behind)));   // ~behind != 0: the shift cleared the top bit
        }
        continue;
      }
      u32 prev = val, run = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const u32 d = (w[j >> 2] >> (8 * (j & 3))) & 255u;
        if (d != prev) {
          if (run) atomicAdd(&h[prev], run);
          run = 0;
          prev = d;
        }
        run += b0 + j < cnt ? 1u : 0u;
      }
      if (run) atomicAdd(&h[prev], run);
    }
    wave_lds_fence();
    const u32x4 c = *reinterpret_cast<const u32x4*>(&h[4 * lane]);
    u32x2 o; o.x = c.x | (c.y << 16); o.y = c.z | (c.w << 16);
    *reinterpret_cast<u32x2*>(counts + (size_t)tile * kBins + 4 * lane) = o;
  }
}
// ---- scan ------------------------------------------------------------------------------------------------------------
// 1. per block of kTilesPerBlock tiles: column sums.  Wave w takes tiles w, w+4, ...; lane l the bins 4l..4l+3.
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_blocksums(const uint16_t* __restrict__ counts, u32 ntiles, u32* __restrict__ blocksum) {
  __shared__ u32 part[kSortWaves][kBins];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  const u32 t0 = blockIdx.x * kTilesPerBlock, t1 = t0 + kTilesPerBlock < ntiles ? t0 + kTilesPerBlock : ntiles;
  u32 acc[4] = {0, 0, 0, 0};
  for (u32 t = t0 + wib; t < t1; t += kSortWaves) {
    const u32x2 c = *reinterpret_cast<const u32x2*>(counts + (size_t)t * kBins + 4 * lane);
    acc[0] += c.x & 0xFFFFu; acc[1] += c.x >> 16; acc[2] += c.y & 0xFFFFu; acc[3] += c.y >> 16;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) part[wib][4 * lane + j] = acc[j];
  __syncthreads();
  blocksum[(size_t)blockIdx.x * kBins + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
// 2. one workgroup, thread = bin: running sums over the blocks (in place, u64) and the first position of each bin.
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_blockscan(const u32* __restrict__ blocksum, u32 nblocks, u64* __restrict__ blockoff, u64* __restrict__ binbase) {
  __shared__ u64 wsum[kSortWaves];
  const u32 bin = threadIdx.x, lane = bin & (kWave - 1), wib = bin >> 6;
  u64 running = 0;
  for (u32 b0 = 0; b0 < nblocks; b0 += 8) {
    u32 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = b0 + j < nblocks ? blocksum[(size_t)(b0 + j) * kBins + bin] : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (b0 + j < nblocks) blockoff[(size_t)(b0 + j) * kBins + bin] = running;
      running += v[j];
    }
  }
  // exclusive scan of the bin totals over the 256 bins
  u64 inc = running;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const u64 t = shfl_up64(inc, d);
    if (lane >= (u32)d) inc += t;
  }
  if (lane == kWave - 1) wsum[wib] = inc;
  __syncthreads();
  u64 off = 0;
  for (u32 w = 0; w < wib; ++w) off += wsum[w];
  binbase[bin] = off + inc - running;
}
// 3. per block, thread = bin: first output position of every (tile, bin).
template <class IDX>
__global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_tilepos(const uint16_t* __restrict__ counts, u32 ntiles, const u64* __restrict__ blockoff, const u64* __restrict__ binbase,
                   IDX* __restrict__ pos) {
  const u32 bin = threadIdx.x;
  const u32 t0 = blockIdx.x * kTilesPerBlock, t1 = t0 + kTilesPerBlock < ntiles ? t0 + kTilesPerBlock : ntiles;
  u64 running = binbase[bin] + blockoff[(size_t)blockIdx.x * kBins + bin];
  constexpr int kFly = 32;                                    // loads in flight per thread: the walk is latency-bound
  for (u32 t = t0; t < t1; t += kFly) {
    u32 v[kFly];
#pragma unroll
    for (int j = 0; j < kFly; ++j) v[j] = t + j < t1 ? counts[(size_t)(t + j) * kBins + bin] : 0;
#pragma unroll
    for (int j = 0; j < kFly; ++j) {
      if (t + j < t1) pos[(size_t)(t + j) * kBins + bin] = (IDX)running;
      running += v[j];
    }
  }
}

// ---- scatter -----------------------------------------------------------------------------------------------------------
template <int THREADS, int ROUNDS>
struct SweepShape {
  static constexpr int T = THREADS * ROUNDS, NW = THREADS / kWave, PER_WAVE = T / NW;
  // LDS: stage 24 T | gdelta 256 x u64 | whist NW x 256 x u32 | misc 16 x u32 | sbin T bytes
  static constexpr size_t lds = 24 * (size_t)T + 8 * kBins + 4 * (size_t)NW * kBins + 64 + (size_t)T;
};

// field / shift: this pass's digit; nfield / nshift: the next pass's (nfield > 2: there is none, no side stream).
// WMODE 0: the real thing.  WMODE 2 (probe builds, -DIBU_SORT_PROBE, WRONG output): the permuted tile goes out linearly.
template <int THREADS, int ROUNDS, class IDX, int WMODE>
__global__ void __launch_bounds__(THREADS)
ibu_k_sort_scatter(const u64* __restrict__ src, u64* __restrict__ dst, u64 n, u32 field, u32 shift, u32 nfield, u32 nshift,
                   const IDX* __restrict__ pos, uint8_t* __restrict__ digits) {
  typedef SweepShape<THREADS, ROUNDS> S;
  constexpr int T = S::T, NW = S::NW, PER_WAVE = S::PER_WAVE;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u64* stage = reinterpret_cast<u64*>(smem);                 // the tile: first in input order, then in digit order
  u64* gdelta = stage + 3 * T;                               // global record index of slot p of bin d = gdelta[d] + p
  u32* whist = reinterpret_cast<u32*>(gdelta + kBins);       // per wave: running count while ranking, then base slot of (wave, bin)
  u32* misc = whist + NW * kBins;                            // [0..3] scan scratch
  uint8_t* sbin = reinterpret_cast<uint8_t*>(misc + 16);     // digit of each slot of the permuted tile
  const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wib = tid >> 6;
  const u64 lt_mask = (1ull << lane) - 1;
  // XCD-aware tile order (speed only): hardware deals workgroup b to XCD b % 8, so XCD x gets the CONSECUTIVE tiles
  // [x * gridDim/8, (x+1) * gridDim/8) in dispatch order.  The run of bin d of tile t+1 continues where tile t's ended,
  // usually in the middle of a 128-byte line: with both tiles on one XCD, close in time, the two halves meet in that
  // XCD's L2 and the line leaves it once, whole (in identity order every boundary line is written twice, by
  // two XCDs, as partial lines).
  const u32 tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // the grid is a multiple of 8
  const u64 tbase = (u64)tile * T;
  if (tbase >= n) return;                                     // block-uniform: padding of the grid
  const u32 cnt = n - tbase < (u64)T ? (u32)(n - tbase) : (u32)T;
  // this tile's first output position per bin: loaded now, needed after the permutation
  const u64 mypos = tid < (u32)kBins ? (u64)pos[(size_t)tile * kBins + tid] : 0;

  // 1. stage the tile (coalesced) and clear the per-wave counters
  if (cnt == (u32)T && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0)) {
    const u32x4* g = reinterpret_cast<const u32x4*>(src + 3 * tbase);
    u32x4* s = reinterpret_cast<u32x4*>(stage);
    constexpr int kTileChunks = T * 24 / 16;                 // dwordx4 per tile (1.5 T)
    constexpr int kChunks = (kTileChunks + THREADS - 1) / THREADS;
    u32x4 v[kChunks];
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const u32 c = tid + THREADS * k;
      v[k] = ld16(g + (c < (u32)kTileChunks ? c : (u32)kTileChunks - 1));   // unconditional load, clamped (kcommon.hpp)
    }
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const u32 c = tid + THREADS * k;
      if (kTileChunks % THREADS == 0 || c < (u32)kTileChunks) s[c] = v[k];
    }
  } else {
    for (u32 w = tid; w < 3 * cnt; w += THREADS) stage[w] = src[3 * tbase + w];
  }
#pragma unroll
  for (int k = 0; k < kBins / kWave; ++k) whist[wib * kBins + lane + kWave * k] = 0;
  __syncthreads();

  // 2. rank every record among the records of its wave with the same digit (stable: slot order)
  u64 r0[ROUNDS], r1[ROUNDS], r2[ROUNDS];
  u32 dig[ROUNDS], rk[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const u32 slot = wib * PER_WAVE + r * kWave + lane;
    const bool valid = slot < cnt;
    r0[r] = r1[r] = r2[r] = 0;
    if (valid) { r0[r] = stage[3 * slot]; r1[r] = stage[3 * slot + 1]; r2[r] = stage[3 * slot + 2]; }
    const u64 key = field == 0 ? r0[r] : (field == 1 ? r1[r] : r2[r]);
    const u32 d = (u32)(key >> shift) & 255u;
    u64 m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const u64 bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const u32 before = (u32)__popcll(m & lt_mask);
    const u32 prev = valid ? whist[wib * kBins + d] : 0;
    wave_lds_fence();                                        // every lane has read before the leaders write
    if (valid && before == 0) whist[wib * kBins + d] = prev + (u32)__popcll(m);
    wave_lds_fence();
    dig[r] = d;
    rk[r] = prev + before;
  }
  __syncthreads();                                           // counters complete; the tile now lives in registers

  // 3. bin totals of the tile -> slot bases per (wave, bin); global record index of slot p of bin d = gdelta[d] + p
  {
    u32 c[NW], tot = 0;
    if (tid < (u32)kBins) {
#pragma unroll
      for (int w = 0; w < NW; ++w) { c[w] = whist[w * kBins + tid]; tot += c[w]; }
    }
    u32 all;
    const u32 tb = block_exclusive_scan(tid < (u32)kBins ? tot : 0u, misc, &all);
    if (tid < (u32)kBins) {
      u32 run = tb;
#pragma unroll
      for (int w = 0; w < NW; ++w) { whist[w * kBins + tid] = run; run += c[w]; }
      gdelta[tid] = mypos - tb;                               // wraps harmlessly: slot >= tb for this bin
    }
  }
  __syncthreads();

  // 4. permute the tile into digit order inside LDS
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const u32 slot = wib * PER_WAVE + r * kWave + lane;
    if (slot < cnt) {
      const u32 p = whist[wib * kBins + dig[r]] + rk[r];
      stage[3 * p] = r0[r]; stage[3 * p + 1] = r1[r]; stage[3 * p + 2] = r2[r];
      sbin[p] = (uint8_t)dig[r];
    }
  }
  __syncthreads();

  // 5. write out (plain stores: the L2 merges the pieces of a run that neighbouring tiles write)
  const u32 nw = 3 * cnt;
  if constexpr (WMODE == 0) {
    // one lane per HALF record (12 bytes, dwordx3): consecutive lanes on consecutive 12-byte pieces of a run, 768 contiguous
    // bytes per wave instruction (the compact last pass's write-out).  One lane per 8-byte word: 512 bytes per instruction,
    // 10.37 instead of 9.45 ms per pass at 1e9 records (profiles r03_v); 16-byte chunks with single-word heads and tails were
    // slower still (more address arithmetic than they save).
    const u32* stage32 = reinterpret_cast<const u32*>(stage);
    for (u32 h = tid; h < 2 * cnt; h += THREADS) {
      const u32 s = h >> 1, j = h & 1u;
      const u64 g = gdelta[sbin[s]] + s;
      u32x3 o;
      o.x = stage32[6 * s + 3 * j]; o.y = stage32[6 * s + 3 * j + 1]; o.z = stage32[6 * s + 3 * j + 2];
      *reinterpret_cast<u32x3_a4*>(reinterpret_cast<uint8_t*>(dst) + 24 * g + 12 * j) = o;
    }
  } else {
    u32x4* o = reinterpret_cast<u32x4*>(dst + 3 * tbase);
    const u32x4* s = reinterpret_cast<const u32x4*>(stage);
    for (u32 c = tid; 2 * c + 1 < nw; c += THREADS) o[c] = s[c];
  }
  // 6. the digit the NEXT pass sorts by, at the record's new position: 1 byte per record instead of a 24-byte re-read
  if (nfield < 3) {
    for (u32 p = tid; p < cnt; p += THREADS) {
      const u64 g = WMODE == 0 ? gdelta[sbin[p]] + p : tbase + p;
      digits[g] = (uint8_t)(stage[3 * p + nfield] >> nshift);
    }
  }
}

// =====================================================================================================
// COMPACT-KEY passes.  A record is 24 bytes, but the census usually finds few of them varying: 16-base barcodes, 12-base
// UMIs and indices below 2^32 vary in 4 + 3 + 4 = 11 bytes, and every other byte is the same in all records.  When at
// most 12 bytes vary (and n < 2^32) the sort runs on 12-BYTE ELEMENTS instead of records:
//
//   compress   records -> elements: element byte j = the j-th least significant varying byte of the key (index bytes
//              lowest, barcode bytes highest), so the element read as a 96-bit little-endian integer orders like the
//              record; also the 1-byte digit side stream of the first pass.                      24 R + 13 W per record
//   passes     LSD over the element bytes that must be sorted (not the index bytes when the input is in index
//              order), counts from the side stream, scan, scatter as above — on half the bytes.  ~26.6 B per record
//   expand     fused into the LAST pass: its scatter writes every element as the 24-byte record it stands for (constant
//              bytes from the census' AND words) straight into the caller's array.            12 R + 24 W per record
//
// 16/12 with a random index: 24 (census) + 37 + 10 x 26.6 + 38 = 365 B/record against 24 + 48 + 11 x 51.9 = 643.
// Both element buffers live in the caller's `tmp` (12 n bytes each), so the second one starts at a 4-byte boundary: every
// element access is a per-lane dwordx3 (64 lanes x 12 B = 768 contiguous bytes), which needs no more than that — and a
// lane that loads whole elements needs no LDS staging in front of the ranking.  The result is the same permutation as the
// 24-byte passes give (stable LSD over the same digits; constant bytes never decide a comparison).
// =====================================================================================================
// Elements of W 32-bit words: W = 3 (12 bytes: at most 12 varying key bytes) or W = 4 (16 bytes: 13 .. 16).  ElemT<W>: in
// memory (4-byte aligned); EV<W>: in registers.
template <int W> struct __attribute__((packed, aligned(4))) ElemT { u32 w[W]; };
typedef ElemT<3> Elem;                                        // the 12-byte element of the C ABI (ibu_records_compact)
static_assert(sizeof(ElemT<3>) == 12 && sizeof(ElemT<4>) == 16, "element sizes");
template <int W> struct EV { u32 w[W]; };
template <int W>
__device__ __forceinline__ EV<W> ld_elem(const ElemT<W>* p) {  // 4-byte aligned: ONE global_load_dwordx3 / x4, read once (nt)
  EV<W> v;
  if constexpr (W == 3) {
    const u32x3 t = __builtin_nontemporal_load(reinterpret_cast<const u32x3_a4*>(p));
    v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z;
  } else {
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4*>(p));
    v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z; v.w[3] = t.w;
  }
  return v;
}
template <int W>
__device__ __forceinline__ void st_elem(ElemT<W>* p, EV<W> v) { __builtin_memcpy(p, &v, 4 * W); }

// Byte gathers as v_perm_b32: a selector byte 0..7 picks a byte of the (hi, lo) register pair, 0x0C gives zero (CompactPlan:
// kernels.h).
template <int W>
__device__ __forceinline__ EV<W> compress_rec(u64 f0, u64 f1, u64 f2, const CompactPlan& pl) {
  EV<W> e;
#pragma unroll
  for (int w = 0; w < W; ++w)
    e.w[w] = __builtin_amdgcn_perm((u32)(f0 >> 32), (u32)f0, pl.csel[w][0]) | __builtin_amdgcn_perm((u32)(f1 >> 32), (u32)f1, pl.csel[w][1]) |
             __builtin_amdgcn_perm((u32)(f2 >> 32), (u32)f2, pl.csel[w][2]);
  return e;
}
template <int W>
__device__ __forceinline__ void expand_elem(EV<W> v, const CompactPlan& pl, u64& f0, u64& f1, u64& f2) {
  u32 d[6], w3 = 0;
  if constexpr (W == 4) w3 = v.w[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = __builtin_amdgcn_perm(v.w[1], v.w[0], pl.xsel[k][0]) | __builtin_amdgcn_perm(w3, v.w[2], pl.xsel[k][1]);
  f0 = pl.base[0] | ((u64)d[1] << 32) | d[0];
  f1 = pl.base[1] | ((u64)d[3] << 32) | d[2];
  f2 = pl.base[2] | ((u64)d[5] << 32) | d[4];
}
template <int W>
__device__ __forceinline__ u32 elem_byte(EV<W> e, u32 byte) {  // byte: uniform
  const u32 w = byte >> 2;
  u32 x = w == 0 ? e.w[0] : w == 1 ? e.w[1] : e.w[2];
  if constexpr (W == 4) x = w == 3 ? e.w[3] : x;
  return (x >> (8 * (byte & 3))) & 255u;
}
// records [0, 128 ntiles) -> elements + first digit; tiled like the census (recs 16-B aligned).  Lane L owns records L and
// L + 64 of the tile: stride-24 ds_read_b64 is conflict-free and each of its two element stores is 768 contiguous bytes.
// CENSUS: the exact census (OR / AND words, order flags: CensusAcc) of the same records is accumulated on the way — the
// speculative path of the sort, whose plan comes from a SAMPLE and is checked against this census afterwards.
template <bool CENSUS, int W>
__global__ void __launch_bounds__(kBlock, CENSUS ? (W == 4 ? 5 : 7) : 8)   // the launcher keeps at most 7 workgroups per CU resident (LaunchCfg); 16-byte elements with the census need 84 VGPRs
ibu_k_sort_compress(const uint8_t* __restrict__ recs, u32 ntiles, CompactPlan pl, u32 first_byte, ElemT<W>* __restrict__ out,
                    uint8_t* __restrict__ digits, u64* __restrict__ census) {
  constexpr int kSlice = CENSUS ? kSliceBytes : kTileBytes;  // with the census: the record in front of the tile is staged too (ibu_k_sort_census)
  constexpr int kLead = CENSUS ? kPrevBytes : 0;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kSlice];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kSlice + kLead;
  const TileRange tr = tile_range(ntiles, wib);             // which tiles this wave sweeps (kcommon.hpp)
  if (tr.t >= tr.end) return;                                // wave-uniform: a wave without tiles adds nothing to the census
  CensusAcc acc;
  const u64* rp = reinterpret_cast<const u64*>(recs);        // row 0 of this launch: the census' reference record (CensusAcc)
  const u64 ref[3] = {CENSUS ? rp[0] : 0, CENSUS ? rp[1] : 0, CENSUS ? rp[2] : 0};
  struct Regs { u32x4 v[CENSUS ? 4 : 3]; };
  sweep_tiles<Regs>(
      tr,
      [&](Regs& g, u32 t) {
        const uint8_t* src = recs + (size_t)t * kTileBytes;
#pragma unroll
        for (int k = 0; k < 3; ++k) g.v[k] = ld16(src + 1024 * k + 16 * lane);
        if constexpr (CENSUS) g.v[3] = ld16(prev_chunk(src, t > 0, lane));
      },
      [&](const Regs& g, u32 t) {
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(tile + 1024 * k + 16 * lane) = g.v[k];
        if constexpr (CENSUS) { if (lane < 2) *reinterpret_cast<u32x4*>(tile - kPrevBytes + 16 * lane) = g.v[3]; }
        wave_lds_fence();
        const u64* r = reinterpret_cast<const u64*>(tile + lane * 24);
        const u64* q = reinterpret_cast<const u64*>(tile + (lane + kWave) * 24);
        if constexpr (CENSUS) {
          acc.rec(r[0], r[1], r[2], ref);
          acc.rec(q[0], q[1], q[2], ref);
          if (lane > 0 || t > 0) acc.pair(r[-3], r[-2], r[-1], r[0], r[1], r[2]);   // lane 0: the record in front of the tile
          acc.pair(q[-3], q[-2], q[-1], q[0], q[1], q[2]);
        }
        const EV<W> e0 = compress_rec<W>(r[0], r[1], r[2], pl), e1 = compress_rec<W>(q[0], q[1], q[2], pl);
        const size_t row = (size_t)t * kTileRecs + lane;
        st_elem<W>(out + row, e0);
        st_elem<W>(out + row + kWave, e1);
        if (digits) {                                        // uniform (NULL: ibu_records_compact, no pass follows)
          digits[row] = (uint8_t)elem_byte<W>(e0, first_byte);
          digits[row + kWave] = (uint8_t)elem_byte<W>(e1, first_byte);
        }
      });
  if constexpr (CENSUS) acc.flush(census, nullptr, ref, true);
}
// the digit stream of a pass from the elements themselves (the speculative path guessed another first pass)
template <int W>
__global__ void ibu_k_sort_digits(const ElemT<W>* __restrict__ in, u64 n, u32 byte, uint8_t* __restrict__ digits) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) digits[i] = (uint8_t)elem_byte<W>(ld_elem<W>(in + i), byte);
}
template <int W>
__global__ void ibu_k_sort_compress_tail(const u64* __restrict__ recs, u64 row0, u64 n, CompactPlan pl, u32 first_byte,
                                         ElemT<W>* __restrict__ out, uint8_t* __restrict__ digits) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const EV<W> e = compress_rec<W>(recs[3 * i], recs[3 * i + 1], recs[3 * i + 2], pl);
  st_elem<W>(out + i, e);
  if (digits) digits[i] = (uint8_t)elem_byte<W>(e, first_byte);
}
// elements -> records [0, 128 nsub) (recs 16-B aligned): ibu_records_expand (the sort itself expands in its last pass).
// Two 128-element sub-tiles per iteration (four element loads per lane in flight behind the current ones); lane L owns
// elements L and L + 64 of a sub-tile.
static constexpr int kExpandSub = 2;
template <int W>
__global__ void __launch_bounds__(kBlock, 8)
ibu_k_sort_expand(const ElemT<W>* __restrict__ in, u32 ntiles /*of 128 * kExpandSub*/, u32 nsub /*128-element sub-tiles in all*/, CompactPlan pl,
                  uint8_t* __restrict__ recs) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kTileBytes;
  const TileRange tr = tile_range(ntiles, wib);
  u32 t = tr.t;
  if (t >= tr.end) return;
  EV<W> a[2 * kExpandSub];
  auto issue = [&](u32 tt, EV<W>* v) {
#pragma unroll
    for (int s = 0; s < kExpandSub; ++s) {
      u32 sub = tt * kExpandSub + s;
      sub = sub < nsub ? sub : nsub - 1;                     // the last tile may be half empty: clamped, unconditional
      v[2 * s] = ld_elem<W>(in + (size_t)sub * kTileRecs + lane);
      v[2 * s + 1] = ld_elem<W>(in + (size_t)sub * kTileRecs + lane + kWave);
    }
  };
  issue(t, a);
  for (;;) {
    const u32 tn = t + tr.stride;
    const bool more = tn < tr.end;
    EV<W> b[2 * kExpandSub];
    issue(more ? tn : t, b);
#pragma unroll
    for (int s = 0; s < kExpandSub; ++s) {
      const u32 sub = t * kExpandSub + s;
      u64 f[6];
      expand_elem<W>(a[2 * s], pl, f[0], f[1], f[2]);
      expand_elem<W>(a[2 * s + 1], pl, f[3], f[4], f[5]);
      wave_lds_fence();                                      // the previous sub-tile's reads precede these writes
      u64* r = reinterpret_cast<u64*>(tile + lane * 24);
      u64* q = reinterpret_cast<u64*>(tile + (lane + kWave) * 24);
      r[0] = f[0]; r[1] = f[1]; r[2] = f[2];
      q[0] = f[3]; q[1] = f[4]; q[2] = f[5];
      wave_lds_fence();
      if (sub < nsub) {                                      // wave-uniform
        uint8_t* dst = recs + (size_t)sub * kTileBytes + 16 * lane;
        st16(dst, *reinterpret_cast<const u32x4*>(tile + 16 * lane));
        st16(dst + 1024, *reinterpret_cast<const u32x4*>(tile + 1024 + 16 * lane));
        st16(dst + 2048, *reinterpret_cast<const u32x4*>(tile + 2048 + 16 * lane));
      }
    }
    if (!more) break;
    t = tn;
#pragma unroll
    for (int k = 0; k < 2 * kExpandSub; ++k) a[k] = b[k];
  }
}
template <int W>
__global__ void ibu_k_sort_expand_tail(const ElemT<W>* __restrict__ in, u64 row0, u64 n, CompactPlan pl, u64* __restrict__ recs) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 f0, f1, f2;
  expand_elem<W>(ld_elem<W>(in + i), pl, f0, f1, f2);
  recs[3 * i] = f0; recs[3 * i + 1] = f1; recs[3 * i + 2] = f2;
}

template <int THREADS, int ROUNDS, int W>
struct CompactShape {
  static constexpr int T = THREADS * ROUNDS, NW = THREADS / kWave, PER_WAVE = T / NW;
  // LDS: stage 4 W T | gdelta 256 x u64 (u32 indices use the low halves' space: sized for the wider) | whist NW x 256 x u32 | misc 16 x u32 | sbin T bytes
  static constexpr size_t lds = 4 * (size_t)W * T + 8 * kBins + 4 * (size_t)NW * kBins + 64 + (size_t)T;
};
// One pass over element byte `byte`; nbyte: the next pass's byte (the digit side stream it leaves behind).
// LAST: the last pass — every element leaves as the 24-byte record it stands for, straight into the caller's array
// (`dst` = the records, `pl` = the expansion; no side stream): the expand kernel and one element round trip are saved.
// IDX: the type of a global element index — u32 below 2^32 elements, u64 from there on (the part holds 1.2e10 records).
template <int THREADS, int ROUNDS, bool LAST, int W, class IDX>
__global__ void __launch_bounds__(THREADS)
ibu_k_sort_scatter_elems(const ElemT<W>* __restrict__ src, void* __restrict__ dst_v, IDX n, u32 byte, u32 nbyte, const IDX* __restrict__ pos,
                         uint8_t* __restrict__ digits, CompactPlan pl) {
  typedef CompactShape<THREADS, ROUNDS, W> S;
  constexpr int T = S::T, NW = S::NW, PER_WAVE = S::PER_WAVE;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32* stage = reinterpret_cast<u32*>(smem);                 // the tile in digit order
  IDX* gdelta = reinterpret_cast<IDX*>(stage + W * T);       // global element index of slot p of bin d = gdelta[d] + p (W T words: 8-byte aligned)
  u32* whist = reinterpret_cast<u32*>(reinterpret_cast<u64*>(stage + W * T) + kBins);
  u32* misc = whist + NW * kBins;
  uint8_t* sbin = reinterpret_cast<uint8_t*>(misc + 16);
  const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wib = tid >> 6;
  const u64 lt_mask = (1ull << lane) - 1;
  const u32 ntiles = (u32)(((u64)n + T - 1) / T);
  struct Win { EV<W> v[ROUNDS]; IDX mypos; };
  // 1. every lane loads its elements (unconditional, clamped) and this tile's first output position per bin
  auto load = [&](u32 tile, Win& w) {
    const IDX tbase = (IDX)((u64)tile * T);
    const u32 cnt = n - tbase < (IDX)T ? (u32)(n - tbase) : (u32)T;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const u32 slot = wib * PER_WAVE + r * kWave + lane;
      w.v[r] = ld_elem<W>(src + tbase + (slot < cnt ? slot : cnt - 1));
    }
    w.mypos = pos[(size_t)tile * kBins + (tid & (kBins - 1))];
  };
  auto body = [&](u32 tile, const Win& w) {
  const EV<W>* v = w.v;
  const IDX mypos = w.mypos;
  const IDX tbase = (IDX)((u64)tile * T);
  const u32 cnt = n - tbase < (IDX)T ? (u32)(n - tbase) : (u32)T;
#pragma unroll
  for (int k = 0; k < kBins / kWave; ++k) whist[wib * kBins + lane + kWave * k] = 0;
  wave_lds_fence();                                          // a wave's counters are its own

  // 2. rank every element among the elements of its wave with the same digit (stable: slot order)
  u32 dig[ROUNDS], rk[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const u32 slot = wib * PER_WAVE + r * kWave + lane;
    const bool valid = slot < cnt;
    const u32 d = elem_byte<W>(v[r], byte);
    u64 m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const u64 bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const u32 before = (u32)__popcll(m & lt_mask);
    const u32 prev = valid ? whist[wib * kBins + d] : 0;
    wave_lds_fence();                                        // every lane has read before the leaders write
    if (valid && before == 0) whist[wib * kBins + d] = prev + (u32)__popcll(m);
    wave_lds_fence();
    dig[r] = d;
    rk[r] = prev + before;
  }
  __syncthreads();

  // 3. bin totals of the tile -> slot bases per (wave, bin)
  {
    u32 c[NW], tot = 0;
    if (tid < (u32)kBins) {
#pragma unroll
      for (int w = 0; w < NW; ++w) { c[w] = whist[w * kBins + tid]; tot += c[w]; }
    }
    u32 all;
    const u32 tb = block_exclusive_scan(tid < (u32)kBins ? tot : 0u, misc, &all);
    if (tid < (u32)kBins) {
      u32 run = tb;
#pragma unroll
      for (int w = 0; w < NW; ++w) { whist[w * kBins + tid] = run; run += c[w]; }
      gdelta[tid] = mypos - tb;                               // wraps harmlessly: slot >= tb for this bin
    }
  }
  __syncthreads();

  // 4. permute into digit order inside LDS
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const u32 slot = wib * PER_WAVE + r * kWave + lane;
    if (slot < cnt) {
      const u32 p = whist[wib * kBins + dig[r]] + rk[r];
#pragma unroll
      for (int w = 0; w < W; ++w) stage[W * p + w] = v[r].w[w];
      sbin[p] = (uint8_t)dig[r];
    }
  }
  __syncthreads();

  // 5. write out: lane = element, consecutive lanes write the consecutive elements of a run (dwordx3 each; plain stores:
  //    the L2 merges the pieces of a run that neighbouring tiles write); 6. the next pass's digit at the new position
  if constexpr (LAST) {
    // The last pass writes 24-byte records.  One lane per HALF record (12 bytes = dwords [3j, 3j+3) of the record, j = lane
    // parity): consecutive lanes write consecutive 12-byte pieces, so a wave's store instruction covers 768 contiguous
    // bytes of a run — the store shape of the element passes, which run at the box's copy rate.  (One lane per record
    // = three 8-byte stores at a 24-byte stride: every instruction touches twelve 128-byte lines for a third of their
    // bytes, three times; measured 8.4-9.0 ms per 1e9 records against 4.8 ms for an element pass of 2/3 the bytes.)
    const u32 j = tid & 1u;
    u32 hsel[3][2], hbase[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      hsel[k][0] = j ? pl.xsel[3 + k][0] : pl.xsel[k][0];
      hsel[k][1] = j ? pl.xsel[3 + k][1] : pl.xsel[k][1];
      const u64 bf = j ? pl.base[(3 + k) >> 1] : pl.base[k >> 1];
      hbase[k] = ((3 * (j ? 1 : 0) + k) & 1) ? (u32)(bf >> 32) : (u32)bf;
    }
#pragma unroll
    for (int r = 0; r < 2 * ROUNDS; ++r) {
      const u32 p = (tid + THREADS * r) >> 1;               // element slot; lanes 2q, 2q+1 share it
      if (p < cnt) {
        const IDX g = gdelta[sbin[p]] + p;
        u32 e[4] = {stage[W * p], stage[W * p + 1], stage[W * p + 2], 0};
        if constexpr (W == 4) e[3] = stage[W * p + 3];
        u32x3 o;
        o.x = hbase[0] | __builtin_amdgcn_perm(e[1], e[0], hsel[0][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[0][1]);
        o.y = hbase[1] | __builtin_amdgcn_perm(e[1], e[0], hsel[1][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[1][1]);
        o.z = hbase[2] | __builtin_amdgcn_perm(e[1], e[0], hsel[2][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[2][1]);
        *reinterpret_cast<u32x3_a4*>(static_cast<uint8_t*>(dst_v) + 24 * (size_t)g + 12 * j) = o;
      }
    }
  } else {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const u32 p = tid + THREADS * r;
    if (p < cnt) {
      const IDX g = gdelta[sbin[p]] + p;
      EV<W> e;
#pragma unroll
      for (int w = 0; w < W; ++w) e.w[w] = stage[W * p + w];
      st_elem<W>(static_cast<ElemT<W>*>(dst_v) + g, e);
      if (nbyte < 4 * W) digits[g] = (uint8_t)elem_byte<W>(e, nbyte);   // uniform; >= 4 W: no pass follows on elements
    }
  }
  }
  };   // body
  // XCD-aware tile order (speed only: see ibu_k_sort_scatter): XCD x = blockIdx % 8 takes the consecutive tiles
  // [x tpp, (x + 1) tpp), its workgroups one after the other.
  const u32 nb = gridDim.x >> 3, tpp = (ntiles + 7u) >> 3;    // the grid is a multiple of 8
  const u32 x0 = (blockIdx.x & 7u) * tpp, xend = x0 + tpp < ntiles ? x0 + tpp : ntiles;
  u32 tile = x0 + (blockIdx.x >> 3);
  if (tile >= xend) return;                                   // block-uniform
  // One tile per workgroup: the grid covers them (nb == tpp).  A persistent form with the next tile's elements prefetched into a
  // second register set was built in round 3 and is not usable at this shape: 174 -> 297 VGPRs, one wave per SIMD.
  (void)nb;
  Win w;
  load(tile, w);
  body(tile, w);
}

// =====================================================================================================
// PREFIX + FINISH (round 3): wide keys.  LSD over all varying bytes costs a pass per byte — 24 passes of ~50 B/record for
// full-range (32,32) records.  But once the records are sorted by their most significant P varying bytes (P LSD passes,
// least significant of the P first), everything that is left to decide lies INSIDE runs of equal prefix ("segments"),
// and for P = ceil(log256(n / 8)) a segment of well-spread keys holds a handful of records.  ibu_k_sort_finish completes the
// sort in ONE more pass: a workgroup takes the segments that START in its tile of T records (from the first segment head
// in the tile to the first head at or behind the tile's end — up to M records of look-ahead), stages them in LDS, ranks
// every record inside its segment by counting the records of the segment that order before it under the full 24-byte key
// (quadratic in the segment length, which is why segments longer than M are refused), permutes in LDS and writes the
// chunk out as consecutive 8-byte words.  4 passes + 1 instead of 24 at 1e9 records.
//   Keys that are NOT well spread (a few heavy prefixes) make long segments: the kernel then raises the overflow flag and the
// host falls back to the full LSD passes (the prefix-sorted records are a permutation of the input; records with equal
// keys are equal byte for byte, so nothing is lost but the time of the P passes).
// =====================================================================================================
#ifndef IBU_FINISH24_T
#define IBU_FINISH24_T 1024
#endif
#ifndef IBU_FINISH24_M
#define IBU_FINISH24_M 256
#endif
// 1024-record tiles + 256 of look-ahead: 37 KiB of LDS, four workgroups per CU.  1e9 full-range (32,32) records (profiles r03_o):
// (2048, 512) 17.2 ms, (1024, 512) 15.7, (1536, 256) 12.8, (1024, 256) 12.4.
static constexpr int kFinishT = IBU_FINISH24_T, kFinishM = IBU_FINISH24_M;
template <int T, int M>
struct FinishShape {
  static constexpr int L = T + M;                             // records staged per workgroup (+ 1 in front)
  // LDS: stage 24 (L + 1) | head u8 [L + 1] (padded) | segstart u16 [L] | seglen u16 [L] | misc 16 x u32
  static constexpr size_t lds = 24 * (size_t)(L + 1) + ((L + 1 + 15) & ~15) + 2 * (size_t)L + 2 * (size_t)L + 64;
};
// PERSIST: persistent grid, the next tile's window prefetched into a second register set while this one is worked on (needs
// 16-byte aligned records; the one-tile form takes any 8-byte aligned input: a shard at an odd record).
template <int T, int M, bool PERSIST>
__global__ void __launch_bounds__(kSortThreads, 4)   // 37 KiB of LDS: four workgroups per CU, if the registers allow (128 VGPRs)
ibu_k_sort_finish(const u64* __restrict__ src, u64* __restrict__ dst, u64 n, u64 pm0, u64 pm1, u64 pm2, u32* __restrict__ overflow) {
  typedef FinishShape<T, M> S;
  constexpr int L = S::L, PER = (L + kSortThreads - 1) / kSortThreads, CH = (3 * L / 2 + kSortThreads - 1) / kSortThreads;
  static_assert((T * 24) % 16 == 0, "tiles must start at 16-byte boundaries of an aligned array");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u64* stage = reinterpret_cast<u64*>(smem) + 3;             // record i of the window at stage[3 i]; record -1 = the one in front
  uint8_t* head = reinterpret_cast<uint8_t*>(stage + 3 * L);  // head[i]: record i starts a segment
  uint16_t* segstart = reinterpret_cast<uint16_t*>(head + ((L + 1 + 15) & ~15));
  uint16_t* seglen = segstart + L;
  u32* misc = reinterpret_cast<u32*>(seglen + L);             // [0] first head in the tile, [1] first head at / behind T, [2] too long
  const u32 tid = threadIdx.x, lane = tid & (kWave - 1);
  const u64 ntiles = (n + T - 1) / T;
  u64 tile = blockIdx.x;
  if (tile >= ntiles) return;
  struct Win { u32x4 v[CH]; u64 front; };
  // a window's loads, all issued before anything waits for them (unconditional, clamped)
  auto load = [&](u64 t, Win& w) {
    const u64 base = t * T;
    const u32 len = n - base < (u64)L ? (u32)(n - base) : (u32)L;
    const u32x4* g4 = reinterpret_cast<const u32x4*>(src + 3 * base);
    const u32 nch = (3 * len) >> 1;                           // 16-byte chunks of the window
#pragma unroll
    for (int r = 0; r < CH; ++r) {
      const u32 c = tid + kSortThreads * r;
      w.v[r] = ld16(g4 + (c < nch ? c : (nch ? nch - 1 : 0)));
    }
    w.front = src[base > 0 ? 3 * base - 3 + (tid < 3 ? tid : 0) : 0];   // threads 0..2: the record in front of the window
  };
  auto work = [&](u64 t, const Win* w) {                      // w == nullptr: stage straight from memory (one-tile form)
    const u64 base = t * T;
    const u32 len = n - base < (u64)L ? (u32)(n - base) : (u32)L;   // records of the window that exist
    const u64* g = src + 3 * base;
    // 1. stage the window (and the record in front of it)
    if (w) {
      const u32 nch = (3 * len) >> 1;
#pragma unroll
      for (int r = 0; r < CH; ++r) {
        const u32 c = tid + kSortThreads * r;
        if (c < nch) {                                        // stage is 8 (mod 16)-aligned: two halves
          stage[2 * c] = ((u64)w->v[r].y << 32) | w->v[r].x;
          stage[2 * c + 1] = ((u64)w->v[r].w << 32) | w->v[r].z;
        }
      }
      if (tid == 0 && ((3 * len) & 1u)) stage[3 * len - 1] = g[3 * len - 1];
      if (tid < 3) stage[(int)tid - 3] = base > 0 ? w->front : 0;
    } else {
      for (u32 k = tid; k < 3 * len; k += kSortThreads) stage[k] = g[k];
      if (tid < 3) stage[(int)tid - 3] = base > 0 ? g[(int)tid - 3] : 0;
    }
    if (tid < 3) misc[tid] = tid == 2 ? 0u : 0xFFFFFFFFu;
    __syncthreads();
    // 2. segment heads: the prefix differs from the predecessor's (row 0 of the array is a head).  With short runs nearly every
    //    record is one: the first head of a wave's 64 goes to the LDS word, not 64 same-address atomics.
    //    misc[0]: first head among the tile's first M records, misc[1]: first head in the look-ahead [T, T + M).
    for (u32 i0 = tid - lane; i0 < len; i0 += kSortThreads) {
      const u32 i = i0 + lane;
      bool h = false;
      if (i < len) {
        const u64* r = stage + 3 * i;
        h = (base + i == 0) || (((r[0] ^ r[-3]) & pm0) | ((r[1] ^ r[-2]) & pm1) | ((r[2] ^ r[-1]) & pm2)) != 0;
        head[i] = h;
      }
      const u64 lo = __ballot(h && i < (u32)M), hi = __ballot(h && i >= (u32)T);
      if (lane == 0) {
        if (lo) atomicMin(&misc[0], i0 + (u32)__builtin_ctzll(lo));
        if (hi) atomicMin(&misc[1], i0 + (u32)__builtin_ctzll(hi));
      }
    }
    __syncthreads();
    // WHO WRITES WHAT: see ibu_k_sort_finish_elems (the same ownership rule: [begin, end) from the first heads among the first M
    // records of this tile and of the next; runs of at most M records between two heads are ranked, everything else is part of a
    // long run, passed through as it stands and checked for order).
    // (The array's end closes a run like a head does: a last tile of at most M elements without a head is all tail of the
    // previous tile's last run — the previous tile, whose window then reaches the array's end, finishes it.)
    const u32 begin = misc[0] != 0xFFFFFFFFu ? misc[0] : ((len <= (u32)M && base + len == n) ? len : 0u);
    u32 end;
    bool end_is_head = true;
    if (len <= (u32)T) end = len;                             // the array ends in this tile
    else if (misc[1] != 0xFFFFFFFFu) end = misc[1];
    else if (base + len == n) end = len;                      // ... or inside the look-ahead
    else { end = (u32)T; end_is_head = false; }
    // 3. short runs: every head walks to the next one; segstart for the members, seglen at the head
    for (u32 i = begin + tid; i < end; i += kSortThreads)
      if (head[i] || i == begin) {                            // (begin without a head: the part of a long run this tile owns)
        u32 j = i + 1;
        while (j < end && !head[j]) ++j;
        if (head[i] && j - i <= (u32)M && (j < end || end_is_head)) {
          for (u32 k = i; k < j; ++k) segstart[k] = (uint16_t)i;
          seglen[i] = (uint16_t)(j - i);
        } else {
          for (u32 k = i; k < j; ++k) segstart[k] = 0xFFFFu;   // part of a long run
        }
      }
    __syncthreads();
    // 4. rank inside the short runs under the full key (ties: window order — equal keys are equal records); long runs: identity
    //    + order check
    u64 k0[PER], k1[PER], k2[PER];
    u32 target[PER];
    bool inversion = false;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const u32 i = begin + tid + kSortThreads * r;
      target[r] = 0xFFFFFFFFu;
      if (i < end) {
        const u64* me = stage + 3 * i;
        k0[r] = me[0]; k1[r] = me[1]; k2[r] = me[2];
        const u32 s0 = segstart[i];
        if (s0 == 0xFFFFu) {                                  // part of a long run
          target[r] = i;
          if (!head[i]) {
            const u32 lt0 = k0[r] < me[-3], eq0 = k0[r] == me[-3], lt1 = k1[r] < me[-2], eq1 = k1[r] == me[-2], lt2 = k2[r] < me[-1];
            inversion = inversion || (lt0 | (eq0 & (lt1 | (eq1 & lt2)))) != 0;
          }
        } else {
          const u32 s1 = s0 + seglen[s0];
          u32 cnt = 0;
          // four candidates per step, their LDS reads issued together (the lanes of a segment read the same record: broadcasts), and
          // the comparison as mask arithmetic — the short-circuit form compiled to five branches per candidate and one LDS round
          // trip per iteration: 159 ms per 1e9 records instead of ~15.  A segment of one record costs nothing.
          if (s1 - s0 > 1)
            for (u32 j = s0; j < s1; j += 4) {
              u64 cb[4], cu[4], cx[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const u32 jj = j + q < s1 ? j + q : s1 - 1;  // clamped: in the window, not counted
                const u64* o = stage + 3 * jj;
                cb[q] = o[0]; cu[q] = o[1]; cx[q] = o[2];
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const u32 lt0 = cb[q] < k0[r], eq0 = cb[q] == k0[r], lt1 = cu[q] < k1[r], eq1 = cu[q] == k1[r], lt2 = cx[q] < k2[r], eq2 = cx[q] == k2[r];
                const u32 before = lt0 | (eq0 & (lt1 | (eq1 & (lt2 | (eq2 & (u32)(j + q < i))))));   // orders before me (ties: window order)
                cnt += before & (u32)(j + q < s1);
              }
            }
          target[r] = s0 + cnt;
        }
      }
    }
    if (inversion) misc[2] = 1u;
    __syncthreads();                                          // every record is in registers: permute in place
    if (misc[2]) {                                            // a long run that is not in order: not this kernel's to sort
      if (tid == 0) *overflow = 1u;
      return;
    }
#pragma unroll
    for (int r = 0; r < PER; ++r)
      if (target[r] != 0xFFFFFFFFu) {
        u64* o = stage + 3 * target[r];
        o[0] = k0[r]; o[1] = k1[r]; o[2] = k2[r];
      }
    __syncthreads();
    // 5. the chunk [begin, end) leaves as half records (dwordx3): consecutive lanes, consecutive 12-byte pieces
    uint8_t* out = reinterpret_cast<uint8_t*>(dst + 3 * (base + begin));
    const u32* in = reinterpret_cast<const u32*>(stage + 3 * begin);
    for (u32 h = tid; h < 2 * (end - begin); h += kSortThreads) {
      u32x3 o;
      o.x = in[3 * h]; o.y = in[3 * h + 1]; o.z = in[3 * h + 2];
      *reinterpret_cast<u32x3_a4*>(out + 12 * (size_t)h) = o;
    }
  };
  if constexpr (!PERSIST) {
    work(tile, nullptr);                                      // one tile per workgroup (the grid covers them)
  } else {
    Win wa, wb;
    load(tile, wa);
    for (;;) {                                                // two register sets take turns (kcommon.hpp, sweep_tiles)
      u64 next = tile + gridDim.x;
      bool more = next < ntiles;
      load(more ? next : tile, wb);
      work(tile, &wa);
      if (!more) break;
      tile = next;
      __syncthreads();                                        // step 5's LDS reads precede the next tile's stage writes
      next = tile + gridDim.x;
      more = next < ntiles;
      load(more ? next : tile, wa);
      work(tile, &wb);
      if (!more) break;
      tile = next;
      __syncthreads();
    }
  }
}

// ---- the same on compact elements (W words): P element passes, then this kernel ranks inside the runs of equal prefix, and
// every element leaves as the 24-byte record it stands for (the chunk is contiguous in the output: one lane per half record,
// dwordx3, fully coalesced).  Elements compare as W-word little-endian integers, which is the record order (COMPACT-KEY
// passes); index bytes that the passes do not sort on (input in index order) take part in the comparison here — the same
// result, because the passes are stable and the input's index order is the element order on those bytes.
template <int W, int T, int M>
struct FinishElemShape {
  static constexpr int L = T + M;
  static constexpr size_t lds = 4 * (size_t)W * (L + 1) + ((L + 1 + 15) & ~15) + 2 * (size_t)L + 2 * (size_t)L + 64;
};
template <int W>
__device__ __forceinline__ u32 elem_before(const u32* a, const u32* b, u32 tie) {   // a orders before b (W-word integers; tie: what equal elements answer)
  u32 r = tie;
#pragma unroll
  for (int w = 0; w < W; ++w) r = (u32)(a[w] < b[w]) | ((u32)(a[w] == b[w]) & r);   // from the least significant word up
  return r;
}
// (12-byte elements: four workgroups per CU fit the LDS, so the registers must too — 128 VGPRs; the kernel sat at 128 when the
// tile shape was chosen and drifted to 135 with later edits, which silently cost a workgroup per CU: 7.8 -> 9.9 ms.)
template <int W, int T, int M>
__global__ void __launch_bounds__(kSortThreads, W == 3 ? 4 : 3)
ibu_k_sort_finish_elems(const ElemT<W>* __restrict__ src, void* __restrict__ dst_v, u64 n, EV<W> pm, CompactPlan pl, u32* __restrict__ overflow) {
  typedef FinishElemShape<W, T, M> S;
  constexpr int L = S::L, PER = (L + kSortThreads - 1) / kSortThreads;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32* stage = reinterpret_cast<u32*>(smem) + W;             // element i of the window at stage[W i]; element -1 = the one in front
  uint8_t* head = reinterpret_cast<uint8_t*>(stage + W * L);
  uint16_t* segstart = reinterpret_cast<uint16_t*>(head + ((L + 1 + 15) & ~15));
  uint16_t* seglen = segstart + L;
  u32* misc = reinterpret_cast<u32*>(seglen + L);
  const u32 tid = threadIdx.x, lane = tid & (kWave - 1);
  const u32 ntiles = (u32)((n + T - 1) / T);
  // which tiles this workgroup sweeps: b, b + grid, ...  (Every XCD owning one contiguous eighth of the tiles — so that the
  // boundary lines two workgroups write meet in one XCD's L2 — measured no different: the ranges are written whole lines.)
  u32 tile = blockIdx.x;
  const u32 tstride = gridDim.x, tend = ntiles;
  if (tile >= tend) return;
  // A window's loads: one element per lane and step (dwordx3 / dwordx4, consecutive lanes on consecutive elements), ALL issued
  // before anything waits for them (unconditional, clamped) — and the NEXT tile's window is loaded while this one is worked on
  // (persistent grid, two register sets).  As a load-then-store loop in a one-tile workgroup the kernel paid 18 memory
  // latencies per tile: 18.8 ms per 1e9 records; loads issued together 14.5 ms; prefetched as here: see profiles/README.md.
  auto load = [&](u32 t, EV<W>* v, EV<W>& front) {
    const u64 base = (u64)t * T;
    const u32 len = n - base < (u64)L ? (u32)(n - base) : (u32)L;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const u32 i = tid + kSortThreads * r;
      v[r] = ld_elem<W>(src + base + (i < len ? i : len - 1));
    }
    front = ld_elem<W>(src + (base > 0 ? base - 1 : 0));      // every lane the same element (one line); used by thread 0
  };
  auto work = [&](u32 t, const EV<W>* v, const EV<W>& front) {
    const u64 base = (u64)t * T;
    const u32 len = n - base < (u64)L ? (u32)(n - base) : (u32)L;
    // 1. stage
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const u32 i = tid + kSortThreads * r;
      if (i < len) {
#pragma unroll
        for (int w = 0; w < W; ++w) stage[W * i + w] = v[r].w[w];
      }
    }
    if (tid == 0) {
#pragma unroll
      for (int w = 0; w < W; ++w) stage[w - W] = base > 0 ? front.w[w] : 0u;
    }
    if (tid < 3) misc[tid] = tid == 2 ? 0u : 0xFFFFFFFFu;
    __syncthreads();
    // 2. heads (with short runs nearly every element is one: the first head of a wave's 64 goes to the LDS word, not 64 atomics).
    //    misc[0]: first head among the tile's first M elements, misc[1]: first head in the look-ahead [T, T + M).
    for (u32 i0 = tid - lane; i0 < len; i0 += kSortThreads) {
      const u32 i = i0 + lane;
      bool h = false;
      if (i < len) {
        u32 diff = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) diff |= (stage[W * i + w] ^ stage[W * i + w - W]) & pm.w[w];
        h = (base + i == 0) || diff != 0;
        head[i] = h;
      }
      const u64 lo = __ballot(h && i < (u32)M), hi = __ballot(h && i >= (u32)T);
      if (lane == 0) {
        if (lo) atomicMin(&misc[0], i0 + (u32)__builtin_ctzll(lo));
        if (hi) atomicMin(&misc[1], i0 + (u32)__builtin_ctzll(hi));
      }
    }
    __syncthreads();
    // WHO WRITES WHAT.  A workgroup owns [begin, end) of its window: begin = the first head among the tile's first M elements (the
    // elements in front of it are the tail of a run the previous tile finishes), or 0 if there is none (then the run that crosses
    // the tile's start is longer than M: nobody ranks it, every tile passes its own part through); end = likewise at the next tile's
    // start, seen through the look-ahead.  Both neighbours look at the same M elements, so the ranges tile the array.
    // Inside the range a run of at most M elements between two heads is RANKED; everything else is part of a long run and is
    // passed through as it stands, provided it is in order already — which is what a stable sort leaves when the input was (equal
    // (barcode, umi) groups of read-order input keep their index order) — and checked: one inversion raises the overflow flag.
    // (The array's end closes a run like a head does: a last tile of at most M elements without a head is all tail of the
    // previous tile's last run — the previous tile, whose window then reaches the array's end, finishes it.)
    const u32 begin = misc[0] != 0xFFFFFFFFu ? misc[0] : ((len <= (u32)M && base + len == n) ? len : 0u);
    u32 end;
    bool end_is_head = true;
    if (len <= (u32)T) end = len;                             // the array ends in this tile
    else if (misc[1] != 0xFFFFFFFFu) end = misc[1];
    else if (base + len == n) end = len;                      // ... or inside the look-ahead
    else { end = (u32)T; end_is_head = false; }
    // 3. short runs: every head walks to the next one; segstart for the members, seglen at the head
    for (u32 i = begin + tid; i < end; i += kSortThreads)
      if (head[i] || i == begin) {                            // (begin without a head: the part of a long run this tile owns)
        u32 j = i + 1;
        while (j < end && !head[j]) ++j;
        if (head[i] && j - i <= (u32)M && (j < end || end_is_head)) {   // closed by heads (or the array's end) and short enough
          for (u32 k = i; k < j; ++k) segstart[k] = (uint16_t)i;
          seglen[i] = (uint16_t)(j - i);
        } else {
          for (u32 k = i; k < j; ++k) segstart[k] = 0xFFFFu;   // part of a long run
        }
      }
    __syncthreads();
    // 4. rank inside the short runs (a run of one record — the usual case — costs nothing); long runs: identity + order check
    u32 me[PER][W];
    u32 target[PER];
    bool inversion = false;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const u32 i = begin + tid + kSortThreads * r;
      target[r] = 0xFFFFFFFFu;
      if (i < end) {
#pragma unroll
        for (int w = 0; w < W; ++w) me[r][w] = stage[W * i + w];
        const u32 s0 = segstart[i];
        if (s0 == 0xFFFFu) {                                  // part of a long run
          target[r] = i;
          if (!head[i]) {                                     // same run as the element in front (i = 0: the one in front of the window)
            u32 prev[W];
#pragma unroll
            for (int w = 0; w < W; ++w) prev[w] = stage[W * i + w - W];
            inversion = inversion || elem_before<W>(me[r], prev, 0u);
          }
        } else {
          const u32 m = seglen[s0];
          u32 cnt = 0;
          if (m > 1)
            for (u32 j = s0; j < s0 + m; j += 2) {
              const u32 j1 = j + 1 < s0 + m ? j + 1 : j;       // clamped: in the window, not counted
              u32 a[W], b[W];
#pragma unroll
              for (int w = 0; w < W; ++w) { a[w] = stage[W * j + w]; b[w] = stage[W * j1 + w]; }
              cnt += elem_before<W>(a, me[r], (u32)(j < i));
              cnt += elem_before<W>(b, me[r], (u32)(j1 < i)) & (u32)(j + 1 < s0 + m);
            }
          target[r] = s0 + cnt;
        }
      }
    }
    if (inversion) misc[2] = 1u;
    __syncthreads();
    if (misc[2]) {                                            // a long run that is not in order: not this kernel's to sort
      if (tid == 0) *overflow = 1u;
      return;
    }
#pragma unroll
    for (int r = 0; r < PER; ++r)
      if (target[r] != 0xFFFFFFFFu) {
#pragma unroll
        for (int w = 0; w < W; ++w) stage[W * target[r] + w] = me[r][w];
      }
    __syncthreads();
    // 5. the chunk [begin, end) leaves as records: one lane per half record (ibu_k_sort_scatter_elems' last-pass write-out).
    //    The lane's half (its parity; kSortThreads is even) is selected HERE, per tile: nine registers that would otherwise live
    //    across the whole loop are what stands between this kernel and its fourth workgroup per CU.
    const u32 hj = tid & 1u;
    u32 hsel[3][2], hbase[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      hsel[k][0] = hj ? pl.xsel[3 + k][0] : pl.xsel[k][0];
      hsel[k][1] = hj ? pl.xsel[3 + k][1] : pl.xsel[k][1];
      const u64 bf = hj ? pl.base[(3 + k) >> 1] : pl.base[k >> 1];
      hbase[k] = ((3 * (hj ? 1 : 0) + k) & 1) ? (u32)(bf >> 32) : (u32)bf;
    }
    uint8_t* out = static_cast<uint8_t*>(dst_v) + 24 * (size_t)base;
    for (u32 h = 2 * begin + tid; h < 2 * end; h += kSortThreads) {   // kSortThreads is even: a lane keeps its half
      const u32 p = h >> 1;
      u32 e[4] = {stage[W * p], stage[W * p + 1], stage[W * p + 2], 0};
      if constexpr (W == 4) e[3] = stage[W * p + 3];
      u32x3 o;
      o.x = hbase[0] | __builtin_amdgcn_perm(e[1], e[0], hsel[0][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[0][1]);
      o.y = hbase[1] | __builtin_amdgcn_perm(e[1], e[0], hsel[1][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[1][1]);
      o.z = hbase[2] | __builtin_amdgcn_perm(e[1], e[0], hsel[2][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[2][1]);
      *reinterpret_cast<u32x3_a4*>(out + 24 * (size_t)p + 12 * hj) = o;
    }
  };
  EV<W> va[PER], vb[PER], fa, fb;
  load(tile, va, fa);
  for (;;) {                                                  // two register sets take turns (kcommon.hpp, sweep_tiles)
    u32 next = tile + tstride;
    bool more = next < tend;
    load(more ? next : tile, vb, fb);
    work(tile, va, fa);
    if (!more) break;
    tile = next;
    __syncthreads();                                          // step 5's LDS reads precede the next tile's stage writes
    next = tile + tstride;
    more = next < tend;
    load(more ? next : tile, va, fa);
    work(tile, vb, fb);
    if (!more) break;
    tile = next;
    __syncthreads();
  }
}

// How long are the runs of equal prefix going to be?  Estimated BEFORE the path is chosen, from the sample ranges the
// speculative census reads anyway: every sample record is compressed on the fly, and for each candidate prefix length
// P = 1 .. kMaxPrefix its top P element bytes are inserted into an exact (64-bit hashed, open addressing) table; the
// number of PAIRS of sample records with equal prefix comes out per P.  With m sample records out of n, a record shares
// its prefix with about 1 + (n / m) * 2 pairs / m records of the whole input — for well-spread keys that is 1 + n / 256^P,
// for keys with few distinct prefixes (barcodes from a whitelist) it is large, and the sort then takes a longer prefix
// or the plain passes.  The pair count is a MEAN; a single heavy prefix (one barcode holding 0.1 % of the records) barely
// moves it and still makes runs far longer than the finishing kernel accepts — so the most frequent prefix of the sample is
// reported too (pairs[kMaxPrefix + P - 1]): four or more sample records with one prefix mean a run of tens of thousands.  (The samples are contiguous ranges: grouped input over-estimates, which errs on the safe side;
// an under-estimate is caught by the finishing kernel's overflow flag.)
static constexpr int kMaxPrefix = 8;
static constexpr u32 kPairSlotsMax = 1u << 18;                // per P: 98 304 sample records -> load factor 0.375
template <int W>
__global__ void ibu_k_sort_sample_pairs(const u64* __restrict__ recs, u64 range_stride, u32 nranges, u32 per_range, CompactPlan pl, u32 k,
                                        u32 first /*table q holds the prefixes of first + q + 1 bytes*/, u32 kPairSlots /*power of two*/, u64* __restrict__ keys /*[kMaxPrefix][slots]*/, u32* __restrict__ cnts,
                                        u64* __restrict__ pairs) {
  // the sample: `nranges` ranges of `per_range` consecutive records, evenly spaced over the input (range r starts at r * range_stride)
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nranges * per_range) return;
  const u32 rg = t / per_range;
  const u64 row = (u64)rg * range_stride + (t - rg * per_range);
  const EV<W> e = compress_rec<W>(recs[3 * row], recs[3 * row + 1], recs[3 * row + 2], pl);
  for (u32 q = 0; q < (u32)kMaxPrefix && first + q + 1 <= k; ++q) {
    const u32 P = first + q + 1;
    u64 h = 0x9E3779B97F4A7C15ull * P;                        // hash of element bytes [k - P, k)
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int lo = (int)(k - P) - 4 * w;                    // first prefix byte inside word w (may be <= 0: whole word, >= 4: none)
      const u32 mask = lo >= 4 ? 0u : lo <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << (8 * lo));
      h = (h ^ (u64)(e.w[w] & mask)) * 0xBF58476D1CE4E5B9ull;
      h ^= h >> 29;
    }
    h = (h ^ (h >> 32)) * 0x94D049BB133111EBull;
    h ^= h >> 31;
    if (h == 0) h = 1;
    u64* kt = keys + (size_t)q * kPairSlots;
    u32* ct = cnts + (size_t)q * kPairSlots;
    for (u32 slot = (u32)h & (kPairSlots - 1), probes = 0; probes < kPairSlots; slot = (slot + 1) & (kPairSlots - 1), ++probes) {
      const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&kt[slot]), 0ull, (unsigned long long)h);
      if (old == 0 || old == h) {
        const u32 before = atomicAdd(&ct[slot], 1u);          // records with this prefix seen so far: that many new pairs
        if (before) {
          atomicAdd(reinterpret_cast<unsigned long long*>(&pairs[q]), (unsigned long long)before);
          if (before >= 3) atomicMax(reinterpret_cast<unsigned long long*>(&pairs[kMaxPrefix + q]), (unsigned long long)(before + 1));   // the most frequent prefix, from four sample records on (three of 98 304 happen by chance)
        }
        break;
      }
    }
  }
}

// The same estimate for 24-byte records (more than 16 varying bytes): the prefix of length P is the P most significant VARYING
// key bytes, given as (field, shift) pairs, most significant first.
struct PrefixBytes { uint8_t field[24], shift[24]; u32 count, first; };   // `count` bytes listed; table q holds the prefixes of first + q + 1 bytes
extern "C" __global__ void ibu_k_sort_sample_pairs_recs(const u64* __restrict__ recs, u64 range_stride, u32 nranges, u32 per_range, PrefixBytes pb,
                                                        u32 kPairSlots /*power of two*/, u64* __restrict__ keys, u32* __restrict__ cnts,
                                                        u64* __restrict__ pairs) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nranges * per_range) return;
  const u32 rg = t / per_range;
  const u64 row = (u64)rg * range_stride + (t - rg * per_range);
  const u64 f[3] = {recs[3 * row], recs[3 * row + 1], recs[3 * row + 2]};
  u64 h = 0x9E3779B97F4A7C15ull;
  for (u32 P = 1; P <= pb.count; ++P) {                       // the hash of the first P bytes extends the hash of the first P - 1
    const u32 fi = pb.field[P - 1];
    const u64 byte = ((fi == 0 ? f[0] : fi == 1 ? f[1] : f[2]) >> pb.shift[P - 1]) & 255u;
    h = (h ^ (byte + 0x100ull * P)) * 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    if (P <= pb.first) continue;                               // hashed, not counted: an earlier window's prefixes
    const u32 q = P - pb.first - 1;
    u64 key = (h ^ (h >> 32)) * 0x94D049BB133111EBull;
    key ^= key >> 31;
    if (key == 0) key = 1;
    u64* kt = keys + (size_t)q * kPairSlots;
    u32* ct = cnts + (size_t)q * kPairSlots;
    for (u32 slot = (u32)key & (kPairSlots - 1), probes = 0; probes < kPairSlots; slot = (slot + 1) & (kPairSlots - 1), ++probes) {
      const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&kt[slot]), 0ull, (unsigned long long)key);
      if (old == 0 || old == key) {
        const u32 before = atomicAdd(&ct[slot], 1u);
        if (before) {
          atomicAdd(reinterpret_cast<unsigned long long*>(&pairs[q]), (unsigned long long)before);
          if (before >= 3) atomicMax(reinterpret_cast<unsigned long long*>(&pairs[kMaxPrefix + q]), (unsigned long long)(before + 1));
        }
        break;
      }
    }
  }
}

// =====================================================================================================
// Host side.  Scratch layout (bytes), all offsets 256-byte aligned:
//   census u64[64][8] | binbase u64[256] | blocksum u32[nblocks][256] | blockoff u64[nblocks][256] | counts u16[ntiles][256]
//   | pos IDX[ntiles][256] | digits u8[ntiles * T]
static constexpr size_t kMiscBytes = 256;                     // behind the census slots: [0] the finishing kernel's overflow flag (u32)
struct SortLayout {
  size_t misc, binbase, blocksum, blockoff, counts, pos, digits, total;
  u32 ntiles, nblocks;
  bool idx64;
};
static SortLayout sort_layout(const LaunchCfg& cfg, size_t n, int tile) {
  SortLayout L;
  const u64 nt = (n + tile - 1) / tile;
  L.ntiles = (u32)nt;
  L.nblocks = (u32)((nt + kTilesPerBlock - 1) / kTilesPerBlock);
  L.idx64 = n >= (1ull << 32) || cfg.sort_idx64;             // cfg.sort_idx64: a test knob (the 64-bit index kernels at small sizes)
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t o = kCensusBytes;                                   // the census slots sit in front
  L.misc = o; o = up(o + kMiscBytes);
  L.binbase = o; o = up(o + 8 * kBins);
  L.blocksum = o; o = up(o + 4 * (size_t)L.nblocks * kBins);
  L.blockoff = o; o = up(o + 8 * (size_t)L.nblocks * kBins);
  L.counts = o; o = up(o + 2 * (size_t)nt * kBins);
  L.pos = o; o = up(o + (L.idx64 ? 8 : 4) * (size_t)nt * kBins);
  L.digits = o; o = up(o + (size_t)nt * tile + 1024);
  L.total = o;
  return L;
}

struct SweepVariant {
  int threads, tile;
  size_t lds;
  const void* scatter32;
  const void* scatter64;
  void (*counts_recs)(const uint8_t*, u32, u32, u32, uint16_t*, uint8_t*);
  void (*counts_tail)(const u64*, u64, u32, u32, u32, uint16_t*, u64*);
  void (*counts_bytes)(const uint8_t*, u64, u32, uint16_t*);
};
#ifdef IBU_SORT_PROBE
#define IBU_WM 2
#else
#define IBU_WM 0
#endif
template <int TH, int R, int WM>
static SweepVariant sweep_variant() {
  typedef SweepShape<TH, R> S;
  return {TH, S::T, S::lds, reinterpret_cast<const void*>(ibu_k_sort_scatter<TH, R, u32, WM>),
          reinterpret_cast<const void*>(ibu_k_sort_scatter<TH, R, u64, WM>), ibu_k_sort_tilecounts_recs<S::T>,
          ibu_k_sort_tilecounts_recs_tail<S::T>, ibu_k_sort_tilecounts_bytes<S::T>};
}
// cfg.sort_variant: tile shape (A/B through ibu_ctx_set_option(ctx, "sort_variant", k))
static const SweepVariant kSweep[] = {
    sweep_variant<256, 10, 0>(),  // 0 (default since round 3): 2560-record tiles (70 KiB), 4 waves x 10 records per lane, two workgroups per CU — with the half-record write-out 4 % faster than 2048 (1e9 full-range (32,32): 0.0615 against 0.0642 s)
    sweep_variant<512, 4, 0>(),   // 1: 2048-record tiles, 8 waves
    sweep_variant<1024, 4, 0>(),  // 2: 4096-record tiles, one workgroup per CU
    sweep_variant<256, 4, 0>(),   // 3: 1024-record tiles
    sweep_variant<256, 6, 0>(),   // 4: 1536-record tiles, three workgroups per CU
    sweep_variant<256, 8, 0>(),   // 5: 2048-record tiles (the default of rounds 1-2)
    sweep_variant<256, 12, 0>(),  // 6: 3072-record tiles, one workgroup per CU
#ifdef IBU_SORT_PROBE
    sweep_variant<256, 8, 2>(), sweep_variant<512, 4, 2>(),  // 7, 8: linear write-out (probe builds only, WRONG output)
#endif
};
static constexpr int kNumSweep = sizeof(kSweep) / sizeof(kSweep[0]);
int sort_num_variants() { return kNumSweep; }
static const SweepVariant& pick_variant(const LaunchCfg& cfg) {
  return kSweep[cfg.sort_variant >= 0 && cfg.sort_variant < kNumSweep ? cfg.sort_variant : 0];
}


// compact-key passes: tile shapes (cfg.sort_compact = 1 + index; 0 = never take the compact path); times: the whole sort,
// 1e9 records 16/12 with a random 30-bit index (profiles/r02_aj_*)
struct CompactVariant {
  int threads, tile;
  size_t lds;
  const void* scatter;
  const void* scatter_last;
  void (*counts_bytes)(const uint8_t*, u64, u32, uint16_t*);
  const void* scatter64;        // the same kernels with 64-bit element indices (2^32 elements and more): the default shapes only
  const void* scatter_last64;
};
template <int TH, int R, int W, bool WIDE = false>
static CompactVariant compact_variant() {
  typedef CompactShape<TH, R, W> S;
  CompactVariant v = {TH, S::T, S::lds, reinterpret_cast<const void*>(ibu_k_sort_scatter_elems<TH, R, false, W, u32>),
                      reinterpret_cast<const void*>(ibu_k_sort_scatter_elems<TH, R, true, W, u32>), ibu_k_sort_tilecounts_bytes<S::T>, nullptr, nullptr};
  if constexpr (WIDE) {
    v.scatter64 = reinterpret_cast<const void*>(ibu_k_sort_scatter_elems<TH, R, false, W, u64>);
    v.scatter_last64 = reinterpret_cast<const void*>(ibu_k_sort_scatter_elems<TH, R, true, W, u64>);
  }
  return v;
}
static const CompactVariant kCompact[] = {   // 12-byte elements
    compact_variant<256, 20, 3, true>(),   // 1 (default): 5120-element tiles (60 KiB), 4 waves, two workgroups per CU (1e9 records: 73.7 ms)
    compact_variant<256, 16, 3>(),   // 2: 4096-element tiles (75.2 ms)
    compact_variant<256, 8, 3>(),    // 3: 2048-element tiles (94 ms: runs of 8 elements = 96 bytes)
    compact_variant<512, 8, 3>(),    // 4: 4096-element tiles, 8 waves (79-83 ms)
    compact_variant<512, 16, 3>(),   // 5: 8192-element tiles, 8 waves, one workgroup per CU (99 ms)
    compact_variant<1024, 8, 3>(),   // 6: 8192-element tiles, 16 waves (97 ms)
    compact_variant<1024, 4, 3>(),   // 7: 4096-element tiles, 16 waves (79-83 ms)
    compact_variant<256, 12, 3>(),   // 8: 3072-element tiles, three workgroups per CU (86 ms)
};
static constexpr int kNumCompact = sizeof(kCompact) / sizeof(kCompact[0]);
// 16-byte elements (13 .. 16 varying bytes); shape = kCompact16[sort_compact - 1] where there is one, the first otherwise
static const CompactVariant kCompact16s[] = {
    compact_variant<256, 16, 4, true>(),   // 4096-element tiles = 64 KiB of elements, two workgroups per CU
    compact_variant<256, 12, 4>(),   // 3072-element tiles
    compact_variant<512, 8, 4>(),    // 4096-element tiles, 8 waves
    compact_variant<256, 8, 4>(),    // 2048-element tiles, four workgroups per CU
};
static constexpr int kNumCompact16 = sizeof(kCompact16s) / sizeof(kCompact16s[0]);
static const CompactVariant& pick_compact16(const LaunchCfg& cfg) {
  return kCompact16s[cfg.sort_compact >= 1 && cfg.sort_compact <= kNumCompact16 ? cfg.sort_compact - 1 : 0];
}
int sort_num_compact_variants() { return kNumCompact; }
static const CompactVariant* pick_compact(const LaunchCfg& cfg) {
  return cfg.sort_compact >= 1 && cfg.sort_compact <= kNumCompact ? &kCompact[cfg.sort_compact - 1] : nullptr;
}

size_t sort_scratch_bytes(const LaunchCfg& cfg, size_t n) {
  size_t need = sort_layout(cfg, n, pick_variant(cfg).tile).total;
  if (const CompactVariant* cv = pick_compact(cfg)) {
    for (const CompactVariant* v : {cv, &kCompact16s[0], &kCompact16s[1], &kCompact16s[2], &kCompact16s[3]}) {
      const size_t c = sort_layout(cfg, n, v->tile).total;
      if (c > need) need = c;
    }
  }
  return need;
}

// The plan of a set of records from its OR / AND words (one rank's census, or the words of all ranks combined): element byte j
// = the j-th least significant varying byte of the key (index bytes first, barcode bytes last).  Selectors are filled for the
// first 16 varying bytes: k <= 12 fits 12-byte elements, k <= 16 the sort's 16-byte elements.
void compact_plan_init(const uint64_t or_words[3], const uint64_t and_words[3], CompactPlan* pl) {
  for (auto& row : pl->csel) for (uint32_t& v : row) v = 0x0C0C0C0Cu;   // selector 0x0C: a zero byte
  for (auto& row : pl->xsel) for (uint32_t& v : row) v = 0x0C0C0C0Cu;
  static const int kFieldLsbFirst[3] = {2, 1, 0};
  u32 k = 0;
  pl->index_bytes = 0;
  for (int fo = 0; fo < 3; ++fo) {
    const int f = kFieldLsbFirst[fo];
    const u64 varying = or_words[f] ^ and_words[f];
    pl->base[f] = and_words[f];
    for (u32 b = 0; b < 8; ++b)
      if ((varying >> (8 * b)) & 255u) {
        if (k < 16) {                                       // element byte k <- byte b of field f, and back
          uint32_t& cs = pl->csel[k >> 2][f];
          cs = (cs & ~(255u << (8 * (k & 3)))) | (b << (8 * (k & 3)));
          uint32_t& xs = pl->xsel[2 * f + (b >> 2)][k < 8 ? 0 : 1];   // element words (w1, w0) / (w3, w2)
          xs = (xs & ~(255u << (8 * (b & 3)))) | ((k & 7u) << (8 * (b & 3)));
        }
        ++k;
        pl->base[f] &= ~(255ull << (8 * b));
      }
    if (f == 2) pl->index_bytes = k;
  }
  pl->k = k;
}
hipError_t launch_records_census(const LaunchCfg& cfg, const void* recs, size_t n, uint64_t* d_census, hipStream_t st) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(kCensusSlots * 8), 0, st, (u64*)d_census);
  if (n) launch_census(cfg, recs, n, (u64*)d_census, nullptr, st);
  hipLaunchKernelGGL(ibu_k_sort_census_fold, dim3(1), dim3(kCensusSlots), 0, st, (u64*)d_census);
  return hipGetLastError();
}
// records -> elements of W words (pl.k <= 4 W).  Records that start at an odd record of a larger array (8- but not 16-byte
// aligned) are PEELED like everywhere else (kcommon.hpp): one record through the per-record kernel brings the rest to a
// 16-byte boundary for the tiled kernel (the elements need no more than their 4-byte alignment).
template <int W>
static void launch_compress(const LaunchCfg& cfg, const CompactPlan& pl, const void* recs, size_t n, u32 first_byte, ElemT<W>* out,
                            uint8_t* digits, hipStream_t st, u64* census = nullptr) {   // census: 16-byte aligned records only
  const size_t head = (reinterpret_cast<uintptr_t>(recs) & 15u) ? (n ? 1 : 0) : 0;
  const size_t main_rows = ((n - head) / kTileRecs) * kTileRecs;
  if (head)
    hipLaunchKernelGGL(ibu_k_sort_compress_tail<W>, dim3(1), dim3(256), 0, st, (const u64*)recs, (u64)0, (u64)head, pl, first_byte, out, digits);
  if (main_rows) {
    static std::atomic<int> occ[2];
    const u32 nt = (u32)(main_rows / kTileRecs);
    const uint8_t* base = static_cast<const uint8_t*>(recs) + 24 * head;
    if (census)
      hipLaunchKernelGGL((ibu_k_sort_compress<true, W>), dim3(grid_for(nt, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_sort_compress<true, W>, 0, &occ[1]))),
                         dim3(kBlock), 0, st, base, nt, pl, first_byte, out + head, digits ? digits + head : digits, census);
    else
      hipLaunchKernelGGL((ibu_k_sort_compress<false, W>), dim3(grid_for(nt, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_sort_compress<false, W>, 0, &occ[0]))),
                         dim3(kBlock), 0, st, base, nt, pl, first_byte, out + head, digits ? digits + head : digits, (u64*)nullptr);
  }
  if (head + main_rows < n) {
    hipLaunchKernelGGL(ibu_k_sort_compress_tail<W>, dim3(tail_grid(n - head - main_rows)), dim3(256), 0, st, (const u64*)recs,
                       (u64)(head + main_rows), (u64)n, pl, first_byte, out, digits);
    if (census)   // the rest rows of the census (each row also against its predecessor)
      hipLaunchKernelGGL(ibu_k_sort_census_tail, dim3(tail_grid(n - head - main_rows)), dim3(256), 0, st, (const u64*)recs,
                         (u64)(head + main_rows), (u64)n, census, (u32*)nullptr);
  }
}
template <int W>
static void launch_expand_w(const LaunchCfg& cfg, const CompactPlan& pl, const ElemT<W>* elems, size_t n, void* recs, hipStream_t st) {
  const size_t head = (reinterpret_cast<uintptr_t>(recs) & 15u) ? 1 : 0;   // peeled: see launch_compress
  const size_t main_rows = ((n - head) / kTileRecs) * kTileRecs;
  if (head)
    hipLaunchKernelGGL(ibu_k_sort_expand_tail<W>, dim3(1), dim3(256), 0, st, elems, (u64)0, (u64)head, pl, (u64*)recs);
  if (main_rows) {
    static std::atomic<int> occ;
    const u32 nsub = (u32)(main_rows / kTileRecs), nt = (nsub + kExpandSub - 1) / kExpandSub;
    hipLaunchKernelGGL(ibu_k_sort_expand<W>, dim3(grid_for(nt, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_sort_expand<W>, 0, &occ))), dim3(kBlock),
                       0, st, elems + head, nt, nsub, pl, static_cast<uint8_t*>(recs) + 24 * head);
  }
  if (head + main_rows < n)
    hipLaunchKernelGGL(ibu_k_sort_expand_tail<W>, dim3(tail_grid(n - head - main_rows)), dim3(256), 0, st, elems, (u64)(head + main_rows),
                       (u64)n, pl, (u64*)recs);
}
hipError_t launch_compact(const LaunchCfg& cfg, const CompactPlan& pl, const void* recs, size_t n, void* elems, hipStream_t st) {
  (void)hipGetLastError();
  if (n == 0) return hipSuccess;
  if (pl.k > 12 || n >= (1ull << 38)) return hipErrorInvalidValue;
  launch_compress<3>(cfg, pl, recs, n, 0, static_cast<Elem*>(elems), nullptr, st);
  return hipGetLastError();
}
hipError_t launch_expand(const LaunchCfg& cfg, const CompactPlan& pl, const void* elems, size_t n, void* recs, hipStream_t st) {
  (void)hipGetLastError();
  if (n == 0) return hipSuccess;
  if (pl.k > 12 || n >= (1ull << 38)) return hipErrorInvalidValue;
  launch_expand_w<3>(cfg, pl, static_cast<const Elem*>(elems), n, recs, st);
  return hipGetLastError();
}

// The compact-key path of launch_sort_records (see "COMPACT-KEY passes" above), W words per element.
// passes[0 .. npass): the element bytes to sort by, ascending.  compressed: the elements (and the digit stream of
// `digits_byte`) are already in place — the speculative path ran the compress pass itself.
// W = 3: both element buffers live in tmp (12 n bytes each) and the last pass always writes the records.
// W = 4: 16 n + 16 n bytes do not fit in tmp, so the second buffer is the head of the RECORD ARRAY (its contents are dead once
//        the elements exist).  The last pass can write records into that array only while reading from tmp, i.e. when the
//        pass count is odd; with an even count it stays an element pass (recs -> tmp) and an expand pass (tmp -> recs) follows.
// finish_prefix = P > 0: PREFIX + FINISH on elements — only the top P of `passes` run (as element passes), then
// ibu_k_sort_finish_elems completes the runs of equal prefix and writes the records; if it overflows (long runs), all passes run
// after all, starting from the prefix-sorted elements wherever they ended (elems_at).  W = 4 needs an even P (the elements must
// end in tmp: the records are written over the other buffer).
static bool trace_sort();
template <int W>
static hipError_t launch_compact_passes(const LaunchCfg& cfg, const CompactVariant& cv, void* recs, void* tmp, size_t n, uint8_t* sc,
                                        const CompactPlan& pl, const u32* passes, u32 npass, hipStream_t st, bool compressed = false,
                                        u32 digits_byte = 0, u32 finish_prefix = 0, ElemT<W>* elems_at = nullptr) {
  const bool retried = (finish_prefix & 0x80000000u) != 0;    // the one retry with a longer prefix (see the overflow handling below)
  finish_prefix &= 0x7FFFFFFFu;
  const SortLayout L = sort_layout(cfg, n, cv.tile);
  u64* binbase = reinterpret_cast<u64*>(sc + L.binbase);
  u32* blocksum = reinterpret_cast<u32*>(sc + L.blocksum);
  u64* blockoff = reinterpret_cast<u64*>(sc + L.blockoff);
  uint16_t* counts = reinterpret_cast<uint16_t*>(sc + L.counts);
  void* pos = sc + L.pos;                                     // u32 or u64 entries (L.idx64)
  uint8_t* digits = sc + L.digits;
  const void* k_scatter = L.idx64 ? cv.scatter64 : cv.scatter;
  const void* k_scatter_last = L.idx64 ? cv.scatter_last64 : cv.scatter_last;
  if (!k_scatter || !k_scatter_last) return hipErrorInvalidValue;   // (the caller only comes here with a shape that has them)
  ElemT<W>* const half2 = W == 3 ? reinterpret_cast<ElemT<W>*>(static_cast<uint8_t*>(tmp) + 12 * n) : static_cast<ElemT<W>*>(recs);
  ElemT<W>* src = elems_at ? elems_at : static_cast<ElemT<W>*>(tmp);
  ElemT<W>* dst = src == half2 ? static_cast<ElemT<W>*>(tmp) : half2;
  const bool fuse_last = W == 3 || (npass & 1u);

  // every call, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE, and a process may drive
  // several GPUs through several contexts (a few microseconds against a sort of milliseconds)
  hipError_t e;
  if (cv.lds > 48 * 1024) {
    e = hipFuncSetAttribute(k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(k_scatter_last, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.lds);
    if (e != hipSuccess) return e;
  }
  const u32 first_pass = finish_prefix ? npass - finish_prefix : 0;
  if (!compressed) launch_compress<W>(cfg, pl, recs, n, passes[first_pass], src, digits, st);
  else if (digits_byte != passes[first_pass])
    hipLaunchKernelGGL(ibu_k_sort_digits<W>, dim3((u32)cfg.cus * 8), dim3(256), 0, st, (const ElemT<W>*)src, (u64)n, passes[first_pass], digits);
  // passes; the last one writes the records themselves (with a finishing pass behind them, none of them does)
  const u32 wave_grid = (L.ntiles + kSortWaves - 1) / kSortWaves;
  const u32 cap = (u32)cfg.cus * 8;
  for (u32 pi = first_pass; pi < npass; ++pi) {
    const u32 b = passes[pi];
    hipLaunchKernelGGL(cv.counts_bytes, dim3(wave_grid < cap ? wave_grid : cap), dim3(kSortThreads), 0, st, (const uint8_t*)digits, (u64)n,
                       L.ntiles, counts);
    hipLaunchKernelGGL(ibu_k_sort_blocksums, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, blocksum);
    hipLaunchKernelGGL(ibu_k_sort_blockscan, dim3(1), dim3(kSortThreads), 0, st, (const u32*)blocksum, L.nblocks, blockoff, binbase);
    if (L.idx64)
      hipLaunchKernelGGL(ibu_k_sort_tilepos<u64>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles,
                         (const u64*)blockoff, (const u64*)binbase, static_cast<u64*>(pos));
    else
      hipLaunchKernelGGL(ibu_k_sort_tilepos<u32>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles,
                         (const u64*)blockoff, (const u64*)binbase, static_cast<u32*>(pos));
    const bool last = pi + 1 == npass, to_records = last && fuse_last && !finish_prefix;
    u32 n32 = (u32)n, b_arg = b, nb_arg = last ? 4u * W : passes[pi + 1];   // 4 W: no digit stream behind the last pass
    u64 n64 = n;
    const ElemT<W>* src_arg = src;
    void* dst_arg = to_records ? recs : static_cast<void*>(dst);
    const void* pos_arg = pos;
    CompactPlan pl_arg = pl;
    void* args[] = {&src_arg, &dst_arg, L.idx64 ? static_cast<void*>(&n64) : static_cast<void*>(&n32), &b_arg, &nb_arg, &pos_arg, &digits, &pl_arg};
    u32 sgrid = (L.ntiles + 7u) & ~7u;                        // multiple of 8: XCD-aware tile order
    e = hipLaunchKernel(to_records ? k_scatter_last : k_scatter, dim3(sgrid), dim3(cv.threads), args, cv.lds, st);
    if (e != hipSuccess) return e;
    ElemT<W>* t = src; src = dst; dst = t;
  }
  if (finish_prefix) {
#ifndef IBU_FINISH_T
#define IBU_FINISH_T 1792
#endif
#ifndef IBU_FINISH_M
#define IBU_FINISH_M 256
#endif
    // 1792-element tiles + 256 of look-ahead (eight elements per thread): 34 / 43 KiB of LDS and 126 / 156 VGPRs -> four / three
    // workgroups per CU.  Measured at 1e9 records 16/12 (profiles r03_m, r03_n): (4096, 512) 12.2 ms, (3072, 256) 12.1, (2048, 512) 9.4,
    // (2048, 256) 7.8-8.0 while it fitted 128 VGPRs and 9.9 once later edits had pushed it to 135 (three workgroups per CU: r03_ae),
    // (2048, 128) 8.0, (1536, 256) 8.0, (1024, 256) 8.6, (1024, 128) 8.4; (1792, 256) 8.4 on the box where (2048, 256) took 9.9 (r03_af).
    // tests/test_tools.py pins the register budgets.
    constexpr int FT = IBU_FINISH_T, FM = IBU_FINISH_M;
    typedef FinishElemShape<W, FT, FM> FS;
    u32* d_overflow = reinterpret_cast<u32*>(sc + L.misc);
    e = hipMemsetAsync(d_overflow, 0, 4, st);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ibu_k_sort_finish_elems<W, FT, FM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FS::lds);
    if (e != hipSuccess) return e;
    EV<W> pm;                                                 // the prefix bytes as word masks
    for (int w = 0; w < W; ++w) pm.w[w] = 0;
    for (u32 pi = first_pass; pi < npass; ++pi) pm.w[passes[pi] >> 2] |= 255u << (8 * (passes[pi] & 3));
    static std::atomic<int> focc;
    int fper = focc.load(std::memory_order_relaxed);
    if (fper <= 0) {                                          // persistent grid, exactly resident (LDS and registers decide)
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fper, ibu_k_sort_finish_elems<W, FT, FM>, kSortThreads, FS::lds) != hipSuccess || fper <= 0) fper = 1;
      focc.store(fper, std::memory_order_relaxed);
    }
    const u32 ftiles = (u32)((n + FT - 1) / FT), fgrid = (u32)fper * (u32)cfg.cus;
    hipLaunchKernelGGL((ibu_k_sort_finish_elems<W, FT, FM>), dim3(ftiles < fgrid ? ftiles : fgrid), dim3(kSortThreads), FS::lds, st, (const ElemT<W>*)src, recs,
                       (u64)n, pm, pl, d_overflow);
    u32 overflow = 0;
    e = hipMemcpyAsync(&overflow, d_overflow, 4, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    if (!overflow) return hipGetLastError();
    // Long runs of equal prefix.  The prefix-sorted elements are a permutation of the input's, so anything may follow.  A heavy
    // prefix usually is a heavy BARCODE whose records the next key bytes (the UMI) spread again: ONE retry with a prefix that
    // reaches at least two bytes past the barcode (and is at least three bytes longer; W = 4: even, the elements must end in tmp)
    // is cheaper than all passes when it still saves two of them — and if that overflows too, all passes run.
    u32 nbar = 0;                                             // element bytes that come from the barcode (the most significant ones)
    for (u32 j = 0; j < pl.k && j < 4u * W; ++j)
      if (((pl.csel[j >> 2][0] >> (8 * (j & 3))) & 255u) != 0x0Cu) ++nbar;
    u32 longer = finish_prefix + 3 > nbar + 2 ? finish_prefix + 3 : nbar + 2;   // at least two bytes past the barcode
    if (W == 4) longer += longer & 1u;
    if (!retried && longer + 2 <= npass) {
      if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu prefix+finish overflowed (long runs of equal prefix): retrying with prefix_passes=%u of %u\n", n, longer, npass);
      return launch_compact_passes<W>(cfg, cv, recs, tmp, n, sc, pl, passes, npass, st, true, 0xFFFFFFFFu, longer | 0x80000000u, src);
    }
    if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu prefix+finish overflowed (long runs of equal prefix): all %u passes\n", n, npass);
    return launch_compact_passes<W>(cfg, cv, recs, tmp, n, sc, pl, passes, npass, st, true, 0xFFFFFFFFu, 0, src);   // elements: a permutation of the input's
  }
  if (!fuse_last) launch_expand_w<W>(cfg, pl, src, n, recs, st);   // W = 4, even pass count: the elements ended in tmp
  return hipGetLastError();
}

// IBU_TRACE_SORT=1: one line per sort on stderr saying which path it took (tests assert on it; never set in production)
static bool trace_sort() {
  static const bool on = [] { const char* v = getenv("IBU_TRACE_SORT"); return v && *v && *v != '0'; }();
  return on;
}
// PREFIX + FINISH on the elements?  Only if the runs of equal prefix are going to be short: a pair count over a sample says
// (ibu_k_sort_sample_pairs; the tables live in tmp, which nothing uses at that point).  `sorted_bytes`: the passes the plain path
// would run (the element bytes it sorts on are the top `sorted_bytes` of the plan's k).  *P = the prefix to use, 0 = none.
static hipError_t estimate_compact_prefix(const LaunchCfg& cfg, const void* recs, size_t n, void* tmp, const CompactPlan& plan, u32 sorted_bytes,
                                          hipStream_t st, u32* P_out, double* seg_out) {
  *P_out = 0;
  *seg_out = 0;
  if (!cfg.sort_hybrid || (reinterpret_cast<uintptr_t>(tmp) & 7u) != 0) return hipSuccess;
  u32 slots = kPairSlotsMax;
  while (slots > 256 && (size_t)slots * 12 * kMaxPrefix + 128 > n * 24) slots >>= 1;
  if ((size_t)slots * 12 * kMaxPrefix + 128 > n * 24) return hipSuccess;
  // the estimate's own sample: 48 evenly spaced ranges of 2048 records (98 304 records, spread over the input: input that is
  // grouped in stretches is seen for what it is); fewer and shorter ranges while the tables must be small (load factor <= 3/8)
  const size_t cap = (size_t)slots * 3 / 8;
  u32 per_range = 2048, nranges = 48;
  while (nranges > 3 && (size_t)nranges * per_range > cap) nranges /= 2;
  if ((size_t)nranges * per_range > cap) per_range = (u32)(cap / nranges);
  if (per_range < 32 || (size_t)nranges * per_range > n) return hipSuccess;
  const size_t m = (size_t)nranges * per_range;
  const u64 range_stride = (n - per_range) / (nranges - 1);
  uint8_t* tb = static_cast<uint8_t*>(tmp);
  u64* d_pairs = reinterpret_cast<u64*>(tb);
  u64* d_keys = reinterpret_cast<u64*>(tb + 128);
  u32* d_cnts = reinterpret_cast<u32*>(tb + 128 + (size_t)slots * 8 * kMaxPrefix);
  const u64* r64 = static_cast<const u64*>(recs);
  const u32 margin = cfg.sort_hybrid == 2 ? 1u : 3u;
  u32 P = 0;
  // prefixes of 1 .. 8 bytes first; keys that need more (a wide barcode from a whitelist: every barcode byte and then some of the
  // UMI's) get a second and third look at 9 .. 16 and 17 .. 24 bytes, as long as such a prefix would still save passes
  for (u32 first = 0; !P && first < plan.k && first + 1 + margin <= sorted_bytes; first += (u32)kMaxPrefix) {
    hipError_t e = hipMemsetAsync(tb, 0, 128 + (size_t)slots * 12 * kMaxPrefix, st);
    if (e != hipSuccess) return e;
    if (plan.k <= 12)
      hipLaunchKernelGGL(ibu_k_sort_sample_pairs<3>, dim3((u32)((m + 255) / 256)), dim3(256), 0, st, r64, range_stride, nranges, per_range, plan, plan.k,
                         first, slots, d_keys, d_cnts, d_pairs);
    else
      hipLaunchKernelGGL(ibu_k_sort_sample_pairs<4>, dim3((u32)((m + 255) / 256)), dim3(256), 0, st, r64, range_stride, nranges, per_range, plan, plan.k,
                         first, slots, d_keys, d_cnts, d_pairs);
    u64 pairs[2 * kMaxPrefix];                                 // [q]: pairs of equal (first + q + 1)-byte prefix; [kMaxPrefix + q]: the most frequent one's count (0: below 4)
    e = hipMemcpyAsync(pairs, d_pairs, sizeof pairs, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    // a record shares its prefix with about 1 + (n / m) * (2 pairs / m) records: at most ~8 wanted (ranking is quadratic)
    for (u32 q = 0; q < (u32)kMaxPrefix && first + q + 1 <= plan.k; ++q) {
      const double seg = 1.0 + ((double)n / (double)m) * (2.0 * (double)pairs[q] / (double)m);
      const double heaviest = (double)pairs[kMaxPrefix + q] * ((double)n / (double)m);   // estimated longest run
      if (seg <= 8.0 && heaviest <= 128.0) { P = first + q + 1; *seg_out = seg; break; }
    }
  }
  if (P && plan.k > 12 && (P & 1u)) ++P;                       // 16-byte elements must end in tmp: an even number of passes
  // worth it?  The finishing pass costs about as much as two element passes (14 B read with the look-ahead + 24 B written per
  // record), the plain path's last pass half a pass more than the others
  if (P && P + margin > sorted_bytes) P = 0;                   // not worth it / would reach into index bytes the passes do not sort on
  *P_out = P;
  return hipSuccess;
}

// Not purely asynchronous: the census result comes back to the host (one 64-byte read) to pick the passes; everything
// after that is queued on `st`.
hipError_t launch_sort_records(const LaunchCfg& cfg, void* recs, void* tmp, size_t n, void* scratch,
                               size_t scratch_bytes, hipStream_t st) {
  (void)hipGetLastError();
  if (n < 2) return hipSuccess;
  const SweepVariant& sv = pick_variant(cfg);
  if ((n + sv.tile - 1) / sv.tile >= (1ull << 31)) return hipErrorInvalidValue;
  const SortLayout L = sort_layout(cfg, n, sv.tile);
  if (scratch_bytes < L.total) return hipErrorInvalidValue;
  uint8_t* sc = static_cast<uint8_t*>(scratch);
  u64* census = reinterpret_cast<u64*>(sc);
  u64* binbase = reinterpret_cast<u64*>(sc + L.binbase);
  u32* blocksum = reinterpret_cast<u32*>(sc + L.blocksum);
  u64* blockoff = reinterpret_cast<u64*>(sc + L.blockoff);
  uint16_t* counts = reinterpret_cast<uint16_t*>(sc + L.counts);
  void* pos = sc + L.pos;
  uint8_t* digits = sc + L.digits;

  // Compact-key path (see "COMPACT-KEY passes"): records 16-byte aligned (the tiled compress kernel), tmp at least 4-byte
  // aligned; from 2^32 records on (64-bit element indices) the shapes that carry those kernels (the defaults).  Whether at most
  // 12 / 16 key bytes vary is the census' to say.
  const CompactVariant* cv = pick_compact(cfg);
  const bool wide_idx = n >= (1ull << 32) || cfg.sort_idx64;
  const bool compact_ok = cv && n < (1ull << 38) && (!wide_idx || (cv->scatter64 && pick_compact16(cfg).scatter64)) &&
                          (reinterpret_cast<uintptr_t>(recs) & 15u) == 0 &&
                          (reinterpret_cast<uintptr_t>(tmp) & 3u) == 0 && scratch_bytes >= sort_layout(cfg, n, cv->tile).total &&
                          scratch_bytes >= sort_layout(cfg, n, pick_compact16(cfg).tile).total;
  hipError_t e;
  // SPECULATION (large inputs): the census and the compress pass both read all the records.  A census of three SAMPLE
  // ranges (first / middle / last 32 Ki records: tens of microseconds) guesses which bytes vary; the compress pass runs on
  // that guess at once and accumulates the EXACT census on the way; afterwards the guess only has to COVER the truth (every
  // byte that really varies is in the elements: bytes it carried needlessly are constant digits, their passes are skipped).
  // A guess that missed a byte costs the compress pass it wasted, and the sort goes on from the exact census as before.
  bool speculated = false;
  CompactPlan gpl;
  u64 g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gmask[3] = {0, 0, 0};
  u32 gfirst = 0, hybP = 0;
  double hyb_seg = 0;
  static constexpr size_t kSample = 32768;
  // cfg.sort_guess: 0 = never, 1 = inputs of 2^17 records and more (the three sample ranges must fit), k > 1 = of k records and more.
  // (Rounds 1-2 started at 2^23: one read of the records saved against one more host round trip.  With prefix + finish behind the
  // guess the sizes in between gain 2x — 3e5 / 1e6 / 4e6 records: 0.42 / 0.75 / 1.07 ms -> 0.27 / 0.42 / 0.58 ms.)
  const size_t guess_min = cfg.sort_guess == 1 ? 4 * kSample : ((size_t)cfg.sort_guess > 4 * kSample ? (size_t)cfg.sort_guess : 4 * kSample);
  if (compact_ok && cfg.sort_guess && n >= guess_min) {
    hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(kCensusSlots * 8), 0, st, census);
    const size_t starts[3] = {0, (n / 2) & ~(size_t)1, (n - kSample) & ~(size_t)1};   // even rows: 16-byte aligned
    for (size_t s0 : starts) launch_census(cfg, static_cast<const u64*>(recs) + 3 * s0, kSample, census, nullptr, st);
    hipLaunchKernelGGL(ibu_k_sort_census_fold, dim3(1), dim3(kCensusSlots), 0, st, census);
    e = hipMemcpyAsync(g, census, sizeof g, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    compact_plan_init(reinterpret_cast<const uint64_t*>(g), reinterpret_cast<const uint64_t*>(g + 3), &gpl);
    // g[7] == 0: no sample row is smaller than its predecessor — the input may well be sorted already, and then the
    // read-only census below (24 B/record) answers that; a speculative compress pass (37 B/record) would be spent first.
    if (g[7] != 0 && gpl.k >= 1 && gpl.k <= 16) {           // 12-byte elements, or 16-byte ones for 13 .. 16 varying bytes
      for (int f = 0; f < 3; ++f)
        for (u32 b = 0; b < 8; ++b)
          if (((g[f] ^ g[3 + f]) >> (8 * b)) & 255u) gmask[f] |= 255ull << (8 * b);
      gfirst = (g[6] == 0 && gpl.index_bytes < gpl.k) ? gpl.index_bytes : 0;   // the sample's guess of the first sorted byte
      e = estimate_compact_prefix(cfg, recs, n, tmp, gpl, gpl.k - gfirst, st, &hybP, &hyb_seg);
      if (e != hipSuccess) return e;
      if (hybP) gfirst = gpl.k - hybP;                       // the digit stream the compress pass leaves: the first prefix pass's
      hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(kCensusSlots * 8), 0, st, census);
      if (gpl.k <= 12) launch_compress<3>(cfg, gpl, recs, n, gfirst, static_cast<ElemT<3>*>(tmp), sc + sort_layout(cfg, n, cv->tile).digits, st, census);
      else launch_compress<4>(cfg, gpl, recs, n, gfirst, static_cast<ElemT<4>*>(tmp), sc + sort_layout(cfg, n, pick_compact16(cfg).tile).digits, st, census);
      speculated = true;
    } else if (g[7] == 0 && trace_sort()) {
      fprintf(stderr, "ibu sort: n=%zu samples in order: read-only census first\n", n);
    }
  }
  if (!speculated) {
    hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(kCensusSlots * 8), 0, st, census);
    launch_census(cfg, recs, n, census, nullptr, st);
  }
  hipLaunchKernelGGL(ibu_k_sort_census_fold, dim3(1), dim3(kCensusSlots), 0, st, census);
  u64 c[8];
  e = hipMemcpyAsync(c, census, sizeof c, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  if (c[7] == 0) {                   // no record is smaller than its predecessor: already sorted
    if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu already sorted%s\n", n, speculated ? " (a speculative compress pass was spent)" : "");   // spent only when the samples saw a drop and the whole did not: impossible, the samples are rows of the whole
    return hipSuccess;
  }

  // which digits vary.  The sort is stable and the index is the LEAST significant field: if the input already runs in
  // non-decreasing index order (the usual case: records are written in read order), ties on (barcode, umi) keep that
  // order and the index passes are the identity — 7 passes instead of 11 at 16/12.
  struct Pass { u32 field, shift; } passes[kDigits];
  int npass = 0;
  static const int kFieldOrder[3] = {2, 1, 0};  // least significant first: index, umi, barcode
  for (int fo = 0; fo < 3; ++fo) {
    const int f = kFieldOrder[fo];
    if (f == 2 && c[6] == 0) continue;
    const u64 varying = c[f] ^ c[3 + f];        // bits that differ between some two records
    for (u32 b = 0; b < 8; ++b)
      if ((varying >> (8 * b)) & 255u) passes[npass++] = {(u32)f, 8 * b};   // constant digits: the pass would be the identity
  }
  if (compact_ok && npass > 0) {
    u32 ebytes[16], ne = 0;
    if (speculated) {
      bool covered = true;
      for (int f = 0; f < 3; ++f)
        if ((c[f] ^ c[3 + f]) & ~gmask[f]) covered = false;
      if (covered) {                                        // the elements in tmp hold every byte that varies
        CompactPlan pl = gpl;
        u32 j = 0;
        for (int fo = 0; fo < 3; ++fo) {
          const int f = kFieldOrder[fo];
          pl.base[f] = c[3 + f] & ~gmask[f];
          for (u32 b = 0; b < 8; ++b)
            if ((gmask[f] >> (8 * b)) & 255u) {             // element byte j = byte b of field f
              const bool varies = ((c[f] ^ c[3 + f]) >> (8 * b)) & 255u, sorted_on = !(f == 2 && c[6] == 0);
              if (varies && sorted_on) ebytes[ne++] = j;
              ++j;
            }
        }
        // prefix + finish: the prefix the estimate was made for must be the top hybP SORTED bytes of the elements
        if (ne && hybP && hybP < ne && ebytes[ne - hybP] == pl.k - hybP) {
          if (trace_sort())
            fprintf(stderr, "ibu sort: n=%zu path=compact-prefix+finish element_bytes=%d prefix_passes=%u of %u estimated_run=%.2f\n", n,
                    pl.k <= 12 ? 12 : 16, hybP, ne, hyb_seg);
          return pl.k <= 12 ? launch_compact_passes<3>(cfg, *cv, recs, tmp, n, sc, pl, ebytes, ne, st, true, gfirst, hybP)
                            : launch_compact_passes<4>(cfg, pick_compact16(cfg), recs, tmp, n, sc, pl, ebytes, ne, st, true, gfirst, hybP);
        }
        if (ne) {
          if (trace_sort())
            fprintf(stderr, "ibu sort: n=%zu path=compact-speculated element_bytes=%d passes=%u first_digit_guess=%s\n", n, pl.k <= 12 ? 12 : 16, ne,
                    gfirst == ebytes[0] ? "hit" : "miss");
          return pl.k <= 12 ? launch_compact_passes<3>(cfg, *cv, recs, tmp, n, sc, pl, ebytes, ne, st, true, gfirst)
                            : launch_compact_passes<4>(cfg, pick_compact16(cfg), recs, tmp, n, sc, pl, ebytes, ne, st, true, gfirst);
        }
      }
      if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu guess did not cover the varying bytes\n", n);
    }
    CompactPlan pl;
    compact_plan_init(reinterpret_cast<const uint64_t*>(c), reinterpret_cast<const uint64_t*>(c + 3), &pl);
    if (pl.k <= 16) {
      for (u32 j = c[6] == 0 ? pl.index_bytes : 0; j < pl.k; ++j) ebytes[ne++] = j;   // input in index order: the index bytes ride along unsorted
      // prefix + finish on the exact plan (inputs below the speculation threshold, or whose guess was not taken): the same estimate
      if (ne && n >= 8192 && !speculated) {
        u32 P = 0;
        double seg = 0;
        e = estimate_compact_prefix(cfg, recs, n, tmp, pl, ne, st, &P, &seg);
        if (e != hipSuccess) return e;
        if (P && P < ne) {
          if (trace_sort())
            fprintf(stderr, "ibu sort: n=%zu path=compact-prefix+finish element_bytes=%d prefix_passes=%u of %u estimated_run=%.2f (exact plan)\n", n,
                    pl.k <= 12 ? 12 : 16, P, ne, seg);
          return pl.k <= 12 ? launch_compact_passes<3>(cfg, *cv, recs, tmp, n, sc, pl, ebytes, ne, st, false, 0, P)
                            : launch_compact_passes<4>(cfg, pick_compact16(cfg), recs, tmp, n, sc, pl, ebytes, ne, st, false, 0, P);
        }
      }
      if (ne) {
        if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu path=compact element_bytes=%d passes=%u\n", n, pl.k <= 12 ? 12 : 16, ne);
        return pl.k <= 12 ? launch_compact_passes<3>(cfg, *cv, recs, tmp, n, sc, pl, ebytes, ne, st)
                          : launch_compact_passes<4>(cfg, pick_compact16(cfg), recs, tmp, n, sc, pl, ebytes, ne, st);
      }
    }
  }
  const void* scatter = L.idx64 ? sv.scatter64 : sv.scatter32;
  if (sv.lds > 48 * 1024) {   // per call: the attribute is per device (see launch_compact_passes)
    e = hipFuncSetAttribute(scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sv.lds);
    if (e != hipSuccess) return e;
  }

  const u32 nfull = (u32)(n / sv.tile);          // tiles with all T records
  const u32 wave_grid = (L.ntiles + kSortWaves - 1) / kSortWaves;
  const u32 cap = (u32)cfg.cus * 8;
  // LSD passes over ps[0 .. np) (least significant first), ping-pong between recs and tmp.  want_in_tmp: where the result is
  // wanted.  The first pass's counting kernel reads every record anyway: when the parity of np would leave the result in
  // the other array, it also copies the records across (24 B/record) and the passes start from there.
  auto lsd = [&](const Pass* ps, int np, bool want_in_tmp) -> hipError_t {
    u64* src = static_cast<u64*>(recs);
    u64* dst = static_cast<u64*>(tmp);
    const bool stage = ((np & 1) != 0) != want_in_tmp;
    for (int p = 0; p < np; ++p) {
      // counts of every tile: from the records for the first pass, from the digit side stream afterwards
      if (p == 0) {
        const bool aligned = ((reinterpret_cast<uintptr_t>(src) | (stage ? reinterpret_cast<uintptr_t>(dst) : 0)) & 15u) == 0;
        const u32 fast = aligned ? nfull : 0;
        if (fast)
          hipLaunchKernelGGL(sv.counts_recs, dim3(fast < cap ? fast : cap), dim3(kSortThreads), 0, st, (const uint8_t*)src, fast,
                             ps[p].field, ps[p].shift, counts, stage ? reinterpret_cast<uint8_t*>(dst) : (uint8_t*)nullptr);
        if (fast < L.ntiles)
          hipLaunchKernelGGL(sv.counts_tail, dim3(L.ntiles - fast), dim3(kSortThreads), 0, st, (const u64*)src, (u64)n, fast,
                             ps[p].field, ps[p].shift, counts, stage ? dst : (u64*)nullptr);
        if (stage) { u64* t = src; src = dst; dst = t; }   // the records now sit in the other array
      } else {
        hipLaunchKernelGGL(sv.counts_bytes, dim3(wave_grid < cap ? wave_grid : cap), dim3(kSortThreads), 0, st, (const uint8_t*)digits,
                           (u64)n, L.ntiles, counts);
      }
      hipLaunchKernelGGL(ibu_k_sort_blocksums, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, blocksum);
      hipLaunchKernelGGL(ibu_k_sort_blockscan, dim3(1), dim3(kSortThreads), 0, st, (const u32*)blocksum, L.nblocks, blockoff, binbase);
      if (L.idx64)
        hipLaunchKernelGGL(ibu_k_sort_tilepos<u64>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles,
                           (const u64*)blockoff, (const u64*)binbase, static_cast<u64*>(pos));
      else
        hipLaunchKernelGGL(ibu_k_sort_tilepos<u32>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles,
                           (const u64*)blockoff, (const u64*)binbase, static_cast<u32*>(pos));
      const bool last = p + 1 == np;
      const u32 nf = last ? 3u : ps[p + 1].field, ns = last ? 0u : ps[p + 1].shift;
      u64 n_arg = n;
      u32 f_arg = ps[p].field, s_arg = ps[p].shift, nf_arg = nf, ns_arg = ns;
      const u64* src_arg = src;
      void* args[] = {&src_arg, &dst, &n_arg, &f_arg, &s_arg, &nf_arg, &ns_arg, &pos, &digits};
      hipError_t le = hipLaunchKernel(scatter, dim3((L.ntiles + 7u) & ~7u), dim3(sv.threads), args, sv.lds, st);   // multiple of 8: XCD-aware tile order
      if (le != hipSuccess) return le;
      u64* t = src; src = dst; dst = t;
    }
    return hipGetLastError();
  };

  // PREFIX + FINISH (see ibu_k_sort_finish): P = the fewest prefix bytes that leave about 64 records per segment of well-spread
  // keys; worth it when at least three passes are saved.  cfg.sort_hybrid: 0 = never, 1 = auto, 2 = whenever a pass is saved
  // (tests).  The result of the P passes is wanted in tmp: the finishing kernel writes the records back into `recs`.
  {
    // P: ranking inside a segment is quadratic in its length (measured at 1e9 records: 1.5 ms per record of average segment
    // length, against 10.3 ms for one more prefix pass), so the prefix is chosen to leave at most ~8 records per segment
    int P = 1;
    for (u64 segs = 256; n / segs > 8 && P < 8; segs <<= 8) ++P;
    // ... of WELL-SPREAD keys.  From 2^17 records on the sample ranges say whether they are (ibu_k_sort_sample_pairs_recs: pairs of
    // equal prefix and the most frequent prefix among 3 x 32 Ki sample records, tables in tmp): the shortest prefix with at most
    // ~8 records per run and no heavy prefix is taken, which may be longer than the one n suggests — or none (P = 0: all passes).
    static constexpr size_t kSampleW = 32768;
    if (cfg.sort_hybrid && n >= 4 * kSampleW && (reinterpret_cast<uintptr_t>(tmp) & 7u) == 0) {
      PrefixBytes pb;
      pb = PrefixBytes();
      for (int k = 0; k < npass && k < 24; ++k) { pb.field[k] = (uint8_t)passes[npass - 1 - k].field; pb.shift[k] = (uint8_t)passes[npass - 1 - k].shift; }
      const u32 slots = kPairSlotsMax;                       // 25 MB of tables in tmp: from 1.05 M records on (below: P from n alone)
      const size_t table_bytes = 128 + (size_t)slots * 12 * kMaxPrefix;
      if (table_bytes <= n * 24) {
        uint8_t* tb = static_cast<uint8_t*>(tmp);
        const u32 per_range = 2048, nranges = 48;             // 48 evenly spaced ranges of 2048 records
        const size_t m = (size_t)nranges * per_range;
        const u64 range_stride = (n - per_range) / (nranges - 1);
        const int est_margin = cfg.sort_hybrid == 2 ? 1 : 3;
        int Pest = 0;
        // prefixes of 1 .. 8 bytes, then (wide barcodes from a whitelist: all their bytes and some of the UMI's) 9 .. 16 and 17 .. 24,
        // as long as such a prefix would still save passes
        for (int first = 0; !Pest && first + 1 + est_margin <= npass; first += kMaxPrefix) {
          pb.first = (u32)first;
          pb.count = (u32)(npass < first + kMaxPrefix ? npass : first + kMaxPrefix);
          e = hipMemsetAsync(tb, 0, table_bytes, st);
          if (e != hipSuccess) return e;
          hipLaunchKernelGGL(ibu_k_sort_sample_pairs_recs, dim3((u32)((m + 255) / 256)), dim3(256), 0, st, (const u64*)recs, range_stride, nranges,
                             per_range, pb, slots, reinterpret_cast<u64*>(tb + 128),
                             reinterpret_cast<u32*>(tb + 128 + (size_t)slots * 8 * kMaxPrefix), reinterpret_cast<u64*>(tb));
          u64 pairs[2 * kMaxPrefix];
          e = hipMemcpyAsync(pairs, tb, sizeof pairs, hipMemcpyDeviceToHost, st);
          if (e != hipSuccess) return e;
          e = hipStreamSynchronize(st);
          if (e != hipSuccess) return e;
          for (u32 q = 0; q < (u32)kMaxPrefix && pb.first + q + 1 <= pb.count; ++q) {
            const double seg = 1.0 + ((double)n / (double)m) * (2.0 * (double)pairs[q] / (double)m);
            const double heaviest = (double)pairs[kMaxPrefix + q] * ((double)n / (double)m);
            if (seg <= 8.0 && heaviest <= 128.0) { Pest = (int)(pb.first + q + 1); break; }
          }
        }
        if (trace_sort() && Pest != P) fprintf(stderr, "ibu sort: n=%zu sample estimate: prefix_passes=%d (well-spread keys would take %d)\n", n, Pest, P);
        P = Pest ? Pest : npass;                              // npass: never worth it below
      }
    }
    const int margin = cfg.sort_hybrid == 2 ? 1 : 3;
    if (cfg.sort_hybrid && npass >= P + margin && n < (1ull << 40)) {
      u32* d_overflow = reinterpret_cast<u32*>(sc + L.misc);
      e = hipMemsetAsync(d_overflow, 0, 4, st);
      if (e != hipSuccess) return e;
      const Pass* ps = passes + (npass - P);     // the P most significant varying bytes
      u64 pm[3] = {0, 0, 0};
      for (int p = 0; p < P; ++p) pm[ps[p].field] |= 255ull << ps[p].shift;
      e = lsd(ps, P, true);
      if (e != hipSuccess) return e;
      typedef FinishShape<kFinishT, kFinishM> FS;
      const u64 nblk = (n + kFinishT - 1) / kFinishT;
      if ((reinterpret_cast<uintptr_t>(tmp) & 15u) == 0) {    // persistent, prefetching form
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ibu_k_sort_finish<kFinishT, kFinishM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FS::lds);
        if (e != hipSuccess) return e;
        static std::atomic<int> focc;
        int fper = focc.load(std::memory_order_relaxed);
        if (fper <= 0) {
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fper, ibu_k_sort_finish<kFinishT, kFinishM, true>, kSortThreads, FS::lds) != hipSuccess || fper <= 0) fper = 1;
          focc.store(fper, std::memory_order_relaxed);
        }
        const u64 fgrid = (u64)fper * (u64)cfg.cus;
        hipLaunchKernelGGL((ibu_k_sort_finish<kFinishT, kFinishM, true>), dim3((u32)(nblk < fgrid ? nblk : fgrid)), dim3(kSortThreads), FS::lds, st, (const u64*)tmp,
                           static_cast<u64*>(recs), (u64)n, pm[0], pm[1], pm[2], d_overflow);
      } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ibu_k_sort_finish<kFinishT, kFinishM, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FS::lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ibu_k_sort_finish<kFinishT, kFinishM, false>), dim3((u32)nblk), dim3(kSortThreads), FS::lds, st, (const u64*)tmp,
                           static_cast<u64*>(recs), (u64)n, pm[0], pm[1], pm[2], d_overflow);
      }
      u32 overflow = 0;
      e = hipMemcpyAsync(&overflow, d_overflow, 4, hipMemcpyDeviceToHost, st);
      if (e != hipSuccess) return e;
      e = hipStreamSynchronize(st);
      if (e != hipSuccess) return e;
      if (!overflow) {
        if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu path=prefix+finish prefix_passes=%d of %d varying bytes\n", n, P, npass);
        return hipSuccess;
      }
      // segments too long for the finishing kernel (heavy prefixes): the prefix-sorted records in tmp are a permutation of the
      // input — copy them back and run every pass
      if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu prefix+finish overflowed (long runs of equal prefix): all %d passes\n", n, npass);
      e = launch_copy(cfg, tmp, recs, n * 24, st);
      if (e != hipSuccess) return e;
    }
  }
  if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu path=24-byte passes=%d\n", n, npass);
  return lsd(passes, npass, false);
}

// =====================================================================================================
// Splitter search of the multi-GPU sample sort: thread j finds the first record >= key j in the sorted records
// (log2 n probes of 24 bytes each; k is the number of ranks minus one).
extern "C" __global__ void ibu_k_lower_bound(const u64* __restrict__ recs, u64 n, const u64* __restrict__ keys, u32 k, u64* __restrict__ pos) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const u64 kb = keys[3 * j], ku = keys[3 * j + 1], kx = keys[3 * j + 2];
  u64 lo = 0, hi = n;
  while (lo < hi) {
    const u64 mid = lo + (hi - lo) / 2;
    if (rec_less(recs[3 * mid], recs[3 * mid + 1], recs[3 * mid + 2], kb, ku, kx)) lo = mid + 1;
    else hi = mid;
  }
  pos[j] = lo;
}
hipError_t launch_lower_bound(const void* recs, size_t n, const void* keys, size_t k, uint64_t* pos, hipStream_t st) {
  (void)hipGetLastError();
  if (k == 0) return hipSuccess;
  hipLaunchKernelGGL(ibu_k_lower_bound, dim3((u32)((k + 63) / 64)), dim3(64), 0, st, (const u64*)recs, (u64)n, (const u64*)keys, (u32)k,
                     (u64*)pos);
  return hipGetLastError();
}

}  // namespace ibu
