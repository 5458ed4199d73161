// sort.hip — device sort of 24-byte records by (barcode, umi, index): the order `derive(Ord)` gives
// Record (src/constructs/record.rs:58-66) and the header's sorted flag promises (header.rs:111-113).
//
// Stable LSD radix sort, 8-bit digits, over the 192-bit key — but only over the digits that actually
// vary: a census kernel ORs and ANDs each field over all records, and a digit whose bits are equal in
// OR and AND is constant, so its pass would be the identity and is skipped.  16-base barcodes, 12-base
// UMIs and indices below 2^32 need 4 + 3 + 4 passes instead of 24 — and 4 + 3 when the input already runs in index
// order (the sort is stable and the index is the least significant field), which the same census detects.
//
// "Onesweep" structure (round 2; round 1 read every record twice per pass):
//   census     one streaming read: OR / AND per field, "already sorted", "already in index order"
//   histogram  ONE streaming read builds the 256-bin histogram of EVERY varying digit at once (LDS atomics),
//              so no pass has to count before it scatters
//   pass       one kernel per varying digit: a workgroup takes the next tile by ticket, stages it in LDS, ranks it with
//              wave-level match-any (8 ballots per record), publishes the tile's bin counts, permutes the tile into
//              digit order inside LDS, learns its global offsets by DECOUPLED LOOK-BACK over the status words of the
//              tiles before it, and writes the runs out as 16-byte chunks.
// HBM traffic: 24 B/record twice up front, then 24 B read + 24 B written per record and pass (+ 2 x 2 KiB of status words per
// tile).  Cross-workgroup hand-off: one 8-byte {tag, value} word per (tile, bin), written by ONE relaxed agent-scope store and
// polled by relaxed agent-scope loads (sc1: past the non-coherent per-XCD L2s) — the word IS the flag, no fence needed
// (MI355X_MICROARCH.md, "Valid forms": 8-B agent atomics both sides).  Tiles start in ticket order, so the tile with the
// smallest unfinished ticket never waits on a tile that has not started: no dependence on dispatch order or placement.
// Every spin is bounded and aborts the whole sort through a global flag (no write is issued from a tile that gave up).
#include <stdio.h>

#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {

static constexpr int kSortThreads = 256;
static constexpr int kSortWaves = kSortThreads / kWave;       // 4
static constexpr int kBins = 256;
static constexpr int kDigits = 24;                            // 3 fields x 8 bytes

// ---- scratch layout (bytes) --------------------------------------------------------------------------
//   census u64[8] @0     OR x3, AND x3, index-order flag, any-inversion flag
//   ticket u32[24] @64   next tile of each pass        err u32 @160   a look-back gave up
//   hist   u64[24][256] @256   digit histograms, scanned in place into exclusive bin bases
//   status u64[kStatusRows][256] @kOffStatus   ring of per-tile status rows (tile t uses row t % kStatusRows)
static constexpr size_t kOffTicket = 64, kOffErr = 160, kOffHist = 256, kOffStatus = kOffHist + 8 * kDigits * kBins;
// A row is reused every kStatusRows tiles.  At most 2048 workgroups are resident (256 CUs x 8) and tiles start in
// ticket order, so when tile t + kStatusRows starts every tile up to t + kStatusRows - 2048 has finished, and no tile
// still looking back can reach as far back as t.  The generation t / kStatusRows is part of the tag anyway.
static constexpr u32 kStatusRowsLog2 = 15, kStatusRows = 1u << kStatusRowsLog2;
// status word: [63:62] flag | [61:56] epoch = pass + 1 | [55:40] generation (tile / kStatusRows) | [39:0] value
static constexpr u64 kFlagAgg = 1, kFlagIncl = 2, kValueMask = (1ull << 40) - 1;
__device__ __forceinline__ u64 status_tag(u64 flag, u32 epoch, u32 tile) {
  return (flag << 62) | ((u64)(epoch & 63u) << 56) | ((u64)((tile >> kStatusRowsLog2) & 0xFFFFu) << 40);
}
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) u32 gu32;
__device__ __forceinline__ u64 ld_agent(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ld_agent32(const u32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
struct CensusAcc {
  u64 o[3] = {0, 0, 0}, a[3] = {~0ull, ~0ull, ~0ull};
  bool index_drops = false, order_drops = false;
  __device__ __forceinline__ void rec(u64 b, u64 u, u64 x) { o[0] |= b; o[1] |= u; o[2] |= x; a[0] &= b; a[1] &= u; a[2] &= x; }
  __device__ __forceinline__ void pair(u64 pb, u64 pu, u64 px, u64 b, u64 u, u64 x) {
    if (x < px) index_drops = true;                       // input not in index order: the index passes are needed
    if (rec_less(b, u, x, pb, pu, px)) order_drops = true;  // not already sorted
  }
  // c == nullptr: only the order flag is wanted (ibu_is_sorted); flag32 != nullptr receives it
  __device__ __forceinline__ void flush(u64* c, u32* flag32) {
    const u32 lane = threadIdx.x & (kWave - 1);
    if (flag32 && __ballot(order_drops) && lane == 0) atomicOr(flag32, 1u);
    if (!c) return;
    if (__ballot(index_drops) && lane == 0) atomicOr(&c[6], 1ull);
    if (__ballot(order_drops) && lane == 0) atomicOr(&c[7], 1ull);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
      for (int f = 0; f < 3; ++f) { o[f] |= shfl_xor64(o[f], m); a[f] &= shfl_xor64(a[f], m); }
    if (lane == 0)
#pragma unroll
      for (int f = 0; f < 3; ++f) { atomicOr(&c[f], o[f]); atomicAnd(&c[3 + f], a[f]); }
  }
};

extern "C" __global__ void ibu_k_sort_census_init(u64* c) {
  if (threadIdx.x < 3) c[threadIdx.x] = 0;
  else if (threadIdx.x < 6) c[threadIdx.x] = ~0ull;
  else if (threadIdx.x < 8) c[threadIdx.x] = 0;  // [6]: some index smaller than its predecessor's; [7]: some record smaller
}
// recs0: row 0 of the caller's array (8-B aligned); the tiles start at row `row0` (16-B aligned there).
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_sort_census(const u64* __restrict__ recs0, u64 row0, u32 ntiles, u64* __restrict__ c, u32* __restrict__ flag32) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kTileBytes;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  const uint8_t* base = reinterpret_cast<const uint8_t*>(recs0 + 3 * row0);
  CensusAcc acc;
  u32 t = logical_block() * kWavesPerBlock + wib;
  if (t < ntiles) {
    const uint8_t* src = base + (size_t)t * kTileBytes + 16 * lane;
    u32x4 a0 = ld16(src), a1 = ld16(src + 1024), a2 = ld16(src + 2048);
    for (;;) {
      const u32 tn = t + nwaves;
      const bool more = tn < ntiles;               // wave-uniform; the prefetch is unconditional (kcommon.hpp)
      src = base + (size_t)(more ? tn : t) * kTileBytes + 16 * lane;
      const u32x4 b0 = ld16(src), b1 = ld16(src + 1024), b2 = ld16(src + 2048);
      // the record before this lane's pair: lane 0 fetches the one before the tile (if any) from global memory
      const u64 grow = row0 + (u64)t * kTileRecs;  // global row of the tile's first record
      u64 p[3] = {0, 0, 0};
      const bool has_prev = lane > 0 || grow > 0;
      if (lane == 0 && grow > 0) { const u64* q = recs0 + 3 * (grow - 1); p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; }
      wave_lds_fence();
      *reinterpret_cast<u32x4*>(tile + 16 * lane) = a0;
      *reinterpret_cast<u32x4*>(tile + 1024 + 16 * lane) = a1;
      *reinterpret_cast<u32x4*>(tile + 2048 + 16 * lane) = a2;
      wave_lds_fence();
      const u64* r = reinterpret_cast<const u64*>(tile + (2 * lane) * 24);  // records 2L, 2L+1 (and 2L-1 just below)
      if (lane > 0) { p[0] = r[-3]; p[1] = r[-2]; p[2] = r[-1]; }
      const u64 x0 = r[0], x1 = r[1], x2 = r[2], y0 = r[3], y1 = r[4], y2 = r[5];
      acc.rec(x0, x1, x2);
      acc.rec(y0, y1, y2);
      if (has_prev) acc.pair(p[0], p[1], p[2], x0, x1, x2);
      acc.pair(x0, x1, x2, y0, y1, y2);
      if (!more) break;
      t = tn;
      a0 = b0; a1 = b1; a2 = b2;
    }
  }
  acc.flush(c, flag32);
}
// rows [row0, n), one thread per row (the n % 128 rest, a peeled first row); compares with row - 1 as well
extern "C" __global__ void ibu_k_sort_census_tail(const u64* __restrict__ recs, u64 row0, u64 n, u64* __restrict__ c,
                                                  u32* __restrict__ flag32) {
  CensusAcc acc;
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const u64 b = recs[3 * i], u = recs[3 * i + 1], x = recs[3 * i + 2];
    acc.rec(b, u, x);
    if (i > 0) acc.pair(recs[3 * i - 3], recs[3 * i - 2], recs[3 * i - 1], b, u, x);
  }
  acc.flush(c, flag32);
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
// Histogram of every varying digit in ONE streaming read.  No LDS staging: the wave stride (3072 B = 384 u64) is a
// multiple of 3, so the u64 a lane finds in slot (k, h) of its dwordx4 loads always belongs to the same field
// (k_records.hip, ibu_k_reduce).  vary[f] has bit b set when byte b of field f differs between some two records; only
// those digits are counted: ds_add_u32 into a [24][256] table per workgroup, flushed with one u64 atomic per
// non-zero counter.
// =====================================================================================================
__device__ __forceinline__ void hist_word(u32* h, u64 v, u32 field, u32 vmask, u32 any_mask) {
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if (!((any_mask >> b) & 1u)) continue;         // wave-uniform: no field varies in this byte
    if ((vmask >> b) & 1u) atomicAdd(&h[(field * 8 + b) * kBins + ((u32)(v >> (8 * b)) & 255u)], 1u);
  }
}
extern "C" __global__ void __launch_bounds__(kBlock, 6)
ibu_k_sort_hist(const uint8_t* __restrict__ recs, u32 ntiles, u32 vary0, u32 vary1, u32 vary2, u64* __restrict__ hist) {
  __shared__ u32 h[kDigits * kBins];
  for (u32 i = threadIdx.x; i < kDigits * kBins; i += kBlock) h[i] = 0;
  __syncthreads();
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  const u32 any_mask = vary0 | vary1 | vary2;
  u32 fld[3][2], vm[3][2];                          // field and varying-byte mask of this lane's six u64 slots
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const u32 f = (2 * (64 * k + lane) + hh) % 3;
      fld[k][hh] = f;
      vm[k][hh] = f == 0 ? vary0 : (f == 1 ? vary1 : vary2);
    }
  u32 t = logical_block() * kWavesPerBlock + wib;
  if (t < ntiles) {
    const uint8_t* p = recs + (size_t)t * kTileBytes + 16 * lane;
    u32x4 a[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = ld16(p + 1024 * k);
    for (;;) {
      const u32 tn = t + nwaves;
      const bool more = tn < ntiles;
      p = recs + (size_t)(more ? tn : t) * kTileBytes + 16 * lane;
      u32x4 b[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) b[k] = ld16(p + 1024 * k);   // unconditional prefetch (kcommon.hpp)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        hist_word(h, ((u64)a[k].y << 32) | a[k].x, fld[k][0], vm[k][0], any_mask);
        hist_word(h, ((u64)a[k].w << 32) | a[k].z, fld[k][1], vm[k][1], any_mask);
      }
      if (!more) break;
      t = tn;
#pragma unroll
      for (int k = 0; k < 3; ++k) a[k] = b[k];
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < kDigits * kBins; i += kBlock) {
    const u32 v = h[i];
    if (v) atomicAdd(&hist[i], (u64)v);
  }
}
extern "C" __global__ void ibu_k_sort_hist_tail(const u64* __restrict__ recs, u64 row0, u64 n, u32 vary0, u32 vary1, u32 vary2,
                                                u64* __restrict__ hist) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 vary[3] = {vary0, vary1, vary2};
#pragma unroll
  for (int f = 0; f < 3; ++f) {
    const u64 v = recs[3 * i + f];
    for (int b = 0; b < 8; ++b)
      if ((vary[f] >> b) & 1u) atomicAdd(&hist[(f * 8 + b) * kBins + ((u32)(v >> (8 * b)) & 255u)], 1ull);
  }
}

// Block-wide exclusive scan of one u32 per thread over the first 256 threads of the block (every thread of the block
// must call; threads >= 256 pass 0 and get garbage).  Leaves the total in *total.
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* wsum /*[4] shared*/, u32* total) {
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const u32 t = __shfl_up(inc, d);
    if (lane >= (u32)d) inc += t;
  }
  if (lane == kWave - 1 && wib < 4) wsum[wib] = inc;
  __syncthreads();
  u32 off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const u32 s = wsum[w];
    if ((u32)w < wib) off += s;
    tot += s;
  }
  __syncthreads();  // wsum may be reused by the caller's next scan
  *total = tot;
  return off + inc - v;
}
// hist[d][0..255] -> exclusive prefix in place (one 256-thread block per digit)
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_scan_hist(u64* __restrict__ hist) {
  __shared__ u64 wsum[kSortWaves];
  u64* row = hist + (size_t)blockIdx.x * kBins;
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  const u64 v = row[threadIdx.x];
  u64 inc = v;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const u64 t = shfl_up64(inc, d);
    if (lane >= (u32)d) inc += t;
  }
  if (lane == kWave - 1) wsum[wib] = inc;
  __syncthreads();
  u64 off = 0;
  for (u32 w = 0; w < wib; ++w) off += wsum[w];
  row[threadIdx.x] = off + inc - v;
}

// Row `bin` of a u32 table -> exclusive prefix within the row (in place) and the row total (used by the run counting below).
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_scan_rows(u32* __restrict__ table, u32 nchunks, u32* __restrict__ rowsum) {
  __shared__ u32 wsum[kSortWaves];
  u32* row = table + (size_t)blockIdx.x * nchunks;
  u32 carry = 0;
  for (u32 base = 0; base < nchunks; base += 4 * kSortThreads) {
    const u32 i0 = base + 4 * threadIdx.x;
    u32 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i0 + j < nchunks ? row[i0 + j] : 0;
    u32 tot;
    u32 ex = carry + block_exclusive_scan(v[0] + v[1] + v[2] + v[3], wsum, &tot);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0 + j < nchunks) row[i0 + j] = ex;
      ex += v[j];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) rowsum[blockIdx.x] = carry;
}

// =====================================================================================================
// One radix pass.  THREADS x ROUNDS records per tile; WMODE 0 = 8-byte words, 1 = 16-byte chunks (write-out).
// =====================================================================================================
template <int THREADS, int ROUNDS>
struct SweepShape {
  static constexpr int T = THREADS * ROUNDS, NW = THREADS / kWave, PER_WAVE = T / NW;
  // LDS: stage 24 T | gdelta 256 x u64 | whist NW x 256 x u32 | misc 16 x u32 | sbin T bytes
  static constexpr size_t lds = 24 * (size_t)T + 8 * kBins + 4 * (size_t)NW * kBins + 64 + (size_t)T;
};

#ifndef IBU_SORT_NT_STORE
#define IBU_SORT_NT_STORE 0
#endif
#ifndef IBU_SORT_LOOKBACK
#define IBU_SORT_LOOKBACK 8
#endif
static constexpr int kLookBack = IBU_SORT_LOOKBACK;          // status rows polled per round trip of the look-back
__device__ __forceinline__ void sort_st16(u64* p, u64 a, u64 b) {
  u32x4 v; v.x = (u32)a; v.y = (u32)(a >> 32); v.z = (u32)b; v.w = (u32)(b >> 32);
#if IBU_SORT_NT_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
#else
  *reinterpret_cast<u32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void sort_st8(u64* p, u64 a) {
#if IBU_SORT_NT_STORE
  __builtin_nontemporal_store(a, p);
#else
  *p = a;
#endif
}

template <int THREADS, int ROUNDS, int WMODE>
__global__ void __launch_bounds__(THREADS)
ibu_k_sort_onesweep(const u64* __restrict__ src, u64* __restrict__ dst, u64 n, u32 field, u32 shift, u32 epoch,
                    const u64* __restrict__ gbase, u64* status, u32* ticket, u32* err) {
  typedef SweepShape<THREADS, ROUNDS> S;
  constexpr int T = S::T, NW = S::NW, PER_WAVE = S::PER_WAVE;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u64* stage = reinterpret_cast<u64*>(smem);                 // the tile: first in input order, then in digit order
  u64* gdelta = stage + 3 * T;                               // global record index of slot p of bin d = gdelta[d] + p
  u32* whist = reinterpret_cast<u32*>(gdelta + kBins);       // per wave: running count while ranking, then base slot of (wave, bin)
  u32* misc = whist + NW * kBins;                            // [0..3] scan scratch, [8] tile id, [9] abort
  uint8_t* sbin = reinterpret_cast<uint8_t*>(misc + 16);     // digit of each slot of the permuted tile
  const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wib = tid >> 6;
  const u64 lt_mask = (1ull << lane) - 1;

  // 0. next tile, in ticket order (dispatch order is not a contract)
  if (tid == 0) { misc[8] = atomicAdd(ticket, 1u); misc[9] = ld_agent32(err); }
  __syncthreads();
  const u32 tile = misc[8];
  const u64 tbase = (u64)tile * T;
  if (tbase >= n || misc[9]) return;                         // block-uniform (cannot happen for grid == ntiles; abort: see below)
  const u32 cnt = n - tbase < (u64)T ? (u32)(n - tbase) : (u32)T;

  // 1. stage the tile (coalesced) and clear the per-wave counters
  if (cnt == (u32)T && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0)) {
    const u32x4* g = reinterpret_cast<const u32x4*>(src + 3 * tbase);
    u32x4* s = reinterpret_cast<u32x4*>(stage);
    constexpr int kChunks = T * 24 / 16 / THREADS;           // dwordx4 per thread per tile (T*24/16 = 1.5 T)
    u32x4 v[kChunks];
#pragma unroll
    for (int k = 0; k < kChunks; ++k) v[k] = ld16(g + tid + THREADS * k);
#pragma unroll
    for (int k = 0; k < kChunks; ++k) s[tid + THREADS * k] = v[k];
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

  // 3. bin totals of the tile -> published (AGGREGATE); slot bases per (wave, bin)
  u32 tot = 0, tb = 0;
  {
    u32 c[NW];
    if (tid < (u32)kBins) {
#pragma unroll
      for (int w = 0; w < NW; ++w) { c[w] = whist[w * kBins + tid]; tot += c[w]; }
    }
    u32 all;
    tb = block_exclusive_scan(tid < (u32)kBins ? tot : 0u, misc, &all);
    if (tid < (u32)kBins) {
      st_agent(&status[(size_t)(tile & (kStatusRows - 1)) * kBins + tid], status_tag(kFlagAgg, epoch, tile) | tot);
      u32 run = tb;
#pragma unroll
      for (int w = 0; w < NW; ++w) { whist[w * kBins + tid] = run; run += c[w]; }
    }
  }
  __syncthreads();

  // 4. permute the tile into digit order inside LDS (no global offset needed yet: predecessors get time to publish)
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const u32 slot = wib * PER_WAVE + r * kWave + lane;
    if (slot < cnt) {
      const u32 p = whist[wib * kBins + dig[r]] + rk[r];
      stage[3 * p] = r0[r]; stage[3 * p + 1] = r1[r]; stage[3 * p + 2] = r2[r];
      sbin[p] = (uint8_t)dig[r];
    }
  }

  // 5. decoupled look-back: records of bin `tid` in all earlier tiles
  if (tid < (u32)kBins) {
    u64 excl = 0;
    bool gave_up = false;
    const u64 want_agg = status_tag(kFlagAgg, epoch, 0) >> 56, want_incl = status_tag(kFlagIncl, epoch, 0) >> 56;
    u32 spins = 0;
#ifdef IBU_SORT_PROBE
    u32 dbg_steps = 0;
    const u64 dbg_t0 = __builtin_amdgcn_s_memtime();
#endif
    // A WINDOW of kLookBack predecessors is polled at once (independent loads in flight together), then examined in
    // order.  One row at a time the walk is latency-bound and long: a tile's INCLUSIVE word appears one walk after its
    // AGGREGATE, so a walk has to cross every tile that started within the last walk's duration — measured ~19 rows
    // of ~0.5 us each per tile (profiles/README.md, r02).  W rows per round trip shorten the walk W-fold in time and, through
    // the earlier INCLUSIVE words, in length too.
    u32 t = (WMODE == 3 ? 0u : tile);                         // next row to examine is t - 1; WMODE 3 (probe build): no look-back
    while (t > 0) {
      u64 s[kLookBack];
#pragma unroll
      for (int j = 0; j < kLookBack; ++j) {
        const u32 q = t > (u32)j ? t - 1 - (u32)j : 0u;       // rows past tile 0 re-read row 0 and are ignored below
        s[j] = ld_agent(&status[(size_t)(q & (kStatusRows - 1)) * kBins + tid]);
      }
      bool done = false, stalled = false;
      u32 used = 0;
#pragma unroll
      for (int j = 0; j < kLookBack; ++j) {
        if (done || stalled || (u32)j >= t) continue;
        const u32 q = t - 1 - (u32)j;
        const bool mine = ((s[j] >> 40) & 0xFFFFu) == (u64)((q >> kStatusRowsLog2) & 0xFFFFu);
        const u64 top = s[j] >> 56;
        if (mine && top == want_incl) { excl += s[j] & kValueMask; done = true; }
        else if (mine && top == want_agg) {
          excl += s[j] & kValueMask; ++used;
#ifdef IBU_SORT_PROBE
          ++dbg_steps;
#endif
        } else stalled = true;                                  // not published yet: poll again from this row
      }
      if (done) break;
      t -= used;
      if (stalled) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0 && (spins >= (1u << 21) || ld_agent32(err))) { gave_up = true; break; }
      }
    }
#ifdef IBU_SORT_PROBE
    if (tid == 0 && (tile & 63u) == 0) {  // probe build (every 64th tile): look-back anatomy of bin 0 (steps over AGGREGATE rows, not-ready polls, cycles)
      u64* dbg = reinterpret_cast<u64*>(reinterpret_cast<uint8_t*>(err) + 32);
      atomicAdd(&dbg[0], (u64)dbg_steps);
      atomicAdd(&dbg[1], (u64)spins);
      atomicMax(&dbg[2], (u64)dbg_steps);
      atomicAdd(&dbg[3], (u64)(__builtin_amdgcn_s_memtime() - dbg_t0));
      atomicAdd(&dbg[4], 1ull);
    }
#endif
    if (gave_up) {
      atomicOr(err, 1u);
      misc[9] = 1;                                            // this workgroup issues no store
    } else {
      st_agent(&status[(size_t)(tile & (kStatusRows - 1)) * kBins + tid], status_tag(kFlagIncl, epoch, tile) | ((excl + tot) & kValueMask));
      gdelta[tid] = gbase[tid] + excl - tb;                   // wraps harmlessly: slot >= tb for this bin
    }
  }
  __syncthreads();
  if (misc[9]) return;

  // 6. write out
  const u32 nw = 3 * cnt;
  if constexpr (WMODE == 0) {
    for (u32 w = tid; w < nw; w += THREADS) {                 // consecutive lanes write consecutive 8-byte words of each run
      const u32 s = (u32)(((u64)w * 0xAAAAAAABull) >> 33);    // w / 3
      const u64 g = gdelta[sbin[s]] + s;
      sort_st8(dst + 3 * g + (w - 3 * s), stage[w]);
    }
  } else if constexpr (WMODE >= 2) {
    // probe builds only (-DIBU_SORT_PROBE, results are WRONG): the permuted tile goes out linearly as dwordx4 — what the
    // pass would cost if the scatter were free (WMODE 3: and the look-back too)
    u32x4* o = reinterpret_cast<u32x4*>(dst + 3 * tbase);
    const u32x4* s = reinterpret_cast<const u32x4*>(stage);
    for (u32 c = tid; 2 * c + 1 < nw; c += THREADS) o[c] = s[c];
  } else {
    // 16-byte chunks of the OUTPUT: word w of the permuted tile goes to output word W(w); consecutive words of a run are
    // consecutive output words.  A word whose output address is 16-B aligned and whose successor continues the run
    // starts a dwordx4 store; a word the chunks leave over (run head at an odd word, run tail at an even word) goes
    // out alone as a dwordx2.  Each thread looks at the word pair (2j, 2j+1): at most one chunk starts in it.
    const u32 par = (u32)(reinterpret_cast<uintptr_t>(dst) >> 3) & 1u;
    for (u32 j = tid; 2 * j < nw; j += THREADS) {
      const u32 wa = 2 * j, wb = wa + 1;
      const u32 sl = wa ? (u32)(((u64)(wa - 1) * 0xAAAAAAABull) >> 33) : 0u;   // slot of word wa-1; words wa-1..wb+1 span sl, sl+1
      const u32 s2 = sl + 1 < cnt ? sl + 1 : sl;
      const u64 g1 = gdelta[sbin[sl]] + sl, g2 = gdelta[sbin[s2]] + s2;
      auto W = [&](u32 w) -> u64 {                            // output word of tile word w (w within [wa-1, wb+1])
        const u32 s = (u32)(((u64)w * 0xAAAAAAABull) >> 33);
        return 3 * (s == sl ? g1 : g2) + (w - 3 * s);
      };
      const bool has_b = wb < nw, has_c = wb + 1 < nw;
      const u64 Wa = W(wa);
      const u64 Wp = wa ? W(wa - 1) : Wa - 2;                 // anything but Wa - 1
      const u64 Wb = has_b ? W(wb) : Wa + 2;
      const u64 Wc = has_c ? W(wb + 1) : Wb + 2;
      const bool same_pa = Wp + 1 == Wa, same_ab = Wa + 1 == Wb, same_bc = Wb + 1 == Wc;
      const bool even_a = (((u32)Wa + par) & 1u) == 0, even_b = (((u32)Wb + par) & 1u) == 0;
      const bool chunk_a = even_a && same_ab, chunk_b = has_b && even_b && same_bc;
      if (chunk_a || chunk_b) {
        const u32 w = chunk_a ? wa : wb;
        sort_st16(dst + (chunk_a ? Wa : Wb), stage[w], stage[w + 1]);
      }
      if (even_a ? !same_ab : !same_pa) sort_st8(dst + Wa, stage[wa]);
      if (has_b && (even_b ? !same_bc : !same_ab)) sort_st8(dst + Wb, stage[wb]);
    }
  }
}

// =====================================================================================================
size_t sort_scratch_bytes(const LaunchCfg&, size_t n) {
  (void)n;
  return kOffStatus + sizeof(u64) * kBins * (size_t)kStatusRows;   // 64 MiB ring + 48 KiB of histograms, whatever n
}

typedef void (*SweepFn)(const u64*, u64*, u64, u32, u32, u32, const u64*, u64*, u32*, u32*);
struct SweepVariant { SweepFn fn; int threads, tile; size_t lds; };
template <int TH, int R, int WM>
static SweepVariant sweep_variant() { return {ibu_k_sort_onesweep<TH, R, WM>, TH, SweepShape<TH, R>::T, SweepShape<TH, R>::lds}; }
// cfg.sort_variant: tile shape x write-out mode (A/B through ibu_ctx_set_option(ctx, "sort_variant", k))
static const SweepVariant kSweep[] = {
    sweep_variant<512, 4, 1>(),   // 0 (default): 2048-record tiles, 8 waves, 16-byte chunks
    sweep_variant<512, 4, 0>(),   // 1: same, 8-byte words
    sweep_variant<256, 4, 1>(),   // 2: 1024-record tiles
    sweep_variant<256, 4, 0>(),   // 3
    sweep_variant<256, 8, 1>(),   // 4: 2048-record tiles, 4 waves
    sweep_variant<256, 8, 0>(),   // 5
    sweep_variant<1024, 4, 1>(),  // 6: 4096-record tiles, one workgroup per CU
    sweep_variant<1024, 4, 0>(),  // 7
#ifdef IBU_SORT_PROBE
    sweep_variant<512, 4, 2>(), sweep_variant<512, 4, 3>(), sweep_variant<1024, 4, 2>(), sweep_variant<1024, 4, 3>(),  // 8..11
#endif
};
static constexpr int kNumSweep = sizeof(kSweep) / sizeof(kSweep[0]);
int sort_num_variants() { return kNumSweep; }

// Not purely asynchronous: the census result comes back to the host (one 64-byte read) to pick the passes, and the
// give-up flag of the look-back is read back at the end (hipErrorLaunchFailure if any tile gave up).
hipError_t launch_sort_records(const LaunchCfg& cfg, void* recs, void* tmp, size_t n, void* scratch,
                               size_t scratch_bytes, hipStream_t st) {
  (void)hipGetLastError();
  if (n < 2) return hipSuccess;
  if (n > kValueMask || scratch_bytes < sort_scratch_bytes(cfg, n)) return hipErrorInvalidValue;
  uint8_t* sc = static_cast<uint8_t*>(scratch);
  u64* census = reinterpret_cast<u64*>(sc);
  u32* ticket = reinterpret_cast<u32*>(sc + kOffTicket);
  u32* err = reinterpret_cast<u32*>(sc + kOffErr);
  u64* hist = reinterpret_cast<u64*>(sc + kOffHist);
  u64* status = reinterpret_cast<u64*>(sc + kOffStatus);

  hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(64), 0, st, census);
  launch_census(cfg, recs, n, census, nullptr, st);
  u64 c[8];
  hipError_t e = hipMemcpyAsync(c, census, sizeof c, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  if (c[7] == 0) return hipSuccess;  // no record is smaller than its predecessor: already sorted

  // which digits vary.  The sort is stable and the index is the LEAST significant field: if the input already runs in
  // non-decreasing index order (the usual case: records are written in read order), ties on (barcode, umi) keep that
  // order and the index passes are the identity — 7 passes instead of 11 at 16/12.
  u32 vary[3];
  for (int f = 0; f < 3; ++f) {
    const u64 varying = c[f] ^ c[3 + f];        // bits that differ between some two records
    vary[f] = 0;
    for (int b = 0; b < 8; ++b)
      if ((varying >> (8 * b)) & 255u) vary[f] |= 1u << b;
  }
  if (c[6] == 0) vary[2] = 0;

  const SweepVariant& sv = kSweep[cfg.sort_variant >= 0 && cfg.sort_variant < kNumSweep ? cfg.sort_variant : 0];
  const u64 ntiles64 = (n + sv.tile - 1) / sv.tile;
  if (ntiles64 >= (1ull << 31)) return hipErrorInvalidValue;
  const u32 ntiles = (u32)ntiles64;
  const size_t rows = ntiles < kStatusRows ? ntiles : kStatusRows;
  e = hipMemsetAsync(sc + kOffTicket, 0, kOffStatus - kOffTicket, st);           // tickets, give-up flag, histograms
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(status, 0, rows * kBins * sizeof(u64), st);                 // tags of an earlier call must not match
  if (e != hipSuccess) return e;
  {  // histogram of every varying digit, one read
    const Span sp[1] = {{recs, 24}};
    const RowSplit rs = split_rows(sp, 1, n, kTileRecs);
    if (rs.head)
      hipLaunchKernelGGL(ibu_k_sort_hist_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, (const u64*)recs, (u64)0, (u64)rs.head,
                         vary[0], vary[1], vary[2], hist);
    if (rs.main) {
      const u32 nt = (u32)(rs.main / kTileRecs);
      static std::atomic<int> occ;
      hipLaunchKernelGGL(ibu_k_sort_hist, dim3(grid_for(nt, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_sort_hist, 0, &occ))),
                         dim3(kBlock), 0, st, adv((const uint8_t*)recs, 24 * rs.head), nt, vary[0], vary[1], vary[2], hist);
    }
    if (rs.head + rs.main < n)
      hipLaunchKernelGGL(ibu_k_sort_hist_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, (const u64*)recs,
                         (u64)(rs.head + rs.main), (u64)n, vary[0], vary[1], vary[2], hist);
    hipLaunchKernelGGL(ibu_k_sort_scan_hist, dim3(kDigits), dim3(kSortThreads), 0, st, hist);
  }
  static std::atomic<bool> lds_set[kNumSweep];
  const int vi = (int)(&sv - kSweep);
  if (sv.lds > 48 * 1024 && !lds_set[vi].load(std::memory_order_relaxed)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sv.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sv.lds);
    if (e != hipSuccess) return e;
    lds_set[vi].store(true, std::memory_order_relaxed);
  }

  u64* src = static_cast<u64*>(recs);
  u64* dst = static_cast<u64*>(tmp);
  static const int kFieldOrder[3] = {2, 1, 0};  // least significant first: index, umi, barcode
  u32 pass = 0;
  for (int fo = 0; fo < 3; ++fo) {
    const int f = kFieldOrder[fo];
    for (u32 b = 0; b < 8; ++b) {
      if (!((vary[f] >> b) & 1u)) continue;     // constant digit: the pass would be the identity
      hipLaunchKernelGGL(sv.fn, dim3(ntiles), dim3(sv.threads), sv.lds, st, (const u64*)src, dst, (u64)n, (u32)f, 8 * b, pass + 1,
                         (const u64*)(hist + (size_t)(f * 8 + b) * kBins), status, ticket + pass, err);
      ++pass;
      u64* t = src; src = dst; dst = t;
    }
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (src != static_cast<u64*>(recs)) {          // odd number of passes
    e = launch_copy(cfg, src, recs, n * 24, st);
    if (e != hipSuccess) return e;
  }
  u32 gave_up = 0;
  e = hipMemcpyAsync(&gave_up, err, sizeof gave_up, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
#ifdef IBU_SORT_PROBE
  {
    u64 dbg[5];
    (void)hipMemcpy(dbg, sc + kOffErr + 32, sizeof dbg, hipMemcpyDeviceToHost);
    fprintf(stderr, "[sort probe] variant %d tiles*passes %llu: AGG steps/tile %.2f (max %llu), not-ready polls/tile %.2f, look-back cycles/tile %.0f\n",
            vi, (unsigned long long)dbg[4], (double)dbg[0] / (double)(dbg[4] ? dbg[4] : 1), (unsigned long long)dbg[2],
            (double)dbg[1] / (double)(dbg[4] ? dbg[4] : 1), (double)dbg[3] / (double)(dbg[4] ? dbg[4] : 1));
  }
#endif
  return gave_up ? hipErrorLaunchFailure : hipSuccess;
}

// =====================================================================================================
// Per-barcode aggregation on SORTED records: the device form of the reference's BarcodeAnalyzer
// processor (src/parallel.rs:72-98: HashMap<barcode, count> merged in on_batch_complete).  On sorted
// input a barcode is a run, so the map is a run-length encoding: barcodes[k], counts[k] and — the
// UMI-dedup figure single-cell pipelines want from exactly this layout — unique_umis[k] = number of
// distinct (barcode, umi) pairs in the run.  Output order = ascending barcode (the map's sorted keys).
//
// Each wave owns one 8 Ki-record segment and needs no LDS and no barrier: run heads are found with a
// lane shuffle (+ one extra load for lane 0), ranked with __ballot/popcount.  Pass 1 counts heads per
// segment, the [2][nseg] table is scanned, pass 2 emits each run's barcode, first record and pair rank with plain
// stores, and a last small kernel turns neighbouring entries into counts (no atomics anywhere: the first version
// used two per run and took 1 s on 0.9e9 runs of length one).
// =====================================================================================================
static constexpr int kSegRecs = 8192;

__device__ __forceinline__ u64 shfl_up64(u64 v) {
  u32 lo = __shfl_up((u32)v, 1), hi = __shfl_up((u32)(v >> 32), 1);
  return ((u64)hi << 32) | lo;
}
// heads of one 64-record step of a segment: h1 = first record of a barcode run, h2 = first record of a
// (barcode, umi) run.  Lanes past `end` are neither.
__device__ __forceinline__ void run_heads(const u64* __restrict__ recs, u64 i, u64 end, u32 lane, u64& b, bool& h1, bool& h2) {
  const bool valid = i < end;
  b = valid ? recs[3 * i] : 0;
  const u64 u = valid ? recs[3 * i + 1] : 0;
  u64 pb = shfl_up64(b), pu = shfl_up64(u);
  if (lane == 0 && valid && i > 0) { pb = recs[3 * (i - 1)]; pu = recs[3 * (i - 1) + 1]; }
  h1 = valid && (i == 0 || b != pb);
  h2 = valid && (h1 || u != pu);
}

extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_runs_count(const u64* __restrict__ recs, u64 n, u32 nseg, u32* __restrict__ seg_heads /*[2][nseg]*/) {
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 seg = blockIdx.x * kSortWaves + (threadIdx.x >> 6);
  if (seg >= nseg) return;                                  // wave-uniform
  const u64 base = (u64)seg * kSegRecs;
  const u64 end = base + kSegRecs < n ? base + kSegRecs : n;
  u32 c1 = 0, c2 = 0;
  for (u64 i0 = base; i0 < end; i0 += kWave) {
    u64 b; bool h1, h2;
    run_heads(recs, i0 + lane, end, lane, b, h1, h2);
    c1 += (u32)__popcll(__ballot(h1));
    c2 += (u32)__popcll(__ballot(h2));
  }
  if (lane == 0) { seg_heads[seg] = c1; seg_heads[nseg + seg] = c2; }
}

extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_runs_emit(const u64* __restrict__ recs, u64 n, u32 nseg, const u32* __restrict__ seg_base /*[2][nseg], scanned*/,
                u64* __restrict__ barcodes, u64* __restrict__ starts, u64* __restrict__ pair_rank) {
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 seg = blockIdx.x * kSortWaves + (threadIdx.x >> 6);
  if (seg >= nseg) return;
  const u64 lt_mask = (1ull << lane) - 1;
  const u64 base = (u64)seg * kSegRecs;
  const u64 end = base + kSegRecs < n ? base + kSegRecs : n;
  u64 p1 = seg_base[seg], p2 = seg_base[nseg + seg];        // runs / pairs that start before this segment
  for (u64 i0 = base; i0 < end; i0 += kWave) {
    const u64 i = i0 + lane;
    u64 b; bool h1, h2;
    run_heads(recs, i, end, lane, b, h1, h2);
    const u64 m1 = __ballot(h1), m2 = __ballot(h2);
    if (h1) {
      const u64 k = p1 + (u64)__popcll(m1 & lt_mask);       // index of the run that starts at record i
      barcodes[k] = b;
      starts[k] = i;                                        // first record of run k
      if (pair_rank) pair_rank[k] = p2 + (u64)__popcll(m2 & lt_mask);  // (barcode, umi) pairs that start before i
    }
    p1 += (u64)__popcll(m1);
    p2 += (u64)__popcll(m2);
  }
}
// counts[k] = start(k+1) - start(k), unique_umis[k] = pair_rank(k+1) - pair_rank(k); entry n_runs is the sentinel.
extern "C" __global__ void ibu_k_runs_finish(const u64* __restrict__ starts, const u64* __restrict__ pair_rank, u64 n_runs, u64 n,
                                             u64 n_pairs, u64* __restrict__ counts, u64* __restrict__ uniq) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n_runs; k += stride) {
    const bool last = k + 1 == n_runs;
    counts[k] = (last ? n : starts[k + 1]) - starts[k];
    if (uniq) uniq[k] = (last ? n_pairs : pair_rank[k + 1]) - pair_rank[k];
  }
}

size_t runs_scratch_bytes(size_t n) {
  const size_t nseg = (n + kSegRecs - 1) / kSegRecs;
  return 64 + 2 * sizeof(u32) * (nseg ? nseg : 1);
}
// Pass 1 + scan.  Leaves the scanned table in `scratch`; totals[0] = runs, totals[1] = (barcode, umi) pairs
// are read back by the caller from scratch[0..1] (u32 each; n < 2^32).
hipError_t launch_runs_count(const LaunchCfg&, const void* recs, size_t n, void* scratch, size_t scratch_bytes, hipStream_t st) {
  (void)hipGetLastError();
  if (n == 0 || n >= (1ull << 32) || scratch_bytes < runs_scratch_bytes(n)) return hipErrorInvalidValue;
  const u32 nseg = (u32)((n + kSegRecs - 1) / kSegRecs);
  u32* totals = static_cast<u32*>(scratch);
  u32* table = reinterpret_cast<u32*>(static_cast<uint8_t*>(scratch) + 64);
  hipLaunchKernelGGL(ibu_k_runs_count, dim3((nseg + kSortWaves - 1) / kSortWaves), dim3(kSortThreads), 0, st, (const u64*)recs,
                     (u64)n, nseg, table);
  hipLaunchKernelGGL(ibu_k_sort_scan_rows, dim3(2), dim3(kSortThreads), 0, st, table, nseg, totals);
  return hipGetLastError();
}
hipError_t launch_runs_emit(const LaunchCfg& cfg, const void* recs, size_t n, const void* scratch, void* run_scratch, uint64_t n_runs,
                            uint64_t n_pairs, uint64_t* barcodes, uint64_t* counts, uint64_t* uniq, hipStream_t st) {
  (void)hipGetLastError();
  const u32 nseg = (u32)((n + kSegRecs - 1) / kSegRecs);
  const u32* table = reinterpret_cast<const u32*>(static_cast<const uint8_t*>(scratch) + 64);
  u64* starts = static_cast<u64*>(run_scratch);             // n_runs entries each (run_scratch_bytes)
  u64* pair_rank = uniq ? starts + n_runs : nullptr;
  hipLaunchKernelGGL(ibu_k_runs_emit, dim3((nseg + kSortWaves - 1) / kSortWaves), dim3(kSortThreads), 0, st, (const u64*)recs,
                     (u64)n, nseg, table, (u64*)barcodes, starts, pair_rank);
  u64 blocks = (n_runs + 255) / 256;
  const u64 cap = (u64)cfg.cus * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(ibu_k_runs_finish, dim3((u32)(blocks ? blocks : 1)), dim3(256), 0, st, (const u64*)starts, (const u64*)pair_rank,
                     (u64)n_runs, (u64)n, (u64)n_pairs, (u64*)counts, (u64*)uniq);
  return hipGetLastError();
}
size_t runs_emit_scratch_bytes(uint64_t n_runs) { return 16 * (size_t)(n_runs ? n_runs : 1); }

}  // namespace ibu
