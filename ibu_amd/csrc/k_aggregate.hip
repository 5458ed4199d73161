// k_aggregate.hip — per-barcode aggregation of SORTED records: the device form of the reference's BarcodeAnalyzer
// processor (src/parallel.rs:72-98).  Launchers: launch_runs_count / launch_runs_emit (kernels.h); C ABI: ibu_barcode_counts
// (device.cpp).  Split from sort.hip in round 3: nothing here depends on the sort.
#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {

static constexpr int kSortThreads = 256;                      // one wave per segment, four waves per workgroup
static constexpr int kSortWaves = kSortThreads / kWave;

// =====================================================================================================
// Per-barcode aggregation on SORTED records: the device form of the reference's BarcodeAnalyzer
// processor (src/parallel.rs:72-98: HashMap<barcode, count> merged in on_batch_complete).  On sorted
// input a barcode is a run, so the map is a run-length encoding: barcodes[k], counts[k] and — the
// UMI-dedup figure single-cell pipelines want from exactly this layout — unique_umis[k] = number of
// distinct (barcode, umi) pairs in the run.  Output order = ascending barcode (the map's sorted keys).
//
// The rows are cut into SEGMENTS, one per wave, no barrier anywhere: segment 0 = the peeled rows in front of the first
// 16-B aligned record (at most one), segments 1 .. S = 8 Ki records each (64 tiles), segment S+1 = the n % 128 rest.
// The 8 Ki segments are TILED like every streaming kernel here (round 2; the first version read two stride-24 u64 per
// lane and step with nothing in flight): three coalesced dwordx4 loads stage 128 records in the wave's LDS slice while
// the next tile's loads are in flight, lane L owns records 2L and 2L+1 and reads record 2L-1 from the slice (lane 0: the
// last record of the previous tile, kept in registers; the first tile of a segment: one global load).  Run heads are
// ranked with __ballot / popcount.  Pass 1 counts heads per segment, the [2][nseg] table is scanned, pass 2 emits each
// run's barcode, first record and pair rank with plain stores, and a last small kernel turns neighbouring entries into
// counts (no atomics anywhere: the first version used two per run and took 1 s on 0.9e9 runs of length one).
// =====================================================================================================
//
// Round 5: ONE read of the records where runs are long.  The count pass also keeps the first kStashHeads run heads of every segment
// (barcode, row inside the segment, pair rank inside the segment: 16 bytes each, plain stores — there are few) in a stash in the
// scratch; the emit pass serves a segment with that many heads or fewer FROM THE STASH (a few lanes, no record read) and walks only
// the others again.  Whitelist barcodes (1e5 runs in 1e9 records, 0.8 heads per segment): 24 B/record instead of 48; every record
// its own barcode: as before.
// =====================================================================================================
static constexpr int kSegRecs = 8192;
static constexpr u32 kStashHeads = 32;
struct __attribute__((aligned(16))) RunStash { u64 barcode; u32 row_off; u32 pair_local; };

__device__ __forceinline__ u64 shfl_up64(u64 v) {
  u32 lo = __shfl_up((u32)v, 1), hi = __shfl_up((u32)(v >> 32), 1);
  return ((u64)hi << 32) | lo;
}
// heads of one 64-record step of an untiled segment: h1 = first record of a barcode run, h2 = first record of a
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

struct SegPlan { u64 head, main, n; u32 nseg; };            // rows [0, head) | [head, head + main) tiled | rest
static inline u32 runs_nseg(size_t main_rows) { return (u32)((main_rows + kSegRecs - 1) / kSegRecs) + 2; }
__device__ __forceinline__ u64 seg_first_row(const SegPlan& sp, u32 seg) {
  return seg == 0 ? 0 : (seg == sp.nseg - 1 ? sp.head + sp.main : sp.head + (u64)(seg - 1) * kSegRecs);
}

// One wave walks one segment and hands every run head to `emit(k, barcode, row, pair_rank)`; returns the number of
// barcode heads / pair heads through c1 / c2.  EMIT = false: counting only (p1, p2 unused).
template <bool EMIT, class F>
__device__ __forceinline__ void runs_segment(const u64* __restrict__ recs, const SegPlan& sp, u32 seg, uint8_t* tile, u32 lane, u64 p1,
                                             u64 p2, u64& c1, u64& c2, F emit) {
  const u64 lt_mask = (1ull << lane) - 1;
  c1 = c2 = 0;
  if (seg == 0 || seg == sp.nseg - 1) {                     // wave-uniform: the untiled ends (< 128 rows each)
    const u64 base = seg == 0 ? 0 : sp.head + sp.main;
    const u64 end = seg == 0 ? sp.head : sp.n;
    for (u64 i0 = base; i0 < end; i0 += kWave) {
      const u64 i = i0 + lane;
      u64 b; bool h1, h2;
      run_heads(recs, i, end, lane, b, h1, h2);
      const u64 m1 = __ballot(h1), m2 = __ballot(h2);
      if (EMIT && h1) emit(p1 + c1 + (u64)__popcll(m1 & lt_mask), b, i, p2 + c2 + (u64)__popcll(m2 & lt_mask));
      c1 += (u64)__popcll(m1);
      c2 += (u64)__popcll(m2);
    }
    return;
  }
  const u64 begin = sp.head + (u64)(seg - 1) * kSegRecs;    // 16-B aligned row
  const u64 stop = sp.head + sp.main;
  const u32 ntiles = (u32)(((begin + kSegRecs < stop ? begin + kSegRecs : stop) - begin) / kTileRecs);   // >= 1
  const uint8_t* src = reinterpret_cast<const uint8_t*>(recs + 3 * begin) + 16 * lane;
  u64 cb = 0, cu = 0;                                       // the record in front of the tile (barcode, umi)
  bool have_prev = begin > 0;
  if (have_prev) { cb = recs[3 * (begin - 1)]; cu = recs[3 * (begin - 1) + 1]; }
  u32x4 a0 = ld16(src), a1 = ld16(src + 1024), a2 = ld16(src + 2048);
  for (u32 t = 0;;) {
    const bool more = t + 1 < ntiles;                       // wave-uniform; the prefetch is unconditional (kcommon.hpp)
    const uint8_t* nx = src + (size_t)(more ? t + 1 : t) * kTileBytes;
    const u32x4 b0 = ld16(nx), b1 = ld16(nx + 1024), b2 = ld16(nx + 2048);
    wave_lds_fence();
    *reinterpret_cast<u32x4*>(tile + 16 * lane) = a0;
    *reinterpret_cast<u32x4*>(tile + 1024 + 16 * lane) = a1;
    *reinterpret_cast<u32x4*>(tile + 2048 + 16 * lane) = a2;
    wave_lds_fence();
    const u64* r = reinterpret_cast<const u64*>(tile + (2 * lane) * 24);  // records 2L, 2L+1 (and 2L-1 just below)
    u64 pb = cb, pu = cu;
    if (lane > 0) { pb = r[-3]; pu = r[-2]; }
    const u64 x0 = r[0], x1 = r[1], y0 = r[3], y1 = r[4];
    const bool ha1 = !(lane > 0 || have_prev) || x0 != pb, ha2 = ha1 || x1 != pu;
    const bool hb1 = y0 != x0, hb2 = hb1 || y1 != x1;
    const u64 ma1 = __ballot(ha1), mb1 = __ballot(hb1), ma2 = __ballot(ha2), mb2 = __ballot(hb2);
    if (EMIT) {
      const u64 row = begin + (u64)t * kTileRecs + 2 * lane;
      const u64 k = p1 + c1 + (u64)(__popcll(ma1 & lt_mask) + __popcll(mb1 & lt_mask));
      const u64 q = p2 + c2 + (u64)(__popcll(ma2 & lt_mask) + __popcll(mb2 & lt_mask));
      if (ha1) emit(k, x0, row, q);
      if (hb1) emit(k + (ha1 ? 1 : 0), y0, row + 1, q + (ha2 ? 1 : 0));
    }
    c1 += (u64)(__popcll(ma1) + __popcll(mb1));
    c2 += (u64)(__popcll(ma2) + __popcll(mb2));
    const u64* last = reinterpret_cast<const u64*>(tile + (kTileRecs - 1) * 24);
    cb = last[0]; cu = last[1];                              // same address in every lane: one broadcast read
    have_prev = true;
    if (!more) break;
    ++t;
    a0 = b0; a1 = b1; a2 = b2;
  }
}

extern "C" __global__ void __launch_bounds__(kSortThreads, 8)
ibu_k_runs_count(const u64* __restrict__ recs, SegPlan sp, u32* __restrict__ seg_heads /*[2][nseg]*/) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kSortWaves * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  const u32 seg = blockIdx.x * kSortWaves + wib;
  if (seg >= sp.nseg) return;                               // wave-uniform
  u64 c1, c2;
  runs_segment<false>(recs, sp, seg, lds + wib * kTileBytes, lane, 0, 0, c1, c2, [](u64, u64, u64, u64) {});
  if (lane == 0) { seg_heads[seg] = (u32)c1; seg_heads[sp.nseg + seg] = (u32)c2; }
}
// ... and keeping the segment's first kStashHeads heads for the emit pass (see the top of the file)
extern "C" __global__ void __launch_bounds__(kSortThreads, 8)
ibu_k_runs_count_stash(const u64* __restrict__ recs, SegPlan sp, u32* __restrict__ seg_heads /*[2][nseg]*/, RunStash* __restrict__ stash /*[nseg][kStashHeads]*/) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kSortWaves * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  const u32 seg = blockIdx.x * kSortWaves + wib;
  if (seg >= sp.nseg) return;                               // wave-uniform
  u64 c1, c2;
  const u64 row0 = seg_first_row(sp, seg);
  RunStash* mine = stash + (size_t)seg * kStashHeads;
  runs_segment<true>(recs, sp, seg, lds + wib * kTileBytes, lane, 0, 0, c1, c2, [=](u64 k, u64 b, u64 row, u64 q) {
    if (k < kStashHeads) { RunStash e; e.barcode = b; e.row_off = (u32)(row - row0); e.pair_local = (u32)q; mine[k] = e; }
  });
  if (lane == 0) { seg_heads[seg] = (u32)c1; seg_heads[sp.nseg + seg] = (u32)c2; }
}

extern "C" __global__ void __launch_bounds__(kSortThreads, 8)
ibu_k_runs_emit(const u64* __restrict__ recs, SegPlan sp, const u64* __restrict__ seg_base /*[2][nseg], scanned*/,
                const u32* __restrict__ seg_heads /*[2][nseg]*/, const RunStash* __restrict__ stash /*[nseg][kStashHeads] or null*/,
                u64* __restrict__ barcodes, u64* __restrict__ starts, u64* __restrict__ pair_rank) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kSortWaves * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  const u32 seg = blockIdx.x * kSortWaves + wib;
  if (seg >= sp.nseg) return;
  if (stash) {                                              // wave-uniform: the heads the count pass kept are all of them
    const u32 heads = seg_heads[seg];
    if (heads <= kStashHeads) {
      if (lane < heads) {
        const RunStash e = stash[(size_t)seg * kStashHeads + lane];
        const u64 k = seg_base[seg] + lane;
        barcodes[k] = e.barcode;
        starts[k] = seg_first_row(sp, seg) + e.row_off;
        if (pair_rank) pair_rank[k] = seg_base[sp.nseg + seg] + e.pair_local;
      }
      return;
    }
  }
  u64 c1, c2;
  // seg_base: runs / pairs that start before this segment
  runs_segment<true>(recs, sp, seg, lds + wib * kTileBytes, lane, seg_base[seg], seg_base[sp.nseg + seg], c1, c2,
                     [=](u64 k, u64 b, u64 row, u64 q) {
                       barcodes[k] = b;
                       starts[k] = row;                      // first record of run k
                       if (pair_rank) pair_rank[k] = q;      // (barcode, umi) pairs that start before it
                     });
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

// seg_heads u32 [2][nseg] -> seg_base u64 [2][nseg] (exclusive prefix per row) and the two row totals.  One workgroup per row;
// u64 sums: 2^32 or more records (and then possibly 2^32 or more runs) fit in 288 GB.
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_runs_scan(const u32* __restrict__ seg_heads, u32 nseg, u64* __restrict__ seg_base, u64* __restrict__ totals) {
  __shared__ u32 wsum[kSortWaves];
  const u32* row = seg_heads + (size_t)blockIdx.x * nseg;
  u64* out = seg_base + (size_t)blockIdx.x * nseg;
  u64 carry = 0;
  for (u32 base = 0; base < nseg; base += 4 * kSortThreads) {   // 1024 segments per round: at most 2^23 heads, fits u32
    const u32 i0 = base + 4 * threadIdx.x;
    u32 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i0 + j < nseg ? row[i0 + j] : 0;
    u32 tot;
    u32 ex = block_exclusive_scan(v[0] + v[1] + v[2] + v[3], wsum, &tot);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0 + j < nseg) out[i0 + j] = carry + ex;
      ex += v[j];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

static SegPlan seg_plan(const LaunchCfg& cfg, const void* recs, size_t n) {
  const Span span[1] = {{recs, 24}};
  const RowSplit rs = split_rows(cfg, span, 1, n, kTileRecs);    // an 8-B aligned base peels exactly one record
  return {(u64)rs.head, (u64)rs.main, (u64)n, runs_nseg(rs.main)};
}
static inline size_t seg_base_offset(u32 nseg) { return 64 + 2 * sizeof(u32) * (size_t)nseg + ((2 * sizeof(u32) * (size_t)nseg) & 4); }
static inline size_t stash_offset(u32 nseg) { return (seg_base_offset(nseg) + 2 * sizeof(u64) * (size_t)nseg + 15) & ~(size_t)15; }
size_t runs_scratch_bytes(size_t n) {
  const u32 nseg = runs_nseg(n);                             // main <= n
  return stash_offset(nseg) + sizeof(RunStash) * kStashHeads * (size_t)nseg;   // totals u64[2] | seg_heads u32[2][nseg] | pad | seg_base u64[2][nseg] | pad | stash
}
// Pass 1 + scan.  Leaves the scanned table in `scratch`; totals[0] = runs, totals[1] = (barcode, umi) pairs
// are read back by the caller from scratch[0..15] (u64 each).
// keep_heads: the emit pass follows (launch_runs_emit with from_stash = true); false: a size query.
hipError_t launch_runs_count(const LaunchCfg& cfg, const void* recs, size_t n, void* scratch, size_t scratch_bytes, bool keep_heads, hipStream_t st) {
  (void)hipGetLastError();
  if (n == 0 || n / kSegRecs + 2 >= (1ull << 31) || scratch_bytes < runs_scratch_bytes(n)) return hipErrorInvalidValue;
  const SegPlan sp = seg_plan(cfg, recs, n);
  u64* totals = static_cast<u64*>(scratch);
  u32* heads = reinterpret_cast<u32*>(static_cast<uint8_t*>(scratch) + 64);
  u64* base = reinterpret_cast<u64*>(static_cast<uint8_t*>(scratch) + seg_base_offset(sp.nseg));
  if (keep_heads)
    hipLaunchKernelGGL(ibu_k_runs_count_stash, dim3((sp.nseg + kSortWaves - 1) / kSortWaves), dim3(kSortThreads), 0, st, (const u64*)recs, sp,
                       heads, reinterpret_cast<RunStash*>(static_cast<uint8_t*>(scratch) + stash_offset(sp.nseg)));
  else
    hipLaunchKernelGGL(ibu_k_runs_count, dim3((sp.nseg + kSortWaves - 1) / kSortWaves), dim3(kSortThreads), 0, st, (const u64*)recs, sp,
                       heads);
  hipLaunchKernelGGL(ibu_k_runs_scan, dim3(2), dim3(kSortThreads), 0, st, (const u32*)heads, sp.nseg, base, totals);
  return hipGetLastError();
}
hipError_t launch_runs_emit(const LaunchCfg& cfg, const void* recs, size_t n, const void* scratch, bool from_stash, void* run_scratch, uint64_t n_runs,
                            uint64_t n_pairs, uint64_t* barcodes, uint64_t* counts, uint64_t* uniq, hipStream_t st) {
  (void)hipGetLastError();
  const SegPlan sp = seg_plan(cfg, recs, n);
  const u32* heads = reinterpret_cast<const u32*>(static_cast<const uint8_t*>(scratch) + 64);
  const u64* base = reinterpret_cast<const u64*>(static_cast<const uint8_t*>(scratch) + seg_base_offset(sp.nseg));
  const RunStash* stash = from_stash ? reinterpret_cast<const RunStash*>(static_cast<const uint8_t*>(scratch) + stash_offset(sp.nseg)) : nullptr;
  u64* starts = static_cast<u64*>(run_scratch);             // n_runs entries each (run_scratch_bytes)
  u64* pair_rank = uniq ? starts + n_runs : nullptr;
  hipLaunchKernelGGL(ibu_k_runs_emit, dim3((sp.nseg + kSortWaves - 1) / kSortWaves), dim3(kSortThreads), 0, st, (const u64*)recs, sp,
                     base, heads, stash, (u64*)barcodes, starts, pair_rank);
  u64 blocks = (n_runs + 255) / 256;
  const u64 cap = (u64)cfg.cus * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(ibu_k_runs_finish, dim3((u32)(blocks ? blocks : 1)), dim3(256), 0, st, (const u64*)starts, (const u64*)pair_rank,
                     (u64)n_runs, (u64)n, (u64)n_pairs, (u64*)counts, (u64*)uniq);
  return hipGetLastError();
}
size_t runs_emit_scratch_bytes(uint64_t n_runs) { return 16 * (size_t)(n_runs ? n_runs : 1); }

}  // namespace ibu
