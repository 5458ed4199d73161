// sort.hip — device sort of 24-byte records by (barcode, umi, index): the order `derive(Ord)` gives
// Record (src/constructs/record.rs:58-66) and the header's sorted flag promises (header.rs:111-113).
//
// Stable LSD radix sort, 8-bit digits, over the 192-bit key — but only over the digits that actually
// vary: a census kernel ORs and ANDs each field over all records, and a digit whose bits are equal in
// OR and AND is constant, so its pass would be the identity and is skipped.  16-base barcodes, 12-base
// UMIs and indices below 2^32 need 4 + 3 + 4 passes instead of 24 — and 4 + 3 when the input already runs in index
// order (the sort is stable and the index is the least significant field), which the same census detects.
//
// One pass = histogram (one 256-bin row per 32 Ki-record chunk) -> exclusive scan of the
// [bin][chunk] table -> scatter.  The scatter stages a 1 Ki-record tile in LDS, ranks it with
// wave-level match-any (8 ballots per record), permutes the tile IN LDS into digit order and
// writes it out as runs of consecutive 8-byte words, so a chunk's 256 output runs are each written
// front to back by one workgroup (L2 merges the pieces) instead of as scattered 24-byte records.
// HBM traffic per pass: 24 B read (histogram) + 24 B read + 24 B write (scatter) per record.
// Where the scatter's time goes (profiles/README.md, r01_e): with the write-out replaced by a linear tile store the
// kernel runs at 5.2 TB/s (staging, ranking and the LDS permutation are not the limit); the real, run-wise write-out
// costs +60 %.  Tile size (512..2048), workgroups per CU (2..5) and nontemporal stores do not move it (or hurt).
#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {

static constexpr int kSortThreads = 256;
static constexpr int kSortWaves = kSortThreads / kWave;       // 4
#ifndef IBU_SORT_TILE
#define IBU_SORT_TILE 1024
#endif
static constexpr int kSortTile = IBU_SORT_TILE;               // records per LDS tile (24 KiB at 1024)
static constexpr int kSortRounds = kSortTile / kSortThreads;  // records per thread per tile
static constexpr int kSortTilesPerChunk = 32768 / kSortTile;
static constexpr int kSortChunk = kSortTile * kSortTilesPerChunk;  // records per histogram row
static constexpr int kBins = 256;

// scratch layout (bytes): census u64[8] @0 (OR x3, AND x3, index-order flag, any-inversion flag) | rowsum u32[256] @64 | binbase u32[256] @1088 | table @2112
static constexpr size_t kOffRowsum = 64, kOffBinbase = kOffRowsum + 4 * kBins, kOffTable = kOffBinbase + 4 * kBins;

__device__ __forceinline__ u64 shfl_xor64(u64 v, int m) {
  u32 lo = __shfl_xor((u32)v, m), hi = __shfl_xor((u32)(v >> 32), m);
  return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 shfl_up64_1(u64 v) {
  u32 lo = __shfl_up((u32)v, 1), hi = __shfl_up((u32)(v >> 32), 1);
  return ((u64)hi << 32) | lo;
}

// ---- census: OR and AND of each field --------------------------------------------------------------
extern "C" __global__ void ibu_k_sort_census_init(u64* c) {
  if (threadIdx.x < 3) c[threadIdx.x] = 0;
  else if (threadIdx.x < 6) c[threadIdx.x] = ~0ull;
  else if (threadIdx.x < 8) c[threadIdx.x] = 0;  // [6]: some index smaller than its predecessor's; [7]: some record smaller
}
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_census(const u64* __restrict__ recs, u64 n, u64* __restrict__ c) {
  u64 o[3] = {0, 0, 0}, a[3] = {~0ull, ~0ull, ~0ull};
  bool index_drops = false, order_drops = false;
  const u64 stride = (u64)gridDim.x * kSortThreads;
  const u32 lane = threadIdx.x & (kWave - 1);
  // wave-uniform trip count (lanes past n carry neutral values) so the neighbour shuffles below are well defined
  for (u64 i0 = (u64)blockIdx.x * kSortThreads + (threadIdx.x & ~(u32)(kWave - 1)); i0 < n; i0 += stride) {
    const u64 i = i0 + lane;
    const bool valid = i < n;
    const u64 b = valid ? recs[3 * i] : 0, u = valid ? recs[3 * i + 1] : 0, x = valid ? recs[3 * i + 2] : 0;
    if (valid) { o[0] |= b; o[1] |= u; o[2] |= x; a[0] &= b; a[1] &= u; a[2] &= x; }
    // predecessor = the previous lane's record (one shuffle per half word); lane 0 reads it from memory
    u64 pb = shfl_up64_1(b), pu = shfl_up64_1(u), px = shfl_up64_1(x);
    if (lane == 0 && valid && i > 0) { pb = recs[3 * (i - 1)]; pu = recs[3 * (i - 1) + 1]; px = recs[3 * (i - 1) + 2]; }
    if (valid && i > 0) {
      if (x < px) index_drops = true;       // input not in index order: the index passes are needed
      if (b != pb ? b < pb : (u != pu ? u < pu : x < px)) order_drops = true;  // not already sorted
    }
  }
  if (__ballot(index_drops) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(&c[6], 1ull);
  if (__ballot(order_drops) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(&c[7], 1ull);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int f = 0; f < 3; ++f) { o[f] |= shfl_xor64(o[f], m); a[f] &= shfl_xor64(a[f], m); }
  if ((threadIdx.x & (kWave - 1)) == 0)
#pragma unroll
    for (int f = 0; f < 3; ++f) { atomicOr(&c[f], o[f]); atomicAnd(&c[3 + f], a[f]); }
}

// ---- histogram: one 256-bin row per chunk, table[bin][chunk] ---------------------------------------
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_hist(const u64* __restrict__ recs, u64 n, u32 field, u32 shift, u32 nchunks, u32* __restrict__ table) {
  __shared__ u32 h[kSortWaves][kBins];
  const u32 tid = threadIdx.x, wib = tid >> 6;
#pragma unroll
  for (int w = 0; w < kSortWaves; ++w) h[w][tid] = 0;
  __syncthreads();
  const u64 base = (u64)blockIdx.x * kSortChunk;
  const u64 end = base + kSortChunk < n ? base + kSortChunk : n;
  for (u64 i = base + tid; i < end; i += kSortThreads)
    atomicAdd(&h[wib][(u32)(recs[3 * i + field] >> shift) & 255u], 1u);
  __syncthreads();
  u32 s = 0;
#pragma unroll
  for (int w = 0; w < kSortWaves; ++w) s += h[w][tid];
  table[(size_t)tid * nchunks + blockIdx.x] = s;
}

// ---- exclusive scans --------------------------------------------------------------------------------
// Block-wide exclusive scan of one u32 per thread (256 threads); returns the exclusive prefix and
// leaves the block total in *total.
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* wsum /*[kSortWaves] shared*/, u32* total) {
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const u32 t = __shfl_up(inc, d);
    if (lane >= (u32)d) inc += t;
  }
  if (lane == kWave - 1) wsum[wib] = inc;
  __syncthreads();
  u32 off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kSortWaves; ++w) {
    const u32 s = wsum[w];
    if ((u32)w < wib) off += s;
    tot += s;
  }
  __syncthreads();  // wsum may be reused by the caller's next scan
  *total = tot;
  return off + inc - v;
}

// Row `bin` of the table -> exclusive prefix within the row (in place) and the row total.
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
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_scan_bins(const u32* __restrict__ rowsum, u32* __restrict__ binbase) {
  __shared__ u32 wsum[kSortWaves];
  u32 tot;
  binbase[threadIdx.x] = block_exclusive_scan(rowsum[threadIdx.x], wsum, &tot);
}

// ---- scatter ----------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_scatter(const u64* __restrict__ src, u64* __restrict__ dst, u64 n, u32 field, u32 shift, u32 nchunks,
                   const u32* __restrict__ table, const u32* __restrict__ binbase) {
  __shared__ __attribute__((aligned(16))) u64 stage[kSortTile * 3];  // the tile, first in input then in digit order
  __shared__ u32 dest[kSortTile];          // global record index of each slot of the permuted tile
  __shared__ u32 whist[kSortWaves][kBins]; // per wave: running count while ranking, then base slot of (wave, bin)
  __shared__ u32 cursor[kBins];            // next global record index of each bin for this chunk
  __shared__ u32 gdelta[kBins];            // global index of a slot = gdelta[bin] + slot
  __shared__ u32 wsum[kSortWaves];
  const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wib = tid >> 6;
  const u64 lt_mask = (1ull << lane) - 1;
  cursor[tid] = binbase[tid] + table[(size_t)tid * nchunks + blockIdx.x];
  const u64 rec0 = (u64)blockIdx.x * kSortChunk;
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;

  // Full, 16-B aligned tiles are prefetched one tile ahead into registers (the global-load latency of tile
  // k+1 hides behind the ranking of tile k); the ragged last tile of the input takes the plain 8-byte path.
  constexpr int kPre = kSortTile * 24 / 16 / kSortThreads;   // dwordx4 per thread per tile
  u32x4 pre[kPre];
  auto full_tile = [&](u64 tb) { return aligned && tb + kSortTile <= n; };
  auto prefetch = [&](u64 tb) {
    const u32x4* g = reinterpret_cast<const u32x4*>(src + 3 * tb);
#pragma unroll
    for (int k = 0; k < kPre; ++k) pre[k] = ld16(g + tid + kSortThreads * k);
  };
  if (full_tile(rec0)) prefetch(rec0);

  for (int tile = 0; tile < kSortTilesPerChunk; ++tile) {
    const u64 tbase = rec0 + (u64)tile * kSortTile;
    if (tbase >= n) break;                                   // block-uniform
    const u32 cnt = n - tbase < (u64)kSortTile ? (u32)(n - tbase) : (u32)kSortTile;
    __syncthreads();                                         // previous tile fully written out
    // a. stage the tile (coalesced) and clear the per-wave counters
    if (full_tile(tbase)) {
      u32x4* s = reinterpret_cast<u32x4*>(stage);
#pragma unroll
      for (int k = 0; k < kPre; ++k) s[tid + kSortThreads * k] = pre[k];
      const u64 nb = tbase + kSortTile;
      if (tile + 1 < kSortTilesPerChunk && full_tile(nb)) prefetch(nb);
    } else {
      for (u32 w = tid; w < 3 * cnt; w += kSortThreads) stage[w] = src[3 * tbase + w];
    }
#pragma unroll
    for (int k = 0; k < kBins / kWave; ++k) whist[wib][lane + kWave * k] = 0;
    __syncthreads();
    // b. rank every record among the records of its wave with the same digit (stable: slot order)
    u64 r0[kSortRounds], r1[kSortRounds], r2[kSortRounds];
    u32 dig[kSortRounds], rk[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
      const u32 slot = wib * (kSortTile / kSortWaves) + r * kWave + lane;
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
      const u32 prev = valid ? whist[wib][d] : 0;
      wave_lds_fence();                                      // every lane has read before the leaders write
      if (valid && before == 0) whist[wib][d] = prev + (u32)__popcll(m);
      wave_lds_fence();
      dig[r] = d;
      rk[r] = prev + before;
    }
    __syncthreads();                                         // counters complete; tile now lives in registers
    // c. bin totals of the tile -> slot bases per (wave, bin), global index delta per bin
    {
      const u32 c0 = whist[0][tid], c1 = whist[1][tid], c2 = whist[2][tid], c3 = whist[3][tid];
      u32 tot;
      const u32 tb = block_exclusive_scan(c0 + c1 + c2 + c3, wsum, &tot);
      whist[0][tid] = tb; whist[1][tid] = tb + c0; whist[2][tid] = tb + c0 + c1; whist[3][tid] = tb + c0 + c1 + c2;
      const u32 g = cursor[tid];
      gdelta[tid] = g - tb;                                  // wraps harmlessly: slot >= tb for this bin
      cursor[tid] = g + c0 + c1 + c2 + c3;
    }
    __syncthreads();
    // d. permute the tile into digit order inside LDS
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
      const u32 slot = wib * (kSortTile / kSortWaves) + r * kWave + lane;
      if (slot < cnt) {
        const u32 p = whist[wib][dig[r]] + rk[r];
        stage[3 * p] = r0[r]; stage[3 * p + 1] = r1[r]; stage[3 * p + 2] = r2[r];
        dest[p] = gdelta[dig[r]] + p;
      }
    }
    __syncthreads();
    // e. write out: consecutive lanes write consecutive 8-byte words of each run
    for (u32 w = tid; w < 3 * cnt; w += kSortThreads) {
      const u32 s = (u32)(((u64)w * 0xAAAAAAABull) >> 33);   // w / 3
      dst[3 * (u64)dest[s] + (w - 3 * s)] = stage[w];  // plain store: L2 merges the pieces of a run (nontemporal: -20 %)
    }
  }
}

// =====================================================================================================
size_t sort_scratch_bytes(const LaunchCfg&, size_t n) {
  const size_t nchunks = (n + kSortChunk - 1) / kSortChunk;
  return kOffTable + sizeof(u32) * kBins * (nchunks ? nchunks : 1);
}

// Not purely asynchronous: the census result comes back to the host (one 48-byte read) to pick the
// passes; everything after that is queued on `st`.
hipError_t launch_sort_records(const LaunchCfg& cfg, void* recs, void* tmp, size_t n, void* scratch,
                               size_t scratch_bytes, hipStream_t st) {
  (void)hipGetLastError();
  if (n < 2) return hipSuccess;
  if (n >= (1ull << 32) || scratch_bytes < sort_scratch_bytes(cfg, n)) return hipErrorInvalidValue;
  uint8_t* sc = static_cast<uint8_t*>(scratch);
  u64* census = reinterpret_cast<u64*>(sc);
  u32* rowsum = reinterpret_cast<u32*>(sc + kOffRowsum);
  u32* binbase = reinterpret_cast<u32*>(sc + kOffBinbase);
  u32* table = reinterpret_cast<u32*>(sc + kOffTable);
  const u32 nchunks = (u32)((n + kSortChunk - 1) / kSortChunk);

  hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(64), 0, st, census);
  u64 cblocks = (n + kSortThreads - 1) / kSortThreads;
  const u64 ccap = (u64)cfg.cus * 8;
  if (cblocks > ccap) cblocks = ccap;
  hipLaunchKernelGGL(ibu_k_sort_census, dim3((u32)cblocks), dim3(kSortThreads), 0, st, (const u64*)recs, (u64)n, census);
  u64 c[8];
  hipError_t e = hipMemcpyAsync(c, census, sizeof c, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  if (c[7] == 0) return hipSuccess;  // no record is smaller than its predecessor: already sorted

  u64* src = static_cast<u64*>(recs);
  u64* dst = static_cast<u64*>(tmp);
  static const int kFieldOrder[3] = {2, 1, 0};  // least significant first: index, umi, barcode
  for (int fo = 0; fo < 3; ++fo) {
    const int f = kFieldOrder[fo];
    // The sort is stable and the index is the LEAST significant field: if the input already runs in non-decreasing
    // index order (the usual case: records are written in read order), ties on (barcode, umi) keep that order and
    // the index passes are the identity — 7 passes instead of 11 at 16/12.
    if (f == 2 && c[6] == 0) continue;
    const u64 varying = c[f] ^ c[3 + f];        // bits that differ between some two records
    for (u32 shift = 0; shift < 64; shift += 8) {
      if (((varying >> shift) & 255u) == 0) continue;  // constant digit: the pass would be the identity
      hipLaunchKernelGGL(ibu_k_sort_hist, dim3(nchunks), dim3(kSortThreads), 0, st, (const u64*)src, (u64)n, (u32)f, shift,
                         nchunks, table);
      hipLaunchKernelGGL(ibu_k_sort_scan_rows, dim3(kBins), dim3(kSortThreads), 0, st, table, nchunks, rowsum);
      hipLaunchKernelGGL(ibu_k_sort_scan_bins, dim3(1), dim3(kSortThreads), 0, st, (const u32*)rowsum, binbase);
      hipLaunchKernelGGL(ibu_k_sort_scatter, dim3(nchunks), dim3(kSortThreads), 0, st, (const u64*)src, dst, (u64)n, (u32)f,
                         shift, nchunks, (const u32*)table, (const u32*)binbase);
      u64* t = src; src = dst; dst = t;
    }
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (src != static_cast<u64*>(recs)) return launch_copy(cfg, src, recs, n * 24, st);  // odd number of passes
  return hipSuccess;
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
