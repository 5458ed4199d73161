// sort_census.hpp — census of the varying key bytes, the sorted check.
// Part of sort.hip's translation unit: included there, inside namespace ibu, after the shared definitions (kSortThreads, kBins,
// rec_less, ...).  Not a header to include anywhere else.
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
  const RowSplit rs = split_rows(cfg, sp, 1, n, kTileRecs);   // an 8-B aligned base peels exactly one record
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
