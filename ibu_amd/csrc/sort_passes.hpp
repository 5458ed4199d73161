// sort_passes.hpp — one radix pass on 24-byte records: tile counts, scan, scatter.
// Part of sort.hip's translation unit: included there, inside namespace ibu, after the shared definitions (kSortThreads, kBins,
// rec_less, ...).  Not a header to include anywhere else.
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
#define IBU_TILES_PER_BLOCK 256   // the most (large inputs).  1024: the position walk of a block (ibu_k_sort_tilepos) took 0.30 ms per pass at 1e9 records; 256: 3 ms less per sort
#endif
static constexpr int kTilesPerBlock = IBU_TILES_PER_BLOCK;                   // tiles per scan block at most (sort.hip: tiles_per_block)
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
// 1. per block of tpb tiles: column sums.  Wave w takes tiles w, w+4, ...; lane l the bins 4l..4l+3.
extern "C" __global__ void __launch_bounds__(kSortThreads)
ibu_k_sort_blocksums(const uint16_t* __restrict__ counts, u32 ntiles, u32 tpb, u32* __restrict__ blocksum) {
  __shared__ u32 part[kSortWaves][kBins];
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  const u32 t0 = blockIdx.x * tpb, t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
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
ibu_k_sort_tilepos(const uint16_t* __restrict__ counts, u32 ntiles, u32 tpb, const u64* __restrict__ blockoff, const u64* __restrict__ binbase,
                   IDX* __restrict__ pos) {
  const u32 bin = threadIdx.x;
  const u32 t0 = blockIdx.x * tpb, t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
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
// DSTREAM: this pass's digit is not a key byte but comes FROM the side stream (digits[i] of record i of `src`: the key range a
// record belongs to — the multi-GPU sort's partition pass, sort.hip launch_partition_records; there is no next pass then: nfield > 2).
// WMODE 0: the real thing.  WMODE 2 (probe builds, -DIBU_SORT_PROBE, WRONG output): the permuted tile goes out linearly.
// DSTREAM: a separate instantiation — as a run-time case of the ordinary kernel the extra load made the register allocator aim for
// twice the occupancy the LDS allows and spill 256 bytes per lane (full-range (32,32) sort 0.063 -> 0.100 s until the code objects'
// metadata was looked at again).
template <int THREADS, int ROUNDS, class IDX, int WMODE, bool DSTREAM = false>
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
    u32 d;
    if constexpr (DSTREAM) d = valid ? (u32)digits[tbase + slot] : 0u;
    else d = (u32)(key >> shift) & 255u;
    const DigitPeers pe = match_digit(d, __ballot(valid));
    const u32 before = pe.before;
    const u32 prev = valid ? whist[wib * kBins + d] : 0;
    wave_lds_fence();                                        // every lane has read before the leaders write
    if (valid && before == 0) whist[wib * kBins + d] = prev + pe.count;
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
