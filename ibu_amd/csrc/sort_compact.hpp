// sort_compact.hpp — compact keys: plan, compress / expand, element passes.
// Part of sort.hip's translation unit: included there, inside namespace ibu, after the shared definitions (kSortThreads, kBins,
// rec_less, ...).  Not a header to include anywhere else.
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
// The key range of a 12-byte element among up to 255 splitter elements staged in LDS (the multi-GPU sort: sort.hip, "partition
// first"): how many splitters are not above it — the element as a 96-bit integer orders like the record.
__device__ __forceinline__ u32 range_of(const u32* sp, u32 nsplit, const EV<3>& e) {
  u32 lo = 0, hi = nsplit;
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    const u32 s2 = sp[3 * mid + 2], s1 = sp[3 * mid + 1], s0 = sp[3 * mid];
    const bool le = s2 != e.w[2] ? s2 < e.w[2] : (s1 != e.w[1] ? s1 < e.w[1] : s0 <= e.w[0]);
    if (le) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// STAMP (W = 3, at most 11 varying bytes): every element also gets its key range among the `nsplit` splitter elements at `split`
// in its free top byte, and the digit stream holds the ranges — the compress and the stamp step of the multi-GPU sort in one read
// of the records (with CENSUS: and the exact census of a plan that was guessed from samples).
template <bool CENSUS, int W, bool STAMP = false>
__global__ void __launch_bounds__(kBlock, CENSUS ? (W == 4 ? 5 : 7) : (STAMP ? 7 : 8))   // the launcher keeps at most 7 workgroups per CU resident (LaunchCfg); 16-byte elements with the census need 84 VGPRs
ibu_k_sort_compress(const uint8_t* __restrict__ recs, u32 ntiles, CompactPlan pl, u32 first_byte, ElemT<W>* __restrict__ out,
                    uint8_t* __restrict__ digits, u64* __restrict__ census, const ElemT<3>* __restrict__ split = nullptr, u32 nsplit = 0) {
  static_assert(!STAMP || W == 3, "ranges are stamped into 12-byte elements");
  constexpr int kSlice = CENSUS ? kSliceBytes : kTileBytes;  // with the census: the record in front of the tile is staged too (ibu_k_sort_census)
  constexpr int kLead = CENSUS ? kPrevBytes : 0;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kSlice];
  __shared__ u32 sp[STAMP ? 3 * 256 : 1];
  if constexpr (STAMP) {
    for (u32 i = threadIdx.x; i < 3 * nsplit; i += kBlock) sp[i] = reinterpret_cast<const u32*>(split)[i];
    __syncthreads();                                         // (before any wave leaves)
  }
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
        EV<W> e0 = compress_rec<W>(r[0], r[1], r[2], pl), e1 = compress_rec<W>(q[0], q[1], q[2], pl);
        const size_t row = (size_t)t * kTileRecs + lane;
        if constexpr (STAMP) {
          const u32 g0 = range_of(sp, nsplit, e0), g1 = range_of(sp, nsplit, e1);
          e0.w[2] |= g0 << 24;
          e1.w[2] |= g1 << 24;
          st_elem<W>(out + row, e0);
          st_elem<W>(out + row + kWave, e1);
          digits[row] = (uint8_t)g0;
          digits[row + kWave] = (uint8_t)g1;
          return;
        }
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
  const u32 ntiles = (u32)(((u64)n + T - 1) / T);
  struct Win { EV<W> v[ROUNDS]; IDX mypos; };
  // 1. every lane loads its elements (unconditional, clamped) and this tile's first output position per bin
  auto load = [&](u32 tile, Win& win) {
    const IDX tbase = (IDX)((u64)tile * T);
    const u32 cnt = n - tbase < (IDX)T ? (u32)(n - tbase) : (u32)T;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const u32 slot = wib * PER_WAVE + r * kWave + lane;
      win.v[r] = ld_elem<W>(src + tbase + (slot < cnt ? slot : cnt - 1));
    }
    win.mypos = pos[(size_t)tile * kBins + (tid & (kBins - 1))];
  };
  auto body = [&](u32 tile, const Win& win) {
  const EV<W>* v = win.v;
  const IDX mypos = win.mypos;
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
    const DigitPeers pe = match_digit<W == 4>(d, __ballot(valid));
    const u32 before = pe.before;
    const u32 prev = valid ? whist[wib * kBins + d] : 0;
    wave_lds_fence();                                        // every lane has read before the leaders write
    if (valid && before == 0) whist[wib * kBins + d] = prev + pe.count;
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
