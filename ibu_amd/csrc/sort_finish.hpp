// sort_finish.hpp — prefix + finish: the finishing kernels and the run-length estimate.
// Part of sort.hip's translation unit: included there, inside namespace ibu, after the shared definitions (kSortThreads, kBins,
// rec_less, ...).  Not a header to include anywhere else.
// =====================================================================================================
// PREFIX + FINISH (round 3): wide keys.  LSD over all varying bytes costs a pass per byte — 24 passes of ~50 B/record for
// full-range (32,32) records.  But once the records are sorted by their most significant P varying bytes (P LSD passes,
// least significant of the P first), everything that is left to decide lies INSIDE runs of equal prefix ("segments"),
// and for P = ceil(log256(n / 8)) a segment of well-spread keys holds a handful of records.  ibu_k_sort_finish completes the
// sort in ONE more pass: a workgroup takes the segments that START in its tile of T records (from the first segment head
// in the tile to the first head at or behind the tile's end — up to M records of look-ahead), stages them in LDS, marks the
// segment heads in a bitmap, lets every record find its segment there and its place inside it under the full 24-byte key
// (segments of one and two records on the spot, longer ones from a worklist by counting — quadratic in the segment length,
// which is why segments longer than M are not ranked), permutes in LDS and writes the chunk out as half records.
// 4 passes + 1 instead of 24 at 1e9 records.
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
// 1024-record tiles + 256 of look-ahead: 36 KiB of LDS, four workgroups per CU.  1e9 full-range (32,32) records (profiles r03_o):
// (2048, 512) 17.2 ms, (1024, 512) 15.7, (1536, 256) 12.8, (1024, 256) 12.4.
static constexpr int kFinishT = IBU_FINISH24_T, kFinishM = IBU_FINISH24_M;
// The run of equal prefix that element i of the window belongs to, from the head bitmap (bit i of hw: element i differs from its
// predecessor in a prefix byte): s0 = its head (found: there is one at or below i), nx = the next head above i (kNoHead: none in the
// window).  The words at and next to i's answer for runs of up to 64 elements; longer ones walk on, word by word.
static constexpr u32 kNoHead = 0xFFFFFFFFu;
__device__ __forceinline__ void run_around(const u64* hw /*[-1 .. nwords]*/, u32 i, u32 nwords, u32& s0, bool& found, u32& nx) {
  const u32 wi = i >> 6, bi = i & 63u;
  const u64 w0 = hw[wi];
  const u64 below = w0 & ((2ull << bi) - 1ull);              // heads at 64 wi .. i (bi = 63: 2 << 63 wraps to 0, minus 1 = all)
  found = true;
  if (below) s0 = (wi << 6) + 63u - (u32)__builtin_clzll(below);
  else {
    const u64 wp = hw[(int)wi - 1];
    if (wp) s0 = ((wi - 1u) << 6) + 63u - (u32)__builtin_clzll(wp);
    else {
      found = false;
      s0 = 0;
      for (int k = (int)wi - 2; k >= 0; --k) {
        const u64 w = hw[k];
        if (w) { s0 = ((u32)k << 6) + 63u - (u32)__builtin_clzll(w); found = true; break; }
      }
    }
  }
  const u64 above = (w0 >> bi) >> 1;                         // bit 0: element i + 1
  if (above) nx = i + 1u + (u32)__builtin_ctzll(above);
  else {
    const u64 wn = hw[wi + 1u];
    if (wn) nx = ((wi + 1u) << 6) + (u32)__builtin_ctzll(wn);
    else {
      nx = kNoHead;
      for (u32 k = wi + 2u; k < nwords; ++k) {
        const u64 w = hw[k];
        if (w) { nx = (k << 6) + (u32)__builtin_ctzll(w); break; }
      }
    }
  }
}
// record a orders before record b under the full key (mask arithmetic, no branches); tie: what equal records answer
__device__ __forceinline__ u32 rec_before(u64 a0, u64 a1, u64 a2, u64 b0, u64 b1, u64 b2, u32 tie) {
  const u32 lt0 = a0 < b0, eq0 = a0 == b0, lt1 = a1 < b1, eq1 = a1 == b1, lt2 = a2 < b2, eq2 = a2 == b2;
  return lt0 | (eq0 & (lt1 | (eq1 & (lt2 | (eq2 & tie)))));
}
template <int T, int M>
struct FinishShape {
  static constexpr int L = T + M;                             // records staged per workgroup (+ 1 in front)
  static constexpr int HW = (L + 63) / 64;                     // 64-bit words of the head bitmap (one zero word in front, one behind)
  // LDS: stage 24 (L + 1) | head bitmap u64 [HW + 2] | worklist u16 [L] | targets u16 [L] | misc 16 x u32
  static constexpr size_t lds = 24 * (size_t)(L + 1) + 8 * (size_t)(HW + 2) + 2 * (size_t)L + 2 * (size_t)L + 64;
};
// PERSIST: persistent grid, the next tile's window prefetched into a second register set while this one is worked on (needs
// 16-byte aligned records; the one-tile form takes any 8-byte aligned input: a shard at an odd record).
template <int T, int M, bool PERSIST>
__global__ void __launch_bounds__(kSortThreads, 4)   // 36 KiB of LDS: four workgroups per CU, if the registers allow (128 VGPRs)
ibu_k_sort_finish(const u64* __restrict__ src, u64* __restrict__ dst, u64 n, u64 pm0, u64 pm1, u64 pm2, u32* __restrict__ overflow) {
  typedef FinishShape<T, M> S;
  constexpr int L = S::L, PER = (L + kSortThreads - 1) / kSortThreads, CH = (3 * L / 2 + kSortThreads - 1) / kSortThreads;
  static_assert((T * 24) % 16 == 0, "tiles must start at 16-byte boundaries of an aligned array");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u64* stage = reinterpret_cast<u64*>(smem) + 3;             // record i of the window at stage[3 i]; record -1 = the one in front
  u64* hw = stage + 3 * L + 1;                               // head bitmap (bit i: record i starts a segment), hw[-1] and hw[HW] stay zero
  uint16_t* wl = reinterpret_cast<uint16_t*>(hw + S::HW + 1);  // worklist: the records of runs of three and more
  uint16_t* tgt = wl + L;                                    // where each record of [begin, end) goes
  u32* misc = reinterpret_cast<u32*>(tgt + L);               // [0] first head in the tile, [1] first head at / behind T, [2] inversion seen, [3] worklist length
  static_assert(T % 64 == 0 && M % 64 == 0, "a word of the head bitmap lies on one side of M and of T");
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
    if (tid < (u32)S::HW + 2u) hw[(int)tid - 1] = 0ull;
    if (tid < 4) misc[tid] = tid >= 2 ? 0u : 0xFFFFFFFFu;
    __syncthreads();
    // 2. segment heads: the prefix differs from the predecessor's (row 0 of the array is a head).  A wave's 64 heads are one word
    //    of the bitmap; misc[0]: first head among the tile's first M records, misc[1]: first head in the look-ahead [T, T + M).
    const u32 nwords = (len + 63u) >> 6;
#pragma unroll 1
    for (u32 i0 = tid - lane; i0 < len; i0 += kSortThreads) {
      const u32 i = i0 + lane;
      bool h = false;
      if (i < len) {
        const u64* r = stage + 3 * i;
        h = (base + i == 0) || (((r[0] ^ r[-3]) & pm0) | ((r[1] ^ r[-2]) & pm1) | ((r[2] ^ r[-1]) & pm2)) != 0;
      }
      const u64 hm = __ballot(h);
      if (lane == 0) {
        hw[i0 >> 6] = hm;
        const u64 lo = i0 < (u32)M ? hm : 0ull, hi = i0 >= (u32)T ? hm : 0ull;
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
    // 3. every record finds its run in the bitmap and its place in the run (ibu_k_sort_finish_elems, step 3: runs of one and two
    //    settled on the spot, longer short runs onto the worklist, long runs identity + order check)
    auto short_run = [&](u32 s0, bool found, u32 nx, u32& m) -> bool {
      const u32 stop = nx < end ? nx : end;
      m = stop - s0;
      return found && s0 >= begin && m <= (u32)M && (nx < end || end_is_head);
    };
    bool inversion = false;
#pragma unroll 1
    for (u32 i = begin + tid; i < end; i += kSortThreads) {
      const u64* me = stage + 3 * i;
      const u64 m0 = me[0], m1 = me[1], m2 = me[2];
      u32 s0, nx, m;
      bool found;
      run_around(hw, i, nwords, s0, found, nx);
      if (short_run(s0, found, nx, m)) {
        if (m >= 3u) {
          wl[atomicAdd(&misc[3], 1u)] = (uint16_t)i;          // its place comes in step 4
        } else {
          const u32 p = m == 2u ? s0 + (u32)(i == s0) : i;     // the other record of a pair (a run of one: itself, which counts nothing)
          const u64* o = stage + 3 * p;
          tgt[i] = (uint16_t)(s0 + rec_before(o[0], o[1], o[2], m0, m1, m2, (u32)(p < i)));
        }
      } else {                                                // part of a long run
        tgt[i] = (uint16_t)i;
        if (!((hw[i >> 6] >> (i & 63u)) & 1ull)) inversion = inversion || rec_before(m0, m1, m2, me[-3], me[-2], me[-1], 0u);
      }
    }
    if (inversion) misc[2] = 1u;
    __syncthreads();
    // 4. the worklist: rank by counting inside the run, four candidates per step (their LDS reads issued together; the lanes of a
    //    run read the same records: broadcasts), the comparison as mask arithmetic — the short-circuit form compiled to five
    //    branches per candidate and one LDS round trip per iteration: 159 ms per 1e9 records instead of ~15
    const u32 nwl = misc[3];
    for (u32 t2 = tid; t2 < nwl; t2 += kSortThreads) {
      const u32 i = wl[t2];
      u32 s0, nx, m;
      bool found;
      run_around(hw, i, nwords, s0, found, nx);
      (void)short_run(s0, found, nx, m);
      const u64* me = stage + 3 * i;
      const u64 m0 = me[0], m1 = me[1], m2 = me[2];
      const u32 s1 = s0 + m;
      u32 cnt = 0;
      for (u32 j = s0; j < s1; j += 4) {
        u64 cb[4], cu[4], cx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32 jj = j + q < s1 ? j + q : s1 - 1;        // clamped: in the window, not counted
          const u64* o = stage + 3 * jj;
          cb[q] = o[0]; cu[q] = o[1]; cx[q] = o[2];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cnt += rec_before(cb[q], cu[q], cx[q], m0, m1, m2, (u32)(j + q < i)) & (u32)(j + q < s1);   // ties: window order
      }
      tgt[i] = (uint16_t)(s0 + cnt);
    }
    __syncthreads();
    if (misc[2]) {                                            // a long run that is not in order: not this kernel's to sort
      if (tid == 0) *overflow = 1u;
      return;
    }
    // every record of the range into registers, then to its place: permuted in place
    u64 k0[PER], k1[PER], k2[PER];
    u32 to[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const u32 i = begin + tid + kSortThreads * r;
      to[r] = 0xFFFFFFFFu;
      if (i < end) {
        const u64* me = stage + 3 * i;
        k0[r] = me[0]; k1[r] = me[1]; k2[r] = me[2];
        to[r] = tgt[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r)
      if (to[r] != 0xFFFFFFFFu) {
        u64* o = stage + 3 * to[r];
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
  static constexpr int HW = (L + 63) / 64;                     // 64-bit words of the head bitmap (one zero word in front, one behind)
  static constexpr size_t stage_bytes = (4 * (size_t)W * (L + 1) + 7) & ~(size_t)7;
  static constexpr size_t lds = stage_bytes + 8 * (size_t)(HW + 2) + 2 * (size_t)L + 2 * (size_t)L + 64;
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
  u64* hw = reinterpret_cast<u64*>(smem + S::stage_bytes) + 1;   // head bitmap, hw[-1] and hw[HW] stay zero
  uint16_t* wl = reinterpret_cast<uint16_t*>(hw + S::HW + 1);    // worklist: the elements of runs of three and more
  uint16_t* tgt = wl + L;                                    // ... and where they go
  u32* misc = reinterpret_cast<u32*>(tgt + L);               // [0] [1] first heads, [2] inversion seen, [3] worklist length
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
    // 1. stage; the head bitmap starts empty
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
    if (tid < (u32)S::HW + 2u) hw[(int)tid - 1] = 0ull;
    if (tid < 4) misc[tid] = tid >= 2 ? 0u : 0xFFFFFFFFu;
    __syncthreads();
    // 2. heads: element i differs from its predecessor in a prefix byte.  A wave's 64 heads are one word of the bitmap; misc[0]: first
    //    head among the tile's first M elements, misc[1]: first head in the look-ahead.  (Steps 2 and 3 are rolled loops over LDS on
    //    purpose: unrolled over the register copies they cost 9000 lines of code and 228 bytes of scratch per lane.)
    const u32 nwords = (len + 63u) >> 6;
#pragma unroll 1
    for (u32 i0 = tid - lane; i0 < len; i0 += kSortThreads) {
      const u32 i = i0 + lane;
      bool h = false;
      if (i < len) {
        u32 diff = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) diff |= (stage[W * i + w] ^ stage[W * i + w - W]) & pm.w[w];
        h = (base + i == 0) || diff != 0;
      }
      const u64 hm = __ballot(h);
      if (lane == 0) {
        hw[i0 >> 6] = hm;
        const u64 lo = i0 < (u32)M ? hm : 0ull, hi = i0 >= (u32)T ? hm : 0ull;   // M and T are multiples of 64: a word lies on one side
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
    // 3. every element finds its run in the bitmap and its place in the run.  Runs of one and two — nearly all of them when the keys
    //    are well spread — are settled on the spot with one comparison; the elements of longer short runs go onto a worklist that is
    //    worked off densely afterwards.  (Round 3's first form let every head walk to the next one and every element loop over its
    //    run: a wave then pays for the longest run among its 64 lanes in every round — 181 VALU and 151 scalar instructions per
    //    element, profiles/README.md r03_sq.)  Long runs: identity + order check.
    // a run [s0, s0 + m) is ranked when it starts with a head inside the range, is closed by a head (or by the array's end) and short
    auto short_run = [&](u32 s0, bool found, u32 nx, u32& m) -> bool {
      const u32 stop = nx < end ? nx : end;
      m = stop - s0;
      return found && s0 >= begin && m <= (u32)M && (nx < end || end_is_head);
    };
    bool inversion = false;
#pragma unroll 1
    for (u32 i = begin + tid; i < end; i += kSortThreads) {
      {
        u32 me[W];
#pragma unroll
        for (int w = 0; w < W; ++w) me[w] = stage[W * i + w];
        u32 s0, nx, m;
        bool found;
        run_around(hw, i, nwords, s0, found, nx);
        if (short_run(s0, found, nx, m)) {
          if (m >= 3u) {
            wl[atomicAdd(&misc[3], 1u)] = (uint16_t)i;    // its place comes in step 4
          } else {
            const u32 p = m == 2u ? s0 + (u32)(i == s0) : i;   // the other element of a pair (a run of one: itself, which counts nothing)
            u32 a[W];
#pragma unroll
            for (int w = 0; w < W; ++w) a[w] = stage[W * p + w];
            tgt[i] = (uint16_t)(s0 + elem_before<W>(a, me, (u32)(p < i)));
          }
        } else {                                              // part of a long run
          tgt[i] = (uint16_t)i;
          if (!((hw[i >> 6] >> (i & 63u)) & 1ull)) {          // same run as the element in front (i = 0: the one in front of the window)
            u32 prev[W];
#pragma unroll
            for (int w = 0; w < W; ++w) prev[w] = stage[W * i + w - W];
            inversion = inversion || elem_before<W>(me, prev, 0u);
          }
        }
      }
    }
    if (inversion) misc[2] = 1u;
    __syncthreads();
    // 4. the worklist: rank by counting inside the run (mask arithmetic, two candidates per step)
    const u32 nwl = misc[3];
    for (u32 t2 = tid; t2 < nwl; t2 += kSortThreads) {
      const u32 i = wl[t2];
      u32 s0, nx, m;
      bool found;
      run_around(hw, i, nwords, s0, found, nx);
      (void)short_run(s0, found, nx, m);
      u32 me[W];
#pragma unroll
      for (int w = 0; w < W; ++w) me[w] = stage[W * i + w];
      u32 cnt = 0;
      for (u32 j = s0; j < s0 + m; j += 2) {
        const u32 j1 = j + 1 < s0 + m ? j + 1 : j;             // clamped: in the window, not counted
        u32 a[W], b[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { a[w] = stage[W * j + w]; b[w] = stage[W * j1 + w]; }
        cnt += elem_before<W>(a, me, (u32)(j < i));
        cnt += elem_before<W>(b, me, (u32)(j1 < i)) & (u32)(j + 1 < s0 + m);
      }
      tgt[i] = (uint16_t)(s0 + cnt);
    }
    __syncthreads();
    if (misc[2]) {                                            // a long run that is not in order: not this kernel's to sort
      if (tid == 0) *overflow = 1u;
      return;
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const u32 i = tid + kSortThreads * r;
      if (i >= begin && i < end) {
        const u32 to = tgt[i];
#pragma unroll
        for (int w = 0; w < W; ++w) stage[W * to + w] = v[r].w[w];
      }
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
#pragma unroll 1
    for (u32 h = 2 * begin + tid; h < 2 * end; h += kSortThreads) {   // kSortThreads is even: a lane keeps its half
      const u32 p = h >> 1;
      u32 e[4] = {stage[W * p], stage[W * p + 1], stage[W * p + 2], 0};
      if constexpr (W == 4) e[3] = stage[W * p + 3];
      u32x3 o;
      o.x = hbase[0] | __builtin_amdgcn_perm(e[1], e[0], hsel[0][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[0][1]);
      o.y = hbase[1] | __builtin_amdgcn_perm(e[1], e[0], hsel[1][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[1][1]);
      o.z = hbase[2] | __builtin_amdgcn_perm(e[1], e[0], hsel[2][0]) | __builtin_amdgcn_perm(e[3], e[2], hsel[2][1]);
      __builtin_nontemporal_store(o, reinterpret_cast<u32x3_a4*>(out + 24 * (size_t)p + 12 * hj));   // the chunk is contiguous: nothing for the L2 to merge
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
