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
// longer prefix, then all passes).  One translation unit in five files (the host side takes the kernels' addresses): the kernels in
// sort_census.hpp | sort_passes.hpp (24-byte passes) | sort_compact.hpp (compress, element passes, expand) | sort_finish.hpp
// (finishing kernels + the sample estimate), included below; this file: shared definitions, host side (layout, variants,
// launch_compact_passes, launch_sort_records), splitter search.  Per-barcode aggregation: k_aggregate.hip.
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
// Match-any on an 8-bit digit: the lanes of the wave (among `valid`) whose digit equals this lane's, as two 32-bit halves, and from
// them the lane's rank among its peers and the peers' count.  Written for the instructions it should become — per bit one
// v_bfe_i32 (0 or ~0), one v_cmp (the ballot), two v_xnor with the ballot's halves as scalar operands, two v_and: six VALU —
// because the scatter kernels turned out VALU-bound, not memory-bound (SQ counters, profiles/README.md r03_sq: 40 % of every wave's
// cycles issuing VALU at two waves per SIMD), and the plain `m &= bit ? bal : ~bal` on a 64-bit m compiled to eleven per bit.
struct DigitPeers { u32 before, count; };
// PLAIN = true: the straightforward 64-bit form, kept for the 16-byte element passes, which it suits better (measured on one box,
// same process order: 6.32 ms per pass against 6.82 with the six-instruction form — those passes are memory-bound and the
// compiler's schedule of the longer form happens to overlap their loads better; the 12-byte passes gain 3.5 % from the short form).
template <bool PLAIN = false>
__device__ __forceinline__ DigitPeers match_digit(u32 d, u64 valid) {
  DigitPeers r;
  if constexpr (PLAIN) {
    u64 m = valid;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const u64 bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    r.before = (u32)__popcll(m & ((1ull << (threadIdx.x & (kWave - 1))) - 1ull));
    r.count = (u32)__popcll(m);
    return r;
  }
  u32 mlo = (u32)valid, mhi = (u32)(valid >> 32);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const u32 s = (u32)__builtin_amdgcn_sbfe((int)d, (u32)b, 1u);
    const u64 bal = __ballot(s != 0u);
    mlo &= ~((u32)bal ^ s);                                   // s = ~0: the lanes with the bit set; s = 0: those without
    mhi &= ~((u32)(bal >> 32) ^ s);
  }
  r.before = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));   // peers in the lanes below this one
  r.count = (u32)__builtin_popcount(mlo) + (u32)__builtin_popcount(mhi);
  return r;
}
__device__ __forceinline__ bool rec_less(u64 b, u64 u, u64 x, u64 pb, u64 pu, u64 px) {  // (b,u,x) < (pb,pu,px): record.rs:58
  return b != pb ? b < pb : (u != pu ? u < pu : x < px);
}

#include "sort_census.hpp"    // census of the varying key bytes, the sorted check
#include "sort_passes.hpp"    // a radix pass on 24-byte records: tile counts, scan, scatter
#include "sort_compact.hpp"   // compact keys: plan, compress / expand, element passes
#include "sort_finish.hpp"    // prefix + finish: finishing kernels, run-length estimate

// =====================================================================================================
// Host side.  Scratch layout (bytes), all offsets 256-byte aligned:
//   census u64[64][8] | binbase u64[256] | blocksum u32[nblocks][256] | blockoff u64[nblocks][256] | counts u16[ntiles][256]
//   | pos IDX[ntiles][256] | digits u8[ntiles * T]
static constexpr size_t kMiscBytes = 256;                     // behind the census slots: [0] the finishing kernel's overflow flag (u32)
struct SortLayout {
  size_t misc, binbase, blocksum, blockoff, counts, pos, digits, total;
  u32 ntiles, nblocks, tpb;                                   // tpb: tiles per scan block
  bool idx64;
};
// Tiles per scan block: the position walk of a block (ibu_k_sort_tilepos) is a serial chain over its tiles and the scan over the
// blocks (ibu_k_sort_blockscan) one over the blocks, so small inputs want short blocks (both chains ~ sqrt(tiles): 1e6 records,
// 196 tiles: 16 per block; the fixed 256 left ONE workgroup walking all of them, 26 of a 400-us sort, three times) and large ones
// the 256 that bounds the scan's chain (1e9 records: 763 blocks).
static u32 tiles_per_block(u64 ntiles) {
  u32 t = 8;
  while (t < (u32)kTilesPerBlock && (u64)t * t < ntiles) t <<= 1;
  return t;
}
static SortLayout sort_layout(const LaunchCfg& cfg, size_t n, int tile) {
  SortLayout L;
  const u64 nt = (n + tile - 1) / tile;
  L.ntiles = (u32)nt;
  L.tpb = tiles_per_block(nt);
  L.nblocks = (u32)((nt + L.tpb - 1) / L.tpb);
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
  const void* scatter32_stream;   // the digit from the side stream (launch_partition_records)
  const void* scatter64_stream;
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
          ibu_k_sort_tilecounts_recs_tail<S::T>, ibu_k_sort_tilecounts_bytes<S::T>,
          reinterpret_cast<const void*>(ibu_k_sort_scatter<TH, R, u32, WM, true>), reinterpret_cast<const void*>(ibu_k_sort_scatter<TH, R, u64, WM, true>)};
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
// ... for n records: inputs of fewer than 512 default tiles (2.6 M records: less than the two workgroups per CU the chip takes)
// run the 12-byte passes on 2048-element tiles — 2.5 times the workgroups, each done sooner (1e6 records: three passes of 14 us
// on 196 workgroups).  The default shape only (an explicit choice is an A/B), and not under the 64-bit index test knob (the
// small shape has no 64-bit kernels).
static constexpr size_t kSmallInput = (size_t)512 * 5120;
static const CompactVariant* pick_compact_for(const LaunchCfg& cfg, size_t n) {
  const CompactVariant* cv = pick_compact(cfg);
  if (cv == &kCompact[0] && n < kSmallInput && !cfg.sort_idx64) return &kCompact[2];
  return cv;
}

size_t sort_scratch_bytes(const LaunchCfg& cfg, size_t n) {
  size_t need = sort_layout(cfg, n, pick_variant(cfg).tile).total;
  if (const CompactVariant* cv = pick_compact_for(cfg, n)) {
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
// The census of three SAMPLE ranges (first / middle / last 32 Ki records) — what the sort's speculation guesses its plan from; inputs
// too small for three ranges get the exact census.  `exact` says which it was.
static constexpr size_t kCensusSample = 32768;
hipError_t launch_records_census_sample(const LaunchCfg& cfg, const void* recs, size_t n, uint64_t* d_census, bool* exact, hipStream_t st) {
  (void)hipGetLastError();
  *exact = n < 4 * kCensusSample || (reinterpret_cast<uintptr_t>(recs) & 15u) != 0;
  if (*exact) return launch_records_census(cfg, recs, n, d_census, st);
  hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(kCensusSlots * 8), 0, st, (u64*)d_census);
  const size_t starts[3] = {0, (n / 2) & ~(size_t)1, (n - kCensusSample) & ~(size_t)1};   // even rows: 16-byte aligned
  for (size_t s0 : starts) launch_census(cfg, static_cast<const u64*>(recs) + 3 * s0, kCensusSample, (u64*)d_census, nullptr, st);
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
    hipLaunchKernelGGL(ibu_k_sort_blocksums, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb, blocksum);
    hipLaunchKernelGGL(ibu_k_sort_blockscan, dim3(1), dim3(kSortThreads), 0, st, (const u32*)blocksum, L.nblocks, blockoff, binbase);
    if (L.idx64)
      hipLaunchKernelGGL(ibu_k_sort_tilepos<u64>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb,
                         (const u64*)blockoff, (const u64*)binbase, static_cast<u64*>(pos));
    else
      hipLaunchKernelGGL(ibu_k_sort_tilepos<u32>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb,
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
bool trace_sort() {
  static const bool on = [] { const char* v = getenv("IBU_TRACE_SORT"); return v && *v && *v != '0'; }();
  return on;
}
// PREFIX + FINISH on the elements?  Only if the runs of equal prefix are going to be short: a pair count over a sample says
// (ibu_k_sort_sample_pairs; the tables live in tmp, which nothing uses at that point).  `sorted_bytes`: the passes the plain path
// would run (the element bytes it sorts on are the top `sorted_bytes` of the plan's k).  *P = the prefix to use, 0 = none.
// n_scale: the size of the whole the runs are estimated for — n, or (the multi-GPU sort) the records of ALL shards, of which these
// n are taken for a sample.
static hipError_t estimate_compact_prefix(const LaunchCfg& cfg, const void* recs, size_t n, void* tmp, const CompactPlan& plan, u32 sorted_bytes,
                                          hipStream_t st, u32* P_out, double* seg_out, size_t n_scale = 0) {
  if (!n_scale) n_scale = n;
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
  while (nranges > 3 && (size_t)nranges * per_range > n / 32) nranges /= 2;   // small inputs: a thirty-second of them is sample enough (the pair count of 49 152 samples was 75 of a 1e6-record sort's 255 us of kernels)
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
      const double seg = 1.0 + ((double)n_scale / (double)m) * (2.0 * (double)pairs[q] / (double)m);
      const double heaviest = (double)pairs[kMaxPrefix + q] * ((double)n_scale / (double)m);   // estimated longest run
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

// ---- the multi-GPU sort on 12-byte elements: partition first, sort once (multi_sort.cpp, round 4) -----------------------------------
// Round 3 sorted every shard, exchanged the pieces between the splitters and sorted every owner's pieces AGAIN.  Now a shard is only
// PARTITIONED before the exchange: its records become 12-byte elements (one plan for all shards, at most 11 varying bytes), every
// element gets — in the same kernel — the number of its key range — how many of the (up to 255) splitters are not above it — in its free top byte, and one
// ordinary element pass on that byte (count from the side stream, scan, scatter: the kernels of the sort) moves the elements into
// range order; the scan's bin starts are the range boundaries, from which the caller cuts the owners' pieces.  The owner sorts the elements it received straight into records (launch_sort_elems: the
// passes of the sort without its census and compress steps — the sender made the elements).
extern "C" __global__ void __launch_bounds__(256)
ibu_k_sort_stamp_bucket(ElemT<3>* __restrict__ elems, u64 n, const ElemT<3>* __restrict__ split, u32 nsplit, uint8_t* __restrict__ digits) {
  __shared__ u32 sp[3 * 256];
  for (u32 i = threadIdx.x; i < 3 * nsplit; i += blockDim.x) sp[i] = reinterpret_cast<const u32*>(split)[i];
  __syncthreads();
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    EV<3> e = ld_elem<3>(elems + i);
    e.w[2] &= 0x00FFFFFFu;                                    // (zero already: at most 11 bytes vary)
    const u32 g = range_of(sp, nsplit, e);
    e.w[2] |= g << 24;
    st_elem<3>(elems + i, e);
    digits[i] = (uint8_t)g;
  }
}
// records -> stamped elements + the digit stream of the ranges: the tiled rows in ONE kernel (ibu_k_sort_compress<.., STAMP>), the
// peeled head row and the rest rows (fewer than 129 in all) through the tail compress kernel and the stamp kernel above.
// census (nullable; 16-byte aligned records only): the exact census words are accumulated on the way, as in the sort's speculative path.
static void launch_compress_stamped(const LaunchCfg& cfg, const CompactPlan& pl, const void* recs, size_t n, ElemT<3>* out, uint8_t* digits,
                                    const ElemT<3>* split, u32 nsplit, hipStream_t st, u64* census = nullptr) {
  const size_t head = (reinterpret_cast<uintptr_t>(recs) & 15u) ? (n ? 1 : 0) : 0;
  const size_t main_rows = ((n - head) / kTileRecs) * kTileRecs;
  if (head) {
    hipLaunchKernelGGL(ibu_k_sort_compress_tail<3>, dim3(1), dim3(256), 0, st, (const u64*)recs, (u64)0, (u64)head, pl, 0u, out, (uint8_t*)nullptr);
    hipLaunchKernelGGL(ibu_k_sort_stamp_bucket, dim3(1), dim3(256), 0, st, out, (u64)head, split, nsplit, digits);
  }
  if (main_rows) {
    static std::atomic<int> occ[2];
    const u32 nt = (u32)(main_rows / kTileRecs);
    if (census)
      hipLaunchKernelGGL((ibu_k_sort_compress<true, 3, true>), dim3(grid_for(nt, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_sort_compress<true, 3, true>, 0, &occ[1]))),
                         dim3(kBlock), 0, st, static_cast<const uint8_t*>(recs) + 24 * head, nt, pl, 0u, out + head, digits + head, census, split, nsplit);
    else
      hipLaunchKernelGGL((ibu_k_sort_compress<false, 3, true>), dim3(grid_for(nt, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_sort_compress<false, 3, true>, 0, &occ[0]))),
                         dim3(kBlock), 0, st, static_cast<const uint8_t*>(recs) + 24 * head, nt, pl, 0u, out + head, digits + head, (u64*)nullptr, split, nsplit);
  }
  const size_t done = head + main_rows;
  if (done < n) {
    hipLaunchKernelGGL(ibu_k_sort_compress_tail<3>, dim3(tail_grid(n - done)), dim3(256), 0, st, (const u64*)recs, (u64)done, (u64)n, pl, 0u, out,
                       (uint8_t*)nullptr);
    hipLaunchKernelGGL(ibu_k_sort_stamp_bucket, dim3(1), dim3(256), 0, st, out + done, (u64)(n - done), split, nsplit, digits + done);
    if (census)   // the rest rows of the census (each row also against its predecessor)
      hipLaunchKernelGGL(ibu_k_sort_census_tail, dim3(tail_grid(n - done)), dim3(256), 0, st, (const u64*)recs, (u64)done, (u64)n, census, (u32*)nullptr);
  }
}
static const CompactVariant* elems_variant(const LaunchCfg& cfg, size_t n, size_t scratch_bytes) {
  const CompactVariant* cv = pick_compact_for(cfg, n);
  if (!cv) cv = &kCompact[0];                                 // (sort_compact = 0 on this context: the default shape)
  const bool wide_idx = n >= (1ull << 32) || cfg.sort_idx64;
  if (n >= (1ull << 38) || (wide_idx && !cv->scatter64) || scratch_bytes < sort_layout(cfg, n, cv->tile).total) return nullptr;
  return cv;
}
hipError_t launch_estimate_prefix(const LaunchCfg& cfg, const void* recs, size_t n, size_t n_scale, void* tmp, const CompactPlan& pl,
                                  uint32_t* prefix_passes, hipStream_t st) {
  (void)hipGetLastError();
  double seg = 0;
  u32 P = 0;
  *prefix_passes = 0;
  if (n < 8192 || pl.k == 0 || pl.k > 12) return hipSuccess;
  const hipError_t e = estimate_compact_prefix(cfg, recs, n, tmp, pl, pl.k, st, &P, &seg, n_scale);   // (synchronises st)
  if (e == hipSuccess) *prefix_passes = P;
  return e;
}
hipError_t launch_partition_elems(const LaunchCfg& cfg, const CompactPlan& pl, const void* recs, void* elems, size_t n, const void* d_split,
                                  uint32_t nsplit, void* out, void* scratch, size_t scratch_bytes, const uint64_t** d_starts,
                                  const uint64_t** d_census, hipStream_t st, const PartitionEarly* early) {
  (void)hipGetLastError();
  if (n == 0 || nsplit > 255 || pl.k > 11) return hipErrorInvalidValue;
  if (d_census && (reinterpret_cast<uintptr_t>(recs) & 15u)) return hipErrorInvalidValue;
  const CompactVariant* cv = elems_variant(cfg, n, scratch_bytes);
  if (!cv) return hipErrorInvalidValue;
  uint8_t* sc = static_cast<uint8_t*>(scratch);
  const SortLayout L = sort_layout(cfg, n, cv->tile);
  u64* binbase = reinterpret_cast<u64*>(sc + L.binbase);
  u32* blocksum = reinterpret_cast<u32*>(sc + L.blocksum);
  u64* blockoff = reinterpret_cast<u64*>(sc + L.blockoff);
  uint16_t* counts = reinterpret_cast<uint16_t*>(sc + L.counts);
  void* pos = sc + L.pos;
  uint8_t* digits = sc + L.digits;
  const void* k_scatter = L.idx64 ? cv->scatter64 : cv->scatter;
  hipError_t e;
  if (cv->lds > 48 * 1024) {
    e = hipFuncSetAttribute(k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv->lds);
    if (e != hipSuccess) return e;
  }
  const u32 cap = (u32)cfg.cus * 8;
  u64* census = d_census ? reinterpret_cast<u64*>(sc) : nullptr;   // the census slots sit at the head of the scratch (SortLayout)
  if (census) hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(kCensusSlots * 8), 0, st, census);
  launch_compress_stamped(cfg, pl, recs, n, static_cast<ElemT<3>*>(elems), digits, static_cast<const ElemT<3>*>(d_split), nsplit, st, census);
  if (census) {
    hipLaunchKernelGGL(ibu_k_sort_census_fold, dim3(1), dim3(kCensusSlots), 0, st, census);
    *d_census = reinterpret_cast<const uint64_t*>(census);    // u64[8]: OR x 3, AND x 3, index drops, order drops — of exactly these n records
  }
  const u32 wave_grid = (L.ntiles + kSortWaves - 1) / kSortWaves;
  hipLaunchKernelGGL(cv->counts_bytes, dim3(wave_grid < cap ? wave_grid : cap), dim3(kSortThreads), 0, st, (const uint8_t*)digits, (u64)n, L.ntiles,
                     counts);
  hipLaunchKernelGGL(ibu_k_sort_blocksums, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb, blocksum);
  hipLaunchKernelGGL(ibu_k_sort_blockscan, dim3(1), dim3(kSortThreads), 0, st, (const u32*)blocksum, L.nblocks, blockoff, binbase);
  if (early) {                                                 // the host's share of the pass is complete here: hand it over before the scatter
    e = hipMemcpyAsync(early->h_starts, binbase, 8 * kBins, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && census) e = hipMemcpyAsync(early->h_words, census, 64, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(early->ready, st);
    if (e != hipSuccess) return e;
  }
  if (L.idx64)
    hipLaunchKernelGGL(ibu_k_sort_tilepos<u64>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb,
                       (const u64*)blockoff, (const u64*)binbase, static_cast<u64*>(pos));
  else
    hipLaunchKernelGGL(ibu_k_sort_tilepos<u32>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb,
                       (const u64*)blockoff, (const u64*)binbase, static_cast<u32*>(pos));
  u32 n32 = (u32)n, b_arg = 11, nb_arg = 12;                  // the owner byte; 12 = no digit stream behind this pass
  u64 n64 = n;
  const void* src_arg = elems;
  void* dst_arg = out;
  const void* pos_arg = pos;
  CompactPlan pl_arg = CompactPlan();                         // (only a last pass expands)
  void* args[] = {&src_arg, &dst_arg, L.idx64 ? static_cast<void*>(&n64) : static_cast<void*>(&n32), &b_arg, &nb_arg, &pos_arg, &digits, &pl_arg};
  e = hipLaunchKernel(k_scatter, dim3((L.ntiles + 7u) & ~7u), dim3(cv->threads), args, cv->lds, st);
  if (e != hipSuccess) return e;
  *d_starts = reinterpret_cast<const uint64_t*>(binbase);                                  // u64[256]: first element of every owner's piece
  return hipGetLastError();
}
// The same partition pass on 24-byte RECORDS (keys of more than 11 varying bytes, or buffers the element kernels cannot take): a
// record's key range — how many of the up to 255 splitter records are not above it — goes into the digit side stream, and one
// ordinary 24-byte pass whose digit comes from that stream (ibu_k_sort_scatter, field > 2) moves the records into range order.
// CENSUS: the exact census words of the records (OR / AND of every field) are accumulated on the way — the owners' sorts of the
// multi-GPU form then need no census pass of their own (the order flags are not taken: pieces of several shards interleave).
template <bool CENSUS>
__global__ void __launch_bounds__(256)
ibu_k_sort_stamp_records(const u64* __restrict__ recs, u64 n, const u64* __restrict__ split, u32 nsplit, uint8_t* __restrict__ digits,
                         u64* __restrict__ census) {
  __shared__ u64 sp[3 * 256];
  for (u32 i = threadIdx.x; i < 3 * nsplit; i += blockDim.x) sp[i] = split[i];
  __syncthreads();
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const u64 ref[3] = {recs[0], recs[1], recs[2]};            // n >= 1; uniform address: scalar loads
  CensusAcc acc;
  const bool any_rows = (u64)blockIdx.x * blockDim.x + (threadIdx.x & ~(u32)(kWave - 1)) < n;   // wave-uniform: the wave's first row exists
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const u64 b = recs[3 * i], u = recs[3 * i + 1], x = recs[3 * i + 2];
    if constexpr (CENSUS) acc.rec(b, u, x, ref);
    u32 lo = 0, hi = nsplit;                                  // range = splitters <= record
    while (lo < hi) {
      const u32 mid = (lo + hi) >> 1;
      if (!rec_less(b, u, x, sp[3 * mid], sp[3 * mid + 1], sp[3 * mid + 2])) lo = mid + 1; else hi = mid;
    }
    digits[i] = (uint8_t)lo;
  }
  if constexpr (CENSUS) acc.flush(census, nullptr, ref, any_rows);
}
hipError_t launch_partition_records(const LaunchCfg& cfg, const void* recs, size_t n, const void* d_split, uint32_t nsplit, void* out, void* scratch,
                                    size_t scratch_bytes, const uint64_t** d_starts, const uint64_t** d_census, hipStream_t st,
                                    const PartitionEarly* early) {
  (void)hipGetLastError();
  if (n == 0 || nsplit > 255) return hipErrorInvalidValue;
  const SweepVariant& sv = pick_variant(cfg);
  if ((n + sv.tile - 1) / sv.tile >= (1ull << 31)) return hipErrorInvalidValue;
  const SortLayout L = sort_layout(cfg, n, sv.tile);
  if (scratch_bytes < L.total) return hipErrorInvalidValue;
  uint8_t* sc = static_cast<uint8_t*>(scratch);
  u64* binbase = reinterpret_cast<u64*>(sc + L.binbase);
  u32* blocksum = reinterpret_cast<u32*>(sc + L.blocksum);
  u64* blockoff = reinterpret_cast<u64*>(sc + L.blockoff);
  uint16_t* counts = reinterpret_cast<uint16_t*>(sc + L.counts);
  void* pos = sc + L.pos;
  uint8_t* digits = sc + L.digits;
  const void* scatter = L.idx64 ? sv.scatter64_stream : sv.scatter32_stream;
  hipError_t e;
  if (sv.lds > 48 * 1024) {
    e = hipFuncSetAttribute(scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sv.lds);
    if (e != hipSuccess) return e;
  }
  const u32 cap = (u32)cfg.cus * 8;
  const u64 want = (n + 255) / 256;
  u64* census = d_census ? reinterpret_cast<u64*>(sc) : nullptr;   // the census slots sit at the head of the scratch (SortLayout)
  if (census) {
    hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(kCensusSlots * 8), 0, st, census);
    hipLaunchKernelGGL(ibu_k_sort_stamp_records<true>, dim3((u32)(want < cap ? want : cap)), dim3(256), 0, st, static_cast<const u64*>(recs), (u64)n,
                       static_cast<const u64*>(d_split), nsplit, digits, census);
    hipLaunchKernelGGL(ibu_k_sort_census_fold, dim3(1), dim3(kCensusSlots), 0, st, census);
    *d_census = reinterpret_cast<const uint64_t*>(census);    // u64[8]: OR x 3, AND x 3 of exactly these n records ([6], [7]: not taken)
  } else {
    hipLaunchKernelGGL(ibu_k_sort_stamp_records<false>, dim3((u32)(want < cap ? want : cap)), dim3(256), 0, st, static_cast<const u64*>(recs), (u64)n,
                       static_cast<const u64*>(d_split), nsplit, digits, (u64*)nullptr);
  }
  const u32 wave_grid = (L.ntiles + kSortWaves - 1) / kSortWaves;
  hipLaunchKernelGGL(sv.counts_bytes, dim3(wave_grid < cap ? wave_grid : cap), dim3(kSortThreads), 0, st, (const uint8_t*)digits, (u64)n, L.ntiles, counts);
  hipLaunchKernelGGL(ibu_k_sort_blocksums, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb, blocksum);
  hipLaunchKernelGGL(ibu_k_sort_blockscan, dim3(1), dim3(kSortThreads), 0, st, (const u32*)blocksum, L.nblocks, blockoff, binbase);
  if (early) {                                                 // the host's share of the pass is complete here: hand it over before the scatter
    e = hipMemcpyAsync(early->h_starts, binbase, 8 * kBins, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && census) e = hipMemcpyAsync(early->h_words, census, 64, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(early->ready, st);
    if (e != hipSuccess) return e;
  }
  if (L.idx64)
    hipLaunchKernelGGL(ibu_k_sort_tilepos<u64>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb,
                       (const u64*)blockoff, (const u64*)binbase, static_cast<u64*>(pos));
  else
    hipLaunchKernelGGL(ibu_k_sort_tilepos<u32>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb,
                       (const u64*)blockoff, (const u64*)binbase, static_cast<u32*>(pos));
  u64 n_arg = n;
  u32 f_arg = 3, s_arg = 0, nf_arg = 3, ns_arg = 0;           // this pass's digit: from the side stream; no pass follows
  const u64* src_arg = static_cast<const u64*>(recs);
  u64* dst_arg = static_cast<u64*>(out);
  void* args[] = {&src_arg, &dst_arg, &n_arg, &f_arg, &s_arg, &nf_arg, &ns_arg, &pos, &digits};
  e = hipLaunchKernel(scatter, dim3((L.ntiles + 7u) & ~7u), dim3(sv.threads), args, sv.lds, st);
  if (e != hipSuccess) return e;
  *d_starts = reinterpret_cast<const uint64_t*>(binbase);     // u64[256]: first record of every range
  return hipGetLastError();
}
bool sort_elems_supported(const LaunchCfg& cfg, const void* recs, const void* tmp, size_t capacity) {
  return cfg.sort_compact != 0 && capacity < (1ull << 38) && (reinterpret_cast<uintptr_t>(recs) & 15u) == 0 && (reinterpret_cast<uintptr_t>(tmp) & 15u) == 0;
}
hipError_t launch_sort_elems(const LaunchCfg& cfg, const CompactPlan& pl, void* recs, void* tmp, size_t n, uint32_t prefix_passes, void* scratch,
                             size_t scratch_bytes, hipStream_t st) {
  (void)hipGetLastError();
  if (n == 0) return hipSuccess;
  if (pl.k > 12) return hipErrorInvalidValue;
  if (n == 1 || pl.k == 0) return launch_expand(cfg, pl, tmp, n, recs, st);   // one record, or all of them the same
  const CompactVariant* cv = elems_variant(cfg, n, scratch_bytes);
  if (!cv) return hipErrorInvalidValue;
  u32 ebytes[12];
  for (u32 j = 0; j < pl.k; ++j) ebytes[j] = j;               // every varying byte: pieces of different shards interleave in the index too
  u32 P = prefix_passes;
  if (n < 8192 || P + 1 >= pl.k) P = 0;
  if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu path=elements-received element_bytes=12 prefix_passes=%u of %u\n", n, P, pl.k);
  return launch_compact_passes<3>(cfg, *cv, recs, tmp, n, static_cast<uint8_t*>(scratch), pl, ebytes, pl.k, st, true, 0xFFFFFFFFu, P,
                                  static_cast<ElemT<3>*>(tmp));
}

// What launch_sort_records' sampled prefix estimate keeps in the head of `tmp` for n records (0: it would not sample) — for a caller
// that hands it some other scratch as `tmp` (only_estimate).
size_t sort_prefix_estimate_tables(const LaunchCfg& cfg, size_t n) {
  const size_t table_bytes = 128 + (size_t)kPairSlotsMax * 12 * kMaxPrefix;
  return (cfg.sort_hybrid && n >= 4 * (size_t)32768 && table_bytes <= n * 24) ? table_bytes : 0;
}

// Not purely asynchronous: the census result comes back to the host (one 64-byte read) to pick the passes; everything
// after that is queued on `st`.
// known_words (nullable): census words the caller already has for a SUPERSET of these records (the multi-GPU sort: the partition pass
// took them over all shards) — OR x 3, AND x 3; no census pass runs, no record is assumed in index order or sorted, and the bytes
// that vary in the superset get their passes (a byte that happens to be constant here costs one identity pass).
// known_prefix (with known_words; the multi-GPU sort): >= 0 = the prefix length of the 24-byte prefix + finish path as somebody already
// estimated it for the WHOLE these records are a key range of (0: all passes); -1: estimate here.  only_estimate (nullable): do nothing
// but that estimate — for n_scale records like these — and return it in *only_estimate (tmp's head is the table scratch).
hipError_t launch_sort_records(const LaunchCfg& cfg, void* recs, void* tmp, size_t n, void* scratch,
                               size_t scratch_bytes, hipStream_t st, const uint64_t* known_words, int known_prefix, int* only_estimate,
                               size_t n_scale) {
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
  const CompactVariant* cv = pick_compact_for(cfg, n);
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
  if (!known_words && compact_ok && cfg.sort_guess && n >= guess_min) {
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
  u64 c[8];
  if (known_words) {
    for (int w = 0; w < 6; ++w) c[w] = known_words[w];
    c[6] = c[7] = 1;                 // nothing is known about the order
    if (trace_sort()) fprintf(stderr, "ibu sort: n=%zu census words given by the caller: no census pass\n", n);
  } else {
    if (!speculated) {
      hipLaunchKernelGGL(ibu_k_sort_census_init, dim3(1), dim3(kCensusSlots * 8), 0, st, census);
      launch_census(cfg, recs, n, census, nullptr, st);
    }
    hipLaunchKernelGGL(ibu_k_sort_census_fold, dim3(1), dim3(kCensusSlots), 0, st, census);
    e = hipMemcpyAsync(c, census, sizeof c, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
  }
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
  if (compact_ok && npass > 0 && !only_estimate) {
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
      hipLaunchKernelGGL(ibu_k_sort_blocksums, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb, blocksum);
      hipLaunchKernelGGL(ibu_k_sort_blockscan, dim3(1), dim3(kSortThreads), 0, st, (const u32*)blocksum, L.nblocks, blockoff, binbase);
      if (L.idx64)
        hipLaunchKernelGGL(ibu_k_sort_tilepos<u64>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb,
                           (const u64*)blockoff, (const u64*)binbase, static_cast<u64*>(pos));
      else
        hipLaunchKernelGGL(ibu_k_sort_tilepos<u32>, dim3(L.nblocks), dim3(kSortThreads), 0, st, (const uint16_t*)counts, L.ntiles, L.tpb,
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
    const size_t n_est = n_scale ? n_scale : n;               // the size the runs are estimated for
    int P = 1;
    for (u64 segs = 256; n_est / segs > 8 && P < 8; segs <<= 8) ++P;
    // ... of WELL-SPREAD keys.  From 2^17 records on the sample ranges say whether they are (ibu_k_sort_sample_pairs_recs: pairs of
    // equal prefix and the most frequent prefix among 3 x 32 Ki sample records, tables in tmp): the shortest prefix with at most
    // ~8 records per run and no heavy prefix is taken, which may be longer than the one n suggests — or none (P = 0: all passes).
    static constexpr size_t kSampleW = 32768;
    if (known_prefix >= 0) {
      P = known_prefix ? (known_prefix < npass ? known_prefix : npass) : npass;
    } else if (cfg.sort_hybrid && n >= 4 * kSampleW && (reinterpret_cast<uintptr_t>(tmp) & 7u) == 0) {
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
            const double seg = 1.0 + ((double)n_est / (double)m) * (2.0 * (double)pairs[q] / (double)m);
            const double heaviest = (double)pairs[kMaxPrefix + q] * ((double)n_est / (double)m);
            if (seg <= 8.0 && heaviest <= 128.0) { Pest = (int)(pb.first + q + 1); break; }
          }
        }
        if (trace_sort() && Pest != P) fprintf(stderr, "ibu sort: n=%zu sample estimate: prefix_passes=%d (well-spread keys would take %d)\n", n, Pest, P);
        P = Pest ? Pest : npass;                              // npass: never worth it below
      }
    }
    if (only_estimate) { *only_estimate = P >= npass ? 0 : P; return hipGetLastError(); }
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
