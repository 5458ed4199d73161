// k_decode.hip — K2 fused decode (AoS records -> barcode ASCII, UMI ASCII, index column) and
// the single-column 2-bit unpack.  Design notes: kcommon.hpp.  Reference semantics: the 2-bit
// table of src/constructs/record.rs:19-27 applied to the fields of a cast_slice'd &[Record]
// (reader.rs:301, mmap.rs:268); see include/ibu_hip.h.
#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {

// Records per wave iteration = 128 * NT.  With NT = 2 every ASCII column whose length is a multiple of 4 is a whole
// number of 64-lane store rounds (16*len chunks), so no round is partial, and a wave keeps twice the bytes in flight:
// 1.0-1.7 % faster for the dword-path specialisations with at least 20 bases per record ((16,12) in every one of ten
// placements, (32,32), (12,8), (32,12)); slower for (8,8) by 1 %, so
// NT is a property of the instantiation: dec_nt<BC, UM>().  -DIBU_DECODE_NT=1|2 forces one value everywhere (A/B builds).
constexpr bool dword_len(int len) { return len > 0 && (len & 3) == 0; }
#ifndef IBU_DECODE_GEN_NT
#define IBU_DECODE_GEN_NT 1
#endif
template <int BC, int UM>
constexpr int dec_nt() {
#ifdef IBU_DECODE_NT
  return IBU_DECODE_NT;
#else
  if (BC == 0 && UM == 0) return IBU_DECODE_GEN_NT;   // both lengths at run time: the code-stream path holds few registers
  return (dword_len(BC) && dword_len(UM) && BC + UM >= 20) ? 2 : 1;
#endif
}

template <int NT>
struct DecRegs {
  u32x4 v[3 * NT];                                   // dwordx4 loads per lane per iteration
  __device__ __forceinline__ void load(const uint8_t* src) {
#pragma unroll
    for (int k = 0; k < 3 * NT; ++k) v[k] = ld16(src + 1024 * k);
  }
};

// Stage one AoS tile held in registers into the wave's LDS slice and expand it.
template <int BC, int UM, bool MSB>
__device__ __forceinline__ void decode_tile(uint8_t* tile, const DecRegs<dec_nt<BC, UM>()>& a, u32 t, u32 bc_len, u32 umi_len,
                                            uint8_t* bc_out, uint8_t* umi_out, u64* idx_out, u32 lane) {
  constexpr int kDecodeNT = dec_nt<BC, UM>(), kDecLoads = 3 * kDecodeNT, kDecRecs = kTileRecs * kDecodeNT;
  wave_lds_fence();                            // previous tile's LDS reads precede these writes
#pragma unroll
  for (int k = 0; k < kDecLoads; ++k) *reinterpret_cast<u32x4*>(tile + 1024 * k + 16 * lane) = a.v[k];
  wave_lds_fence();
  if constexpr (MSB) {                         // MSB-first code words -> the order the expansion reads
    rev_pairs_tile<kDecodeNT>(tile, 24, 0, bc_len, lane);
    rev_pairs_tile<kDecodeNT>(tile, 24, 8, umi_len, lane);
    wave_lds_fence();
  }
  // a runtime-length field goes through its code stream (kcommon.hpp), kept behind the AoS tile in the wave's LDS slice
  u32* const stream_bc = reinterpret_cast<u32*>(tile + kTileBytes * kDecodeNT);
  u32* const stream_umi = reinterpret_cast<u32*>(tile + kTileBytes * kDecodeNT + (BC == 0 ? stream_bytes(kDecodeNT) : 0));
  if (bc_out) expand_field<BC, kDecodeNT>(tile, 24, 0, bc_len, bc_out + (size_t)t * kDecRecs * bc_len, lane, stream_bc);
  if (umi_out) expand_field<UM, kDecodeNT>(tile, 24, 8, umi_len, umi_out + (size_t)t * kDecRecs * umi_len, lane, stream_umi);
  if (idx_out) {                               // chunk c = indices of records 2c, 2c+1
#pragma unroll
    for (int j = 0; j < kDecodeNT; ++j) {
      const u32 c = lane + 64 * j;
      u64 i0 = *reinterpret_cast<const u64*>(tile + (2 * c) * 24 + 16);
      u64 i1 = *reinterpret_cast<const u64*>(tile + (2 * c + 1) * 24 + 16);
      u32x4 o; o.x = (u32)i0; o.y = (u32)(i0 >> 32); o.z = (u32)i1; o.w = (u32)(i1 >> 32);
      st16(reinterpret_cast<uint8_t*>(idx_out) + (size_t)t * (8 * kDecRecs) + 16 * c, o);
    }
  }
}

// BC / UM: compile-time barcode / UMI length, or 0 for "runtime length" (generic kernel).
// Two register sets take turns (phase A expands `a` while `b` is in flight, phase B the
// reverse): the next tile's loads are issued FIRST in each phase, never copied, never
// behind a wait.
// Register budget: the specialisations (all on the dword path since round 4) fit 72 VGPRs (7 waves/SIMD), measured as fast as 64
// (8) for the streaming kernels (profiles/r01_b sweep); NT = 2 holds twice the tile registers: 5 waves/SIMD, and 4 where a
// length-12 field is in it (its gathers hold more addresses than the dword reads of 16 / 32: round 3 found those three with
// scratch in the code objects' metadata — tools/kernel_resources.py, a CPU test).  A runtime-length field goes through its code
// stream and needs no more registers than a specialised one (68 VGPRs at (0,0)): six waves/SIMD (168 VGPRs and three until round 4).
#ifndef IBU_DECODE_GEN_WAVES
#define IBU_DECODE_GEN_WAVES 6
#endif
template <int BC, int UM, bool MSB>
constexpr int dec_waves() {
  if (BC == 0 || UM == 0) return (BC == 12 || UM == 12) ? 5 : IBU_DECODE_GEN_WAVES;     // a runtime-length field: the code stream
  if (dec_nt<BC, UM>() > 1) return (BC == 12 || UM == 12) ? 4 : (MSB ? 4 : 5);
  return MSB ? 6 : 7;
}
template <int BC, int UM, bool MSB>
__global__ void __launch_bounds__(kBlock, (dec_waves<BC, UM, MSB>()))
ibu_k_decode(const uint8_t* __restrict__ recs, u32 ntiles, u32 bc_len, u32 umi_len,
             uint8_t* __restrict__ bc_out, uint8_t* __restrict__ umi_out, u64* __restrict__ idx_out) {
  constexpr int kDecBytes = kTileBytes * dec_nt<BC, UM>();
  constexpr int kWaveLds = kDecBytes + ((BC == 0) + (UM == 0)) * (int)stream_bytes(dec_nt<BC, UM>());
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kWaveLds];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kWaveLds;
  const TileRange tr = tile_range(ntiles, wib);
  const u32 nwaves = tr.stride;
  u32 t = tr.t;
  ntiles = tr.end;
  if (t >= ntiles) return;
  if (BC > 0) bc_len = BC;
  if (UM > 0) umi_len = UM;

  DecRegs<dec_nt<BC, UM>()> a, b;
  a.load(recs + (size_t)t * kDecBytes + 16 * lane);
  for (;;) {
    u32 tn = t + nwaves;
    bool more = tn < ntiles;                   // wave-uniform
    b.load(recs + (size_t)(more ? tn : t) * kDecBytes + 16 * lane);
    decode_tile<BC, UM, MSB>(tile, a, t, bc_len, umi_len, bc_out, umi_out, idx_out, lane);
    if (!more) break;
    t = tn;
    tn = t + nwaves;
    more = tn < ntiles;
    a.load(recs + (size_t)(more ? tn : t) * kDecBytes + 16 * lane);
    decode_tile<BC, UM, MSB>(tile, b, t, bc_len, umi_len, bc_out, umi_out, idx_out, lane);
    if (!more) break;
    t = tn;
  }
}

// Single u64 column -> ASCII (stride-8 "records").  NT 128-code tiles per wave iteration: with one (1 KiB of codes, a single
// 16-byte load per lane and iteration) the waves sat parked on their loads 45 % of the time (SQ counters, profiles r03_sq) and the
// kernel ran at 5.0 TB/s; the decode kernel had gained the same way (dec_nt).
#ifndef IBU_UNPACK_GEN_NT
#define IBU_UNPACK_GEN_NT 2
#endif
// Round 4: four tiles at six waves per SIMD for rows of at most 16 bases (+1 % on 1e9 codes, profiles/r04_t; 32-base rows would spill
// there and keep two tiles at eight waves).
constexpr int unpack_nt(int len) { return len == 0 ? IBU_UNPACK_GEN_NT : len <= 16 ? 4 : 2; }
constexpr int unpack_waves(int len) { return len == 0 ? 6 : len <= 16 ? 6 : 8; }
template <int LEN, bool MSB>
__global__ void __launch_bounds__(kBlock, unpack_waves(LEN))
ibu_k_unpack(const u64* __restrict__ codes, u32 ntiles /*of NT x 128 codes*/, u32 len, uint8_t* __restrict__ out) {
  constexpr int NT = unpack_nt(LEN);
  constexpr int kWaveLds = 1024 * NT + (LEN == 0 ? (int)stream_bytes(NT) : 0);   // a runtime length goes through the code stream
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kWaveLds];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kWaveLds;
  if (LEN > 0) len = LEN;
  const uint8_t* base = reinterpret_cast<const uint8_t*>(codes) + 16 * lane;
  struct Regs { u32x4 v[NT]; };
  sweep_tiles<Regs>(
      tile_range(ntiles, wib),                                // which tiles this wave sweeps (kcommon.hpp)
      [&](Regs& g, u32 t) {
#pragma unroll
        for (int k = 0; k < NT; ++k) g.v[k] = ld16(base + ((size_t)t * NT + k) * 1024);
      },
      [&](const Regs& g, u32 t) {
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < NT; ++k)                          // lane's chunk = code words 2L, 2L+1 of sub-tile k
          *reinterpret_cast<u32x4*>(tile + 1024 * k + 16 * lane) = MSB ? rev_pairs_x2(g.v[k], len) : g.v[k];
        wave_lds_fence();
        expand_field<LEN, NT>(tile, 8, 0, len, out + (size_t)t * NT * kTileRecs * len, lane, reinterpret_cast<u32*>(tile + 1024 * NT));
      });
}

// ---- tails: one thread per record, any alignment -----------------------------------------------
__device__ __forceinline__ void unpack_row_bytes(u64 code, u32 len, u32 msb, uint8_t* out) {
  if (msb) code = rev_pairs(code, len);
  for (u32 i = 0; i < len; ++i) {  // per-base extract with v_bfe_u32 on the half of the code word that holds base i
    const u32 half = (u32)(code >> (i & 16u ? 32 : 0));
    out[i] = (uint8_t)__builtin_amdgcn_ubfe(kPool, 8 * __builtin_amdgcn_ubfe(half, 2 * (i & 15u), 2), 8);
  }
}
extern "C" __global__ void ibu_k_decode_tail(const u64* __restrict__ recs, u64 row0, u64 n, u32 bc_len,
                                             u32 umi_len, u32 msb, uint8_t* bc_out, uint8_t* umi_out, u64* idx_out) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (bc_out) unpack_row_bytes(recs[3 * i], bc_len, msb, bc_out + i * bc_len);
  if (umi_out) unpack_row_bytes(recs[3 * i + 1], umi_len, msb, umi_out + i * umi_len);
  if (idx_out) idx_out[i] = recs[3 * i + 2];
}
extern "C" __global__ void ibu_k_unpack_tail(const u64* codes, u64 row0, u64 n, u32 len, u32 msb, uint8_t* out) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unpack_row_bytes(codes[i], len, msb, out + i * len);
}

// ---- launchers ------------------------------------------------------------------------------------
typedef void (*DecFn)(const uint8_t*, u32, u32, u32, uint8_t*, uint8_t*, u64*);
template <int B, int U, bool M>
static constexpr DecFn dec_entry() { return ibu_k_decode<len_of_mode(B), len_of_mode(U), M>; }
#define IBU_DEC_ROW(B, M) {dec_entry<B, 0, M>(), dec_entry<B, 1, M>(), dec_entry<B, 2, M>(), dec_entry<B, 3, M>(), dec_entry<B, 4, M>()}
#define IBU_DEC_TABLE(M) {IBU_DEC_ROW(0, M), IBU_DEC_ROW(1, M), IBU_DEC_ROW(2, M), IBU_DEC_ROW(3, M), IBU_DEC_ROW(4, M)}
static const DecFn kDecTable[2][kNumLenModes][kNumLenModes] = {IBU_DEC_TABLE(false), IBU_DEC_TABLE(true)};  // [base_order][bc][umi]
// records per wave iteration of each instantiation (the same for both base orders)
template <int B, int U>
static constexpr int dec_recs() { return kTileRecs * dec_nt<len_of_mode(B), len_of_mode(U)>(); }
#define IBU_DEC_RECS_ROW(B) {dec_recs<B, 0>(), dec_recs<B, 1>(), dec_recs<B, 2>(), dec_recs<B, 3>(), dec_recs<B, 4>()}
static const int kDecRecsTable[kNumLenModes][kNumLenModes] = {IBU_DEC_RECS_ROW(0), IBU_DEC_RECS_ROW(1), IBU_DEC_RECS_ROW(2),
                                                              IBU_DEC_RECS_ROW(3), IBU_DEC_RECS_ROW(4)};

hipError_t launch_decode(const LaunchCfg& cfg, const void* recs, size_t n, uint32_t bc_len, uint32_t umi_len,
                         uint8_t* bc, uint8_t* umi, uint64_t* idx, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[4] = {{recs, 24}, {bc, bc_len}, {umi, umi_len}, {idx, 8}};
  const size_t kDecRecs = (size_t)kDecRecsTable[mode_of_len(bc_len)][mode_of_len(umi_len)];
  const RowSplit rs = split_rows(cfg, sp, 4, n, kDecRecs);   // peel rows until every array is 16-B aligned
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_decode_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, (const u64*)recs, (u64)0,
                       (u64)rs.head, bc_len, umi_len, cfg.base_order, bc, umi, (u64*)idx);
  if (rs.main) {
    const u32 ntiles = (u32)(rs.main / kDecRecs);
    const int mb = mode_of_len(bc_len), mu = mode_of_len(umi_len);
    const int mo = cfg.base_order ? 1 : 0;
    const DecFn fn = kDecTable[mo][mb][mu];
    static std::atomic<int> occ[2][kNumLenModes][kNumLenModes];
    hipLaunchKernelGGL(fn, dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, fn, 0, &occ[mo][mb][mu]))), dim3(kBlock), 0,
                       st, adv((const uint8_t*)recs, 24 * rs.head), ntiles, bc_len, umi_len, adv(bc, rs.head * bc_len),
                       adv(umi, rs.head * umi_len), adv((u64*)idx, 8 * rs.head));
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_decode_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, (const u64*)recs,
                       (u64)(rs.head + rs.main), (u64)n, bc_len, umi_len, cfg.base_order, bc, umi, (u64*)idx);
  return hipGetLastError();
}

typedef void (*UnpFn)(const u64*, u32, u32, uint8_t*);
#define IBU_UNP_ROW(M) {ibu_k_unpack<len_of_mode(0), M>, ibu_k_unpack<len_of_mode(1), M>, ibu_k_unpack<len_of_mode(2), M>, \
                        ibu_k_unpack<len_of_mode(3), M>, ibu_k_unpack<len_of_mode(4), M>}
static const UnpFn kUnpTable[2][kNumLenModes] = {IBU_UNP_ROW(false), IBU_UNP_ROW(true)};

hipError_t launch_unpack(const LaunchCfg& cfg, const uint64_t* codes, size_t n, uint32_t len, uint8_t* out,
                         hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[2] = {{codes, 8}, {out, len}};
  const size_t tile_recs = (size_t)kTileRecs * unpack_nt(len_of_mode(mode_of_len(len)));
  const RowSplit rs = split_rows(cfg, sp, 2, n, tile_recs);
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_unpack_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, (const u64*)codes, (u64)0, (u64)rs.head,
                       len, cfg.base_order, out);
  if (rs.main) {
    const u32 ntiles = (u32)(rs.main / tile_recs);
    const int m = mode_of_len(len), mo = cfg.base_order ? 1 : 0;
    static std::atomic<int> occ[2][kNumLenModes];
    hipLaunchKernelGGL(kUnpTable[mo][m], dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, kUnpTable[mo][m], 0, &occ[mo][m]))),
                       dim3(kBlock), 0, st, adv((const u64*)codes, 8 * rs.head), ntiles, len, adv(out, rs.head * len));
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_unpack_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, (const u64*)codes,
                       (u64)(rs.head + rs.main), (u64)n, len, cfg.base_order, out);
  return hipGetLastError();
}

}  // namespace ibu
