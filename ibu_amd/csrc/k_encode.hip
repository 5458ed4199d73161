// k_encode.hip — K3 fused encode (barcode ASCII, UMI ASCII, index column -> AoS records) and
// the single-column 2-bit pack.  Design notes: kcommon.hpp.  Reference semantics: the 2-bit
// table of src/constructs/record.rs:19-27, Record::new (record.rs:87-93) and the cast_slice of
// Writer::write_batch (writer.rs:315-318); invalid bytes follow bitnuc's InvalidBase contract
// (README.md:45) — see include/ibu_hip.h.
#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {

// BC / UM: compile-time barcode / UMI length, or 0 for "runtime length" (generic kernel).
// Dynamic LDS per wave: max(3072, the two fields' staging areas: 128*len bytes of ASCII, or the 32*len + 16 bytes of the
// code stream of a runtime-length field); the AoS tile reuses the staging area once every row has been packed
// (in-order DS makes that safe).
// Lane L owns the ADJACENT rows 2L, 2L+1: its index pair is one 16-B chunk of the column and
// its two records are 48 contiguous bytes of the AoS tile (3 x ds_write_b128 at stride 48 B,
// conflict-free).  All global loads of a tile are issued back to back into registers; right
// after they have been copied to LDS the same registers are re-loaded with the NEXT tile, so
// one tile of HBM reads is always in flight behind the packing of the current one.
#ifndef IBU_ENCODE_MINWAVES
#define IBU_ENCODE_MINWAVES 6
#endif
// Register budget (waves/SIMD): short rows fit 80 VGPRs; 32-base rows need 128 — none of the instantiations spills.
// A runtime-length field (0) lands as its code stream and costs no more registers than a specialised one (round 4; it was
// packed byte by byte under a 168-VGPR budget before, like the length-10 specialisation that round 4 retired).
#ifndef IBU_ENCODE_GEN_MINWAVES
#define IBU_ENCODE_GEN_MINWAVES 4
#endif
constexpr int encode_minwaves(int bc, int um) {
  if (bc == 0 || um == 0) return IBU_ENCODE_GEN_MINWAVES;
  return (bc <= 16 && um <= 16) ? IBU_ENCODE_MINWAVES : 4;
}
// LDS of one field of one 128-row tile: the ASCII rows, or the code stream of a runtime-length field.
__host__ __device__ constexpr u32 enc_field_lds(int spec_len, u32 len) { return spec_len > 0 ? 128u * len : 32u * len + kStreamPad; }
// Rows per wave iteration = 128 * NT.  Round 4 measured two tiles per iteration for the dword-path rows of at most 32 bases
// (at (8,8) one tile is 3 KiB of loads in flight per wave against the 6 KiB of decode's two), same box, same
// arrays, one process (profiles/r04_d_kbench_enc_nt.jsonl): (8,8) 0.756 -> 0.759 of peak, (12,8) 0.752 -> 0.757, (12,12) and
// (16,12) equal, (16,16) needs 128 VGPRs and four waves per SIMD to tie — the kernel does not wait for more bytes in flight
// (its probe builds without ALU or LDS traffic are not faster either: profiles/README.md r02_i).  NT stays 1;
// -DIBU_ENCODE_NT=2 builds the other form on every dword-path instantiation for the next A/B.
template <int BC, int UM>
constexpr int enc_nt() {
#ifdef IBU_ENCODE_NT
  return (BC > 0 && UM > 0 && (BC & 3) == 0 && (UM & 3) == 0) ? IBU_ENCODE_NT : 1;
#else
  return 1;
#endif
}
#ifndef IBU_ENCODE_NT2_MINWAVES
#define IBU_ENCODE_NT2_MINWAVES 5
#endif
template <int BC, int UM>
constexpr int enc_waves() { return enc_nt<BC, UM>() > 1 ? IBU_ENCODE_NT2_MINWAVES : encode_minwaves(BC, UM); }

template <int BC, int UM, bool MSB>
__global__ void __launch_bounds__(kBlock, (enc_waves<BC, UM>()))
ibu_k_encode(const uint8_t* __restrict__ bc_in, const uint8_t* __restrict__ umi_in,
             const u64* __restrict__ idx_in, u64 first_index, u64 row_base, u32 ntiles, u32 bc_len, u32 umi_len,
             u32 wave_lds_bytes, uint8_t* __restrict__ recs, u64* __restrict__ status) {
  // row_base: rows the launcher peeled off in front of this launch (kcommon.hpp, "Peeling"); bad rows are reported
  // in the caller's numbering, and first_index already includes it
  constexpr int NT = enc_nt<BC, UM>(), kRecs = kTileRecs * NT;
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  if (BC > 0) bc_len = BC;
  if (UM > 0) umi_len = UM;
  uint8_t* area = dyn_lds + wib * wave_lds_bytes;
  uint8_t* asc_bc = area;
  uint8_t* asc_umi = area + enc_field_lds(BC, bc_len) * NT;
  const TileRange tr = tile_range(ntiles, wib);
  const u32 nwaves = tr.stride;
  u32 t = tr.t;
  ntiles = tr.end;
  if (t >= ntiles) return;
  // idx_in == NULL: the loads below read the (valid, 16-B aligned) barcode column instead and
  // the result is ignored, so the instruction stream has no branch around a load.
  const uint8_t* idx_src = idx_in ? reinterpret_cast<const uint8_t*>(idx_in) + 16 * lane : bc_in;
  const size_t idx_tile = idx_in ? 1024 * NT : 0, idx_sub = idx_in ? 1024 : 0;

  AsciiStage<BC, NT> sb;
  AsciiStage<UM, NT> su;
  u32x4 vi[NT];
  sb.issue(bc_in + (size_t)t * kRecs * bc_len, bc_len, lane);
  su.issue(umi_in + (size_t)t * kRecs * umi_len, umi_len, lane);
#pragma unroll
  for (int j = 0; j < NT; ++j) vi[j] = ld16(idx_src + (size_t)t * idx_tile + j * idx_sub);
  BadRows bad;
  for (;;) {
    const size_t row0 = (size_t)t * kRecs;
    wave_lds_fence();                          // previous tile's AoS reads precede these writes
    const bool okcb = sb.land(asc_bc, bc_len, lane), okcu = su.land(asc_umi, umi_len, lane);
    const bool chunks_ok = okcb && okcu;       // false only from a runtime-length field
    u64 ix[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      ix[j][0] = ((u64)vi[j].y << 32) | vi[j].x;
      ix[j][1] = ((u64)vi[j].w << 32) | vi[j].z;
      if (!idx_in) { ix[j][0] = first_index + row0 + 128 * j + 2 * lane; ix[j][1] = ix[j][0] + 1; }
    }
    const u32 tn = t + nwaves;
    const bool more = tn < ntiles;             // wave-uniform
    const u32 tp = more ? tn : t;              // registers are free again: next tile goes in flight
    sb.issue(bc_in + (size_t)tp * kRecs * bc_len, bc_len, lane);   // unconditional, see kcommon.hpp
    su.issue(umi_in + (size_t)tp * kRecs * umi_len, umi_len, lane);
#pragma unroll
    for (int j = 0; j < NT; ++j) vi[j] = ld16(idx_src + (size_t)tp * idx_tile + j * idx_sub);
    wave_lds_fence();
    u64 b[NT][2], u[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32 r = 128 * j + 2 * lane + h;
        bool okb = true, oku = true;
#if IBU_PROBE == 1      // measurement build (WRONG output): the LDS row reads stay, the packing ALU goes
        b[j][h] = *reinterpret_cast<const u32*>(asc_bc + r * bc_len);
        u[j][h] = *reinterpret_cast<const u32*>(asc_umi + r * umi_len);
#elif IBU_PROBE == 2    // measurement build (WRONG output): no LDS row reads either
        b[j][h] = row0 + r;
        u[j][h] = b[j][h] * 3;
#else
        b[j][h] = pack_row<BC>(asc_bc, r, bc_len, okb);
        u[j][h] = pack_row<UM>(asc_umi, r, umi_len, oku);
        if constexpr (BC == 0 || UM == 0) {
          // A runtime-length field was packed chunk by chunk, which does not say WHICH row holds the offending byte: a tile
          // with one (wave-uniform test, never taken on valid input) packs the rows of those fields again, byte by byte from
          // the column in global memory, as the tail kernel does.
          if (__ballot(!chunks_ok) != 0) {
            if constexpr (BC == 0) b[j][h] = pack_row_bytes(bc_in + (row0 + r) * bc_len, bc_len, okb);
            if constexpr (UM == 0) u[j][h] = pack_row_bytes(umi_in + (row0 + r) * umi_len, umi_len, oku);
          }
        }
#endif
        if (!okb) b[j][h] = 0;
        if (!oku) u[j][h] = 0;
        bad.note(!(okb && oku), row_base + row0 + r);
      }
    wave_lds_fence();                          // all ASCII reads done before the area is reused
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      u64* r = reinterpret_cast<u64*>(area + (128 * j + 2 * lane) * 24);
      r[0] = b[j][0]; r[1] = u[j][0]; r[2] = ix[j][0]; r[3] = b[j][1]; r[4] = u[j][1]; r[5] = ix[j][1];
    }
    wave_lds_fence();
    if constexpr (MSB) {                       // first base most significant: rewrite the staged AoS tile
      rev_pairs_tile<NT>(area, 24, 0, bc_len, lane);   // (lane L owns records 128j + 2L, 2L+1 = the 48 bytes it just wrote)
      rev_pairs_tile<NT>(area, 24, 8, umi_len, lane);
      wave_lds_fence();
    }
    uint8_t* dst = recs + (size_t)t * (kTileBytes * NT) + 16 * lane;
#pragma unroll
    for (int k = 0; k < 3 * NT; ++k) st16(dst + 1024 * k, *reinterpret_cast<const u32x4*>(area + 1024 * k + 16 * lane));
    if (!more) break;
    t = tn;
  }
  bad.flush(status);
}

// Single ASCII column -> u64 codes.  NT 128-row tiles per wave iteration: two for the specialised rows of at most 16 bases, at five
// or four waves per SIMD instead of eight (round 3 measured two tiles at eight waves: they spilled) — same arrays, one process, 1e9 rows
// (profiles/r04_t_kbench_pack_unpack.jsonl): pack<16> 5.17 -> 5.37 TB/s, pack<12> 5.15 -> 5.29, pack<8> 5.24 -> 5.32.
// Runtime lengths (LEN == 0) are staged as their code stream, whose dword c is chunk c of the column whatever the length: GNT
// tiles and RND load rounds per iteration chosen by the launcher so that every round carries chunks — 4 tiles for rows of at most
// 8 bases, 2 beyond, and ceil(len * GNT / 8) rounds.  Round 4 staged one tile four rounds deep at any length: at 5
// bases three of the four loads of a lane re-read the column's last chunk (2.0 vector loads per row where 0.31 carry data; SQ
// pass profiles/r05_b_sq5) and the kernel ran 0.86 of its own 5R:8W yardstick.
constexpr int pack_nt(int len) { return (len > 0 && len <= 16) ? 2 : 1; }
template <int LEN, bool MSB, int GNT = 1, int RND = 0>
__global__ void __launch_bounds__(kBlock, LEN == 0 ? (RND >= 7 ? 5 : (GNT >= 4 || RND >= 5) ? 6 : 8) : LEN <= 12 ? 5 : 4)   // (what two tiles of rows leave: 69-80 / 108 VGPRs; four
                                                                                              // tiles of runtime-length rows spill under 64)
ibu_k_pack(const uint8_t* __restrict__ in, u64 row_base, u32 ntiles /*of NT x 128 rows*/, u32 len, u64* __restrict__ codes,
           u64* __restrict__ status) {
  constexpr int NT = LEN > 0 ? pack_nt(LEN) : GNT, kRows = kTileRecs * NT;
  // a wave's slice: the ASCII rows of its tiles, or (LEN == 0) their code stream — 4 bytes per chunk, 64 chunks per round, padded
  // for stream_row's third dword
  constexpr u32 kWaveLds = LEN > 0 ? (u32)(kRows * LEN) : (u32)(256 * (RND > 0 ? RND : 4)) + kStreamPad;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kWaveLds];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* asc = lds + wib * kWaveLds;
  const TileRange tr = tile_range(ntiles, wib);   // which tiles this wave sweeps (kcommon.hpp)
  const u32 nwaves = tr.stride;
  u32 t = tr.t;
  ntiles = tr.end;
  if (LEN > 0) len = LEN;
  if (t >= ntiles) return;
  AsciiStage<LEN, NT, RND> sv;
  sv.issue(in + (size_t)t * kRows * len, len, lane);
  BadRows bad;
  for (;;) {
    const size_t row0 = (size_t)t * kRows;
    wave_lds_fence();
    const bool chunks_ok = sv.land(asc, len, lane);
    const u32 tn = t + nwaves;
    const bool more = tn < ntiles;
    sv.issue(in + (size_t)(more ? tn : t) * kRows * len, len, lane);  // unconditional, see kcommon.hpp
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const u32 r = 128 * j + 2 * lane;
      bool ok0 = true, ok1 = true;
      u64 v0 = pack_row<LEN>(asc, r, len, ok0);
      u64 v1 = pack_row<LEN>(asc, r + 1, len, ok1);
      if constexpr (LEN == 0) {                    // a tile with an offending byte: its rows again, byte by byte (see ibu_k_encode)
        if (__ballot(!chunks_ok) != 0) {
          v0 = pack_row_bytes(in + (row0 + r) * len, len, ok0);
          v1 = pack_row_bytes(in + (row0 + r + 1) * len, len, ok1);
        }
      }
      if constexpr (MSB) { v0 = rev_pairs(v0, len); v1 = rev_pairs(v1, len); }
      if (!ok0) v0 = 0;
      if (!ok1) v1 = 0;
      bad.note(!ok0, row_base + row0 + r);
      bad.note(!ok1, row_base + row0 + r + 1);
      u32x4 o; o.x = (u32)v0; o.y = (u32)(v0 >> 32); o.z = (u32)v1; o.w = (u32)(v1 >> 32);
      st16(reinterpret_cast<uint8_t*>(codes) + (row0 + 128 * j) * 8 + 16 * lane, o);
    }
    if (!more) break;
    t = tn;
  }
  bad.flush(status);
}

// ---- tails: one thread per record, any alignment -----------------------------------------------
extern "C" __global__ void ibu_k_encode_tail(const uint8_t* bc_in, const uint8_t* umi_in, const u64* idx_in,
                                             u64 first_index, u64 row0, u64 n, u32 bc_len, u32 umi_len, u32 msb,
                                             u64* __restrict__ recs, u64* status) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool okb = true, oku = true;
  u64 b = pack_row_bytes(bc_in + i * bc_len, bc_len, okb);
  u64 u = pack_row_bytes(umi_in + i * umi_len, umi_len, oku);
  if (msb) { b = rev_pairs(b, bc_len); u = rev_pairs(u, umi_len); }
  if (!okb) b = 0;
  if (!oku) u = 0;
  if (!(okb && oku)) { atomicMin(&status[0], i); atomicAdd(&status[1], 1ull); }
  recs[3 * i] = b; recs[3 * i + 1] = u; recs[3 * i + 2] = idx_in ? idx_in[i] : first_index + i;
}
extern "C" __global__ void ibu_k_pack_tail(const uint8_t* in, u64 row0, u64 n, u32 len, u32 msb, u64* codes, u64* status) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = true;
  u64 v = pack_row_bytes(in + i * len, len, ok);
  if (msb) v = rev_pairs(v, len);
  if (!ok) { v = 0; atomicMin(&status[0], i); atomicAdd(&status[1], 1ull); }
  codes[i] = v;
}

// ---- launchers ------------------------------------------------------------------------------------
typedef void (*EncFn)(const uint8_t*, const uint8_t*, const u64*, u64, u64, u32, u32, u32, u32, uint8_t*, u64*);
template <int B, int U, bool M>
static constexpr EncFn enc_entry() { return ibu_k_encode<len_of_mode(B), len_of_mode(U), M>; }
#define IBU_ENC_ROW(B, M) {enc_entry<B, 0, M>(), enc_entry<B, 1, M>(), enc_entry<B, 2, M>(), enc_entry<B, 3, M>(), enc_entry<B, 4, M>()}
#define IBU_ENC_TABLE(M) {IBU_ENC_ROW(0, M), IBU_ENC_ROW(1, M), IBU_ENC_ROW(2, M), IBU_ENC_ROW(3, M), IBU_ENC_ROW(4, M)}
static const EncFn kEncTable[2][kNumLenModes][kNumLenModes] = {IBU_ENC_TABLE(false), IBU_ENC_TABLE(true)};  // [base_order][bc][umi]
// 128-row tiles per wave iteration of each instantiation (the same for both base orders)
template <int B, int U>
static constexpr int enc_nt_of() { return enc_nt<len_of_mode(B), len_of_mode(U)>(); }
#define IBU_ENC_NT_ROW(B) {enc_nt_of<B, 0>(), enc_nt_of<B, 1>(), enc_nt_of<B, 2>(), enc_nt_of<B, 3>(), enc_nt_of<B, 4>()}
static const int kEncNtTable[kNumLenModes][kNumLenModes] = {IBU_ENC_NT_ROW(0), IBU_ENC_NT_ROW(1), IBU_ENC_NT_ROW(2),
                                                            IBU_ENC_NT_ROW(3), IBU_ENC_NT_ROW(4)};

hipError_t launch_encode(const LaunchCfg& cfg, const uint8_t* bc, const uint8_t* umi, const uint64_t* idx,
                         uint64_t first_index, size_t n, uint32_t bc_len, uint32_t umi_len, void* recs,
                         uint64_t* status, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[4] = {{recs, 24}, {bc, bc_len}, {umi, umi_len}, {idx, 8}};
  const int mb = mode_of_len(bc_len), mu = mode_of_len(umi_len);
  const u32 nt = (u32)kEncNtTable[mb][mu];
  const size_t tile_recs = (size_t)kTileRecs * nt;
  const RowSplit rs = split_rows(cfg, sp, 4, n, tile_recs);   // peel rows until every array is 16-B aligned
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_encode_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, bc, umi, (const u64*)idx, (u64)first_index,
                       (u64)0, (u64)rs.head, bc_len, umi_len, cfg.base_order, (u64*)recs, (u64*)status);
  if (rs.main) {
    const u32 ntiles = (u32)(rs.main / tile_recs);
    u32 wave_lds = (enc_field_lds(len_of_mode(mb), bc_len) + enc_field_lds(len_of_mode(mu), umi_len)) * nt;
    if (wave_lds < (u32)kTileBytes * nt) wave_lds = kTileBytes * nt;
    const int mo = cfg.base_order ? 1 : 0;
    const EncFn fn = kEncTable[mo][mb][mu];
    static std::atomic<int> occ[2][33][33];  // LDS depends on the actual lengths, not only on the mode
    const int nb = resident_blocks<kBlock>(cfg, fn, wave_lds * kWavesPerBlock, &occ[mo][bc_len][umi_len]);
    hipLaunchKernelGGL(fn, dim3(grid_for(ntiles, cfg.cus, nb)), dim3(kBlock), wave_lds * kWavesPerBlock, st,
                       adv(bc, rs.head * bc_len), adv(umi, rs.head * umi_len), adv((const u64*)idx, 8 * rs.head),
                       (u64)(first_index + rs.head), (u64)rs.head, ntiles, bc_len, umi_len, wave_lds,
                       adv((uint8_t*)recs, 24 * rs.head), (u64*)status);
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_encode_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, bc, umi, (const u64*)idx,
                       (u64)first_index, (u64)(rs.head + rs.main), (u64)n, bc_len, umi_len, cfg.base_order, (u64*)recs, (u64*)status);
  return hipGetLastError();
}

typedef void (*PackFn)(const uint8_t*, u64, u32, u32, u64*, u64*);
#define IBU_PACK_ROW(M) {nullptr /* runtime lengths: pack_gen_entry */, ibu_k_pack<len_of_mode(1), M>, ibu_k_pack<len_of_mode(2), M>, \
                         ibu_k_pack<len_of_mode(3), M>, ibu_k_pack<len_of_mode(4), M>}
static const PackFn kPackTable[2][kNumLenModes] = {IBU_PACK_ROW(false), IBU_PACK_ROW(true)};
// Runtime lengths: tiles per iteration (4 for rows of at most 8 bases, 2 beyond) and load rounds = ceil(len * tiles / 8).
struct PackShape { int nt, rounds; };
static inline PackShape pack_shape(uint32_t len) {
  const int nt = len <= 8 ? 4 : 2;
  return {nt, (int)(len * nt + 7) / 8};
}
template <bool M>
static PackFn pack_gen_entry(int nt, int rounds) {
  switch (nt * 10 + rounds) {
    case 41: return ibu_k_pack<0, M, 4, 1>;   // 1-2 bases
    case 42: return ibu_k_pack<0, M, 4, 2>;   // 3-4
    case 43: return ibu_k_pack<0, M, 4, 3>;   // 5-6
    case 44: return ibu_k_pack<0, M, 4, 4>;   // 7 (8 has its specialisation)
    case 23: return ibu_k_pack<0, M, 2, 3>;   // 9-11 (12 specialised)
    case 24: return ibu_k_pack<0, M, 2, 4>;   // 13-15
    case 25: return ibu_k_pack<0, M, 2, 5>;   // 17-20 (round 5 measured one tile, three rounds at 20 bases: 0.93 of its 5R:2W yardstick)
    case 26: return ibu_k_pack<0, M, 2, 6>;   // 21-24
    case 27: return ibu_k_pack<0, M, 2, 7>;   // 25-28 (one tile, four rounds at 25 bases: 0.66 of peak where 24 ran 0.73)
    default: return ibu_k_pack<0, M, 2, 8>;   // 29-31
  }
}

hipError_t launch_pack(const LaunchCfg& cfg, const uint8_t* in, size_t n, uint32_t len, uint64_t* codes,
                       uint64_t* status, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[2] = {{in, len}, {codes, 8}};
  const int m = mode_of_len(len), mo = cfg.base_order ? 1 : 0;
  const PackShape shape = m ? PackShape{pack_nt(len_of_mode(m)), 0} : pack_shape(len);
  const size_t tile_rows = (size_t)kTileRecs * shape.nt;
  const RowSplit rs = split_rows(cfg, sp, 2, n, tile_rows);
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_pack_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, in, (u64)0, (u64)rs.head, len, cfg.base_order,
                       (u64*)codes, (u64*)status);
  if (rs.main) {
    const u32 ntiles = (u32)(rs.main / tile_rows);
    const PackFn fn = m ? kPackTable[mo][m] : (mo ? pack_gen_entry<true>(shape.nt, shape.rounds) : pack_gen_entry<false>(shape.nt, shape.rounds));
    static std::atomic<int> occ[2][kNumLenModes], occ_gen[2][33];
    std::atomic<int>* oc = m ? &occ[mo][m] : &occ_gen[mo][len];
    hipLaunchKernelGGL(fn, dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, fn, 0, oc))),
                       dim3(kBlock), 0, st, adv(in, rs.head * len), (u64)rs.head, ntiles, len, adv((u64*)codes, 8 * rs.head),
                       (u64*)status);
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_pack_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, in, (u64)(rs.head + rs.main),
                       (u64)n, len, cfg.base_order, (u64*)codes, (u64*)status);
  return hipGetLastError();
}

}  // namespace ibu
