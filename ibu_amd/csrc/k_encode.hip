// k_encode.hip — K3 fused encode (barcode ASCII, UMI ASCII, index column -> AoS records) and
// the single-column 2-bit pack.  Design notes: kcommon.hpp.  Reference semantics: the 2-bit
// table of src/constructs/record.rs:19-27, Record::new (record.rs:87-93) and the cast_slice of
// Writer::write_batch (writer.rs:315-318); invalid bytes follow bitnuc's InvalidBase contract
// (README.md:45) — see include/ibu_hip.h.
#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {

// BC / UM: compile-time barcode / UMI length, or 0 for "runtime length" (generic kernel).
// Dynamic LDS per wave: max(3072, 128*(bc_len+umi_len)) bytes; the AoS tile reuses the ASCII
// staging area once every row has been packed (in-order DS makes that safe).
// Lane L owns the ADJACENT rows 2L, 2L+1: its index pair is one 16-B chunk of the column and
// its two records are 48 contiguous bytes of the AoS tile (3 x ds_write_b128 at stride 48 B,
// conflict-free).  All global loads of a tile are issued back to back into registers; right
// after they have been copied to LDS the same registers are re-loaded with the NEXT tile, so
// one tile of HBM reads is always in flight behind the packing of the current one.
#ifndef IBU_ENCODE_MINWAVES
#define IBU_ENCODE_MINWAVES 6
#endif
// Register budget (waves/SIMD): short dword-path rows fit 80 VGPRs; 32-base rows need 128;
// byte-path (len % 4 != 0) and generic kernels get 168 — none of the instantiations spills.
constexpr int encode_minwaves(int bc, int um) {
  const bool dw = bc > 0 && um > 0 && (bc & 3) == 0 && (um & 3) == 0;
  return !dw ? 3 : (bc <= 16 && um <= 16) ? IBU_ENCODE_MINWAVES : 4;
}
template <int BC, int UM, bool MSB>
__global__ void __launch_bounds__(kBlock, encode_minwaves(BC, UM))
ibu_k_encode(const uint8_t* __restrict__ bc_in, const uint8_t* __restrict__ umi_in,
             const u64* __restrict__ idx_in, u64 first_index, u64 row_base, u32 ntiles, u32 bc_len, u32 umi_len,
             u32 wave_lds_bytes, uint8_t* __restrict__ recs, u64* __restrict__ status) {
  // row_base: rows the launcher peeled off in front of this launch (kcommon.hpp, "Peeling"); bad rows are reported
  // in the caller's numbering, and first_index already includes it
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  if (BC > 0) bc_len = BC;
  if (UM > 0) umi_len = UM;
  uint8_t* area = dyn_lds + wib * wave_lds_bytes;
  uint8_t* asc_bc = area;
  uint8_t* asc_umi = area + kTileRecs * bc_len;
  const TileRange tr = tile_range(ntiles, wib);
  const u32 nwaves = tr.stride;
  u32 t = tr.t;
  ntiles = tr.end;
  if (t >= ntiles) return;
  // idx_in == NULL: the loads below read the (valid, 16-B aligned) barcode column instead and
  // the result is ignored, so the instruction stream has no branch around a load.
  const uint8_t* idx_src = idx_in ? reinterpret_cast<const uint8_t*>(idx_in) + 16 * lane : bc_in;
  const size_t idx_tile = idx_in ? 1024 : 0;

  AsciiStage<BC> sb;
  AsciiStage<UM> su;
  sb.issue(bc_in + (size_t)t * kTileRecs * bc_len, bc_len, lane);
  su.issue(umi_in + (size_t)t * kTileRecs * umi_len, umi_len, lane);
  u32x4 vi = ld16(idx_src + (size_t)t * idx_tile);
  BadRows bad;
  for (;;) {
    const size_t row0 = (size_t)t * kTileRecs;
    wave_lds_fence();                          // previous tile's AoS reads precede these writes
    sb.land(asc_bc, bc_len, lane);
    su.land(asc_umi, umi_len, lane);
    u64 i0 = ((u64)vi.y << 32) | vi.x, i1 = ((u64)vi.w << 32) | vi.z;
    if (!idx_in) { i0 = first_index + row0 + 2 * lane; i1 = i0 + 1; }
    const u32 tn = t + nwaves;
    const bool more = tn < ntiles;             // wave-uniform
    const u32 tp = more ? tn : t;              // registers are free again: next tile goes in flight
    sb.issue(bc_in + (size_t)tp * kTileRecs * bc_len, bc_len, lane);   // unconditional, see kcommon.hpp
    su.issue(umi_in + (size_t)tp * kTileRecs * umi_len, umi_len, lane);
    vi = ld16(idx_src + (size_t)tp * idx_tile);
    wave_lds_fence();
    bool okb0 = true, oku0 = true, okb1 = true, oku1 = true;
#if IBU_PROBE == 1      // measurement build (WRONG output): the LDS row reads stay, the packing ALU goes
    u64 b0 = *reinterpret_cast<const u32*>(asc_bc + (2 * lane) * bc_len), u0 = *reinterpret_cast<const u32*>(asc_umi + (2 * lane) * umi_len);
    u64 b1 = *reinterpret_cast<const u32*>(asc_bc + (2 * lane + 1) * bc_len), u1 = *reinterpret_cast<const u32*>(asc_umi + (2 * lane + 1) * umi_len);
#elif IBU_PROBE == 2    // measurement build (WRONG output): no LDS row reads either
    u64 b0 = row0 + lane, u0 = b0 * 3, b1 = b0 + 7, u1 = b0 ^ 5;
#else
    u64 b0 = pack_row<BC>(asc_bc + (2 * lane) * bc_len, bc_len, okb0);
    u64 u0 = pack_row<UM>(asc_umi + (2 * lane) * umi_len, umi_len, oku0);
    u64 b1 = pack_row<BC>(asc_bc + (2 * lane + 1) * bc_len, bc_len, okb1);
    u64 u1 = pack_row<UM>(asc_umi + (2 * lane + 1) * umi_len, umi_len, oku1);
#endif
    if (!okb0) b0 = 0;
    if (!oku0) u0 = 0;
    if (!okb1) b1 = 0;
    if (!oku1) u1 = 0;
    bad.note(!(okb0 && oku0), row_base + row0 + 2 * lane);
    bad.note(!(okb1 && oku1), row_base + row0 + 2 * lane + 1);
    wave_lds_fence();                          // all ASCII reads done before the area is reused
    u64* r = reinterpret_cast<u64*>(area + lane * 48);
    r[0] = b0; r[1] = u0; r[2] = i0; r[3] = b1; r[4] = u1; r[5] = i1;
    wave_lds_fence();
    if constexpr (MSB) {                       // first base most significant: rewrite the staged AoS tile
      rev_pairs_tile(area, 24, 0, bc_len, lane);   // (lane L owns records 2L, 2L+1 = the 48 bytes it just wrote)
      rev_pairs_tile(area, 24, 8, umi_len, lane);
      wave_lds_fence();
    }
    uint8_t* dst = recs + (size_t)t * kTileBytes + 16 * lane;
    st16(dst, *reinterpret_cast<const u32x4*>(area + 16 * lane));
    st16(dst + 1024, *reinterpret_cast<const u32x4*>(area + 1024 + 16 * lane));
    st16(dst + 2048, *reinterpret_cast<const u32x4*>(area + 2048 + 16 * lane));
    if (!more) break;
    t = tn;
  }
  bad.flush(status);
}

// Single ASCII column -> u64 codes.
template <int LEN, bool MSB>
__global__ void __launch_bounds__(kBlock, (LEN > 0 && LEN <= 16 && (LEN & 3) == 0) ? 8 : 4)
ibu_k_pack(const uint8_t* __restrict__ in, u64 row_base, u32 ntiles, u32 len, u64* __restrict__ codes,
           u64* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kTileRecs * 32];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* asc = lds + wib * kTileRecs * 32;
  const TileRange tr = tile_range(ntiles, wib);   // which tiles this wave sweeps (kcommon.hpp)
  const u32 nwaves = tr.stride;
  u32 t = tr.t;
  ntiles = tr.end;
  if (LEN > 0) len = LEN;
  if (t >= ntiles) return;
  AsciiStage<LEN> sv;
  sv.issue(in + (size_t)t * kTileRecs * len, len, lane);
  BadRows bad;
  for (;;) {
    const size_t row0 = (size_t)t * kTileRecs;
    wave_lds_fence();
    sv.land(asc, len, lane);
    const u32 tn = t + nwaves;
    const bool more = tn < ntiles;
    sv.issue(in + (size_t)(more ? tn : t) * kTileRecs * len, len, lane);  // unconditional, see kcommon.hpp
    wave_lds_fence();
    bool ok0 = true, ok1 = true;
    u64 v0 = pack_row<LEN>(asc + (2 * lane) * len, len, ok0);
    u64 v1 = pack_row<LEN>(asc + (2 * lane + 1) * len, len, ok1);
    if constexpr (MSB) { v0 = rev_pairs(v0, len); v1 = rev_pairs(v1, len); }
    if (!ok0) v0 = 0;
    if (!ok1) v1 = 0;
    bad.note(!ok0, row_base + row0 + 2 * lane);
    bad.note(!ok1, row_base + row0 + 2 * lane + 1);
    u32x4 o; o.x = (u32)v0; o.y = (u32)(v0 >> 32); o.z = (u32)v1; o.w = (u32)(v1 >> 32);
    st16(reinterpret_cast<uint8_t*>(codes) + row0 * 8 + 16 * lane, o);
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
#define IBU_ENC_ROW(B, M) {enc_entry<B, 0, M>(), enc_entry<B, 1, M>(), enc_entry<B, 2, M>(), enc_entry<B, 3, M>(), enc_entry<B, 4, M>(), enc_entry<B, 5, M>()}
#define IBU_ENC_TABLE(M) {IBU_ENC_ROW(0, M), IBU_ENC_ROW(1, M), IBU_ENC_ROW(2, M), IBU_ENC_ROW(3, M), IBU_ENC_ROW(4, M), IBU_ENC_ROW(5, M)}
static const EncFn kEncTable[2][kNumLenModes][kNumLenModes] = {IBU_ENC_TABLE(false), IBU_ENC_TABLE(true)};  // [base_order][bc][umi]

hipError_t launch_encode(const LaunchCfg& cfg, const uint8_t* bc, const uint8_t* umi, const uint64_t* idx,
                         uint64_t first_index, size_t n, uint32_t bc_len, uint32_t umi_len, void* recs,
                         uint64_t* status, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[4] = {{recs, 24}, {bc, bc_len}, {umi, umi_len}, {idx, 8}};
  const RowSplit rs = split_rows(sp, 4, n, kTileRecs);   // peel rows until every array is 16-B aligned
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_encode_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, bc, umi, (const u64*)idx, (u64)first_index,
                       (u64)0, (u64)rs.head, bc_len, umi_len, cfg.base_order, (u64*)recs, (u64*)status);
  if (rs.main) {
    const u32 ntiles = (u32)(rs.main / kTileRecs);
    u32 wave_lds = kTileRecs * (bc_len + umi_len);
    if (wave_lds < (u32)kTileBytes) wave_lds = kTileBytes;
    const int mb = mode_of_len(bc_len), mu = mode_of_len(umi_len);
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
#define IBU_PACK_ROW(M) {ibu_k_pack<len_of_mode(0), M>, ibu_k_pack<len_of_mode(1), M>, ibu_k_pack<len_of_mode(2), M>, \
                         ibu_k_pack<len_of_mode(3), M>, ibu_k_pack<len_of_mode(4), M>, ibu_k_pack<len_of_mode(5), M>}
static const PackFn kPackTable[2][kNumLenModes] = {IBU_PACK_ROW(false), IBU_PACK_ROW(true)};

hipError_t launch_pack(const LaunchCfg& cfg, const uint8_t* in, size_t n, uint32_t len, uint64_t* codes,
                       uint64_t* status, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[2] = {{in, len}, {codes, 8}};
  const RowSplit rs = split_rows(sp, 2, n, kTileRecs);
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_pack_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, in, (u64)0, (u64)rs.head, len, cfg.base_order,
                       (u64*)codes, (u64*)status);
  if (rs.main) {
    const u32 ntiles = (u32)(rs.main / kTileRecs);
    const int m = mode_of_len(len), mo = cfg.base_order ? 1 : 0;
    static std::atomic<int> occ[2][kNumLenModes];
    hipLaunchKernelGGL(kPackTable[mo][m], dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, kPackTable[mo][m], 0, &occ[mo][m]))),
                       dim3(kBlock), 0, st, adv(in, rs.head * len), (u64)rs.head, ntiles, len, adv((u64*)codes, 8 * rs.head),
                       (u64*)status);
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_pack_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, in, (u64)(rs.head + rs.main),
                       (u64)n, len, cfg.base_order, (u64*)codes, (u64*)status);
  return hipGetLastError();
}

}  // namespace ibu
