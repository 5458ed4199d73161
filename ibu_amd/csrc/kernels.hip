// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for the IBU hot path.
//
// Everything here is HBM-bound integer / byte work: no MFMA.  The design rules are the
// streaming ones: every global access is a fully coalesced 16 B-per-lane wave instruction
// (1 KiB per instruction), the 24-byte record stride is absorbed in LDS (a wave-private
// 128-record = 3 KiB tile; stride-24 ds_read_b64 is bank-conflict-free because
// 6 dwords * k mod 64 is distinct for 32 consecutive k), persistent grid-stride waves keep a
// tile in flight in registers while the previous one is processed, and no workgroup barrier
// is ever needed (waves never share a tile).
//
// Tile geometry (one wave):  128 records  = 3072 B AoS = 3 x (64 lanes x 16 B)
//                             barcode ASCII = 128*bc_len  B = 8*bc_len  16-B chunks
//                             UMI ASCII     = 128*umi_len B = 8*umi_len 16-B chunks
//                             index column  = 1024 B       = 64 chunks
// All tile bases are 16-B aligned whenever the array bases are.  Records beyond the last
// full tile (n % 128) and arrays whose base is not 16-B aligned go through the *_tail
// kernels (one thread per record) — still on the GPU; there is no CPU fallback anywhere.
//
// Reference semantics each kernel stands for are cited at its launcher in device.cpp and in
// include/ibu_hip.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace ibu {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

static constexpr int kWave = 64;
static constexpr int kBlock = 256;             // 4 waves, each with a private LDS slice
static constexpr int kWavesPerBlock = kBlock / kWave;
static constexpr int kTileRecs = 128;          // records per wave tile
static constexpr int kTileBytes = kTileRecs * 24;  // 3072
static constexpr u32 kPool = 0x54474341u;      // "ACGT" little-endian: byte k = base code k

// ---- wave-private LDS ordering ------------------------------------------------------------
// A wave's DS instructions execute in order, so a ds_read issued after a ds_write of the same
// wave observes it.  Only the compiler has to be told not to move LDS traffic across this.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- 2-bit <-> ASCII primitives -----------------------------------------------------------
// One code byte (4 bases, base i at bits [2i,2i+1]) -> 4 ASCII bytes.
__device__ __forceinline__ u32 expand4(u32 x) {
  u32 t = (x | (x << 12)) & 0x000F000Fu;   // nibbles to bytes 0 and 2
  t = (t | (t << 6)) & 0x03030303u;        // 2-bit fields to the low bits of bytes 0..3
  return __builtin_amdgcn_perm(kPool, kPool, t);  // selector bytes 0..3 pick A,C,G,T
}
// 4 ASCII bytes -> one code byte; *ok cleared if any byte is outside ACGTacgt.
__device__ __forceinline__ u32 pack4(u32 w, bool& ok) {
  u32 sel = ((w >> 1) ^ (w >> 2)) & 0x03030303u;  // A/a=0 C/c=1 G/g=2 T/t=3 per byte
  u32 expect = __builtin_amdgcn_perm(kPool, kPool, sel);
  ok = ok && ((w & 0xDFDFDFDFu) == expect);       // upper-cased input must be the letter decoded
  u32 y = sel | (sel >> 6);
  return (y | (y >> 12)) & 0xFFu;
}
__device__ __forceinline__ u32 pack1(u32 c, bool& ok) {
  u32 code = ((c >> 1) ^ (c >> 2)) & 3u;
  ok = ok && ((c & 0xDFu) == ((kPool >> (8 * code)) & 0xFFu));
  return code;
}
__device__ __forceinline__ u64 mask2(u32 len) { return len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1); }

// ---- global <-> register helpers (16 B per lane, coalesced) --------------------------------
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// =============================================================================================
// Output-centric expansion of one field of a staged tile.
//   tile     : wave-private LDS bytes holding 128 "records" of stride `rstride` bytes
//   foff     : byte offset of the u64 field inside a record
//   len      : bases per row (1..32)
//   out_tile : global address of row 0 of this tile (16-B aligned)
// Every lane produces whole 16-byte chunks of the ASCII stream, so every store is a full
// coalesced dwordx4 whatever len is.
// =============================================================================================
__device__ __forceinline__ void expand_field(const uint8_t* tile, u32 rstride, u32 foff, u32 len,
                                             uint8_t* out_tile, u32 lane) {
  const u32 nchunks = 8 * len;  // 128*len / 16
  if ((len & 3) == 0) {
    const u32 l4 = len >> 2;                   // code bytes per row (1..8)
    const u32 magic = 65536u / l4 + 1;         // d / l4 == (d*magic)>>16 for d < 1024
    for (u32 c = lane; c < nchunks; c += kWave) {
      u32x4 o;
      if (l4 == 4) {                           // 16 bases: the chunk is exactly row c
        u32 w = *reinterpret_cast<const u32*>(tile + c * rstride + foff);
        o.x = expand4(w & 0xFF); o.y = expand4((w >> 8) & 0xFF);
        o.z = expand4((w >> 16) & 0xFF); o.w = expand4(w >> 24);
      } else if (l4 == 8) {                    // 32 bases: half a row per chunk
        u32 w = *reinterpret_cast<const u32*>(tile + (c >> 1) * rstride + foff + (c & 1) * 4);
        o.x = expand4(w & 0xFF); o.y = expand4((w >> 8) & 0xFF);
        o.z = expand4((w >> 16) & 0xFF); o.w = expand4(w >> 24);
      } else {                                 // 4,8,12,20,24,28 bases: gather 4 code bytes
        u32 d = 4 * c;
        u32 r = (d * magic) >> 16;
        u32 q = d - r * l4;
        u32 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = expand4(tile[r * rstride + foff + q]);
          if (++q == l4) { q = 0; ++r; }
        }
        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
      }
      st16(out_tile + 16 * (size_t)c, o);
    }
  } else {
    // len not a multiple of 4: rows straddle dwords; resolve every output byte on its own.
    const u32 magic = (1u << 20) / len + 1;    // o / len == (o*magic)>>20 for o < 128*len
    for (u32 c = lane; c < nchunks; c += kWave) {
      u32 v[4];
      u32 o = 16 * c;
      u32 r = (o * magic) >> 20;
      u32 p = o - r * len;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32 w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          u32 code = (tile[r * rstride + foff + (p >> 2)] >> (2 * (p & 3))) & 3u;
          w |= ((kPool >> (8 * code)) & 0xFFu) << (8 * b);
          if (++p == len) { p = 0; ++r; }
        }
        v[j] = w;
      }
      u32x4 ov; ov.x = v[0]; ov.y = v[1]; ov.z = v[2]; ov.w = v[3];
      st16(out_tile + 16 * (size_t)c, ov);
    }
  }
}

// Record-centric packing of one row of `len` ASCII bytes staged at asc + row*len.
template <int L4>
__device__ __forceinline__ u64 pack_row_dwords(const uint8_t* row, bool& ok) {
  u64 v = 0;
#pragma unroll
  for (int q = 0; q < L4; ++q) {
    u32 w = *reinterpret_cast<const u32*>(row + 4 * q);
    v |= (u64)pack4(w, ok) << (8 * q);
  }
  return v;
}
__device__ __forceinline__ u64 pack_row(const uint8_t* row, u32 len, bool& ok) {
  if ((len & 3) == 0) {
    switch (len >> 2) {
      case 1: return pack_row_dwords<1>(row, ok);
      case 2: return pack_row_dwords<2>(row, ok);
      case 3: return pack_row_dwords<3>(row, ok);
      case 4: return pack_row_dwords<4>(row, ok);
      case 5: return pack_row_dwords<5>(row, ok);
      case 6: return pack_row_dwords<6>(row, ok);
      case 7: return pack_row_dwords<7>(row, ok);
      default: return pack_row_dwords<8>(row, ok);
    }
  }
  u64 v = 0;
  for (u32 i = 0; i < len; ++i) v |= (u64)pack1(row[i], ok) << (2 * i);
  return v;
}

// Stage `nchunks` 16-byte chunks global -> wave-private LDS, linear.
__device__ __forceinline__ void stage_linear(uint8_t* lds, const uint8_t* g, u32 nchunks, u32 lane) {
  for (u32 c = lane; c < nchunks; c += kWave)
    *reinterpret_cast<u32x4*>(lds + 16 * c) = ld16(g + 16 * (size_t)c);
}

// Report offending rows of this wave's tile: one atomicMin + one atomicAdd per wave, and only
// when something is actually wrong (wave-uniform branch on the ballot).
__device__ __forceinline__ void report_bad(bool bad, u64 row_global, u64* status, u32 lane) {
  u64 m = __ballot(bad);
  if (m) {
    if (bad) atomicMin(&status[0], row_global);
    if (lane == (u32)(__ffsll((long long)m) - 1)) atomicAdd(&status[1], (u64)__popcll(m));
  }
}

// =============================================================================================
// K2  fused decode: AoS records -> barcode ASCII, UMI ASCII, index column
// =============================================================================================
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_decode(const uint8_t* __restrict__ recs, u32 ntiles, u32 bc_len, u32 umi_len,
             uint8_t* __restrict__ bc_out, uint8_t* __restrict__ umi_out, u64* __restrict__ idx_out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kTileBytes;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  u32 t = blockIdx.x * kWavesPerBlock + wib;
  if (t >= ntiles) return;

  const uint8_t* src = recs + (size_t)t * kTileBytes + 16 * lane;
  u32x4 a0 = ld16(src), a1 = ld16(src + 1024), a2 = ld16(src + 2048);
  for (;;) {
    const u32 tn = t + nwaves;
    u32x4 b0, b1, b2;
    const bool more = tn < ntiles;             // wave-uniform
    if (more) {                                // next tile in flight while this one is expanded
      const uint8_t* s2 = recs + (size_t)tn * kTileBytes + 16 * lane;
      b0 = ld16(s2); b1 = ld16(s2 + 1024); b2 = ld16(s2 + 2048);
    }
    wave_lds_fence();                          // previous tile's LDS reads precede these writes
    *reinterpret_cast<u32x4*>(tile + 16 * lane) = a0;
    *reinterpret_cast<u32x4*>(tile + 1024 + 16 * lane) = a1;
    *reinterpret_cast<u32x4*>(tile + 2048 + 16 * lane) = a2;
    wave_lds_fence();
    if (bc_out) expand_field(tile, 24, 0, bc_len, bc_out + (size_t)t * kTileRecs * bc_len, lane);
    if (umi_out) expand_field(tile, 24, 8, umi_len, umi_out + (size_t)t * kTileRecs * umi_len, lane);
    if (idx_out) {                             // chunk = indices of records 2*lane, 2*lane+1
      u64 i0 = *reinterpret_cast<const u64*>(tile + (2 * lane) * 24 + 16);
      u64 i1 = *reinterpret_cast<const u64*>(tile + (2 * lane + 1) * 24 + 16);
      u32x4 o; o.x = (u32)i0; o.y = (u32)(i0 >> 32); o.z = (u32)i1; o.w = (u32)(i1 >> 32);
      st16(reinterpret_cast<uint8_t*>(idx_out) + (size_t)t * 1024 + 16 * lane, o);
    }
    if (!more) break;
    t = tn; a0 = b0; a1 = b1; a2 = b2;
  }
}

// =============================================================================================
// K3  fused encode: barcode ASCII, UMI ASCII (+ index column) -> AoS records
// Dynamic LDS per wave: max(3072, 128*(bc_len+umi_len)) bytes; the AoS tile reuses the ASCII
// staging area once every row has been packed (in-order DS makes that safe).
// =============================================================================================
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_encode(const uint8_t* __restrict__ bc_in, const uint8_t* __restrict__ umi_in,
             const u64* __restrict__ idx_in, u64 first_index, u32 ntiles, u32 bc_len, u32 umi_len,
             u32 wave_lds_bytes, uint8_t* __restrict__ recs, u64* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* area = dyn_lds + wib * wave_lds_bytes;
  uint8_t* asc_bc = area;
  uint8_t* asc_umi = area + kTileRecs * bc_len;
  const u32 nwaves = gridDim.x * kWavesPerBlock;

  for (u32 t = blockIdx.x * kWavesPerBlock + wib; t < ntiles; t += nwaves) {
    const size_t row0 = (size_t)t * kTileRecs;
    // index column: records lane and lane+64, 8 B per lane, coalesced
    u64 i0, i1;
    if (idx_in) { i0 = idx_in[row0 + lane]; i1 = idx_in[row0 + 64 + lane]; }
    else        { i0 = first_index + row0 + lane; i1 = i0 + 64; }
    wave_lds_fence();
    stage_linear(asc_bc, bc_in + row0 * bc_len, 8 * bc_len, lane);
    stage_linear(asc_umi, umi_in + row0 * umi_len, 8 * umi_len, lane);
    wave_lds_fence();
    bool okb0 = true, oku0 = true, okb1 = true, oku1 = true;
    u64 b0 = pack_row(asc_bc + lane * bc_len, bc_len, okb0);
    u64 u0 = pack_row(asc_umi + lane * umi_len, umi_len, oku0);
    u64 b1 = pack_row(asc_bc + (lane + 64) * bc_len, bc_len, okb1);
    u64 u1 = pack_row(asc_umi + (lane + 64) * umi_len, umi_len, oku1);
    if (!okb0) b0 = 0;
    if (!oku0) u0 = 0;
    if (!okb1) b1 = 0;
    if (!oku1) u1 = 0;
    report_bad(!(okb0 && oku0), row0 + lane, status, lane);
    report_bad(!(okb1 && oku1), row0 + 64 + lane, status, lane);
    wave_lds_fence();                          // all ASCII reads done before the area is reused
    u64* r0 = reinterpret_cast<u64*>(area + lane * 24);
    u64* r1 = reinterpret_cast<u64*>(area + (lane + 64) * 24);
    r0[0] = b0; r0[1] = u0; r0[2] = i0;
    r1[0] = b1; r1[1] = u1; r1[2] = i1;
    wave_lds_fence();
    uint8_t* dst = recs + (size_t)t * kTileBytes + 16 * lane;
    st16(dst, *reinterpret_cast<const u32x4*>(area + 16 * lane));
    st16(dst + 1024, *reinterpret_cast<const u32x4*>(area + 1024 + 16 * lane));
    st16(dst + 2048, *reinterpret_cast<const u32x4*>(area + 2048 + 16 * lane));
  }
}

// =============================================================================================
// K1  deserialise AoS -> three u64 columns          K1' serialise columns -> AoS
// =============================================================================================
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_deserialize(const uint8_t* __restrict__ recs, u32 ntiles, u64* __restrict__ bc,
                  u64* __restrict__ umi, u64* __restrict__ idx) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kTileBytes;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  u32 t = blockIdx.x * kWavesPerBlock + wib;
  if (t >= ntiles) return;
  const uint8_t* src = recs + (size_t)t * kTileBytes + 16 * lane;
  u32x4 a0 = ld16(src), a1 = ld16(src + 1024), a2 = ld16(src + 2048);
  for (;;) {
    const u32 tn = t + nwaves;
    u32x4 b0, b1, b2;
    const bool more = tn < ntiles;
    if (more) {
      const uint8_t* s2 = recs + (size_t)tn * kTileBytes + 16 * lane;
      b0 = ld16(s2); b1 = ld16(s2 + 1024); b2 = ld16(s2 + 2048);
    }
    wave_lds_fence();
    *reinterpret_cast<u32x4*>(tile + 16 * lane) = a0;
    *reinterpret_cast<u32x4*>(tile + 1024 + 16 * lane) = a1;
    *reinterpret_cast<u32x4*>(tile + 2048 + 16 * lane) = a2;
    wave_lds_fence();
    const u64* r0 = reinterpret_cast<const u64*>(tile + (2 * lane) * 24);
    const u64* r1 = reinterpret_cast<const u64*>(tile + (2 * lane + 1) * 24);
    u64* cols[3] = {bc, umi, idx};
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      u64 x = r0[f], y = r1[f];
      u32x4 o; o.x = (u32)x; o.y = (u32)(x >> 32); o.z = (u32)y; o.w = (u32)(y >> 32);
      st16(reinterpret_cast<uint8_t*>(cols[f]) + (size_t)t * 1024 + 16 * lane, o);
    }
    if (!more) break;
    t = tn; a0 = b0; a1 = b1; a2 = b2;
  }
}

extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_serialize(const u64* __restrict__ bc, const u64* __restrict__ umi, const u64* __restrict__ idx,
                u32 ntiles, uint8_t* __restrict__ recs) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kTileBytes;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  for (u32 t = blockIdx.x * kWavesPerBlock + wib; t < ntiles; t += nwaves) {
    const size_t off = (size_t)t * 1024 + 16 * lane;   // records 2*lane, 2*lane+1 of each column
    u32x4 c0 = ld16(reinterpret_cast<const uint8_t*>(bc) + off);
    u32x4 c1 = ld16(reinterpret_cast<const uint8_t*>(umi) + off);
    u32x4 c2 = ld16(reinterpret_cast<const uint8_t*>(idx) + off);
    wave_lds_fence();
    u32x2* r0 = reinterpret_cast<u32x2*>(tile + (2 * lane) * 24);
    u32x2* r1 = reinterpret_cast<u32x2*>(tile + (2 * lane + 1) * 24);
    r0[0] = c0.xy; r1[0] = c0.zw;
    r0[1] = c1.xy; r1[1] = c1.zw;
    r0[2] = c2.xy; r1[2] = c2.zw;
    wave_lds_fence();
    uint8_t* dst = recs + (size_t)t * kTileBytes + 16 * lane;
    st16(dst, *reinterpret_cast<const u32x4*>(tile + 16 * lane));
    st16(dst + 1024, *reinterpret_cast<const u32x4*>(tile + 1024 + 16 * lane));
    st16(dst + 2048, *reinterpret_cast<const u32x4*>(tile + 2048 + 16 * lane));
  }
}

// =============================================================================================
// Single-column 2-bit unpack / pack (stride-8 "records")
// =============================================================================================
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_unpack(const u64* __restrict__ codes, u32 ntiles, u32 len, uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * 1024];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * 1024;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  for (u32 t = blockIdx.x * kWavesPerBlock + wib; t < ntiles; t += nwaves) {
    u32x4 a = ld16(reinterpret_cast<const uint8_t*>(codes) + (size_t)t * 1024 + 16 * lane);
    wave_lds_fence();
    *reinterpret_cast<u32x4*>(tile + 16 * lane) = a;
    wave_lds_fence();
    expand_field(tile, 8, 0, len, out + (size_t)t * kTileRecs * len, lane);
  }
}

extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_pack(const uint8_t* __restrict__ in, u32 ntiles, u32 len, u64* __restrict__ codes,
           u64* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kTileRecs * 32];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* asc = lds + wib * kTileRecs * 32;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  for (u32 t = blockIdx.x * kWavesPerBlock + wib; t < ntiles; t += nwaves) {
    const size_t row0 = (size_t)t * kTileRecs;
    wave_lds_fence();
    stage_linear(asc, in + row0 * len, 8 * len, lane);
    wave_lds_fence();
    bool ok0 = true, ok1 = true;
    u64 v0 = pack_row(asc + lane * len, len, ok0);
    u64 v1 = pack_row(asc + (lane + 64) * len, len, ok1);
    if (!ok0) v0 = 0;
    if (!ok1) v1 = 0;
    report_bad(!ok0, row0 + lane, status, lane);
    report_bad(!ok1, row0 + 64 + lane, status, lane);
    codes[row0 + lane] = v0;
    codes[row0 + 64 + lane] = v1;
  }
}

// =============================================================================================
// K4  reduce: wrapping sums and XORs of the three fields.  Pure streaming read, no LDS in the
// loop: a lane's dwordx4 always lands on the same two field slots because the wave stride
// (3072 B = 384 u64) is a multiple of 3.
// =============================================================================================
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
  u32 lo = __shfl_xor((u32)v, m), hi = __shfl_xor((u32)(v >> 32), m);
  return ((u64)hi << 32) | lo;
}

extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_reduce(const uint8_t* __restrict__ recs, u32 ntiles, u64 n_total, u64* __restrict__ acc) {
  __shared__ u64 part[kWavesPerBlock][6];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  u64 s[3][2] = {{0, 0}, {0, 0}, {0, 0}}, x[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  u32 t = blockIdx.x * kWavesPerBlock + wib;
  for (; t + nwaves < ntiles; t += 2 * nwaves) {  // two tiles (6 KiB per wave) in flight
    const uint8_t* p = recs + (size_t)t * kTileBytes + 16 * lane;
    const uint8_t* q = recs + (size_t)(t + nwaves) * kTileBytes + 16 * lane;
    u32x4 a[3], b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { a[k] = ld16(p + 1024 * k); b[k] = ld16(q + 1024 * k); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      u64 a0 = ((u64)a[k].y << 32) | a[k].x, a1 = ((u64)a[k].w << 32) | a[k].z;
      u64 b0 = ((u64)b[k].y << 32) | b[k].x, b1 = ((u64)b[k].w << 32) | b[k].z;
      s[k][0] += a0 + b0; s[k][1] += a1 + b1;
      x[k][0] ^= a0 ^ b0; x[k][1] ^= a1 ^ b1;
    }
  }
  if (t < ntiles) {
    const uint8_t* p = recs + (size_t)t * kTileBytes + 16 * lane;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      u32x4 a = ld16(p + 1024 * k);
      u64 a0 = ((u64)a.y << 32) | a.x, a1 = ((u64)a.w << 32) | a.z;
      s[k][0] += a0; s[k][1] += a1;
      x[k][0] ^= a0; x[k][1] ^= a1;
    }
  }
  // slot (k,h) of this lane is flat u64 element 2*(64k+lane)+h of the tile -> field e % 3
  u64 S[3] = {0, 0, 0}, X[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32 f = (2 * (64 * k + lane) + h) % 3;
#pragma unroll
      for (int g = 0; g < 3; ++g)
        if (f == (u32)g) { S[g] += s[k][h]; X[g] ^= x[k][h]; }
    }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int g = 0; g < 3; ++g) { S[g] += shfl_xor_u64(S[g], m); X[g] ^= shfl_xor_u64(X[g], m); }
  if (lane == 0)
#pragma unroll
    for (int g = 0; g < 3; ++g) { part[wib][g] = S[g]; part[wib][3 + g] = X[g]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    u64 v = part[0][threadIdx.x];
    for (int w = 1; w < kWavesPerBlock; ++w)
      v = threadIdx.x < 3 ? v + part[w][threadIdx.x] : v ^ part[w][threadIdx.x];
    if (threadIdx.x < 3) { if (v) atomicAdd(&acc[1 + threadIdx.x], v); }
    else                 { if (v) atomicXor(&acc[1 + threadIdx.x], v); }
  }
  if (blockIdx.x == 0 && threadIdx.x == 6) atomicAdd(&acc[0], n_total);
}

// =============================================================================================
// Synthetic records: flat u64 element e = 3*i + k of the record stream is splitmix64(seed + e)
// masked for k = 0,1 and i for k = 2.  One 16-B chunk (two elements) per thread, coalesced.
// =============================================================================================
__device__ __forceinline__ u64 splitmix64(u64 z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ u64 synth_elem(u64 seed, u64 first, u64 e, u64 mb, u64 mu) {
  const u64 i = e / 3;
  const u32 k = (u32)(e - 3 * i);
  const u64 gi = first + i;
  if (k == 2) return gi;
  const u64 r = splitmix64(seed + 3 * gi + k);
  return k == 0 ? (r & mb) : (r & mu);
}
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_generate(u64 seed, u64 first, u64 n_elems, u32 bc_len, u32 umi_len, u64* __restrict__ out) {
  const u64 mb = mask2(bc_len), mu = mask2(umi_len);
  const u64 stride = (u64)gridDim.x * kBlock;
  const u64 npairs = n_elems >> 1;  // n_elems = 3n; pairs cover the 16-B chunks
  for (u64 c = (u64)blockIdx.x * kBlock + threadIdx.x; c < npairs; c += stride) {
    u64 v0 = synth_elem(seed, first, 2 * c, mb, mu), v1 = synth_elem(seed, first, 2 * c + 1, mb, mu);
    u32x4 o; o.x = (u32)v0; o.y = (u32)(v0 >> 32); o.z = (u32)v1; o.w = (u32)(v1 >> 32);
    st16(reinterpret_cast<uint8_t*>(out) + 16 * c, o);
  }
  if ((n_elems & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    out[n_elems - 1] = synth_elem(seed, first, n_elems - 1, mb, mu);
}

// =============================================================================================
// Tail / unaligned kernels: one thread per record, no alignment assumption beyond the natural
// 8 B of the u64 columns and records.  Used for n % 128 and for misaligned bases only.
// =============================================================================================
__device__ __forceinline__ void unpack_row_bytes(u64 code, u32 len, uint8_t* out) {
  for (u32 i = 0; i < len; ++i) out[i] = (uint8_t)((kPool >> (8 * ((code >> (2 * i)) & 3))) & 0xFF);
}
__device__ __forceinline__ u64 pack_row_bytes(const uint8_t* in, u32 len, bool& ok) {
  u64 v = 0;
  for (u32 i = 0; i < len; ++i) v |= (u64)pack1(in[i], ok) << (2 * i);
  return v;
}
extern "C" __global__ void ibu_k_decode_tail(const u64* __restrict__ recs, u64 row0, u64 n, u32 bc_len,
                                             u32 umi_len, uint8_t* bc_out, uint8_t* umi_out, u64* idx_out) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (bc_out) unpack_row_bytes(recs[3 * i], bc_len, bc_out + i * bc_len);
  if (umi_out) unpack_row_bytes(recs[3 * i + 1], umi_len, umi_out + i * umi_len);
  if (idx_out) idx_out[i] = recs[3 * i + 2];
}
extern "C" __global__ void ibu_k_encode_tail(const uint8_t* bc_in, const uint8_t* umi_in, const u64* idx_in,
                                             u64 first_index, u64 row0, u64 n, u32 bc_len, u32 umi_len,
                                             u64* __restrict__ recs, u64* status) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool okb = true, oku = true;
  u64 b = pack_row_bytes(bc_in + i * bc_len, bc_len, okb);
  u64 u = pack_row_bytes(umi_in + i * umi_len, umi_len, oku);
  if (!okb) b = 0;
  if (!oku) u = 0;
  if (!(okb && oku)) { atomicMin(&status[0], i); atomicAdd(&status[1], 1ull); }
  recs[3 * i] = b; recs[3 * i + 1] = u; recs[3 * i + 2] = idx_in ? idx_in[i] : first_index + i;
}
extern "C" __global__ void ibu_k_deserialize_tail(const u64* __restrict__ recs, u64 row0, u64 n, u64* bc,
                                                  u64* umi, u64* idx) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bc[i] = recs[3 * i]; umi[i] = recs[3 * i + 1]; idx[i] = recs[3 * i + 2];
}
extern "C" __global__ void ibu_k_serialize_tail(const u64* bc, const u64* umi, const u64* idx, u64 row0, u64 n,
                                                u64* __restrict__ recs) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  recs[3 * i] = bc[i]; recs[3 * i + 1] = umi[i]; recs[3 * i + 2] = idx[i];
}
extern "C" __global__ void ibu_k_unpack_tail(const u64* codes, u64 row0, u64 n, u32 len, uint8_t* out) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unpack_row_bytes(codes[i], len, out + i * len);
}
extern "C" __global__ void ibu_k_pack_tail(const uint8_t* in, u64 row0, u64 n, u32 len, u64* codes, u64* status) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = true;
  u64 v = pack_row_bytes(in + i * len, len, ok);
  if (!ok) { v = 0; atomicMin(&status[0], i); atomicAdd(&status[1], 1ull); }
  codes[i] = v;
}
extern "C" __global__ void ibu_k_reduce_tail(const u64* __restrict__ recs, u64 row0, u64 n, u64* acc) {
  // at most a few hundred records: one block, one thread per record, atomics straight to acc
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int f = 0; f < 3; ++f) {
    u64 v = recs[3 * i + f];
    if (v) { atomicAdd(&acc[1 + f], v); atomicXor(&acc[4 + f], v); }
  }
}
extern "C" __global__ void ibu_k_sorted_check(const u64* __restrict__ recs, u64 n, u32* unsorted) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += stride) {
    const u64* a = recs + 3 * i;
    bool gt = a[0] != a[3] ? a[0] > a[3] : (a[1] != a[4] ? a[1] > a[4] : a[2] > a[5]);
    if (gt) atomicOr(unsorted, 1u);
  }
}
extern "C" __global__ void ibu_k_fill_u64(u64* p, u64 v0, u64 v1) { p[0] = v0; p[1] = v1; }

// =============================================================================================
// Launchers
// =============================================================================================
static inline u32 grid_for(u32 ntiles, int cus, int blocks_per_cu) {
  u32 need = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
  u32 cap = (u32)(cus * blocks_per_cu);
  return need < cap ? (need ? need : 1) : cap;
}
// Persistent grids must be exactly resident: a workgroup that has to wait for a slot runs its
// whole share alone at the end.  Ask the runtime how many 256-thread blocks of this kernel fit
// on a CU (registers, LDS), cap it by cfg.blocks_per_cu, remember the answer per kernel.
template <class K>
static inline int resident_blocks(const LaunchCfg& cfg, K kernel, size_t dyn_lds, int* cache) {
  if (*cache <= 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, dyn_lds) != hipSuccess || nb <= 0) nb = 4;
    *cache = nb;
  }
  return *cache < cfg.blocks_per_cu ? *cache : cfg.blocks_per_cu;
}
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline u32 tail_grid(u64 rows) { return (u32)((rows + 255) / 256); }

hipError_t launch_decode(const LaunchCfg& cfg, const void* recs, size_t n, uint32_t bc_len, uint32_t umi_len,
                         uint8_t* bc, uint8_t* umi, uint64_t* idx, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const bool fast = aligned16(recs) && aligned16(bc) && aligned16(umi) && aligned16(idx);
  const size_t n_main = fast ? (n / kTileRecs) * kTileRecs : 0;
  if (n_main) {
    u32 ntiles = (u32)(n_main / kTileRecs);
    static int occ = 0;
    hipLaunchKernelGGL(ibu_k_decode, dim3(grid_for(ntiles, cfg.cus, resident_blocks(cfg, ibu_k_decode, 0, &occ))),
                       dim3(kBlock), 0, st,
                       (const uint8_t*)recs, ntiles, bc_len, umi_len, bc, umi, (u64*)idx);
  }
  if (n_main < n)
    hipLaunchKernelGGL(ibu_k_decode_tail, dim3(tail_grid(n - n_main)), dim3(256), 0, st, (const u64*)recs,
                       (u64)n_main, (u64)n, bc_len, umi_len, bc, umi, (u64*)idx);
  return hipGetLastError();
}

hipError_t launch_encode(const LaunchCfg& cfg, const uint8_t* bc, const uint8_t* umi, const uint64_t* idx,
                         uint64_t first_index, size_t n, uint32_t bc_len, uint32_t umi_len, void* recs,
                         uint64_t* status, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const bool fast = aligned16(recs) && aligned16(bc) && aligned16(umi);
  const size_t n_main = fast ? (n / kTileRecs) * kTileRecs : 0;
  if (n_main) {
    u32 ntiles = (u32)(n_main / kTileRecs);
    u32 wave_lds = kTileRecs * (bc_len + umi_len);
    if (wave_lds < (u32)kTileBytes) wave_lds = kTileBytes;
    static int occ[65] = {0};
    const int nb = resident_blocks(cfg, ibu_k_encode, wave_lds * kWavesPerBlock, &occ[bc_len + umi_len]);
    hipLaunchKernelGGL(ibu_k_encode, dim3(grid_for(ntiles, cfg.cus, nb)), dim3(kBlock),
                       wave_lds * kWavesPerBlock, st, bc, umi, (const u64*)idx, (u64)first_index, ntiles, bc_len,
                       umi_len, wave_lds, (uint8_t*)recs, (u64*)status);
  }
  if (n_main < n)
    hipLaunchKernelGGL(ibu_k_encode_tail, dim3(tail_grid(n - n_main)), dim3(256), 0, st, bc, umi, (const u64*)idx,
                       (u64)first_index, (u64)n_main, (u64)n, bc_len, umi_len, (u64*)recs, (u64*)status);
  return hipGetLastError();
}

hipError_t launch_deserialize(const LaunchCfg& cfg, const void* recs, size_t n, uint64_t* bc, uint64_t* umi,
                              uint64_t* idx, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const bool fast = aligned16(recs) && aligned16(bc) && aligned16(umi) && aligned16(idx);
  const size_t n_main = fast ? (n / kTileRecs) * kTileRecs : 0;
  if (n_main) {
    u32 ntiles = (u32)(n_main / kTileRecs);
    static int occ = 0;
    hipLaunchKernelGGL(ibu_k_deserialize,
                       dim3(grid_for(ntiles, cfg.cus, resident_blocks(cfg, ibu_k_deserialize, 0, &occ))), dim3(kBlock), 0, st,
                       (const uint8_t*)recs, ntiles, (u64*)bc, (u64*)umi, (u64*)idx);
  }
  if (n_main < n)
    hipLaunchKernelGGL(ibu_k_deserialize_tail, dim3(tail_grid(n - n_main)), dim3(256), 0, st, (const u64*)recs,
                       (u64)n_main, (u64)n, (u64*)bc, (u64*)umi, (u64*)idx);
  return hipGetLastError();
}

hipError_t launch_serialize(const LaunchCfg& cfg, const uint64_t* bc, const uint64_t* umi, const uint64_t* idx,
                            size_t n, void* recs, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const bool fast = aligned16(recs) && aligned16(bc) && aligned16(umi) && aligned16(idx);
  const size_t n_main = fast ? (n / kTileRecs) * kTileRecs : 0;
  if (n_main) {
    u32 ntiles = (u32)(n_main / kTileRecs);
    static int occ = 0;
    hipLaunchKernelGGL(ibu_k_serialize, dim3(grid_for(ntiles, cfg.cus, resident_blocks(cfg, ibu_k_serialize, 0, &occ))),
                       dim3(kBlock), 0, st,
                       (const u64*)bc, (const u64*)umi, (const u64*)idx, ntiles, (uint8_t*)recs);
  }
  if (n_main < n)
    hipLaunchKernelGGL(ibu_k_serialize_tail, dim3(tail_grid(n - n_main)), dim3(256), 0, st, (const u64*)bc,
                       (const u64*)umi, (const u64*)idx, (u64)n_main, (u64)n, (u64*)recs);
  return hipGetLastError();
}

hipError_t launch_unpack(const LaunchCfg& cfg, const uint64_t* codes, size_t n, uint32_t len, uint8_t* out,
                         hipStream_t st) {
  if (n == 0) return hipSuccess;
  const bool fast = aligned16(codes) && aligned16(out);
  const size_t n_main = fast ? (n / kTileRecs) * kTileRecs : 0;
  if (n_main) {
    u32 ntiles = (u32)(n_main / kTileRecs);
    static int occ = 0;
    hipLaunchKernelGGL(ibu_k_unpack, dim3(grid_for(ntiles, cfg.cus, resident_blocks(cfg, ibu_k_unpack, 0, &occ))),
                       dim3(kBlock), 0, st,
                       (const u64*)codes, ntiles, len, out);
  }
  if (n_main < n)
    hipLaunchKernelGGL(ibu_k_unpack_tail, dim3(tail_grid(n - n_main)), dim3(256), 0, st, (const u64*)codes,
                       (u64)n_main, (u64)n, len, out);
  return hipGetLastError();
}

hipError_t launch_pack(const LaunchCfg& cfg, const uint8_t* in, size_t n, uint32_t len, uint64_t* codes,
                       uint64_t* status, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const bool fast = aligned16(in);
  const size_t n_main = fast ? (n / kTileRecs) * kTileRecs : 0;
  if (n_main) {
    u32 ntiles = (u32)(n_main / kTileRecs);
    static int occ = 0;
    hipLaunchKernelGGL(ibu_k_pack, dim3(grid_for(ntiles, cfg.cus, resident_blocks(cfg, ibu_k_pack, 0, &occ))),
                       dim3(kBlock), 0, st, in, ntiles, len,
                       (u64*)codes, (u64*)status);
  }
  if (n_main < n)
    hipLaunchKernelGGL(ibu_k_pack_tail, dim3(tail_grid(n - n_main)), dim3(256), 0, st, in, (u64)n_main, (u64)n, len,
                       (u64*)codes, (u64*)status);
  return hipGetLastError();
}

hipError_t launch_reduce(const LaunchCfg& cfg, const void* recs, size_t n, uint64_t* acc, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const bool fast = aligned16(recs);
  const size_t n_main = fast ? (n / kTileRecs) * kTileRecs : 0;
  u32 ntiles = (u32)(n_main / kTileRecs);
  // the main kernel also adds n to the count slot, so it always runs (ntiles may be 0)
  static int occ = 0;
  hipLaunchKernelGGL(ibu_k_reduce, dim3(grid_for(ntiles, cfg.cus, resident_blocks(cfg, ibu_k_reduce, 0, &occ))),
                     dim3(kBlock), 0, st,
                     (const uint8_t*)recs, ntiles, (u64)n, (u64*)acc);
  if (n_main < n)
    hipLaunchKernelGGL(ibu_k_reduce_tail, dim3(tail_grid(n - n_main)), dim3(256), 0, st, (const u64*)recs,
                       (u64)n_main, (u64)n, (u64*)acc);
  return hipGetLastError();
}

hipError_t launch_generate(const LaunchCfg& cfg, uint64_t seed, uint64_t first, size_t n, uint32_t bc_len,
                           uint32_t umi_len, void* recs, hipStream_t st) {
  if (n == 0) return hipSuccess;
  u64 n_elems = 3ull * n;
  u64 blocks = ((n_elems >> 1) + kBlock - 1) / kBlock;
  u64 cap = (u64)cfg.cus * 16;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(ibu_k_generate, dim3((u32)blocks), dim3(kBlock), 0, st, (u64)seed, (u64)first, n_elems, bc_len,
                     umi_len, (u64*)recs);
  return hipGetLastError();
}

hipError_t launch_sorted_check(const LaunchCfg& cfg, const void* recs, size_t n, uint32_t* flag, hipStream_t st) {
  if (n < 2) return hipSuccess;
  u64 blocks = (n + 255) / 256;
  u64 cap = (u64)cfg.cus * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(ibu_k_sorted_check, dim3((u32)blocks), dim3(256), 0, st, (const u64*)recs, (u64)n, flag);
  return hipGetLastError();
}

hipError_t launch_fill2(uint64_t* p, uint64_t v0, uint64_t v1, hipStream_t st) {
  hipLaunchKernelGGL(ibu_k_fill_u64, dim3(1), dim3(1), 0, st, (u64*)p, (u64)v0, (u64)v1);
  return hipGetLastError();
}

}  // namespace ibu
