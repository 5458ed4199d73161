// kcommon.hpp — device-side building blocks shared by the gfx950 kernels (k_*.hip).
//
// Everything on this path is HBM-bound integer / byte work: no MFMA.  The rules are the
// streaming ones:
//   * every global access is a fully coalesced 16 B-per-lane wave instruction (1 KiB each),
//     with the nontemporal hint (each byte is touched once; measured in-process: reduce
//     6.0 -> 6.7 TB/s, decode 5.7 -> 6.0 TB/s — profiles/r01_c);
//   * the 24-byte record stride is absorbed in LDS: a wave-private 128-record = 3 KiB tile;
//     stride-24 ds_read_b64 is bank-conflict-free (6 dwords * k mod 64 distinct for 32 k);
//   * waves never share a tile, so there is no workgroup barrier anywhere;
//   * persistent grid, grid-stride over tiles, next tile's loads in flight while the current
//     one is processed.  The prefetch loads are UNCONDITIONAL (the last iteration re-reads its
//     own tile) and loop trip counts around loads/stores are compile-time constants: hipcc's
//     waitcnt pass tracks the in-order vmcnt counter per path and merges paths conservatively,
//     so a branch around a load, or a store in a runtime-count loop, turns the wait before the
//     LDS write into vmcnt(0) and drains the prefetch every iteration (seen in the ISA of the
//     first version of these kernels).
//
// Tile geometry (one wave):  128 records  = 3072 B AoS = 3 x (64 lanes x 16 B)
//                             ASCII column  = 128*len B  = 8*len 16-B chunks
//                             u64 column    = 1024 B     = 64 chunks
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"

namespace ibu {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

static constexpr int kWave = 64;
#ifndef IBU_BLOCK
#define IBU_BLOCK 256
#endif
static constexpr int kBlock = IBU_BLOCK;           // waves of a workgroup each own a private LDS slice
static constexpr int kWavesPerBlock = kBlock / kWave;
static constexpr int kTileRecs = 128;              // records per wave tile
static constexpr int kTileBytes = kTileRecs * 24;  // 3072
static constexpr u32 kPool = 0x54474341u;          // "ACGT" little-endian: byte k = base code k

// Lengths with a fully specialised (constant-folded, straight-line) kernel; every other length 1..32 runs the runtime-length
// kernels (mode 0: the field goes through its code stream, below).  Length 10 had a specialisation until round 4 — a byte path
// (10 % 4 != 0) under a two- or three-wave register budget; through the code stream the same arrays run (10,10) encode 2.5 %,
// pack 10 % faster and decode level (profiles/r04_i_kbench_len10.jsonl), so it went.
static constexpr int kNumLenModes = 5;
__host__ __device__ constexpr int len_of_mode(int m) { return m == 1 ? 8 : m == 2 ? 12 : m == 3 ? 16 : m == 4 ? 32 : 0; }
static inline int mode_of_len(uint32_t len) {
  switch (len) {
    case 8: return 1;
    case 12: return 2;
    case 16: return 3;
    case 32: return 4;
    default: return 0;
  }
}

// XCD-aware block order.  Hardware places workgroup i on XCD i % 8 (8 XCDs, each with its own
// L2 and TLB hierarchy).  With the identity order every XCD's workgroups are scattered over the
// whole moving front of the grid-stride sweep; remapped, each XCD sweeps ONE contiguous eighth of
// the front in each of the arrays, so its L2 / translation working set is 8x more compact.
#ifndef IBU_XCD_REMAP
#define IBU_XCD_REMAP 1
#endif
__device__ __forceinline__ u32 logical_block() {
#if IBU_XCD_REMAP
  const u32 nb = gridDim.x;
  if (nb & 7u) return blockIdx.x;                  // small grids: identity
  return (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
#else
  return blockIdx.x;
#endif
}

// Which tiles a wave sweeps.  IBU_XCD_STATIC 0: wave w of the (XCD-remapped) grid takes tiles w, w + nwaves, ... — every XCD
// owns an eighth of the MOVING FRONT.  1: every XCD owns one fixed eighth of the arrays (tiles [x N/8, (x+1) N/8)) and
// its workgroups sweep that range: the eight fronts sit an eighth of each array apart instead of side by side.
#ifndef IBU_XCD_STATIC
#define IBU_XCD_STATIC 1
#endif
struct TileRange { u32 t, stride, end; };
// IBU_XCD_SUB (measurement builds): every XCD's eighth is cut into this many sub-ranges, each swept by its own share of the
// XCD's workgroups (8 x SUB fronts instead of 8).
#ifndef IBU_XCD_SUB
#define IBU_XCD_SUB 1
#endif
__device__ __forceinline__ TileRange tile_range(u32 ntiles, u32 wib) {
#if IBU_XCD_STATIC
  if ((gridDim.x & 7u) == 0 && gridDim.x >= 8u * IBU_XCD_SUB) {
    const u32 parts = 8u * IBU_XCD_SUB;
    const u32 tpp = (ntiles + parts - 1) / parts;                 // tiles per part
    const u32 xcd = blockIdx.x & 7u, lb = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const u32 sub = lb % IBU_XCD_SUB, part = xcd * IBU_XCD_SUB + sub;
    const u32 nb_sub = (nbx - sub + IBU_XCD_SUB - 1) / IBU_XCD_SUB;   // workgroups of this XCD on this sub-range
    const u32 t0 = part * tpp, t1 = t0 + tpp < ntiles ? t0 + tpp : ntiles;
    return {t0 + (lb / IBU_XCD_SUB) * (u32)kWavesPerBlock + wib, nb_sub * (u32)kWavesPerBlock, t0 < ntiles ? t1 : 0u};
  }
#endif
  return {logical_block() * (u32)kWavesPerBlock + wib, gridDim.x * (u32)kWavesPerBlock, ntiles};
}

// The sweep of a wave over its tiles with the next tile's loads always in flight: two register sets take turns (phase A
// works on `a` while `b` loads, phase B the reverse), so a set is never COPIED — `a = b` at the end of an iteration makes
// the compiler wait for b's loads right there (v_mov needs the data) and the prefetch is drained once per tile (seen in
// the ISA of round 2's census / compress / expand loops: s_waitcnt vmcnt(..) in front of the copies).
// load(regs, tile): issue the tile's loads, unconditionally (the last iteration re-loads its own tile).  body(regs, tile).
template <class Regs, class Load, class Body>
__device__ __forceinline__ void sweep_tiles(const TileRange tr, Load load, Body body) {
  u32 t = tr.t;
  if (t >= tr.end) return;                         // wave-uniform
  Regs a, b;
  load(a, t);
  for (;;) {
    u32 tn = t + tr.stride;
    bool more = tn < tr.end;                       // wave-uniform
    load(b, more ? tn : t);
    body(a, t);
    if (!more) break;
    t = tn;
    tn = t + tr.stride;
    more = tn < tr.end;
    load(a, more ? tn : t);
    body(b, t);
    if (!more) break;
    t = tn;
  }
}

#ifndef IBU_NT_LOAD
#define IBU_NT_LOAD 1
#endif
#ifndef IBU_NT_STORE
#define IBU_NT_STORE 1
#endif
// IBU_ST_POLICY / IBU_LD_POLICY (measurement builds, tools/kbench.py --so): the cache-policy bits of the streaming
// accesses spelled out, through raw buffer instructions so that the compiler still counts them in vmcnt (inline-asm
// stores are invisible to its waitcnt pass, which then waits for the stores as well).  Value = aux bits + 1:
// bit0 sc0, bit1 nt, bit4 sc1 — e.g. 3 = "nt" (the default policy through the buffer path), 17 = "sc1", 19 = "sc1 nt",
// 18 = "sc0 sc1", 20 = "sc0 sc1 nt", 1 = no bits.  0 = the builtin global path below (nt or plain).
#ifndef IBU_ST_POLICY
#define IBU_ST_POLICY 0
#endif
#ifndef IBU_LD_POLICY
#define IBU_LD_POLICY 0
#endif
struct WaveBuf {  // a 4 GiB window below/above the first active lane's address, as a buffer resource in SGPRs
  __amdgpu_buffer_rsrc_t r;
  u32 voff;
  __device__ __forceinline__ explicit WaveBuf(const void* p) {
    const u64 a = (u64)p;
    // readfirstlane returns int: without the casts the low half is SIGN-extended into the high half
    const u64 first = ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(a >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((u32)a);
    const u64 base = first - (1ull << 30);
    r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, -1, 0x00020000);
    voff = (u32)(a - base);
  }
};
__device__ __forceinline__ u32x4 ld16(const void* p) {
#if IBU_LD_POLICY
  const WaveBuf w(p);
  return __builtin_amdgcn_raw_buffer_load_b128(w.r, (int)w.voff, 0, IBU_LD_POLICY - 1);
#elif IBU_NT_LOAD
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#else
  return *reinterpret_cast<const u32x4*>(p);
#endif
}
__device__ __forceinline__ void st16(void* p, u32x4 v) {
#if IBU_ST_POLICY
  const WaveBuf w(p);
  __builtin_amdgcn_raw_buffer_store_b128(v, w.r, (int)w.voff, 0, IBU_ST_POLICY - 1);
#elif IBU_NT_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
#else
  *reinterpret_cast<u32x4*>(p) = v;
#endif
}

// A wave's DS instructions execute in order, so a ds_read issued after a ds_write of the same
// wave observes it.  Only the compiler has to be told not to move LDS traffic across this.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- bit order of the 2-bit codec ------------------------------------------------------------
// Default (order 0, "LSB first"): base i of a sequence sits at bits [2i, 2i+1] of the code word, "ACGT" -> 0b11100100
// (bitnuc's convention as recalled; nothing in the reference pins it: DESIGN.md §3).  The hedge (order 1, "MSB
// first"): base i at bits [2(len-1-i), 2(len-1-i)+1], the sequence read as a base-4 number, "ACGT" -> 0b00011011.
// The two code words of one sequence are each other's image under rev_pairs (the order of the low `len` 2-bit groups
// reversed, everything at and above bit 2*len dropped), so the MSB-first kernels are the LSB-first kernels with one
// rev_pairs per code word: on the staged tile before expansion (decode), on the staged AoS tile before it is stored
// (encode).  A TEMPLATE parameter of the tiled kernels (`MSB`): the default-order instantiations are instruction for
// instruction the kernels measured in profiles/ (as a kernel argument the branch cost decode<12,12> and encode<16,12>
// their last free VGPRs: both started to spill); the tail kernels take it as an argument.
__device__ __forceinline__ u64 rev_pairs(u64 x, u32 len) {   // len in 1..32
  u64 r = __builtin_bitreverse64(x);                            // v_bfrev_b32 x2
  r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);  // bit-reversal also swapped the two bits of a group
  return r >> (64 - 2 * len);
}
__device__ __forceinline__ u32x4 rev_pairs_x2(u32x4 v, u32 len) {   // two code words held as one 16-B chunk
  const u64 a = rev_pairs(((u64)v.y << 32) | v.x, len), b = rev_pairs(((u64)v.w << 32) | v.z, len);
  u32x4 o; o.x = (u32)a; o.y = (u32)(a >> 32); o.z = (u32)b; o.w = (u32)(b >> 32);
  return o;
}
// Rewrite the code words of a staged tile in place: `words` u64 fields at byte `foff` of records of stride `rstride`,
// lane L owns records 2L and 2L+1 of every 128-record tile staged in `tile`.
template <int NT = 1>
__device__ __forceinline__ void rev_pairs_tile(uint8_t* tile, u32 rstride, u32 foff, u32 len, u32 lane) {
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u64* p = reinterpret_cast<u64*>(tile + (size_t)(128 * j + 2 * lane + h) * rstride + foff);
      *p = rev_pairs(*p, len);
    }
}

// ---- 2-bit <-> ASCII primitives -----------------------------------------------------------
// One code byte (4 bases, base i at bits [2i,2i+1]) -> 4 ASCII bytes.
__device__ __forceinline__ u32 expand4(u32 x) {
  u32 t = (x | (x << 12)) & 0x000F000Fu;          // nibbles to bytes 0 and 2
  t = (t | (t << 6)) & 0x03030303u;               // 2-bit fields to the low bits of bytes 0..3
  return __builtin_amdgcn_perm(kPool, kPool, t);  // selector bytes 0..3 pick A,C,G,T
}
// 4 ASCII bytes -> one code byte; ok cleared if any byte is outside ACGTacgt.
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
__device__ __forceinline__ u32x4 expand16(u32 w) {  // 16 bases
  u32x4 o;
  o.x = expand4(w & 0xFF); o.y = expand4((w >> 8) & 0xFF);
  o.z = expand4((w >> 16) & 0xFF); o.w = expand4(w >> 24);
  return o;
}

// One 16-byte chunk `c` of the ASCII stream of a staged tile (rows of `len` bases, u64 field at
// byte `foff` of records of stride `rstride`).  `len` may be a compile-time constant.
// IBU_PROBE (measurement builds only, results are WRONG): 1 = keep the LDS traffic, drop the expansion ALU;
// 2 = no LDS reads either.  Used with tools/kbench.py --so to tell an HBM bound from an issue/LDS bound.
#ifndef IBU_PROBE
#define IBU_PROBE 0
#endif
__device__ __forceinline__ u32x4 expand_chunk(const uint8_t* tile, u32 rstride, u32 foff, u32 len, u32 c) {
#if IBU_PROBE == 1
  { const u32 w = *reinterpret_cast<const u32*>(tile + (c & 127u) * rstride + foff); u32x4 o; o.x = o.y = o.z = o.w = w; return o; }
#elif IBU_PROBE == 2
  { u32x4 o; o.x = o.y = o.z = o.w = c + len + rstride + foff; return o; }
#endif
  // len is a multiple of 4 (expand_field: every specialised length is; any other length goes through its code stream, below)
  const u32 l4 = len >> 2;                         // code bytes per row (1..8)
  if (l4 == 4) return expand16(*reinterpret_cast<const u32*>(tile + c * rstride + foff));
  if (l4 == 8) return expand16(*reinterpret_cast<const u32*>(tile + (c >> 1) * rstride + foff + (c & 1) * 4));
  const u32 d = 4 * c;                             // 4,8,12,20,24,28 bases: gather 4 code bytes
  u32 r = (d * (65536u / l4 + 1)) >> 16;           // d / l4 for d < 1024
  u32 q = d - r * l4;
  u32 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = expand4(tile[r * rstride + foff + q]);
    if (++q == l4) { q = 0; ++r; }
  }
  u32x4 o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
  return o;
}

__device__ __forceinline__ u64 mask2(u32 len) { return len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1); }

// ---- runtime-length fields: the code stream ----------------------------------------------------------------------------
// A field whose length is only known at run time (any of 1..32 without a specialisation: header.rs:180-185) goes through
// its CODE STREAM: the rows' 2*len code bits back to back, row r at bit r*2*len.  The stream of a 128-row tile is 32*len
// bytes, and its dword c holds exactly the 16 bases of chunk c of the ASCII column — between the stream and the column
// every length looks like length 16 (one ds_read_b32 / ds_write_b32 and four v_perm expansions / SWAR packs per 16-byte
// chunk).  What is ragged sits between the rows and the stream, and costs a fixed handful of instructions per ROW:
//   decode: the row's masked code word shifted to its bit position and OR-ed into 2 (len <= 16) or 3 dwords of a zeroed
//           stream with LDS atomics (ds_or_b32, no return value; a wave's DS instructions execute in order);
//   encode: three dword reads at the row's bit position and two v_alignbit_b32.
// Round 3 resolved every byte of such a field on its own (ds_read_u8 + v_bfe + shift/or per base; pack1 per byte):
// (31,31) decode 0.49 / encode 0.53 of peak where the specialised lengths ran 0.75 / 0.65.
static constexpr u32 kStreamPad = 16;                                   // the third dword of the last row's access
__host__ __device__ constexpr u32 stream_bytes(int nt) { return 1024u * nt + kStreamPad; }   // 32 B/row x 128 x nt rows at len 32

#ifndef IBU_STREAM_PROBE   // measurement builds (WRONG output): 1 = plain stores instead of the LDS atomics, 2 = no stream build at all
#define IBU_STREAM_PROBE 0
#endif
__device__ __forceinline__ void lds_or(u32* p, u32 v) {
#if IBU_STREAM_PROBE == 1
  *p = v;
#else
  (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
}
// Rows of a staged tile (u64 field at byte `foff` of records of stride `rstride`; bits at and above 2*len ignored: F7) ->
// the field's code stream.  Lane L places rows 2L, 2L+1 of each of the NT sub-tiles.
template <int NT>
__device__ __forceinline__ void stream_from_rows(const uint8_t* tile, u32 rstride, u32 foff, u32 len, u32* stream, u32 lane) {
#if IBU_STREAM_PROBE == 2
  return;
#endif
  u32x4 z; z.x = z.y = z.z = z.w = 0;
#pragma unroll
  for (int k = 0; k < NT; ++k) *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(stream) + 1024 * k + 16 * lane) = z;
  if (lane == 0) *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(stream) + 1024 * NT) = z;
  wave_lds_fence();
  const u64 m = mask2(len);
  const bool wide = len > 16;                      // wave-uniform: a row of at most 32 bits touches two dwords
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32 r = 128 * j + 2 * lane + h;
      const u64 v = *reinterpret_cast<const u64*>(tile + (size_t)r * rstride + foff) & m;
      const u32 bit = r * 2 * len, s = bit & 31u;
      u32* d = stream + (bit >> 5);
      const u64 t = v << s;
      lds_or(d, (u32)t);
      lds_or(d + 1, (u32)(t >> 32));
      if (wide) lds_or(d + 2, (u32)((v >> 32) >> (32 - s)));   // the s bits shifted out of t (u64 arithmetic: s == 0 gives 0)
    }
  wave_lds_fence();
}
// The code stream -> the ASCII column of the tile.  Fixed trip count, stores predicated: a store in a runtime-count loop
// would turn the wait in front of the next tile's LDS writes into vmcnt(0) (see the head of this file).
template <int NT>
__device__ __forceinline__ void expand_stream(const u32* stream, u32 len, uint8_t* out_tile, u32 lane) {
  const u32 nchunks = 8 * len * NT;
#pragma unroll
  for (int i = 0; i < 4 * NT; ++i) {
    const u32 c = lane + 64 * i;
    if (c < nchunks) st16(out_tile + 16 * (size_t)c, expand16(stream[c]));
  }
}
// Row r of a code stream (encode side).  Reads up to 8 bytes past the row: the stream area is padded (kStreamPad).
__device__ __forceinline__ u64 stream_row(const u32* stream, u32 r, u32 len) {
  const u32 bit = r * 2 * len, s = bit & 31u;
  const u32* d = stream + (bit >> 5);
  const u32 w0 = d[0], w1 = d[1], w2 = d[2];
  const u32 lo = __builtin_amdgcn_alignbit(w1, w0, s), hi = __builtin_amdgcn_alignbit(w2, w1, s);   // ({w1,w0} >> s) & 0xffffffff
  return (((u64)hi << 32) | lo) & mask2(len);
}

// Output-centric expansion of one field of a staged tile: every lane produces whole 16-byte
// chunks of the ASCII stream, so every store is a full coalesced dwordx4 whatever len is.
//   LEN > 0 : compile-time length; ceil(LEN/8) rounds, straight-line; lanes past the last chunk
//             recompute and re-store the LAST chunk (same bytes, same address: benign) so no
//             store is exec-branched and the store count per tile is exact.
//   LEN == 0: runtime length: rows -> code stream -> chunks (above).
template <int LEN, int NT = 1>  // NT: 128-record tiles staged back to back in `tile`; LEN: 0 or a multiple of 4
__device__ __forceinline__ void expand_field(const uint8_t* tile, u32 rstride, u32 foff, u32 rt_len,
                                             uint8_t* out_tile, u32 lane, u32* stream = nullptr) {
  static_assert((LEN & 3) == 0, "specialised lengths are multiples of 4");
  if constexpr (LEN > 0) {
    constexpr u32 last = 8 * LEN * NT - 1;
    constexpr int rounds = (8 * LEN * NT + kWave - 1) / kWave;
#pragma unroll
    for (int i = 0; i < rounds; ++i) {
      u32 c = lane + 64 * i;
      c = c < last ? c : last;
      st16(out_tile + 16 * (size_t)c, expand_chunk(tile, rstride, foff, (u32)LEN, c));
    }
  } else {                                         // runtime length: through the field's code stream (`stream`: stream_bytes(NT) of LDS)
    stream_from_rows<NT>(tile, rstride, foff, rt_len, stream, lane);
    expand_stream<NT>(stream, rt_len, out_tile, lane);
  }
}

// Record-centric packing of one row of ASCII bytes staged in LDS.
template <int L4>
__device__ __forceinline__ u64 pack_row_dwords(const uint8_t* row, bool& ok) {
  u64 v = 0;
#pragma unroll
  for (int q = 0; q < L4; ++q) v |= (u64)pack4(*reinterpret_cast<const u32*>(row + 4 * q), ok) << (8 * q);
  return v;
}
__device__ __forceinline__ u64 pack_row_bytes(const uint8_t* row, u32 len, bool& ok) {
  u64 v = 0;
  for (u32 i = 0; i < len; ++i) v |= (u64)pack1(row[i], ok) << (2 * i);
  return v;
}
// Row `r` of a field staged at `field` by AsciiStage<LEN>::land: ASCII rows (LEN > 0) or the code stream (LEN == 0, whose
// validity was settled per chunk when it was landed: `ok` is not touched).
template <int LEN>
__device__ __forceinline__ u64 pack_row(const uint8_t* field, u32 r, u32 rt_len, bool& ok) {
  if constexpr (LEN > 0 && (LEN & 3) == 0) {
    return pack_row_dwords<LEN / 4>(field + r * LEN, ok);
  } else {
    static_assert(LEN == 0, "specialised lengths are multiples of 4; every other length is a runtime length");
    return stream_row(reinterpret_cast<const u32*>(field), r, rt_len);
  }
}

// Staging of one ASCII tile global -> registers -> LDS (linear).
//   LEN > 0 : ceil(LEN/8) wave-wide dwordx4 loads in straight-line code; lanes past the last
//             chunk re-read the last chunk (same cache line as their neighbours: no extra HBM
//             traffic) and skip the LDS write.
//   LEN == 0: runtime length; RND rounds (0: four, the most a 128-row tile of 32-base rows needs), the unused ones predicated
//             off; lands as the field's code stream.  A caller that knows the row length class picks NT and RND so that every
//             round carries chunks (ibu_k_pack: short rows staged four rounds deep used a quarter of their loads).
template <int LEN, int NT = 1, int RND = 0>  // NT: 128-row tiles back to back
struct AsciiStage {
  static constexpr int rounds = LEN > 0 ? (LEN * NT + 7) / 8 : (RND > 0 ? RND : 4);
  u32x4 v[rounds];
  __device__ __forceinline__ void issue(const uint8_t* g, u32 rt_len, u32 lane) {
    const u32 last = 8 * (LEN > 0 ? (u32)(LEN * NT) : rt_len * NT) - 1;
#pragma unroll
    for (int i = 0; i < rounds; ++i) {
      u32 c = lane + 64 * i;
      c = c < last ? c : last;
      v[i] = ld16(g + 16 * (size_t)c);
    }
  }
  // Registers -> LDS.  LEN > 0: the ASCII bytes, linear.  LEN == 0: every 16-byte chunk packed to the 32 code bits of its 16
  // bases = dword c of the field's code stream (the rows of the NT tiles are contiguous, so is their stream).  Returns false in
  // a lane one of whose chunks holds a byte outside ACGTacgt (LEN == 0 only; rows are attributed by the caller's slow path).
  __device__ __forceinline__ bool land(uint8_t* lds, u32 rt_len, u32 lane) const {
    const u32 nchunks = 8 * (LEN > 0 ? (u32)(LEN * NT) : rt_len * NT);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < rounds; ++i) {
      const u32 c = lane + 64 * i;
      if constexpr (LEN > 0) {
        if (c < nchunks) *reinterpret_cast<u32x4*>(lds + 16 * c) = v[i];
      } else {
        if (64u * i < nchunks) {                     // wave-uniform: a round past the column costs nothing
          bool okc = true;
          const u32 w = pack4(v[i].x, okc) | (pack4(v[i].y, okc) << 8) | (pack4(v[i].z, okc) << 16) | (pack4(v[i].w, okc) << 24);
          if (c < nchunks) { reinterpret_cast<u32*>(lds)[c] = w; ok = ok && okc; }
        }
      }
    }
    return ok;
  }
};

// Offending rows (a byte outside ACGTacgt) are tallied per lane in registers while a wave sweeps its tiles and
// reported ONCE per wave when it leaves the kernel: one atomicMin (first offending row) + one atomicAdd (count),
// and only if something was wrong.  No atomics in the loop: input that is wrong everywhere costs the same as
// valid input (it used to cost two atomics per tile: 365 ms instead of 10 ms at 1e9 all-invalid rows).
struct BadRows {
  u64 first = ~0ull;
  u32 count = 0;
  __device__ __forceinline__ void note(bool bad, u64 row_global) {
    if (bad) { first = row_global < first ? row_global : first; ++count; }
  }
  __device__ __forceinline__ void flush(u64* status) const {
    if (__ballot(count != 0) == 0) return;         // wave-uniform: the common case leaves here
    u64 f = first;
    u32 c = count;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const u32 flo = __shfl_xor((u32)f, m), fhi = __shfl_xor((u32)(f >> 32), m);
      const u64 fo = ((u64)fhi << 32) | flo;
      f = fo < f ? fo : f;
      c += __shfl_xor(c, m);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { atomicMin(&status[0], f); atomicAdd(&status[1], (u64)c); }
  }
};


// ---- host-side launch helpers ------------------------------------------------------------------
static inline u32 grid_for(u32 ntiles, int cus, int blocks_per_cu) {
  u32 need = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
  u32 cap = (u32)(cus * blocks_per_cu);
  return need < cap ? (need ? need : 1) : cap;
}
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Peeling.  The tiled kernels need every array 16-B aligned.  A resident shard that starts at an odd record of a
// larger buffer (perfectly legal: the reference's split gives `per = len / n`, mmap.rs:297-307) is 8-B but not 16-B
// aligned, and so may be its columns.  Instead of sending the whole shard through the one-thread-per-record tail
// kernel, peel rows off the FRONT until every column is aligned: the smallest h in [0, 16) with
// (p + h * stride) % 16 == 0 for every non-null column (16 rows always restore the phase, so 16 candidates decide).
// -1: no such h (columns whose misalignments are inconsistent, e.g. an odd byte address) -> tail kernel for all rows.
struct Span { const void* p; size_t stride; };
static inline int peel_rows(const Span* s, int k) {
  for (int h = 0; h < 16; ++h) {
    bool ok = true;
    for (int i = 0; i < k; ++i)
      if (s[i].p && ((reinterpret_cast<uintptr_t>(s[i].p) + (uintptr_t)h * s[i].stride) & 15u)) ok = false;
    if (ok) return h;
  }
  return -1;
}
// Split n rows into head (tail kernel) + main (tiled kernel, `tile` rows per tile) + rest (tail kernel).
// Context option "trace_rows" (tests only; a context starts with the value IBU_TRACE_ROWS had when the library first looked):
// one stderr line per split saying how many rows took which kernel, so that a test can assert that an odd-record shard
// still runs tiled WITHOUT timing anything (a wall-clock ratio in the correctness suite is a flake on a pool whose
// placements differ by 20 %).  Not an environment lookup per launch: the ring pipelines launch per slot from several host
// threads.
struct RowSplit { size_t head, main; };
static inline RowSplit split_rows(const LaunchCfg& cfg, const Span* s, int k, size_t n, size_t tile) {
  const int h = peel_rows(s, k);
  RowSplit rs{n, 0};
  if (h >= 0) {
    rs.head = (size_t)h < n ? (size_t)h : n;
    rs.main = ((n - rs.head) / tile) * tile;
  }
  if (cfg.trace_rows) fprintf(stderr, "ibu rows: n=%zu head=%zu tiled=%zu rest=%zu tile=%zu\n", n, rs.head, rs.main, n - rs.head - rs.main, tile);
  return rs;
}
template <class T> static inline T* adv(T* p, size_t bytes) {
  return p ? reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + bytes) : p;
}
static inline u32 tail_grid(u64 rows) { return (u32)((rows + 255) / 256); }

// Block-wide exclusive scan of one u32 per thread over the first 256 threads of the block (every thread of the block
// must call; threads >= 256 pass 0 and get garbage).  Leaves the total in *total.
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* wsum /*[4] shared*/, u32* total) {
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const u32 t = __shfl_up(inc, d);
    if (lane >= (u32)d) inc += t;
  }
  if (lane == kWave - 1 && wib < 4) wsum[wib] = inc;
  __syncthreads();
  u32 off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const u32 s = wsum[w];
    if ((u32)w < wib) off += s;
    tot += s;
  }
  __syncthreads();  // wsum may be reused by the caller's next scan
  *total = tot;
  return off + inc - v;
}

}  // namespace ibu
