// k_records.hip — K1/K1' (AoS <-> u64 columns), K4 reduce, synthetic generator, sortedness check.
// Design notes: kcommon.hpp.  Reference semantics are cited at the C-ABI entry points in device.cpp
// and include/ibu_hip.h (cast_slice of &[Record]: reader.rs:301, writer.rs:317, mmap.rs:268; the
// in-repo processors: lib.rs:117-129, examples/parallel.rs:21-36, examples/roundtrip.rs:84-87).
#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {

// =============================================================================================
// K1  deserialise AoS -> three u64 columns          K1' serialise columns -> AoS
// =============================================================================================
// NT tiles of 128 records per wave iteration (kRecNT).  Decode's recipe (NT = 2: twice the bytes in flight per wave) was tried
// here in round 3 and bought nothing — one process, same buffers, 1e9 records (profiles/r03_c_kbench_rec_nt.jsonl):
// deserialize 7.69 ms (NT 1) / 7.74 (NT 2), serialize 7.74 / 7.83, i.e. 6.2 TB/s = 0.775-0.78 of peak either way, above the
// plain copy kernel on the same box (5.8 TB/s); both kernels' store rounds are full with one tile already.  NT stays 1.
#ifndef IBU_REC_NT
#define IBU_REC_NT 1
#endif
static constexpr int kRecNT = IBU_REC_NT;
static constexpr int kRecTileRecs = kTileRecs * kRecNT, kRecTileBytes = kTileBytes * kRecNT;
struct RecRegs { u32x4 v[3 * kRecNT]; };

extern "C" __global__ void __launch_bounds__(kBlock, kRecNT > 1 ? 5 : 8)
ibu_k_deserialize(const uint8_t* __restrict__ recs, u32 ntiles, u64* __restrict__ bc,
                  u64* __restrict__ umi, u64* __restrict__ idx) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kRecTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kRecTileBytes;
  u64* const cols[3] = {bc, umi, idx};
  sweep_tiles<RecRegs>(                            // two register sets take turns (kcommon.hpp)
      tile_range(ntiles, wib),
      [&](RecRegs& g, u32 t) {
        const uint8_t* src = recs + (size_t)t * kRecTileBytes + 16 * lane;
#pragma unroll
        for (int k = 0; k < 3 * kRecNT; ++k) g.v[k] = ld16(src + 1024 * k);
      },
      [&](const RecRegs& g, u32 t) {
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 3 * kRecNT; ++k) *reinterpret_cast<u32x4*>(tile + 1024 * k + 16 * lane) = g.v[k];
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < kRecNT; ++j) {         // chunk c of a column = records 2c, 2c+1
          const u32 c = lane + 64 * j;
          const u64* r0 = reinterpret_cast<const u64*>(tile + (2 * c) * 24);
          const u64* r1 = r0 + 3;
#pragma unroll
          for (int f = 0; f < 3; ++f) {
            const u64 x = r0[f], y = r1[f];
            u32x4 o; o.x = (u32)x; o.y = (u32)(x >> 32); o.z = (u32)y; o.w = (u32)(y >> 32);
            st16(reinterpret_cast<uint8_t*>(cols[f]) + (size_t)t * (1024 * kRecNT) + 16 * c, o);
          }
        }
      });
}

extern "C" __global__ void __launch_bounds__(kBlock, kRecNT > 1 ? 5 : 8)
ibu_k_serialize(const u64* __restrict__ bc, const u64* __restrict__ umi, const u64* __restrict__ idx,
                u32 ntiles, uint8_t* __restrict__ recs) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kRecTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kRecTileBytes;
  const uint8_t* const cols[3] = {reinterpret_cast<const uint8_t*>(bc), reinterpret_cast<const uint8_t*>(umi),
                                  reinterpret_cast<const uint8_t*>(idx)};
  sweep_tiles<RecRegs>(
      tile_range(ntiles, wib),
      [&](RecRegs& g, u32 t) {                     // v[3 j + f]: records 2c, 2c+1 of column f, c = lane + 64 j
#pragma unroll
        for (int j = 0; j < kRecNT; ++j)
#pragma unroll
          for (int f = 0; f < 3; ++f) g.v[3 * j + f] = ld16(cols[f] + (size_t)t * (1024 * kRecNT) + 16 * (lane + 64 * j));
      },
      [&](const RecRegs& g, u32 t) {
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < kRecNT; ++j) {
          const u32x4 c0 = g.v[3 * j], c1 = g.v[3 * j + 1], c2 = g.v[3 * j + 2];
          u32x4* r = reinterpret_cast<u32x4*>(tile + (lane + 64 * j) * 48);  // two adjacent records = 48 B
          u32x4 w0, w1, w2;
          w0.x = c0.x; w0.y = c0.y; w0.z = c1.x; w0.w = c1.y;     // bc[2c]   umi[2c]
          w1.x = c2.x; w1.y = c2.y; w1.z = c0.z; w1.w = c0.w;     // idx[2c]  bc[2c+1]
          w2.x = c1.z; w2.y = c1.w; w2.z = c2.z; w2.w = c2.w;     // umi[2c+1] idx[2c+1]
          r[0] = w0; r[1] = w1; r[2] = w2;
        }
        wave_lds_fence();
        uint8_t* dst = recs + (size_t)t * kRecTileBytes + 16 * lane;
#pragma unroll
        for (int k = 0; k < 3 * kRecNT; ++k) st16(dst + 1024 * k, *reinterpret_cast<const u32x4*>(tile + 1024 * k + 16 * lane));
      });
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
  u32 t = logical_block() * kWavesPerBlock + wib;
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
  // acc: kReduceSlots x 8 words; a workgroup adds into slot blockIdx % kReduceSlots, ibu_k_reduce_fold folds them at fetch
  // time (one slot: ~1800 same-address atomics per word at the end of a resident grid, ~0.1 ms — most of a small batch)
  u64* slot = acc + 8 * (blockIdx.x & (kReduceSlots - 1));
  if (threadIdx.x < 6) {
    u64 v = part[0][threadIdx.x];
    for (int w = 1; w < kWavesPerBlock; ++w)
      v = threadIdx.x < 3 ? v + part[w][threadIdx.x] : v ^ part[w][threadIdx.x];
    if (threadIdx.x < 3) { if (v) atomicAdd(&slot[1 + threadIdx.x], v); }
    else                 { if (v) atomicXor(&slot[1 + threadIdx.x], v); }
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
// Record-centric like K1': lane L of a wave builds records 2L and 2L+1 of its 128-record tile (4 splitmix64, no
// division, no per-element branch), parks the 48 bytes in the wave's LDS slice (3 x ds_write_b128 at stride 48 B,
// conflict-free) and the wave stores the tile as three coalesced dwordx4.  (The first version derived every u64
// element from its flat index: a 64-bit divide by 3 per element kept it at 4.3 TB/s.)
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_generate(u64 seed, u64 first, u32 ntiles, u32 bc_len, u32 umi_len, uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * kTileBytes];
  const u32 lane = threadIdx.x & (kWave - 1);
  const u32 wib = threadIdx.x >> 6;
  uint8_t* tile = lds + wib * kTileBytes;
  const u32 nwaves = gridDim.x * kWavesPerBlock;
  const u64 mb = mask2(bc_len), mu = mask2(umi_len);
  for (u32 t = logical_block() * kWavesPerBlock + wib; t < ntiles; t += nwaves) {
    const u64 g0 = first + (u64)t * kTileRecs + 2 * lane, g1 = g0 + 1;
    const u64 b0 = splitmix64(seed + 3 * g0) & mb, u0 = splitmix64(seed + 3 * g0 + 1) & mu;
    const u64 b1 = splitmix64(seed + 3 * g1) & mb, u1 = splitmix64(seed + 3 * g1 + 1) & mu;
    wave_lds_fence();                          // previous tile's LDS reads precede these writes
    u32x4* r = reinterpret_cast<u32x4*>(tile + lane * 48);
    u32x4 w0, w1, w2;
    w0.x = (u32)b0; w0.y = (u32)(b0 >> 32); w0.z = (u32)u0; w0.w = (u32)(u0 >> 32);
    w1.x = (u32)g0; w1.y = (u32)(g0 >> 32); w1.z = (u32)b1; w1.w = (u32)(b1 >> 32);
    w2.x = (u32)u1; w2.y = (u32)(u1 >> 32); w2.z = (u32)g1; w2.w = (u32)(g1 >> 32);
    r[0] = w0; r[1] = w1; r[2] = w2;
    wave_lds_fence();
    uint8_t* dst = out + (size_t)t * kTileBytes + 16 * lane;
    st16(dst, *reinterpret_cast<const u32x4*>(tile + 16 * lane));
    st16(dst + 1024, *reinterpret_cast<const u32x4*>(tile + 1024 + 16 * lane));
    st16(dst + 2048, *reinterpret_cast<const u32x4*>(tile + 2048 + 16 * lane));
  }
}
extern "C" __global__ void ibu_k_generate_tail(u64 seed, u64 first, u64 row0, u64 n, u32 bc_len, u32 umi_len, u64* __restrict__ out) {
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 g = first + i;
  out[3 * i] = splitmix64(seed + 3 * g) & mask2(bc_len);
  out[3 * i + 1] = splitmix64(seed + 3 * g + 1) & mask2(umi_len);
  out[3 * i + 2] = g;
}

// =============================================================================================
// Plain streaming copy: the device-side form of the reference's memcpy hot loops (write_slice's
// copy_from_slice, writer.rs:335-347; Writer::ingest's append, writer.rs:477-482) and the
// on-device COPY CEILING every other kernel here is priced against ("measure an
// on-device copy ceiling with a plain dwordx4 copy kernel and report both denominators").
// Nontemporal both ways.
// =============================================================================================
// Tiles of 4 KiB per wave (four dwordx4 per lane), swept like every streaming kernel here: XCD-static ownership, the next
// tile's loads in flight while this one's stores are issued, two register sets taking turns.  (The first form — a grid-stride
// loop with four chunks in flight — ran at 5.7-5.8 TB/s where deserialize, moving the same bytes through LDS, reached 6.2.)
struct CopyRegs { u32x4 v[4]; };
extern "C" __global__ void __launch_bounds__(kBlock, 8)
ibu_k_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, u64 nchunks) {
  const u32 lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
  const u32 ntiles = (u32)(nchunks >> 8);                    // 256 chunks = 4 KiB per tile; the launcher keeps nchunks below 2^40
  sweep_tiles<CopyRegs>(
      tile_range(ntiles, wib),
      [&](CopyRegs& g, u32 t) {
        const uint8_t* p = src + (size_t)t * 4096 + 16 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) g.v[k] = ld16(p + 1024 * k);
      },
      [&](const CopyRegs& g, u32 t) {
        uint8_t* q = dst + (size_t)t * 4096 + 16 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) st16(q + 1024 * k, g.v[k]);
      });
  // the chunks behind the last whole tile (fewer than 256): the first workgroup's threads
  const u64 done = (u64)ntiles << 8;
  if (blockIdx.x == 0) {
    const u64 c = done + threadIdx.x;
    if (c < nchunks) st16(dst + 16 * c, ld16(src + 16 * c));
  }
}
// First differing 8-byte word of two word arrays, or ~0: `a == b` on two record slices (Record derives PartialEq / Eq,
// src/constructs/record.rs:58) with the position a test wants.  VEC 2: both 16-B aligned, dwordx4 loads, four chunks of
// each array in flight; VEC 1: 8-B aligned inputs.  One atomicMin per wave, and only where something differs.
template <int VEC>
__global__ void __launch_bounds__(kBlock, 8)
ibu_k_mismatch(const u64* __restrict__ a, const u64* __restrict__ b, u64 nwords, u64* __restrict__ first) {
  const u64 stride = (u64)gridDim.x * kBlock;
  u64 c = (u64)logical_block() * kBlock + threadIdx.x;
  u64 best = ~0ull;
  if constexpr (VEC == 2) {
    const uint8_t* pa = reinterpret_cast<const uint8_t*>(a);
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(b);
    const u64 nch = nwords >> 1;
    auto cmp = [&](u32x4 x, u32x4 y, u64 chunk) {
      const bool lo = x.x == y.x && x.y == y.y, hi = x.z == y.z && x.w == y.w;
      if (!(lo && hi)) { const u64 w = 2 * chunk + (lo ? 1 : 0); best = w < best ? w : best; }
    };
    for (; c + 3 * stride < nch; c += 4 * stride) {
      const u32x4 x0 = ld16(pa + 16 * c), x1 = ld16(pa + 16 * (c + stride)), x2 = ld16(pa + 16 * (c + 2 * stride)), x3 = ld16(pa + 16 * (c + 3 * stride));
      const u32x4 y0 = ld16(pb + 16 * c), y1 = ld16(pb + 16 * (c + stride)), y2 = ld16(pb + 16 * (c + 2 * stride)), y3 = ld16(pb + 16 * (c + 3 * stride));
      cmp(x0, y0, c); cmp(x1, y1, c + stride); cmp(x2, y2, c + 2 * stride); cmp(x3, y3, c + 3 * stride);
    }
    for (; c < nch; c += stride) cmp(ld16(pa + 16 * c), ld16(pb + 16 * c), c);
    if ((nwords & 1) && blockIdx.x == 0 && threadIdx.x == 0 && a[nwords - 1] != b[nwords - 1]) best = nwords - 1 < best ? nwords - 1 : best;
  } else {
    for (; c < nwords; c += stride)
      if (a[c] != b[c]) best = c < best ? c : best;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const u64 o = shfl_xor_u64(best, m); best = o < best ? o : best; }
  if ((threadIdx.x & (kWave - 1)) == 0 && best != ~0ull) atomicMin(first, best);
}
extern "C" __global__ void ibu_k_copy_bytes(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, u64 off, u64 n) {
  const u64 i = off + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// =============================================================================================
// Tail / unaligned kernels: one thread per record, no alignment assumption beyond the natural
// 8 B of the u64 columns and records.  Used for n % 128 and for misaligned bases only.
// =============================================================================================
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
extern "C" __global__ void ibu_k_reduce_tail(const u64* __restrict__ recs, u64 row0, u64 n, u64* acc) {
  // one thread per record (peeled head rows, the n % 128 rest, or a whole input no peel can align); each wave folds its
  // 64 records with shuffles first, so the atomics are six per WAVE whatever the size
  const u64 i = row0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 v[3] = {0, 0, 0};
  if (i < n) { v[0] = recs[3 * i]; v[1] = recs[3 * i + 1]; v[2] = recs[3 * i + 2]; }
  u64 S[3] = {v[0], v[1], v[2]}, X[3] = {v[0], v[1], v[2]};
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int g = 0; g < 3; ++g) { S[g] += shfl_xor_u64(S[g], m); X[g] ^= shfl_xor_u64(X[g], m); }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    u64* slot = acc + 8 * (blockIdx.x & (kReduceSlots - 1));
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      if (S[g]) atomicAdd(&slot[1 + g], S[g]);
      if (X[g]) atomicXor(&slot[4 + g], X[g]);
    }
  }
}
// slots -> slot 0, the other slots cleared (so that further ibu_reduce calls keep accumulating): one wave, lane = slot
extern "C" __global__ void ibu_k_reduce_fold(u64* acc) {
  const u32 lane = threadIdx.x;
  u64 v[7];
#pragma unroll
  for (int w = 0; w < 7; ++w) v[w] = acc[8 * lane + w];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int w = 0; w < 7; ++w) {
      const u64 o = shfl_xor_u64(v[w], m);
      v[w] = w < 4 ? v[w] + o : v[w] ^ o;
    }
#pragma unroll
  for (int w = 0; w < 7; ++w) acc[8 * lane + w] = lane == 0 ? v[w] : 0;
}
hipError_t launch_reduce_fold(uint64_t* acc, hipStream_t st) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(ibu_k_reduce_fold, dim3(1), dim3(kReduceSlots), 0, st, (u64*)acc);
  return hipGetLastError();
}
extern "C" __global__ void ibu_k_fill_u64(u64* p, u64 v0, u64 v1) { p[0] = v0; p[1] = v1; }

// =============================================================================================
// Launchers
// =============================================================================================
hipError_t launch_deserialize(const LaunchCfg& cfg, const void* recs, size_t n, uint64_t* bc, uint64_t* umi,
                              uint64_t* idx, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[4] = {{recs, 24}, {bc, 8}, {umi, 8}, {idx, 8}};
  const RowSplit rs = split_rows(cfg, sp, 4, n, kRecTileRecs);   // peel rows until every array is 16-B aligned
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_deserialize_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, (const u64*)recs, (u64)0, (u64)rs.head,
                       (u64*)bc, (u64*)umi, (u64*)idx);
  if (rs.main) {
    u32 ntiles = (u32)(rs.main / kRecTileRecs);
    static std::atomic<int> occ;
    hipLaunchKernelGGL(ibu_k_deserialize,
                       dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_deserialize, 0, &occ))), dim3(kBlock), 0, st,
                       adv((const uint8_t*)recs, 24 * rs.head), ntiles, adv((u64*)bc, 8 * rs.head), adv((u64*)umi, 8 * rs.head),
                       adv((u64*)idx, 8 * rs.head));
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_deserialize_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, (const u64*)recs,
                       (u64)(rs.head + rs.main), (u64)n, (u64*)bc, (u64*)umi, (u64*)idx);
  return hipGetLastError();
}

hipError_t launch_serialize(const LaunchCfg& cfg, const uint64_t* bc, const uint64_t* umi, const uint64_t* idx,
                            size_t n, void* recs, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[4] = {{recs, 24}, {bc, 8}, {umi, 8}, {idx, 8}};
  const RowSplit rs = split_rows(cfg, sp, 4, n, kRecTileRecs);
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_serialize_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, (const u64*)bc, (const u64*)umi,
                       (const u64*)idx, (u64)0, (u64)rs.head, (u64*)recs);
  if (rs.main) {
    u32 ntiles = (u32)(rs.main / kRecTileRecs);
    static std::atomic<int> occ;
    hipLaunchKernelGGL(ibu_k_serialize, dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_serialize, 0, &occ))),
                       dim3(kBlock), 0, st, adv((const u64*)bc, 8 * rs.head), adv((const u64*)umi, 8 * rs.head),
                       adv((const u64*)idx, 8 * rs.head), ntiles, adv((uint8_t*)recs, 24 * rs.head));
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_serialize_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, (const u64*)bc,
                       (const u64*)umi, (const u64*)idx, (u64)(rs.head + rs.main), (u64)n, (u64*)recs);
  return hipGetLastError();
}

hipError_t launch_reduce(const LaunchCfg& cfg, const void* recs, size_t n, uint64_t* acc, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[1] = {{recs, 24}};
  const RowSplit rs = split_rows(cfg, sp, 1, n, kTileRecs);   // an 8-B aligned base peels exactly one record
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_reduce_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, (const u64*)recs, (u64)0, (u64)rs.head,
                       (u64*)acc);
  u32 ntiles = (u32)(rs.main / kTileRecs);
  // the main kernel also adds n to the count slot, so it always runs (ntiles may be 0)
  static std::atomic<int> occ;
  hipLaunchKernelGGL(ibu_k_reduce, dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_reduce, 0, &occ))),
                     dim3(kBlock), 0, st, adv((const uint8_t*)recs, 24 * rs.head), ntiles, (u64)n, (u64*)acc);
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_reduce_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, (const u64*)recs,
                       (u64)(rs.head + rs.main), (u64)n, (u64*)acc);
  return hipGetLastError();
}

hipError_t launch_generate(const LaunchCfg& cfg, uint64_t seed, uint64_t first, size_t n, uint32_t bc_len,
                           uint32_t umi_len, void* recs, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  if (n == 0) return hipSuccess;
  const Span sp[1] = {{recs, 24}};
  const RowSplit rs = split_rows(cfg, sp, 1, n, kTileRecs);
  if (rs.head)
    hipLaunchKernelGGL(ibu_k_generate_tail, dim3(tail_grid(rs.head)), dim3(256), 0, st, (u64)seed, (u64)first, (u64)0, (u64)rs.head,
                       bc_len, umi_len, (u64*)recs);
  if (rs.main) {
    const u32 ntiles = (u32)(rs.main / kTileRecs);
    static std::atomic<int> occ;
    hipLaunchKernelGGL(ibu_k_generate, dim3(grid_for(ntiles, cfg.cus, resident_blocks<kBlock>(cfg, ibu_k_generate, 0, &occ))),
                       dim3(kBlock), 0, st, (u64)seed, (u64)(first + rs.head), ntiles, bc_len, umi_len,
                       adv((uint8_t*)recs, 24 * rs.head));
  }
  if (rs.head + rs.main < n)
    hipLaunchKernelGGL(ibu_k_generate_tail, dim3(tail_grid(n - rs.head - rs.main)), dim3(256), 0, st, (u64)seed, (u64)first,
                       (u64)(rs.head + rs.main), (u64)n, bc_len, umi_len, (u64*)recs);
  return hipGetLastError();
}

hipError_t launch_copy(const LaunchCfg& cfg, const void* src, void* dst, size_t bytes, hipStream_t st) {
  (void)hipGetLastError();
  if (bytes == 0) return hipSuccess;
  // same phase modulo 16 (e.g. both buffers start at an odd record): peel the leading bytes, then whole chunks
  const uintptr_t ps = reinterpret_cast<uintptr_t>(src), pd = reinterpret_cast<uintptr_t>(dst);
  const bool fast = ((ps ^ pd) & 15u) == 0;
  size_t head = fast ? (size_t)((16 - (ps & 15u)) & 15u) : 0;
  if (head > bytes) head = bytes;
  if (head)
    hipLaunchKernelGGL(ibu_k_copy_bytes, dim3(1), dim3(256), 0, st, (const uint8_t*)src, (uint8_t*)dst, (u64)0, (u64)head);
  const u64 nchunks = fast ? (bytes - head) / 16 : 0;
  if (nchunks) {
    u64 blocks = (nchunks + kBlock - 1) / kBlock;
    static std::atomic<int> occ;
    const u64 cap = (u64)cfg.cus * resident_blocks<kBlock>(cfg, ibu_k_copy, 0, &occ);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ibu_k_copy, dim3((u32)blocks), dim3(kBlock), 0, st, (const uint8_t*)src + head, (uint8_t*)dst + head, nchunks);
  }
  const u64 done = head + nchunks * 16;
  if (done < bytes) {
    const u64 rest = bytes - done;
    if (rest > (1ull << 31) * 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ibu_k_copy_bytes, dim3((u32)((rest + 255) / 256)), dim3(256), 0, st, (const uint8_t*)src,
                       (uint8_t*)dst, done, (u64)bytes);
  }
  return hipGetLastError();
}

hipError_t launch_fill2(uint64_t* p, uint64_t v0, uint64_t v1, hipStream_t st) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  hipLaunchKernelGGL(ibu_k_fill_u64, dim3(1), dim3(1), 0, st, (u64*)p, (u64)v0, (u64)v1);
  return hipGetLastError();
}

hipError_t launch_mismatch(const LaunchCfg& cfg, const void* a, const void* b, size_t nwords, uint64_t* first, hipStream_t st) {
  (void)hipGetLastError();
  if (nwords == 0) return hipSuccess;
  const bool vec = aligned16(a) && aligned16(b);
  const u64 units = vec ? (nwords >> 1) + 1 : nwords;
  u64 blocks = (units + kBlock - 1) / kBlock;
  static std::atomic<int> occ[2];
  if (vec) {
    const u64 cap = (u64)cfg.cus * resident_blocks<kBlock>(cfg, ibu_k_mismatch<2>, 0, &occ[1]);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ibu_k_mismatch<2>, dim3((u32)blocks), dim3(kBlock), 0, st, (const u64*)a, (const u64*)b, (u64)nwords, (u64*)first);
  } else {
    const u64 cap = (u64)cfg.cus * resident_blocks<kBlock>(cfg, ibu_k_mismatch<1>, 0, &occ[0]);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ibu_k_mismatch<1>, dim3((u32)blocks), dim3(kBlock), 0, st, (const u64*)a, (const u64*)b, (u64)nwords, (u64*)first);
  }
  return hipGetLastError();
}

}  // namespace ibu
