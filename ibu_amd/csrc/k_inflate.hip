// k_inflate.hip — DEFLATE blocks (RFC 1951) inflated ON THE DEVICE, one wave per block: the device half of BGZF ingest (the
// reference reads gzip through niffler, src/io/reader.rs:345-352; a bgzip file is a chain of independent members of at most 64 KiB
// whose sizes stand in their headers, so the COMPRESSED bytes can cross the PCIe link — half of them for a records file — and a
// block each goes to a wave).  Launcher: launch_inflate_blocks (kernels.h); C ABI: ibu_inflate_blocks_device (device.cpp); the
// host walk over the block headers: ibu_bgzf_scan (host_io.cpp).  The host decoder with the same acceptance rules: pgzip.cpp.
//
// A wave decodes its block SEQUENTIALLY — every lane runs the same symbol loop on the same (wave-uniform) state, so there is no
// divergence and nothing to broadcast — and uses its lanes where the work is wide:
//   - the compressed bytes are staged through a 1 KiB ring in LDS, 256 bytes per load instruction, the next chunk always
//     requested one chunk ahead (its dword sits in a register until there is room), so a bit-buffer refill is three LDS reads;
//   - the decode tables live in LDS: 2^10 literal/length and 2^8 distance entries that resolve a code of up to 10 / 8 bits,
//     length or distance extra bits included, in one lookup; the rare longer codes are decoded canonically (counts per length and
//     the symbols in code order: 15 steps at most).  The tables are BUILT by the wave in parallel: code ranks by __ballot /
//     popcount per length, every lane fills the entries of its symbols;
//   - the output goes through a 4 KiB ring in LDS: a match of up to 3584 bytes back copies inside it, 64 bytes per step (the
//     source of a match that overlaps its destination is periodic, so every lane reads bytes that existed before the match); a
//     match from further back reads the block's own output back from global memory (flushed long before: the flush runs 256
//     bytes behind the write position at most).  Full 256-byte chunks leave the ring as one dword store per lane;
//   - the CRC-32 of the output is checked by the same wave: every lane takes 1/64 of the block, the partial values are combined
//     with x^(8 n) mod P (the identity crc32_combine uses), one wave reduction.
// A block is accepted exactly as the host decoder accepts it (pgzip.cpp, RawInflater::inflate): the final deflate block ends on
// the block's last compressed byte, the output has the announced length, the CRC matches; any invalid code, distance or size
// makes the block bad (status 1), a wrong CRC status 2.  Every loop consumes input or produces output and both are bounded by
// the descriptor, so a wave leaves any input — random bytes included — after at most 8 x comp_len + a few iterations.
#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {
namespace {

constexpr int kInfThreads = 256, kInfWaves = kInfThreads / kWave;
constexpr u32 kInRing = 1024, kOutRing = 4096, kNear = 3584;
constexpr u32 kLitRoot = 10, kDistRoot = 8;
constexpr u32 K_LIT = 0, K_BASE = 1, K_EOB = 2, K_LONG = 3, K_BAD = 4;

// Decode-table entry (the host decoder's layout): bits 0-7 the stream bits this step consumes (code + extra bits), bits 8-11 the
// code length alone (where the extra bits start), bits 12-14 the kind, bits 16-31 the payload (literal, base length / distance).
__device__ __forceinline__ u32 mk(u32 payload, u32 f, u32 kind, u32 nbits) { return (payload << 16) | (kind << 12) | (f << 8) | nbits; }

__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ u32 lit_entry(u32 sym, u32 nb) {
  if (sym < 256) return mk(sym, 0, K_LIT, nb);
  if (sym == 256) return mk(0, 0, K_EOB, nb);
  if (sym < 286) return mk(kLenBase[sym - 257], nb, K_BASE, nb + kLenExtra[sym - 257]);
  return mk(0, 0, K_BAD, nb);
}
__device__ __forceinline__ u32 dist_entry(u32 sym, u32 nb) {
  if (sym < 30) return mk(kDistBase[sym], nb, K_BASE, nb + kDistExtra[sym]);
  return mk(0, 0, K_BAD, nb);
}

// Everything the symbol loop computes is the same in all 64 lanes.  Said to the compiler (readfirstlane at every LDS read whose
// address is uniform), the bit buffer, the table entries and the positions live in SGPRs and the loop runs on the scalar unit; left
// unsaid, every one of its ~90 instructions per symbol is a vector instruction over 64 identical lanes (first version: 16.7 ms per
// 64 KiB block, 12 GB/s for the chip).
__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

struct InfWave {                         // one wave's slice of LDS: 11 584 bytes
  u32 lit[1u << kLitRoot];
  u32 dist[1u << kDistRoot];
  u32 in[kInRing / 4];
  u32 out[kOutRing / 4];
  uint16_t perm_lit[288];                // symbols in code order (by length, then value): the canonical decode of long codes
  uint16_t perm_dist[32];
  uint16_t cnt_lit[16], cnt_dist[16];    // codes per length
  uint16_t cl[128];                      // the code-length code: symbol << 8 | length
  uint8_t lens[352];                     // [0, 316): code lengths of a dynamic header; [320, 339): those of the code-length code
};

// The canonical decode tables of one code, built by the wave: lens[0, n) in LDS (n <= 320), root P.  false: a code set zlib rejects.
__device__ __forceinline__ bool build_tables(const uint8_t* lens, u32 n, u32 P, bool is_dist, u32* tab, uint16_t* cnts, uint16_t* perm, u32 lane) {
  const u64 lt = (1ull << lane) - 1;
  u32 run[16];                                              // codes of each length seen so far (wave-uniform)
#pragma unroll
  for (int l = 0; l < 16; ++l) run[l] = 0;
  u32 my_len[5], my_rank[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const u32 s = 64 * c + lane;
    const u32 li = s < n ? lens[s] : 0;
    u32 rk = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
      const u64 m = __ballot(li == (u32)l);
      if (li == (u32)l) rk = run[l] + (u32)__popcll(m & lt);
      run[l] += (u32)__popcll(m);
    }
    my_len[c] = li;
    my_rank[c] = rk;
  }
  u32 maxlen = 0;
#pragma unroll
  for (int l = 1; l < 16; ++l)
    if (run[l]) maxlen = l;
  const u32 root = 1u << P;
  for (u32 i = lane; i < root; i += kWave) tab[i] = mk(0, 0, K_BAD, 1);
  if (lane < 16) cnts[lane] = 0;
  if (maxlen == 0) { wave_lds_fence(); return is_dist; }    // no codes at all: literals only (allowed for distances)
  int left = 1;
#pragma unroll
  for (int l = 1; l < 16; ++l) {
    left = 2 * left - (int)run[l];
    if (left < 0) return false;                             // over-subscribed
  }
  if (left > 0 && maxlen != 1) return false;                // incomplete (allowed: a single 1-bit code)
  u32 next[16], offs[16];                                   // first code / first position in `perm` of each length
  {
    u32 code = 0, o = 0;
    next[0] = offs[0] = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
      code = (code + (l > 1 ? run[l - 1] : 0u)) << 1;
      next[l] = code;
      offs[l] = o;
      o += run[l];
    }
  }
  wave_lds_fence();                                         // the BAD fill above is in place before the entries
#pragma unroll
  for (int l = 1; l < 16; ++l)
    if (lane == 0) cnts[l] = (uint16_t)run[l];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const u32 s = 64 * c + lane, l = my_len[c];
    if (l == 0) continue;
    u32 nx = 0, of = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k)
      if (l == (u32)k) { nx = next[k]; of = offs[k]; }
    perm[of + my_rank[c]] = (uint16_t)s;
    const u32 code = nx + my_rank[c];
    const u32 r = __brev(code) >> (32 - l);                 // LSB-first streams: the code as the bit buffer shows it
    if (l <= P) {
      const u32 e = is_dist ? dist_entry(s, l) : lit_entry(s, l);
      for (u32 i = r; i < root; i += 1u << l) tab[i] = e;
    } else {
      tab[r & (root - 1)] = mk(0, 0, K_LONG, 0);
    }
  }
  wave_lds_fence();
  return true;
}

// A code longer than the root table resolves: the canonical decode, one bit at a time.  0xFFFF: no such code.
__device__ __forceinline__ u32 slow_symbol(const uint16_t* cnts, const uint16_t* perm, u64 buf, u32* nbits) {
  u32 code = 0, first = 0, index = 0;
  for (u32 len = 1; len <= 15; ++len) {
    code |= (u32)(buf >> (len - 1)) & 1u;
    const u32 c = uni(cnts[len]);
    if (code < first + c) { *nbits = len; return uni(perm[index + (code - first)]); }
    index += c;
    first = (first + c) << 1;
    code <<= 1;
  }
  return 0xFFFFu;
}

struct Crc32Pow { u32 x2n[32]; };                            // x^(2^k) mod P, reflected (host_io.cpp fills it)

__device__ __forceinline__ u32 multmodp(u32 a, u32 b) {      // a(x) b(x) mod P(x), reflected CRC-32
  u32 m = 1u << 31, p = 0;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1)) == 0) break;
    }
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
  }
  return p;
}
__device__ __forceinline__ u32 x8nmodp(const Crc32Pow& pw, u32 n) {   // x^(8 n) mod P
  u32 p = 1u << 31, k = 3;
  while (n) {
    if (n & 1u) p = multmodp(pw.x2n[k & 31], p);
    n >>= 1;
    ++k;
  }
  return p;
}

struct Inflate {                                             // the decoder of one block: every field wave-uniform except `pre`
  InfWave* w;
  const uint8_t* comp;                                       // the block's compressed bytes (readable 2 KiB past clen)
  uint8_t* out;                                              // where its output goes
  u32 lane, clen, isize;
  u64 buf;
  u32 cnt, ipos, in_loaded, pre;
  u32 opos, flushed;

  __device__ __forceinline__ u32 gload(u32 off) const {
    u32 v;
    __builtin_memcpy(&v, comp + off + 4 * lane, 4);
    return v;
  }
  __device__ __forceinline__ void in_fill() {                // at least 512 staged bytes in front of ipos; the next chunk requested
    if (in_loaded >= ipos + 512) return;
    do {
      w->in[((in_loaded & (kInRing - 1)) >> 2) + lane] = pre;
      in_loaded += 256;
      pre = gload(in_loaded);
    } while (in_loaded < ipos + 512);
    wave_lds_fence();
  }
  __device__ __forceinline__ void in_start(u32 p) {          // (re)start reading at byte p of the compressed block
    in_loaded = p & ~255u;
    pre = gload(in_loaded);
    buf = 0;
    cnt = 0;
    ipos = p;
  }
  __device__ __forceinline__ u64 in_read64(u32 p) const {
    const u32 i = (p & (kInRing - 1)) >> 2, sh = 8 * (p & 3);
    const u32 a = uni(w->in[i]), b = uni(w->in[(i + 1) & (kInRing / 4 - 1)]), c = uni(w->in[(i + 2) & (kInRing / 4 - 1)]);
    const u64 lo = ((u64)b << 32) | a;
    return sh ? (lo >> sh) | ((u64)c << (64 - sh)) : lo;
  }
  // buf holds cnt valid low bits; what lies above them is zero or the true upcoming bits (a refill ORs the same bits again)
  __device__ __forceinline__ void refill() {
    in_fill();
    buf |= in_read64(ipos) << cnt;
    ipos += (63 - cnt) >> 3;
    cnt |= 56;
  }
  __device__ __forceinline__ void drop(u32 k) { buf >>= k; cnt -= k; }
  __device__ __forceinline__ u32 take(u32 k) { const u32 v = (u32)(buf & ((1ull << k) - 1)); drop(k); return v; }
  __device__ __forceinline__ u32 bitpos() const { return ipos * 8 - cnt; }

  __device__ __forceinline__ void flush_full() {             // whole 256-byte chunks of the ring -> global, a dword per lane
    if (opos - flushed < 256) return;
    wave_lds_fence();
    do {
      const u32 v = w->out[((flushed & (kOutRing - 1)) >> 2) + lane];
      __builtin_memcpy(out + flushed + 4 * lane, &v, 4);
      flushed += 256;
    } while (opos - flushed >= 256);
  }
  __device__ __forceinline__ void flush_rest() {
    wave_lds_fence();
    const uint8_t* ring = reinterpret_cast<const uint8_t*>(w->out);
    for (u32 p = flushed + lane; p < opos; p += kWave) out[p] = ring[p & (kOutRing - 1)];
    flushed = opos;
  }
  // `len` bytes from `dist` back.  The source of an overlapping match is periodic with period dist: every lane reads a byte that
  // existed before the match began.
  __device__ __forceinline__ bool copy_match(u32 len, u32 dist) {
    if (dist > opos || opos + len > isize) return false;
    uint8_t* ring = reinterpret_cast<uint8_t*>(w->out);
    const bool plain = dist >= len;
    if (dist <= kNear) {
      wave_lds_fence();
      for (u32 done = 0; done < len; done += kWave) {
        const u32 i = done + lane;
        if (i < len) {
          const u32 off = plain ? i : i % dist;
          ring[(opos + i) & (kOutRing - 1)] = ring[(opos - dist + off) & (kOutRing - 1)];
        }
      }
    } else {                                                 // from the block's own output in global memory: flushed long ago
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ... and the flush stores have landed
      for (u32 done = 0; done < len; done += kWave) {
        const u32 i = done + lane;
        if (i < len) {
          const u32 off = plain ? i : i % dist;
          ring[(opos + i) & (kOutRing - 1)] = __builtin_nontemporal_load(out + (opos - dist + off));
        }
      }
    }
    opos += len;
    flush_full();
    return true;
  }

  // the code lengths of a dynamic block -> tables.  Behind the 3 block bits.
  __device__ __forceinline__ bool parse_dynamic() {
    refill();
    const u32 hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t* cll = w->lens + 320;
    if (lane < 19) cll[lane] = 0;
    wave_lds_fence();
    refill();
    for (u32 i = 0; i < hclen; ++i) {
      if (cnt < 3) refill();
      const u32 v = take(3);
      if (lane == 0) cll[kClOrder[i]] = (uint8_t)v;
    }
    wave_lds_fence();
    {                                                        // the code-length code: 19 symbols of at most 7 bits, built serially
      u32 count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const u32 mine = lane < 19 ? cll[lane] : 0;
#pragma unroll
      for (int l = 1; l < 8; ++l) count[l] = (u32)__popcll(__ballot(mine == (u32)l));
      int left = 1;
#pragma unroll
      for (int l = 1; l < 8; ++l) {
        left = 2 * left - (int)count[l];
        if (left < 0) return false;
      }
      if (left > 0) return false;                            // zlib: an incomplete code-length code is always an error
      u32 next[8], code = 0;
      next[0] = 0;
#pragma unroll
      for (int l = 1; l < 8; ++l) { code = (code + (l > 1 ? count[l - 1] : 0u)) << 1; next[l] = code; }
      // rank among the symbols of the same length, in symbol order
      const u64 lt = (1ull << lane) - 1;
      u32 rk = 0, nx = 0;
#pragma unroll
      for (int l = 1; l < 8; ++l) {
        const u64 m = __ballot(mine == (u32)l);
        if (mine == (u32)l) { rk = (u32)__popcll(m & lt); nx = next[l]; }
      }
      if (mine) {
        const u32 r = __brev(nx + rk) >> (32 - mine);
        for (u32 i = r; i < 128; i += 1u << mine) w->cl[i] = (uint16_t)((lane << 8) | mine);
      }
      wave_lds_fence();
    }
    const u32 total = hlit + hdist;
    u32 i = 0, prev = 0;
    while (i < total) {
      refill();
      const u32 e = uni(w->cl[buf & 127]);
      drop(e & 0xFF);
      const u32 sym = e >> 8;
      if (sym < 16) {
        if (lane == 0) w->lens[i] = (uint8_t)sym;
        prev = sym;
        ++i;
        continue;
      }
      u32 rep, val = 0;
      if (sym == 16) {
        if (i == 0) return false;
        val = prev;
        rep = 3 + take(2);
      } else if (sym == 17) rep = 3 + take(3);
      else rep = 11 + take(7);
      if (i + rep > total) return false;
      if (lane < rep) w->lens[i + lane] = (uint8_t)val;
      if (lane + 64 < rep) w->lens[i + lane + 64] = (uint8_t)val;
      if (lane + 128 < rep) w->lens[i + lane + 128] = (uint8_t)val;
      prev = val;
      i += rep;
    }
    wave_lds_fence();
    if (uni(w->lens[256]) == 0) return false;                // no end-of-block code
    if (!build_tables(w->lens, hlit, kLitRoot, false, w->lit, w->cnt_lit, w->perm_lit, lane)) return false;
    if (!build_tables(w->lens + hlit, hdist, kDistRoot, true, w->dist, w->cnt_dist, w->perm_dist, lane)) return false;
    return true;
  }
  __device__ __forceinline__ bool fixed_tables() {
    for (u32 s = lane; s < 288; s += kWave) w->lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
    if (lane < 32) w->lens[288 + lane] = 5;
    wave_lds_fence();
    if (!build_tables(w->lens, 288, kLitRoot, false, w->lit, w->cnt_lit, w->perm_lit, lane)) return false;
    return build_tables(w->lens + 288, 32, kDistRoot, true, w->dist, w->cnt_dist, w->perm_dist, lane);
  }

  // a stored block behind its 3 header bits: to the byte boundary, LEN / NLEN, LEN bytes as they are
  __device__ __forceinline__ bool stored() {
    drop(cnt & 7);
    if (cnt < 32) refill();
    const u32 len = take(16), nlen = take(16);
    if ((len ^ nlen) != 0xFFFFu) return false;
    const u32 p = bitpos() >> 3;                             // (a byte boundary)
    if (p + len > clen || opos + len > isize) return false;
    uint8_t* ring = reinterpret_cast<uint8_t*>(w->out);
    const u32 base = opos;
    for (u32 done = 0; done < len; done += kWave) {
      const u32 i = done + lane;
      if (i < len) ring[(base + i) & (kOutRing - 1)] = comp[p + i];
      opos = base + (len - done < kWave ? len : done + kWave);
      flush_full();
    }
    in_start(p + len);
    return true;
  }

  // the symbols of a Huffman block up to its end-of-block code
  __device__ __forceinline__ bool symbols() {
    const u32 lim = clen * 8;
    for (;;) {
      if (cnt < 48) refill();
      if (bitpos() > lim) return false;                      // ran past the block's last byte
      u32 e = uni(w->lit[buf & ((1u << kLitRoot) - 1)]);
      u32 kind = (e >> 12) & 7u;
      if (kind == K_LONG) {
        u32 nb = 0;
        const u32 sym = slow_symbol(w->cnt_lit, w->perm_lit, buf, &nb);
        if (sym == 0xFFFFu) return false;
        e = lit_entry(sym, nb);
        kind = (e >> 12) & 7u;
      }
      if (kind == K_LIT) {
        if (opos >= isize) return false;
        drop(e & 0xFFu);
        if (lane == 0) reinterpret_cast<uint8_t*>(w->out)[opos & (kOutRing - 1)] = (uint8_t)(e >> 16);
        ++opos;
        flush_full();
        continue;
      }
      if (kind == K_EOB) { drop(e & 0xFFu); return true; }
      if (kind != K_BASE) return false;
      const u32 lf = (e >> 8) & 15u, lb = e & 0xFFu;
      const u32 len = (e >> 16) + ((u32)(buf >> lf) & ((1u << (lb - lf)) - 1));
      drop(lb);
      u32 d = uni(w->dist[buf & ((1u << kDistRoot) - 1)]);
      u32 dk = (d >> 12) & 7u;
      if (dk == K_LONG) {
        u32 nb = 0;
        const u32 sym = slow_symbol(w->cnt_dist, w->perm_dist, buf, &nb);
        if (sym == 0xFFFFu) return false;
        d = dist_entry(sym, nb);
        dk = (d >> 12) & 7u;
      }
      if (dk != K_BASE) return false;
      const u32 df = (d >> 8) & 15u, db = d & 0xFFu;
      const u32 dist = (d >> 16) + ((u32)(buf >> df) & ((1u << (db - df)) - 1));
      drop(db);
      if (!copy_match(len, dist)) return false;
    }
  }

  // 0: the block is what its descriptor says; 1: it is not
  __device__ __forceinline__ u32 run() {
    in_start(0);
    opos = flushed = 0;
    for (;;) {
      refill();
      if (bitpos() + 3 > clen * 8) return 1;
      const u32 last = take(1), type = take(2);
      bool ok;
      if (type == 0) ok = stored();
      else if (type == 1) ok = fixed_tables() && symbols();
      else if (type == 2) ok = parse_dynamic() && symbols();
      else ok = false;
      if (!ok) return 1;
      if (last) break;
    }
    if (opos != isize || ((bitpos() + 7) >> 3) != clen) return 1;
    flush_rest();
    return 0;
  }
};

}  // namespace

__global__ void __launch_bounds__(kInfThreads)
ibu_k_inflate_blocks(const uint8_t* __restrict__ comp, const InflateBlockDesc* __restrict__ blocks, u32 nblocks, uint8_t* __restrict__ out_base,
                     u32* __restrict__ status, u32* __restrict__ first_bad, Crc32Pow pw) {
  __shared__ InfWave lds[kInfWaves];
  __shared__ u32 crc_tab[256];
  {
    u32 c = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
    crc_tab[threadIdx.x] = c;
  }
  __syncthreads();
  const u32 lane = threadIdx.x & (kWave - 1), wib = uni(threadIdx.x >> 6);
  const u32 nwaves = gridDim.x * kInfWaves;
  for (u32 b = blockIdx.x * kInfWaves + wib; b < nblocks; b += nwaves) {
    const InflateBlockDesc bd = blocks[b];
    Inflate s;
    s.w = &lds[wib];
    s.comp = comp + bd.coff;
    s.out = out_base + bd.ooff;
    s.lane = lane;
    s.clen = bd.clen;
    s.isize = bd.isize;
    u32 st = bd.isize > 65536u ? 1u : s.run();
    if (st == 0 && bd.isize) {                               // CRC-32 of what was written
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const u32 chunk = (bd.isize + kWave - 1) / kWave;
      const u32 a = lane * chunk < bd.isize ? lane * chunk : bd.isize, e = a + chunk < bd.isize ? a + chunk : bd.isize;
      u32 crc = 0xFFFFFFFFu;
      for (u32 p = a; p < e; ++p) crc = crc_tab[(crc ^ __builtin_nontemporal_load(s.out + p)) & 255u] ^ (crc >> 8);
      crc ^= 0xFFFFFFFFu;
      u32 v = e > a ? multmodp(x8nmodp(pw, bd.isize - e), crc) : 0u;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v ^= __shfl_xor(v, m);
      if (v != bd.crc) st = 2;
    } else if (st == 0 && bd.crc != 0) {
      st = 2;
    }
    if (lane == 0) {
      status[b] = st;
      if (st) atomicMin(first_bad, b);
    }
  }
}

hipError_t launch_inflate_blocks(const LaunchCfg& cfg, const void* d_comp, const InflateBlockDesc* d_blocks, size_t nblocks, void* d_out_base,
                                 uint32_t* d_status, uint32_t* d_first_bad, hipStream_t st) {
  (void)hipGetLastError();
  if (nblocks == 0) return hipSuccess;
  if (nblocks >= (1ull << 31)) return hipErrorInvalidValue;
  static const Crc32Pow pw = [] {
    Crc32Pow t;
    auto mul = [](u32 a, u32 b) {
      u32 m = 1u << 31, p = 0;
      for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
      }
      return p;
    };
    u32 p = 1u << 30;
    t.x2n[0] = p;
    for (int n = 1; n < 32; ++n) t.x2n[n] = p = mul(p, p);
    return t;
  }();
  const u32 want = (u32)((nblocks + kInfWaves - 1) / kInfWaves);
  const u32 cap = (u32)cfg.cus * 3;                          // 47 KB of LDS per workgroup: three fit a CU
  hipLaunchKernelGGL(ibu_k_inflate_blocks, dim3(want < cap ? want : cap), dim3(kInfThreads), 0, st, (const uint8_t*)d_comp, d_blocks, (u32)nblocks,
                     (uint8_t*)d_out_base, d_status, d_first_bad, pw);
  return hipGetLastError();
}

}  // namespace ibu
