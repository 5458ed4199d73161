// k_inflate.hip — DEFLATE blocks (RFC 1951) inflated ON THE DEVICE: the device half of BGZF ingest (the reference reads gzip through
// niffler, src/io/reader.rs:345-352; a bgzip file is a chain of independent members of at most 64 KiB whose sizes stand in their
// headers, so the COMPRESSED bytes can cross the PCIe link — half of them for a records file — and the blocks inflate side by side).
// Launcher: launch_inflate_blocks (kernels.h); C ABI: ibu_inflate_blocks_device (device.cpp); the host walk over the block headers:
// ibu_bgzf_scan (host_io.cpp); the library's user: ibu_load_bgzf_*_to_device (stream.cpp).  The host decoder with the same
// acceptance rules: pgzip.cpp.
//
// ONE LANE PER BLOCK, 64 blocks per wave.  Inflating is sequential inside a block, and a CU issues about one instruction per cycle
// whatever the instruction does: a wave that decodes ONE block with wave-uniform state (the first form: 2^10-entry tables and both
// rings in LDS, 12 blocks in flight per CU) spends that rate on one symbol at a time — 16.7 ms per block, 12 GB/s for the chip, the
// same on the vector and on the scalar unit.  With a block per lane an instruction advances up to 64 symbols.  What makes that fit:
//   - no lookup tables.  A canonical code lives in registers: per code length the number of codes, the first code and the position of
//     its first symbol in code order (10 + 16 + 10 bits, packed: 18 registers a code).  A window of L bits is a code of length L exactly
//     when first[L] <= window < first[L] + count[L] — in whatever order the lengths are tried — so the likeliest lengths (8, 9, 7, 10
//     for a literal of a records file) are tested first, the first four without a branch (decode_index).  The symbols in code order
//     are the only table: 288 + 32 bytes and a 288-bit flag word (the ninth bit) per lane, one read per symbol;
//   - the compressed bytes come a dword at a time from global memory, always one dword ahead (its 64-byte line stays in L1 / L2 for
//     the 15 reads that follow);
//   - the output goes through a 256-byte ring per lane in LDS: literals and matches of up to 64 bytes back never touch global
//     memory; whole 32-byte pieces leave the ring as 8 dword stores, for all lanes together once one of them holds 96 bytes; a match
//     from further back reads the lane's own earlier output from global memory (a wave's vector memory operations execute in order:
//     the byte a lane stored is the byte it loads);
//   - lanes diverge between "literal" and "match": the lanes first decode literals only, parking the length symbol they meet, and
//     serve the parked symbols together (symbols()); the error checks of a run of literals are made once, behind it;
//   - a lane's LDS sits lane after lane at an odd dword stride (a byte's address is base + offset, the lanes' same byte in different
//     banks); what is indexed by symbol is lane-interleaved (element k of lane L at k * 64 + L).
// TWO FORMS of the same kernel (template TL).  TL: the symbol orders in LDS too (47 KB a wave, three waves per CU) — the shortest wave
// (~45 ms for its 64 blocks), taken when one round of those waves holds the whole call (49 152 blocks).  Else: symbol orders and the
// header's code lengths in the caller's scratch, 20 KB of LDS a wave, EIGHT waves per CU, two to a SIMD — a wave takes ~72 ms there,
// but 2048 run at once: the highest rate.  A wave's step is a chain of ~330 dependent instructions in ~2840 cycles with one wave per
// SIMD (SQ counters, r05_az: 170 vector, 130 scalar — the bookkeeping of divergent branches —, 34 branch, 11 LDS, 3 global): what a
// CU needs is waves to interleave, not lanes (five blocks per wave take as long as 64).
// A launch may run AHEAD of the copies that bring its input (`ready`, below): its waves wait for their blocks to arrive.
// When the 64 lanes of a wave have finished, the wave checks the CRC-32 of each of their blocks together: every lane takes 1/64 of a
// block (16 bytes per load), the partial values are combined with x^(8 n) mod P (the identity crc32_combine uses).
// A block is accepted exactly as the host decoder accepts it (pgzip.cpp, RawInflater::inflate): the final deflate block ends on
// the block's last compressed byte, the output has the announced length, the CRC matches; any invalid code, distance or size
// makes the block bad (status 1), a wrong CRC status 2.  Every loop consumes input or produces output and both are bounded by
// the descriptor, so a lane leaves any input — random bytes included — after at most 8 x comp_len + a few iterations.
// Measured (profiles/README.md, r05_ad ... r05_bk; BGZF level 1 of 16/12 records, ratio 0.50): 1e8 records (575 waves, TL) 0.045 s =
// 53 GB/s of records; 3e8 (1724 waves at once) 0.0716 s = 100 GB/s = 4.2 G records/s; the host's 16 inflate threads: 9.6 GB/s.
// On the way, at 1e8 records: bytes straight to global memory, counts in LDS 0.231 s; counts in registers, input one dword ahead, the
// ring 0.124; the CRC pass 16 bytes per load (it was 25 of a wave's 134 ms) 0.109; literal runs batched 0.068; lengths tried likeliest
// first 0.055; symbol orders in LDS for calls of one round 0.046.  The streams do not use the decoder: a ring slot holds a few hundred
// blocks — a handful of waves for 45 ms.
#include "kcommon.hpp"
#include "kernels.h"

namespace ibu {
namespace {

constexpr int kInfThreads = kWave;                           // one wave per workgroup: its LDS is the 64 lanes' tables
constexpr u32 kDistSyms = 32;

// Per-wave tables, lane-interleaved: in LDS behind InfLds (TL) or one per workgroup of the grid in the caller's scratch.
struct InfTables {
  uint8_t sym_lo[288 * kWave];           // literal/length symbols in code order, low 8 bits: [k * 64 + lane]
  u32 sym_hi[9 * kWave];                 // bit k of the lane's 288 bits: symbol k of the order is >= 256
  uint8_t sym_dist[kDistSyms * kWave];   // distance symbols (and the code-length code's) in code order
  uint8_t lens[320 * kWave];             // (the form with its tables in scratch) the code lengths of the header being read
};
// A lane's LDS: [0, 256) the output ring, then construct()'s position per length (u16[16]).  TL (the form for calls of one round, its
// symbol orders in LDS too): the code lengths of a header being read share the lane's bytes [0, 320) with the ring, the positions sit
// at [320, 352) — 89 dwords per lane.  Else the code lengths live in the scratch beside the symbol orders and a lane needs 73 dwords:
// 18.7 KB per wave, EIGHT waves per CU, two to a SIMD (r05_az: one wave per SIMD waits 8.6 cycles from one instruction to the next).
// Both strides are odd: the lanes' same byte sits in different banks, and a byte's address is base + offset.
template <bool TL> struct LaneLds { static constexpr u32 kDwords = TL ? 89 : 73, kOffs = TL ? 320 : 256; };
template <bool TL>
struct InfLds {
  u32 ring[LaneLds<TL>::kDwords * kWave];
  uint16_t base[29 + 30];                // length / distance bases, then their extra bits (shared by the lanes)
  uint8_t extra[29 + 30];
  uint8_t clorder[19];
  u32 crc_tab[256];
};
constexpr u32 kRing = 256, kNear = 64, kPiece = 32, kChunk = 64, kHigh = 96;

__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Crc32Pow { u32 x2n[32]; };                            // x^(2^k) mod P, reflected (the launcher fills it)

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

// One lane's decoder.
template <bool TL>
struct LaneInflate {
  InfLds<TL>* w;
  InfTables* t;
  uint8_t* rb;                                               // this lane's bytes of w->ring
  u32 lane, park_steps;
  const uint8_t* comp;                                       // the block's compressed bytes (readable 2 KiB past clen)
  uint8_t* out;
  u32 clen, isize;
  u64 buf;                                                   // cnt valid low bits
  u32 cnt, ipos, nxt;                                        // nxt: the dword at comp + ipos, already loaded
  u32 opos, flushed, rbase;                                  // ring: bytes [max(rbase, opos - 256 + pending), opos) are in it; [flushed, opos) not yet in global memory
  // a canonical code in registers: per length 1 .. 15 the number of codes and the position of its first symbol in code order (10 bits
  // each, three to a word) and its first code (16 bits, two to a word)
  struct Code { u32 cnt[5], off[5], first[8]; };
  Code cl, cd;                                               // literal/length and distance code

  // ---- LDS columns ----
  __device__ __forceinline__ uint8_t* ring8(u32 b) const { return rb + b; }
  __device__ __forceinline__ uint8_t& lens(u32 s) const { return TL ? *ring8(s) : t->lens[s * kWave + lane]; }
  __device__ __forceinline__ uint16_t& offs(u32 l) const { return *reinterpret_cast<uint16_t*>(ring8(LaneLds<TL>::kOffs + 2 * l)); }
  __device__ __forceinline__ u32 lit_symbol(u32 k) const {
    return (u32)t->sym_lo[k * kWave + lane] | (((t->sym_hi[(k >> 5) * kWave + lane] >> (k & 31)) & 1u) << 8);
  }

  // ---- input ----
  __device__ __forceinline__ u32 load32(u32 at) const {
    u32 v;
    __builtin_memcpy(&v, comp + at, 4);
    return v;
  }
  __device__ __forceinline__ void in_start(u32 p) { buf = 0; cnt = 0; ipos = p; nxt = load32(p); }
  __device__ __forceinline__ void need(u32 n) {              // n <= 32
    if (cnt < n) {
      buf |= (u64)nxt << cnt;
      cnt += 32;
      ipos += 4;
      nxt = load32(ipos);                                    // wanted ~4 symbols from now
    }
  }
  __device__ __forceinline__ u32 bits(u32 n) {
    need(n);
    const u32 v = (u32)(buf & ((1ull << n) - 1));
    buf >>= n;
    cnt -= n;
    return v;
  }
  __device__ __forceinline__ u32 bitpos() const { return ipos * 8 - cnt; }

  // The position in code order of the next code: no memory, no loop-carried state.  A window of L bits is a code of length L exactly when
  // first[L] <= window < first[L] + count[L] — whatever order the lengths are tried in (a shorter code's longer windows lie below the
  // longer lengths' ranges, a longer code's prefixes above the shorter ones').  So the likely lengths go first: a literal of a records
  // file is 8 or 9 bits, and the wave leaves after the slowest lane's hit — ~3 tries instead of the ~12 steps of counting up from 1.
  // LIT: the order for literal/length codes; else for distance (and code-length) codes.  -1: no such code.
  template <bool LIT>
  __device__ __forceinline__ int decode_index(const Code& c) {
    need(15);
    const u32 x = __brev((u32)buf) >> 17;                    // the next 15 bits, first bit on top
    constexpr int kOrderLit[15] = {8, 9, 7, 10, 6, 11, 5, 12, 4, 13, 3, 14, 2, 15, 1};
    constexpr int kOrderDist[15] = {5, 4, 6, 3, 7, 2, 8, 1, 9, 10, 11, 12, 13, 14, 15};
    // the four likeliest lengths without a branch (exactly one length can hit: the selects do not fight), the rest one by one
    u32 hit_len = 0, hit_idx = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int len = LIT ? kOrderLit[i] : kOrderDist[i];
      const u32 cn = (c.cnt[(len - 1) / 3] >> (10 * ((len - 1) % 3))) & 1023u;
      const u32 fi = (c.first[(len - 1) / 2] >> (16 * ((len - 1) % 2))) & 0xFFFFu;
      const u32 d = (x >> (15 - len)) - fi;
      const bool h = d < cn;                                 // (unsigned: a window below `first` wraps to a huge d)
      hit_len = h ? (u32)len : hit_len;
      hit_idx = h ? ((c.off[(len - 1) / 3] >> (10 * ((len - 1) % 3))) & 1023u) + d : hit_idx;
    }
    if (hit_len == 0) {
#pragma unroll
      for (int i = 4; i < 15; ++i) {
        const int len = LIT ? kOrderLit[i] : kOrderDist[i];
        const u32 cn = (c.cnt[(len - 1) / 3] >> (10 * ((len - 1) % 3))) & 1023u;
        const u32 fi = (c.first[(len - 1) / 2] >> (16 * ((len - 1) % 2))) & 0xFFFFu;
        const u32 d = (x >> (15 - len)) - fi;
        if (d < cn) {
          hit_len = (u32)len;
          hit_idx = ((c.off[(len - 1) / 3] >> (10 * ((len - 1) % 3))) & 1023u) + d;
          break;
        }
      }
      if (hit_len == 0) return -1;
    }
    buf >>= hit_len;
    cnt -= hit_len;
    return (int)hit_idx;
  }

  // Counts, first codes and first positions (packed into cc) and symbol order of the code with lengths lens[at, at + n).  How complete it is: 0 complete, > 0 codes left
  // over, < 0 over-subscribed; *maxlen = the longest code.  store(k, s): symbol s is the k-th in code order.
  template <class Store>
  __device__ __forceinline__ int construct(Code& cc, u32 at, u32 n, u32* maxlen, Store store) {
    for (u32 l = 0; l < 16; ++l) offs(l) = 0;
    for (u32 s = 0; s < n; ++s) offs(lens(at + s))++;
    int left = 1;
    u32 ml = 0, o = 0, code = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) cc.cnt[k] = cc.off[k] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cc.first[k] = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
      const u32 c = offs(l);
      cc.cnt[(l - 1) / 3] |= c << (10 * ((l - 1) % 3));
      cc.off[(l - 1) / 3] |= (o & 1023u) << (10 * ((l - 1) % 3));
      cc.first[(l - 1) / 2] |= (code & 0xFFFFu) << (16 * ((l - 1) % 2));   // (a valid code set keeps it below 2^l)
      code = (code + c) << 1;
      if (c) ml = l;
      if (left >= 0) left = 2 * left - (int)c;              // (stays negative once over-subscribed)
      offs(l) = (uint16_t)o;
      o += c;
    }
    *maxlen = ml;
    if (left < 0) return left;
    for (u32 s = 0; s < n; ++s) {
      const u32 l = lens(at + s);
      if (l) {
        const u32 k = offs(l);
        offs(l) = (uint16_t)(k + 1);
        store(k, s);
      }
    }
    return left;
  }
  // the host decoder's acceptance of a code set (pgzip.cpp, build_table)
  __device__ __forceinline__ bool accept(int left, u32 maxlen, bool is_dist) const {
    if (maxlen == 0) return is_dist;                         // no codes at all: literals only (allowed for distances)
    if (left < 0) return false;                              // over-subscribed
    return left == 0 || maxlen == 1;                         // incomplete: only a single 1-bit code
  }
  __device__ __forceinline__ bool build_lit(u32 at, u32 n) {
    for (u32 k = 0; k < 9; ++k) t->sym_hi[k * kWave + lane] = 0;
    u32 ml = 0;
    const int left = construct(cl, at, n, &ml, [&](u32 k, u32 s) {
      t->sym_lo[k * kWave + lane] = (uint8_t)s;
      if (s >= 256) t->sym_hi[(k >> 5) * kWave + lane] |= 1u << (k & 31);
    });
    return accept(left, ml, false);
  }
  __device__ __forceinline__ bool build_dist(u32 at, u32 n) {
    u32 ml = 0;
    const int left = construct(cd, at, n, &ml, [&](u32 k, u32 s) { t->sym_dist[k * kWave + lane] = (uint8_t)s; });
    return accept(left, ml, true);
  }

  // ---- output ----
  // Whole 32-byte pieces of the ring -> global memory, for ALL the lanes that are here together as soon as ONE of them holds 96 bytes:
  // the flush is 8 LDS reads and 8 stores behind a branch, and taken lane by lane (each at its own 64th byte) some lane took it at
  // nearly every step of the wave.  Pending bytes stay below 96 + 64 (a match's chunk), 64 bytes of history behind them: the ring's 256.
  __device__ __forceinline__ void flush_if_due() {
    if (__ballot(opos - flushed >= kHigh) == 0) return;
    while (opos - flushed >= kPiece) {
      const u32* src = reinterpret_cast<const u32*>(rb + (flushed & (kRing - 1)));   // (flushed is a multiple of 32)
#pragma unroll
      for (u32 d = 0; d < kPiece / 4; ++d) {
        const u32 v = src[d];
        __builtin_memcpy(out + flushed + 4 * d, &v, 4);
      }
      flushed += kPiece;
    }
  }
  __device__ __forceinline__ void put_pending() {            // the bytes behind the last whole piece -> global memory (they stay pending)
    for (u32 p = flushed; p < opos; ++p) out[p] = *ring8(p & (kRing - 1));
  }
  __device__ __forceinline__ void get_pending() {            // ... and back into the ring (the header's code lengths were parsed over them)
    for (u32 p = flushed; p < opos; ++p) *ring8(p & (kRing - 1)) = out[p];
    rbase = opos;                                            // nothing older is in the ring any more
  }
  __device__ __forceinline__ bool copy_match(u32 len, u32 dist) {
    if (dist > opos || opos + len > isize) return false;
    const bool near = dist <= kNear && dist <= opos - rbase;
    while (len) {                                            // at most 64 bytes, then (if some lane is due) the whole pieces leave
      const u32 n = len < kChunk ? len : kChunk;
      if (near) {
        for (u32 i = 0; i < n; ++i) *ring8((opos + i) & (kRing - 1)) = *ring8((opos + i - dist) & (kRing - 1));
      } else {                                               // further back than the ring is good for: from global memory what has left it
        for (u32 i = 0; i < n; ++i) {
          const u32 src = opos + i - dist;
          *ring8((opos + i) & (kRing - 1)) = src < flushed ? out[src] : *ring8(src & (kRing - 1));
        }
      }
      opos += n;
      len -= n;
      flush_if_due();
    }
    return true;
  }

  __device__ __forceinline__ bool parse_dynamic() {
    const u32 hlit = bits(5) + 257, hdist = bits(5) + 1, hclen = bits(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    if (TL) put_pending();                                   // the code lengths are parsed in the ring's LDS
    for (u32 i = 0; i < 19; ++i) lens(i) = 0;
    for (u32 i = 0; i < hclen; ++i) lens(w->clorder[i]) = (uint8_t)bits(3);
    {                                                        // the code-length code (in the distance code's place): an incomplete one is always an error
      u32 ml = 0;
      if (construct(cd, 0, 19, &ml, [&](u32 k, u32 s) { t->sym_dist[k * kWave + lane] = (uint8_t)s; }) != 0) return false;
    }
    const u32 total = hlit + hdist;
    u32 i = 0, prev = 0;
    while (i < total) {
      const int k = decode_index<false>(cd);
      if (k < 0) return false;
      const u32 sym = t->sym_dist[(u32)k * kWave + lane];
      if (sym < 16) {
        lens(i) = (uint8_t)sym;
        prev = sym;
        ++i;
        continue;
      }
      u32 rep, val = 0;
      if (sym == 16) {
        if (i == 0) return false;
        val = prev;
        rep = 3 + bits(2);
      } else if (sym == 17) rep = 3 + bits(3);
      else rep = 11 + bits(7);
      if (i + rep > total) return false;
      for (u32 k2 = 0; k2 < rep; ++k2) lens(i + k2) = (uint8_t)val;
      prev = val;
      i += rep;
    }
    if (lens(256) == 0) return false;                        // no end-of-block code
    const bool ok = build_lit(0, hlit) && build_dist(hlit, hdist);
    if (TL) get_pending();
    return ok;
  }
  __device__ __forceinline__ bool fixed_tables() {
    if (TL) put_pending();
    for (u32 s = 0; s < 288; ++s) lens(s) = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
    for (u32 s = 0; s < 32; ++s) lens(288 + s) = 5;
    const bool ok = build_lit(0, 288) && build_dist(288, 32);
    if (TL) get_pending();
    return ok;
  }
  // a stored block behind its 3 header bits: to the byte boundary, LEN / NLEN, LEN bytes as they are
  __device__ __forceinline__ bool stored() {
    const u32 skip = cnt & 7;                                // (ipos * 8 is a byte boundary: the bits in the buffer say where we are)
    buf >>= skip;
    cnt -= skip;
    const u32 len = bits(16), nlen = bits(16);
    if ((len ^ nlen) != 0xFFFFu) return false;
    const u32 p = bitpos() >> 3;
    if (p + len > clen || opos + len > isize) return false;
    for (u32 i = 0; i < len; ++i) {
      *ring8(opos & (kRing - 1)) = comp[p + i];
      ++opos;
      flush_if_due();
    }
    in_start(p + len);
    return true;
  }
  // The symbols of a Huffman block up to its end-of-block code.  Lanes diverge between "literal" and "match": run naively, every step
  // of the wave pays for both.  So the lanes that are here together first decode LITERALS only — a lane that meets a length symbol (or
  // the end) parks it — until all of them are parked (or 6 steps have passed); then the parked symbols are served in one go.  With a
  // match every ~5 symbols a wave then runs the match path every ~5 steps for (nearly) all its lanes, not every step for a fifth.
  __device__ __forceinline__ bool symbols() {
    const u32 lim = clen * 8;
    for (;;) {
      u32 parked = 0xFFFFFFFFu;                              // the non-literal symbol this lane waits with
      bool bad = false;                                      // checked once per run of literals: a lane that has gone wrong writes no further
      for (u32 step = 0; step < park_steps; ++step) {        // byte (the position stops) and reads at most 6 x 2 bytes of the padding
        if (parked == 0xFFFFFFFFu) {
          bad |= bitpos() > lim;                             // ran past the block's last byte
          const int k = decode_index<true>(cl);
          bad |= k < 0;
          const u32 sym = lit_symbol(k < 0 ? 0u : (u32)k);
          if (sym < 256) {
            bad |= opos >= isize;
            *ring8(opos & (kRing - 1)) = (uint8_t)sym;
            opos += bad ? 0u : 1u;
            flush_if_due();
          } else {
            parked = sym;
          }
        }
        if (__ballot(parked == 0xFFFFFFFFu) == 0) break;     // (of the lanes that are in this loop together)
      }
      if (bad) return false;
      if (parked == 0xFFFFFFFFu) continue;
      if (parked == 256) return true;
      const u32 ls = parked - 257;
      if (ls >= 29) return false;
      const u32 len = w->base[ls] + bits(w->extra[ls]);
      const int kd = decode_index<false>(cd);
      if (kd < 0) return false;
      const u32 ds = t->sym_dist[(u32)kd * kWave + lane];
      if (ds >= 30) return false;
      const u32 dist = w->base[29 + ds] + bits(w->extra[29 + ds]);
      if (!copy_match(len, dist)) return false;
    }
  }
  // 0: the block is what its descriptor says; 1: it is not
  __device__ __forceinline__ u32 run() {
    in_start(0);
    opos = flushed = rbase = 0;
    for (;;) {
      if (bitpos() + 3 > clen * 8) return 1;
      const u32 last = bits(1), type = bits(2);
      bool ok;
      if (type == 0) ok = stored();
      else if (type == 1) ok = fixed_tables() && symbols();
      else if (type == 2) ok = parse_dynamic() && symbols();
      else ok = false;
      if (!ok) return 1;
      if (last) break;
    }
    if (opos != isize || ((bitpos() + 7) >> 3) != clen) return 1;
    put_pending();
    return 0;
  }
};

}  // namespace

// TL: the lanes' symbol orders in LDS (47 KB per wave: three waves per CU — the form for inputs that fit one round of them, where the
// symbol's round trip to L2 is a quarter of a wave's step) instead of global scratch (eight waves per CU: the form for large inputs).
template <bool TL>
__global__ void __launch_bounds__(kInfThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))   // (two waves to a SIMD: 256 registers each — the scratch form sits right at that)
ibu_k_inflate_blocks(const uint8_t* __restrict__ comp, const InflateBlockDesc* __restrict__ blocks, u32 nblocks, uint8_t* __restrict__ out_base,
                     u32* __restrict__ status, u32* __restrict__ first_bad, InfTables* __restrict__ tables /*[gridDim.x]*/, u32 park_steps, u32 bpw /*blocks per wave: lanes [0, bpw) decode*/,
                     const unsigned long long* ready /*nullable: compressed bytes that have arrived so far*/, unsigned long long ready_total, Crc32Pow pw) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  InfLds<TL>* w = reinterpret_cast<InfLds<TL>*>(lds_raw);
  const u32 lane = threadIdx.x;
  for (u32 i = lane; i < 29 + 30; i += kWave) {
    w->base[i] = i < 29 ? kLenBase[i] : kDistBase[i - 29];
    w->extra[i] = i < 29 ? kLenExtra[i] : kDistExtra[i - 29];
  }
  if (lane < 19) w->clorder[lane] = kClOrder[lane];
  for (u32 i = lane; i < 256; i += kWave) {
    u32 c = i;
#pragma unroll
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
    w->crc_tab[i] = c;
  }
  __syncthreads();
  const u32 ngroups = (nblocks + bpw - 1) / bpw;
  for (u32 g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const u32 b = g * bpw + lane;
    const bool live = lane < bpw && b < nblocks;
    InflateBlockDesc bd;
    bd.coff = 0; bd.ooff = 0; bd.clen = 0; bd.isize = 0; bd.crc = 0; bd.reserved = 0;
    if (live) bd = blocks[b];
    u32 st = 0;
    bool arrived = true;
    if (ready) {
      // The launch may run AHEAD of the copies that bring its input (ibu_load_bgzf_*_to_device, large files): `*ready` = how many
      // compressed bytes are on the device so far, written by the copy stream behind every piece.  A wave waits — all 64 lanes for the
      // last of their blocks, plus the 4 KiB a lane may read (not use) past its block, so that no line is cached before it is final —
      // looking at a word in PINNED HOST memory that the host's thread writes when it has seen a piece's copy complete (a word in device
      // memory written by the copy stream never changed for a running kernel: ordinary device memory is coherent with the copy engine
      // at kernel boundaries only, and in fine-grained device memory the 8-byte copies themselves waited for the kernel to end), asleep
      // 14 ... 220 us between looks (each one a read over the link), and for ~4 s at most: whatever happens to the host, every wave ends
      // (status 3: its input never came — the caller inflates what such waves left once everything has arrived: a slow disk is no error).
      unsigned long long need = live ? bd.coff + bd.clen + 4096ull : 0ull;
      if (need > ready_total) need = ready_total;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        const u32 lo = (u32)__shfl_xor((int)(u32)need, m), hi = (u32)__shfl_xor((int)(u32)(need >> 32), m);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        need = o > need ? o : need;
      }
      u32 naps = 0, nap = 4;                               // (a nap: 127 x 64 cycles, ~3.4 us)
      while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
        if (naps > 1200000u) { arrived = false; break; }   // ~4 s asleep
        for (u32 k = 0; k < nap; ++k) __builtin_amdgcn_s_sleep(127);
        naps += nap;
        if (nap < 64) nap *= 2;                            // back off to a look every ~220 us: 2048 waves looking every 27 us halved the copies' rate
      }
      if (!arrived && live) st = 3;
    }
    if (live && arrived) {
      LaneInflate<TL> s;
      s.w = w;
      s.t = TL ? reinterpret_cast<InfTables*>(lds_raw + ((sizeof(InfLds<TL>) + 15) & ~(size_t)15)) : tables + blockIdx.x;
      s.rb = reinterpret_cast<uint8_t*>(w->ring + lane * LaneLds<TL>::kDwords);
      s.lane = lane;
      s.park_steps = park_steps;
      s.comp = comp + bd.coff;
      s.out = out_base + bd.ooff;
      s.clen = bd.clen;
      s.isize = bd.isize;
      st = bd.isize > 65536u ? 1u : s.run();
    }
    // the CRC-32 of what each lane wrote, block by block, all lanes on one block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 todo = __ballot(live && st == 0);
    for (u32 j = 0; j < kWave; ++j) {
      if (!((todo >> j) & 1ull)) continue;                   // wave-uniform
      const u32 isz = (u32)__shfl((int)bd.isize, (int)j);
      const u32 want = (u32)__shfl((int)bd.crc, (int)j);
      const u32 lo = (u32)__shfl((int)(u32)bd.ooff, (int)j), hi = (u32)__shfl((int)(u32)((u64)bd.ooff >> 32), (int)j);
      const uint8_t* o = out_base + (int64_t)(((u64)hi << 32) | lo);
      u32 v = 0;
      if (isz) {
        const u32 chunk = (isz + kWave - 1) / kWave;
        const u32 a = lane * chunk < isz ? lane * chunk : isz, e = a + chunk < isz ? a + chunk : isz;
        u32 crc = 0xFFFFFFFFu;
        // 16 bytes per load, the next load in flight while these go through the table (a byte per load made this pass 98 of a wave's
        // 134 ms: 64 blocks x 1024 dependent trips to L2)
        u32 p = a;
        u32x4 cur = {0, 0, 0, 0}, nx = {0, 0, 0, 0};
        if (p + 16 <= e) __builtin_memcpy(&cur, o + p, 16);
        while (p + 16 <= e) {
          if (p + 32 <= e) __builtin_memcpy(&nx, o + p + 16, 16);
          const u32 ws[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            u32 x = ws[q];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              crc = w->crc_tab[(crc ^ x) & 255u] ^ (crc >> 8);
              x >>= 8;
            }
          }
          cur = nx;
          p += 16;
        }
        for (; p < e; ++p) crc = w->crc_tab[(crc ^ o[p]) & 255u] ^ (crc >> 8);
        crc ^= 0xFFFFFFFFu;
        v = e > a ? multmodp(x8nmodp(pw, isz - e), crc) : 0u;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v ^= (u32)__shfl_xor((int)v, m);
      }
      if (lane == j && v != want) st = 2;
    }
    if (live) {
      status[b] = st;
      if (st) atomicMin(first_bad, b);
    }
  }
}

// Always 64 blocks per wave: a wave's time does not shrink with fewer lanes (3678 blocks dealt five to a wave took 42 ms, 36 766 at 48
// to a wave 46 ms: the step is a chain of dependent instructions, not divergence), and dense waves leave room for the next launch.
static inline u32 inflate_bpw(const LaunchCfg&, size_t) { return kWave; }
// one round of three waves per CU takes everything: the form with the symbol orders in LDS; else eight waves per CU, tables in scratch
static inline bool inflate_tables_in_lds(const LaunchCfg& cfg, size_t nblocks) { return (nblocks + kWave - 1) / kWave <= (size_t)cfg.cus * 3; }
static inline u32 inflate_grid(const LaunchCfg& cfg, size_t nblocks) {
  const u32 bpw = inflate_bpw(cfg, nblocks);
  const size_t want = (nblocks + bpw - 1) / bpw;
  const size_t cap = (size_t)cfg.cus * 8;                    // 20 KB of LDS per wave: eight fit a CU
  return (u32)(want < cap ? want : cap);
}
// form: 0 = by size (one round of three waves per CU holds the call: symbol orders in LDS, the shortest wave; else in scratch, eight
// waves per CU, the highest rate); 2 = in scratch whatever the size (launches that are to run beside each other: 20 KB of LDS a wave)
size_t inflate_scratch_bytes(const LaunchCfg& cfg, size_t nblocks, int form) {
  return form != 2 && inflate_tables_in_lds(cfg, nblocks) ? 16 : sizeof(InfTables) * (size_t)inflate_grid(cfg, nblocks);
}
hipError_t launch_inflate_blocks(const LaunchCfg& cfg, const void* d_comp, const InflateBlockDesc* d_blocks, size_t nblocks, void* d_out_base,
                                 uint32_t* d_status, uint32_t* d_first_bad, void* scratch, size_t scratch_bytes, hipStream_t st, int form,
                                 const uint64_t* d_ready, uint64_t ready_total) {
  (void)hipGetLastError();
  if (nblocks == 0) return hipSuccess;
  if (nblocks >= (1ull << 31) || scratch_bytes < inflate_scratch_bytes(cfg, nblocks, form)) return hipErrorInvalidValue;
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
  const bool tl = form != 2 && inflate_tables_in_lds(cfg, nblocks);
  const size_t lds = tl ? ((sizeof(InfLds<true>) + 15) & ~(size_t)15) + sizeof(InfTables) - sizeof(((InfTables*)nullptr)->lens) : sizeof(InfLds<false>);
  const void* fn = tl ? reinterpret_cast<const void*>(ibu_k_inflate_blocks<true>) : reinterpret_cast<const void*>(ibu_k_inflate_blocks<false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const u32 grid = inflate_grid(cfg, nblocks);
  if (tl)
    hipLaunchKernelGGL(ibu_k_inflate_blocks<true>, dim3(grid), dim3(kInfThreads), lds, st, (const uint8_t*)d_comp, d_blocks, (u32)nblocks,
                       (uint8_t*)d_out_base, d_status, d_first_bad, static_cast<InfTables*>(scratch), 6u, inflate_bpw(cfg, nblocks),
                       reinterpret_cast<const unsigned long long*>(d_ready), (unsigned long long)ready_total, pw);
  else
    hipLaunchKernelGGL(ibu_k_inflate_blocks<false>, dim3(grid), dim3(kInfThreads), lds, st, (const uint8_t*)d_comp, d_blocks, (u32)nblocks,
                       (uint8_t*)d_out_base, d_status, d_first_bad, static_cast<InfTables*>(scratch), 6u, inflate_bpw(cfg, nblocks),
                       reinterpret_cast<const unsigned long long*>(d_ready), (unsigned long long)ready_total, pw);
  return hipGetLastError();
}

}  // namespace ibu
