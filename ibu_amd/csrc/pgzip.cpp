// pgzip.cpp — see pgzip.hpp.  Host code only (no device code in this translation unit).
//
// Deflate (RFC 1951) / gzip (RFC 1952) as zlib's inflate accepts them: over-subscribed code sets are errors, incomplete
// ones too unless the set is a single 1-bit code, a literal/length set needs the end-of-block symbol, length symbols
// 286/287 and distance symbols 30/31 are errors when used, a distance may not reach in front of the start of the output.
// The one check this decoder cannot make in marker mode is the last one ACROSS a member boundary (a new member's back
// reference into the previous member's bytes); the member CRC still covers the bytes that come out.
#include "pgzip.hpp"

#include <errno.h>
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define IBU_PGZ_X86 1
#else
#define IBU_PGZ_X86 0
#endif
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

#include "common.hpp"

namespace ibu {
namespace pgz {
namespace {

constexpr size_t kWin = 32768;
constexpr int kLitBits = 11, kDistBits = 8;
constexpr size_t kLitCap = (1u << kLitBits) + 4800, kDistCap = (1u << kDistBits) + 4096;
constexpr size_t kPad = 512;                         // zero bytes readable behind the last compressed byte (> one dynamic header)
constexpr size_t kNone = ~(size_t)0;
constexpr uint32_t K_LIT = 0, K_BASE = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4;

// Decode-table entry: bits 0-7 the number of stream bits this step consumes (code length, for a length / distance
// symbol PLUS its extra bits, so that one shift consumes both; for a subtable link the primary width), bits 8-11 the
// code length alone (where the extra bits start; for a link: the subtable's width), bits 12-14 the kind, bits 16-31 the
// payload (literal, base length, base distance, subtable offset).
inline uint32_t mk(uint32_t payload, uint32_t f, uint32_t kind, uint32_t nbits) {
  return (payload << 16) | (kind << 12) | (f << 8) | nbits;
}
inline uint32_t e_kind(uint32_t e) { return (e >> 12) & 7u; }
inline uint32_t e_bits(uint32_t e) { return e & 0xFFu; }
inline uint32_t e_f(uint32_t e) { return (e >> 8) & 15u; }
inline uint32_t e_val(uint32_t e) { return e >> 16; }

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t rev_bits(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
  return r;
}
inline uint32_t lit_entry(int sym, int nb) {
  if (sym < 256) return mk((uint32_t)sym, 0, K_LIT, (uint32_t)nb);
  if (sym == 256) return mk(0, 0, K_EOB, (uint32_t)nb);
  if (sym < 286) return mk(kLenBase[sym - 257], (uint32_t)nb, K_BASE, (uint32_t)nb + kLenExtra[sym - 257]);
  return mk(0, 0, K_BAD, (uint32_t)nb);
}
inline uint32_t dist_entry(int sym, int nb) {
  if (sym < 30) return mk(kDistBase[sym], (uint32_t)nb, K_BASE, (uint32_t)nb + kDistExtra[sym]);
  return mk(0, 0, K_BAD, (uint32_t)nb);
}

// Two-level decode table of a canonical Huffman code read LSB first: `P` primary bits, longer codes through
// per-prefix subtables.  false: the code set is one zlib rejects.
bool build_table(const uint8_t* lens, int n, int P, bool is_dist, uint32_t* tab, size_t cap) {
  int count[16] = {0};
  for (int i = 0; i < n; ++i) count[lens[i]]++;
  int maxlen = 15;
  while (maxlen > 0 && count[maxlen] == 0) --maxlen;
  const uint32_t bad = mk(0, 0, K_BAD, 1);
  for (size_t i = 0; i < ((size_t)1 << P); ++i) tab[i] = bad;
  if (maxlen == 0) return is_dist;                   // no codes at all: a block of literals only (zlib allows it for distances)
  int left = 1;
  for (int l = 1; l <= 15; ++l) {
    left <<= 1;
    left -= count[l];
    if (left < 0) return false;                      // over-subscribed
  }
  if (left > 0 && maxlen != 1) return false;         // incomplete (allowed: a single 1-bit code)
  uint32_t next[16];
  uint32_t code = 0;
  for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
  // pass 1: subtable width per primary prefix
  uint8_t sub_bits[1u << kLitBits];
  const bool any_long = maxlen > P;
  if (any_long) {
    memset(sub_bits, 0, (size_t)1 << P);
    uint32_t nx[16];
    memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n; ++s) {
      const int l = lens[s];
      if (l <= P) { if (l) nx[l]++; continue; }
      const uint32_t r = rev_bits(nx[l]++, l), pre = r & (((uint32_t)1 << P) - 1);
      if (l - P > sub_bits[pre]) sub_bits[pre] = (uint8_t)(l - P);
    }
    size_t off = (size_t)1 << P;
    for (uint32_t pre = 0; pre < ((uint32_t)1 << P); ++pre) {
      if (!sub_bits[pre]) continue;
      const size_t sz = (size_t)1 << sub_bits[pre];
      if (off + sz > cap || off > 0xFFFF) return false;  // cannot happen for valid code sets (cap is the worst case)
      tab[pre] = mk((uint32_t)off, sub_bits[pre], K_SUB, (uint32_t)P);
      for (size_t i = 0; i < sz; ++i) tab[off + i] = bad;
      off += sz;
    }
  }
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t r = rev_bits(next[l]++, l);
    if (l <= P) {
      const uint32_t e = is_dist ? dist_entry(s, l) : lit_entry(s, l);
      for (uint32_t i = r; i < ((uint32_t)1 << P); i += (uint32_t)1 << l) tab[i] = e;
    } else {
      const uint32_t pre = r & (((uint32_t)1 << P) - 1), link = tab[pre];
      const uint32_t off = e_val(link), sb = e_f(link), e = is_dist ? dist_entry(s, l - P) : lit_entry(s, l - P);
      for (uint32_t i = r >> P; i < ((uint32_t)1 << sb); i += (uint32_t)1 << (l - P)) tab[off + i] = e;
    }
  }
  return true;
}

struct FixedTables {
  uint32_t lit[kLitCap], dist[kDistCap];
  FixedTables() {
    uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    (void)build_table(l, 288, kLitBits, false, lit, kLitCap);
    uint8_t d[32];
    for (int i = 0; i < 32; ++i) d[i] = 5;
    (void)build_table(d, 32, kDistBits, true, dist, kDistCap);
  }
};
const FixedTables& fixed_tables() {
  static const FixedTables t;
  return t;
}

// ---- bit reader (little-endian host; `in` has kPad readable zero bytes behind its end) ------------------------------
// buf holds cnt valid low bits; bits above cnt are zero or the true upcoming bits (refill ORs the same bits again).
struct Bits {
  const uint8_t* in;
  uint64_t buf = 0;
  uint32_t cnt = 0;
  size_t pos = 0;
  Bits(const uint8_t* p, size_t bitpos) : in(p) {
    pos = bitpos >> 3;
    refill();
    drop((uint32_t)(bitpos & 7));
  }
  inline void refill() {
    uint64_t w;
    memcpy(&w, in + pos, 8);
    buf |= w << cnt;
    pos += (63 - cnt) >> 3;
    cnt |= 56;
  }
  inline void drop(uint32_t k) { buf >>= k; cnt -= k; }
  inline uint32_t take(uint32_t k) { const uint32_t v = (uint32_t)(buf & (((uint64_t)1 << k) - 1)); drop(k); return v; }
  inline size_t bitpos() const { return pos * 8 - cnt; }
};

struct Segment { size_t out_end; uint32_t crc, isize; };  // a gzip member ended after out_end output elements

enum St { S_HEADER, S_BLOCK, S_HUFF, S_STORED_LEN, S_STORED, S_TRAILER, S_DONE };
enum Stop { R_BOUNDARY, R_OUT, R_IN, R_END, R_ERR, R_SWITCH };

struct Inflater {
  std::vector<uint32_t> lit_own, dist_own;
  const uint32_t* lit = nullptr;
  const uint32_t* dist = nullptr;
  St st = S_HEADER;
  bool raw = false;                                  // a bare deflate stream (no gzip header / trailer): ends behind its final block
  bool last_block = false;
  uint32_t stored_left = 0;
  size_t bp = 0;                                     // bit position in the compressed buffer where decoding resumes
  size_t scanned = 0, last_marker = 0;               // marker mode: symbols [0, scanned) were searched; the last marker sits in front of last_marker
  std::vector<Segment> segs;
  void use_own() {
    if (lit_own.empty()) { lit_own.resize(kLitCap); dist_own.resize(kDistCap); }
    lit = lit_own.data();
    dist = dist_own.data();
  }
};

// Dynamic block header behind the 3 block bits.  false: invalid.
bool parse_dynamic(Bits& b, Inflater& s) {
  b.refill();
  const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  uint8_t cl[19] = {0};
  b.refill();
  for (uint32_t i = 0; i < hclen; ++i) {
    if (b.cnt < 3) b.refill();
    cl[kClOrder[i]] = (uint8_t)b.take(3);
  }
  uint32_t cltab[128];
  {
    int count[8] = {0};
    for (int i = 0; i < 19; ++i) count[cl[i]]++;
    int left = 1;
    for (int l = 1; l <= 7; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; }
    if (left > 0) return false;                      // zlib: an incomplete code-length code is always an error
    uint32_t next[8], code = 0;
    for (int l = 1; l <= 7; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    for (int sym = 0; sym < 19; ++sym) {
      const int l = cl[sym];
      if (!l) continue;
      const uint32_t r = rev_bits(next[l]++, l);
      for (uint32_t i = r; i < 128; i += (uint32_t)1 << l) cltab[i] = ((uint32_t)sym << 8) | (uint32_t)l;
    }
  }
  uint8_t lens[286 + 30];
  uint32_t i = 0;
  const uint32_t total = hlit + hdist;
  while (i < total) {
    b.refill();
    const uint32_t e = cltab[b.buf & 127];
    b.drop(e & 0xFF);
    const uint32_t sym = e >> 8;
    if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
    uint32_t rep, val = 0;
    if (sym == 16) {
      if (i == 0) return false;
      val = lens[i - 1];
      rep = 3 + b.take(2);
    } else if (sym == 17) rep = 3 + b.take(3);
    else rep = 11 + b.take(7);
    if (i + rep > total) return false;
    memset(lens + i, (int)val, rep);
    i += rep;
  }
  if (lens[256] == 0) return false;                  // no end-of-block code
  s.use_own();
  if (!build_table(lens, (int)hlit, kLitBits, false, s.lit_own.data(), kLitCap)) return false;
  if (!build_table(lens + hlit, (int)hdist, kDistBits, true, s.dist_own.data(), kDistCap)) return false;
  return true;
}

// ---- output buffer: kWin elements of history in front, grows by realloc (no zero fill) -------------------------------
template <class T>
struct OutBuf {
  T* mem = nullptr;
  size_t cap = 0;                                    // elements behind the history, incl. 64 of slack
  size_t n = 0;
  OutBuf() {}
  OutBuf(const OutBuf&) = delete;
  OutBuf& operator=(const OutBuf&) = delete;
  ~OutBuf() { free(mem); }
  T* at0() { return mem + kWin; }
  bool reserve(size_t want) {
    if (want + 64 <= cap) return true;
    size_t nc = cap ? cap : (size_t)1 << 20;
    while (nc < want + 64) nc *= 2;
    T* m = static_cast<T*>(realloc(mem, (kWin + nc) * sizeof(T)));
    if (!m) return false;
    mem = m;
    cap = nc;
    return true;
  }
};

// Copy of a match.  May write up to 16 elements past `len` (the buffers keep 64 elements of slack); most matches of a
// fast-level gzip stream are 3-6 long, so the first piece is stored unconditionally and the loop is the exception.
template <class T>
inline void lz_copy(T* o, size_t dist, size_t len) {
  const T* s = o - dist;
  constexpr size_t W = 16 / sizeof(T);               // elements per 16-byte piece
  if (dist >= W) {                                   // a piece never overlaps its own source
    memcpy(o, s, 16);
    if (len > W) {
      T* const end = o + len;
      o += W; s += W;
      do { memcpy(o, s, 16); o += W; s += W; } while (o < end);
    }
  } else if (dist == 1) {                            // run of one value
    uint64_t v = (uint64_t)s[0];
    v *= sizeof(T) == 1 ? 0x0101010101010101ull : 0x0001000100010001ull;
    memcpy(o, &v, 8);
    memcpy(reinterpret_cast<uint8_t*>(o) + 8, &v, 8);
    if (len > W) {
      T* const end = o + len;
      for (o += W; o < end; o += W) { memcpy(o, &v, 8); memcpy(reinterpret_cast<uint8_t*>(o) + 8, &v, 8); }
    }
  } else {                                           // 2 <= dist < W: the pattern repeats inside a piece
    for (size_t i = 0; i < W; ++i) o[i] = s[i];
    for (size_t i = W; i < len; ++i) o[i] = s[i];
  }
}

// Runs `s` over in[0, in_len) from s.bp.  Stops
//   R_BOUNDARY at the first block boundary (state S_BLOCK) whose bit position is >= target and != s.bp at entry; with
//              target_stored also in state S_STORED_LEN (behind a stored block's 3 header bits and padding, in front of
//              its LEN field) when that byte position is exactly the target and the block is not the final one,
//   R_OUT      in the middle of anything once ob.n >= out_cap,
//   R_IN       when the bytes read so far do not suffice (only if !final),
//   R_END      at the clean end of the input (final, behind a member trailer),
//   R_ERR      on invalid data (or a truncated stream when final),
//   R_SWITCH   (marker mode only) at a block boundary once the last 32 KiB of output hold no marker: nothing decoded
//              from here on can refer to the unknown window, the caller continues in byte mode.
// wlen: history elements in front of output element 0 that exist (distance check; kWin in marker mode).
template <class T>
__attribute__((always_inline)) inline Stop run_body(Inflater& s, const uint8_t* in, size_t in_len, bool final, size_t target,
                                                    bool target_stored, OutBuf<T>& ob, size_t out_cap, size_t wlen, int* oom) {
  const size_t entry_bp = s.bp;
  // positions at which the hot loop gives up: never touch bytes behind in_len + kPad; when more input may come, stop 64 B early
  // (final: from in_len on the exact bit position is compared with the end of the input every iteration)
  const size_t pos_lim = final ? in_len : (in_len > 64 ? in_len - 64 : 0);
  for (;;) {
    switch (s.st) {
      case S_DONE:
        return R_END;
      case S_HEADER: {
        size_t p = (s.bp + 7) >> 3;
        if (p >= in_len) {
          if (!final) return R_IN;
          s.st = S_DONE;
          return R_END;
        }
        const size_t avail = in_len - p;
        auto short_of = [&]() { return final ? R_ERR : R_IN; };
        if (avail < 10) return short_of();
        if (in[p] != 0x1f || in[p + 1] != 0x8b || in[p + 2] != 8 || (in[p + 3] & 0xE0)) return R_ERR;
        const uint8_t flg = in[p + 3];
        size_t q = p + 10;
        if (flg & 4) {                               // FEXTRA
          if (q + 2 > in_len) return short_of();
          const size_t xl = (size_t)in[q] | ((size_t)in[q + 1] << 8);
          q += 2;
          if (q + xl > in_len) return short_of();
          q += xl;
        }
        for (int f = 8; f <= 16; f <<= 1)            // FNAME, FCOMMENT: zero-terminated
          if (flg & f) {
            while (q < in_len && in[q]) ++q;
            if (q >= in_len) return short_of();
            ++q;
          }
        if (flg & 2) {                               // FHCRC
          if (q + 2 > in_len) return short_of();
          q += 2;
        }
        s.bp = q * 8;
        s.st = S_BLOCK;
        s.last_block = false;
        break;
      }
      case S_TRAILER: {
        const size_t p = (s.bp + 7) >> 3;
        if (p + 8 > in_len) return final ? R_ERR : R_IN;
        Segment g;
        g.out_end = ob.n;
        g.crc = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16) | ((uint32_t)in[p + 3] << 24);
        g.isize = (uint32_t)in[p + 4] | ((uint32_t)in[p + 5] << 8) | ((uint32_t)in[p + 6] << 16) | ((uint32_t)in[p + 7] << 24);
        s.segs.push_back(g);
        s.bp = (p + 8) * 8;
        s.st = S_HEADER;
        break;
      }
      case S_BLOCK: {
        if (s.bp >= target && s.bp != entry_bp) return R_BOUNDARY;
        if (ob.n >= out_cap) return R_OUT;
        if (sizeof(T) == 2) {
          // Markers die out quickly (0.7 % of the symbols of a records file) but not monotonically: a long-distance match
          // can copy one forward.  So the symbols written since the last boundary are searched — backwards, stopping at the
          // first marker, four at a time — and byte mode starts once 32 KiB in a row hold none.
          const uint16_t* p = reinterpret_cast<const uint16_t*>(ob.at0());
          size_t i = ob.n;
          while (i > s.scanned) {
            if (i - s.scanned >= 4) {
              uint64_t v;
              memcpy(&v, p + i - 4, 8);
              if (!(v & 0x8000800080008000ull)) { i -= 4; continue; }
            }
            --i;
            if (p[i] & 0x8000u) { s.last_marker = i + 1; break; }
          }
          s.scanned = ob.n;
          if (ob.n >= s.last_marker + kWin) return R_SWITCH;
        }
        // a dynamic header is < 400 bytes; when more input may come, do not start one that may not be whole
        if (!final && (s.bp >> 3) + 400 > in_len) return R_IN;
        if ((s.bp >> 3) >= in_len) return R_ERR;     // final and nothing left: truncated
        Bits b(in, s.bp);
        s.last_block = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 0) {
          s.bp = ((b.bitpos() + 7) >> 3) * 8;        // header bits, then padding to the byte boundary
          s.st = S_STORED_LEN;
        } else if (type == 1) {
          s.lit = fixed_tables().lit;
          s.dist = fixed_tables().dist;
          s.bp = b.bitpos();
          s.st = S_HUFF;
        } else if (type == 2) {
          if (!parse_dynamic(b, s)) return R_ERR;
          s.bp = b.bitpos();
          if (s.bp > in_len * 8) return R_ERR;       // the header ran into the padding: truncated
          s.st = S_HUFF;
        } else {
          return R_ERR;
        }
        break;
      }
      case S_STORED_LEN: {
        if (target_stored && s.bp == target && !s.last_block && s.bp != entry_bp) return R_BOUNDARY;
        const size_t p = s.bp >> 3;
        if (p + 4 > in_len) return final ? R_ERR : R_IN;
        const uint32_t len = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8), nlen = (uint32_t)in[p + 2] | ((uint32_t)in[p + 3] << 8);
        if ((len ^ 0xFFFFu) != nlen) return R_ERR;
        s.stored_left = len;
        s.bp = (p + 4) * 8;
        s.st = S_STORED;
        break;
      }
      case S_STORED: {
        if (s.stored_left == 0) { s.st = s.last_block ? (s.raw ? S_DONE : S_TRAILER) : S_BLOCK; break; }
        if (ob.n >= out_cap) return R_OUT;
        const size_t p = s.bp >> 3;
        size_t k = s.stored_left;
        if (k > out_cap - ob.n) k = out_cap - ob.n;
        const size_t avail = in_len > p ? in_len - p : 0;
        if (k > avail) k = avail;
        if (k == 0) return final ? R_ERR : R_IN;       // the stored bytes are not there (yet)
        if (!ob.reserve(ob.n + k)) { *oom = 1; return R_ERR; }
        T* o = ob.at0() + ob.n;
        for (size_t i = 0; i < k; ++i) o[i] = (T)in[p + i];
        ob.n += k;
        s.bp += k * 8;
        s.stored_left -= (uint32_t)k;
        break;
      }
      case S_HUFF: {
        Bits b(in, s.bp);
        const uint32_t* const lit = s.lit;
        const uint32_t* const dist = s.dist;
        const uint32_t lmask = (1u << kLitBits) - 1, dmask = (1u << kDistBits) - 1;
        if (!ob.reserve(ob.n + (1u << 16))) { *oom = 1; return R_ERR; }
        T* base = ob.at0();
        size_t o = ob.n;
        size_t o_lim = ob.cap - 64 - 600;            // room for three literals + one match + copy overshoot
        if (o_lim > out_cap) o_lim = out_cap;
        Stop why = R_BOUNDARY;
        bool block_done = false;
        size_t o_prev = o;                             // output position in front of the previous iteration's symbols
        for (;;) {
          if (o >= o_lim) {
            if (o >= out_cap) { why = R_OUT; break; }
            ob.n = o;
            if (!ob.reserve(o + (o >> 1) + (1u << 16))) { *oom = 1; return R_ERR; }
            base = ob.at0();
            o_lim = ob.cap - 64 - 600;
            if (o_lim > out_cap) o_lim = out_cap;
            if (o >= o_lim) { why = R_OUT; break; }
          }
          if (b.pos > pos_lim) {
            if (!final) { why = R_IN; break; }
            if (b.bitpos() > in_len * 8) {           // the previous iteration decoded padding: the stream is truncated;
              o = o_prev;                            // what it produced is not part of it
              why = R_ERR;
              break;
            }
          }
          o_prev = o;
          b.refill();
          // >= 56 bits after a refill: three symbols of <= 15 bits fit.  Literals first; the subtable link is tested only
          // on the way out of the literal path.
          uint32_t e = lit[b.buf & lmask];
          if (e_kind(e) == K_LIT) {
            b.drop(e_bits(e));
            base[o++] = (T)e_val(e);
            e = lit[b.buf & lmask];
            if (e_kind(e) == K_LIT) {
              b.drop(e_bits(e));
              base[o++] = (T)e_val(e);
              e = lit[b.buf & lmask];
              if (e_kind(e) == K_LIT) {
                b.drop(e_bits(e));
                base[o++] = (T)e_val(e);
                continue;
              }
            }
          }
          if (e_kind(e) == K_SUB) { b.drop(kLitBits); e = lit[e_val(e) + (b.buf & ((1u << e_f(e)) - 1))]; }
          uint64_t saved = b.buf;
          b.drop(e_bits(e));                           // code + extra bits in one shift; >= 6 bits left in the worst case
          if (e_kind(e) == K_LIT) { base[o++] = (T)e_val(e); continue; }
          if (e_kind(e) == K_BASE) {
            const size_t len = e_val(e) + ((uint32_t)(saved >> e_f(e)) & ((1u << (e_bits(e) - e_f(e))) - 1));
            if (b.cnt < 32) b.refill();                // a distance needs <= 15 + 13 bits
            uint32_t d = dist[b.buf & dmask];
            if (e_kind(d) == K_SUB) { b.drop(kDistBits); d = dist[e_val(d) + (b.buf & ((1u << e_f(d)) - 1))]; }
            saved = b.buf;
            b.drop(e_bits(d));
            if (e_kind(d) != K_BASE) { why = R_ERR; break; }
            const size_t dd = e_val(d) + ((uint32_t)(saved >> e_f(d)) & ((1u << (e_bits(d) - e_f(d))) - 1));
            if (dd > o + wlen) { why = R_ERR; break; }   // reaches in front of the history
            lz_copy(base + o, dd, len);
            o += len;
            continue;
          }
          if (e_kind(e) == K_EOB) { block_done = true; break; }
          why = R_ERR;                                 // an unused code / a length symbol that does not exist
          break;
        }
        ob.n = o;                                      // on an error: everything in front of the bad spot stays
        s.bp = b.bitpos();
        if (block_done) {
          if (s.bp > in_len * 8) { ob.n = o_prev; return R_ERR; }   // the block's last bits came from the padding: truncated
          s.st = s.last_block ? (s.raw ? S_DONE : S_TRAILER) : S_BLOCK;
          break;
        }
        return why;
      }
    }
  }
}

// The same body compiled twice: for any x86-64, and with BMI2 (shrx / bzhi: the variable shifts and masks of the symbol
// loop stop going through CL and the flags), picked once at run time.
template <class T>
Stop run_generic(Inflater& s, const uint8_t* in, size_t in_len, bool final, size_t target, bool target_stored, OutBuf<T>& ob,
                 size_t out_cap, size_t wlen, int* oom) {
  return run_body<T>(s, in, in_len, final, target, target_stored, ob, out_cap, wlen, oom);
}
#if IBU_PGZ_X86
template <class T>
__attribute__((target("bmi2,bmi,lzcnt"))) Stop run_bmi2(Inflater& s, const uint8_t* in, size_t in_len, bool final, size_t target,
                                                        bool target_stored, OutBuf<T>& ob, size_t out_cap, size_t wlen, int* oom) {
  return run_body<T>(s, in, in_len, final, target, target_stored, ob, out_cap, wlen, oom);
}
#endif
template <class T>
Stop run(Inflater& s, const uint8_t* in, size_t in_len, bool final, size_t target, bool target_stored, OutBuf<T>& ob,
         size_t out_cap, size_t wlen, int* oom) {
#if IBU_PGZ_X86
  static const bool bmi2 = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("bmi") && !getenv("IBU_PGZ_NO_BMI2");
  if (bmi2) return run_bmi2<T>(s, in, in_len, final, target, target_stored, ob, out_cap, wlen, oom);
#endif
  return run_generic<T>(s, in, in_len, final, target, target_stored, ob, out_cap, wlen, oom);
}

// A stored block whose LEN field sits at byte P: LEN / NLEN agree, the byte in front ends in BFINAL = 0, BTYPE = 00 and
// zero padding (what zlib, gzip and pigz write), and — LEN ^ NLEN alone is only a 2^-16 test — the block BEHIND it starts
// with a valid stored or dynamic header as well.  (A sync flush, the seam between pigz's pieces, is an empty stored block.)
bool stored_candidate(const uint8_t* in, size_t in_len, size_t P, Inflater& scratch) {
  if (P < 1 || P + 4 > in_len) return false;
  const uint32_t len = (uint32_t)in[P] | ((uint32_t)in[P + 1] << 8), nlen = (uint32_t)in[P + 2] | ((uint32_t)in[P + 3] << 8);
  if ((len ^ 0xFFFFu) != nlen || (in[P - 1] & 0xE0)) return false;
  const size_t Q = P + 4 + len;
  if (Q + 5 > in_len) return false;                  // cannot be confirmed inside this buffer: not a candidate
  const uint32_t type = (in[Q] >> 1) & 3;
  if (type == 0) {
    if (in[Q] >> 3) return false;                    // padding
    const uint32_t l2 = (uint32_t)in[Q + 1] | ((uint32_t)in[Q + 2] << 8), n2 = (uint32_t)in[Q + 3] | ((uint32_t)in[Q + 4] << 8);
    return (l2 ^ 0xFFFFu) == n2;
  }
  if (type == 2) {
    Bits b(in, Q * 8 + 3);
    return parse_dynamic(b, scratch) && b.bitpos() <= in_len * 8;
  }
  return false;
}

// First candidate in [from, to): the bit position of a non-final dynamic block header (complete code sets, an
// end-of-block code), or — *stored set — the bit position of the LEN field of a stored block (see stored_candidate).
// Reads at most ~400 bytes behind `to`; the caller keeps that inside the buffer + padding.
size_t find_block(const uint8_t* in, size_t in_len, size_t from, size_t to, Inflater& scratch, bool* stored) {
  constexpr uint32_t kMaxParses = 1u << 14;
  uint32_t parses = 0;
  *stored = false;
  for (size_t bit = from; bit < to; ++bit) {
    uint64_t w;
    memcpy(&w, in + (bit >> 3), 8);
    if ((bit & 7) == 0 && ((((uint32_t)w ^ (uint32_t)(w >> 16)) & 0xFFFFu) == 0xFFFFu) && stored_candidate(in, in_len, bit >> 3, scratch)) {
      *stored = true;
      return bit;
    }
    const uint64_t x = w >> (bit & 7);
    if ((x & 7) != 4) continue;                      // BFINAL = 0, BTYPE = 2 (bits 0 1 from the LSB: 0, then 01 -> value 0b100)
    if (((x >> 3) & 31) > 29 || ((x >> 8) & 31) > 29) continue;
    const uint32_t ncl = (uint32_t)((x >> 13) & 15) + 4;
    memcpy(&w, in + ((bit + 17) >> 3), 8);
    const uint64_t y = w >> ((bit + 17) & 7);        // 57 bits = 19 code-length-code lengths
    uint32_t kraft = 0;
    for (uint32_t i = 0; i < ncl; ++i) {
      const uint32_t l = (uint32_t)(y >> (3 * i)) & 7;
      if (l) kraft += 128u >> l;
    }
    if (kraft != 128) continue;
    // Highly repetitive compressed data (a long run of one byte) passes the cheap tests at every period of its bit
    // pattern and fails only in the full parse: bound that work; the chunk then has no candidate and the chunk in front
    // of it decodes through.
    if (++parses > kMaxParses) return kNone;
    Bits b(in, bit + 3);
    if (!parse_dynamic(b, scratch)) continue;
    if (b.bitpos() > in_len * 8) continue;
    return bit;
  }
  return kNone;
}

// 16-bit symbols -> bytes.  win: the 32 KiB in front of the chunk.  false: a symbol that is neither a byte nor a marker.
bool resolve(const uint16_t* src, size_t n, const uint8_t* win, uint8_t* dst, uint64_t* markers) {
  size_t i = 0;
  uint64_t m = 0;
  uint32_t bad = 0;
  for (; i + 4 <= n; i += 4) {
    uint64_t v;
    memcpy(&v, src + i, 8);
    if ((v & 0xFF00FF00FF00FF00ull) == 0) {
      dst[i] = (uint8_t)v; dst[i + 1] = (uint8_t)(v >> 16); dst[i + 2] = (uint8_t)(v >> 32); dst[i + 3] = (uint8_t)(v >> 48);
      continue;
    }
    for (int k = 0; k < 4; ++k) {
      const uint16_t s = src[i + k];
      if (s < 256) dst[i + k] = (uint8_t)s;
      else if (s >= 0x8000) { dst[i + k] = win[s - 0x8000]; ++m; }
      else bad = 1;
    }
  }
  for (; i < n; ++i) {
    const uint16_t s = src[i];
    if (s < 256) dst[i] = (uint8_t)s;
    else if (s >= 0x8000) { dst[i] = win[s - 0x8000]; ++m; }
    else bad = 1;
  }
  *markers += m;
  return !bad;
}

// Worker threads that live as long as the decoder: three parallel phases per batch on 32 threads were ~100 thread
// starts per batch (1 ms per phase, an eighth of the wall time at 1e9 records).  run(n, fn) calls fn(0) ... fn(n-1), the
// last index on the calling thread, and returns when all are done; fn must not throw.  Workers that cannot be started
// simply do not exist: their indices are run by the caller.
class Pool {
 public:
  explicit Pool(unsigned workers) {
    try {
      th_.reserve(workers);
      for (unsigned i = 0; i < workers; ++i) th_.emplace_back([this, i] { loop(i); });
    } catch (...) {                                    // EAGAIN under a pids cgroup, bad_alloc: fewer workers
    }
  }
  ~Pool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  template <class F>
  void run(unsigned n, F&& fn) {
    if (n == 0) return;
    const unsigned w = (unsigned)th_.size();
    if (n == 1 || w == 0) { for (unsigned i = 0; i < n; ++i) fn(i); return; }
    std::function<void(unsigned)> f = std::ref(fn);
    {
      std::lock_guard<std::mutex> g(m_);
      fn_ = &f;
      n_ = n;
      next_ = 0;
      pending_ = n;
      ++gen_;
    }
    cv_.notify_all();
    work();                                            // the caller takes indices like everybody else
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      unsigned i;
      {
        std::lock_guard<std::mutex> g(m_);
        if (!fn_ || next_ >= n_) return;
        i = next_++;
      }
      (*fn_)(i);
      bool last;
      { std::lock_guard<std::mutex> g(m_); last = --pending_ == 0; }
      if (last) done_.notify_all();
    }
  }
  void loop(unsigned) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned)>* fn_ = nullptr;
  unsigned n_ = 0, next_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// CRC-32 (the gzip polynomial) by carry-less multiplication: four 128-bit lanes folded over 64 bytes per step, then
// folded together and Barrett-reduced (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ",
// Intel 2009; constants for the bit-reflected polynomial 0xEDB88320).  `crc` in and out are zlib's crc32() values.
// zlib 1.2.11's table-driven crc32 runs at ~0.8 GB/s here, a quarter of the whole parallel decode; this one at memory speed.
#if IBU_PGZ_X86
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t crc, const uint8_t* buf, size_t len) {  // len >= 64, len % 16 == 0
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0};
  alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
  x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
  x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
  x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)~crc));
  x0 = _mm_load_si128((const __m128i*)k1k2);
  buf += 64;
  len -= 64;
  while (len >= 64) {
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
    x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
    x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
    x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
    x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    buf += 64;
    len -= 64;
  }
  x0 = _mm_load_si128((const __m128i*)k3k4);          // four lanes -> one
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (len >= 16) {                                  // remaining whole 16-byte blocks
    x2 = _mm_loadu_si128((const __m128i*)buf);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    buf += 16;
    len -= 16;
  }
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);             // 128 -> 64 bits
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_loadl_epi64((const __m128i*)k5k0);
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_load_si128((const __m128i*)poly);           // Barrett reduction to 32 bits
  x2 = _mm_and_si128(x1, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return ~(uint32_t)_mm_extract_epi32(x1, 1);
}
#endif
bool have_clmul() {
#if IBU_PGZ_X86
  static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  return ok;
#else
  return false;
#endif
}
uint32_t crc32_bytes(uint32_t crc, const uint8_t* buf, size_t len) {
#if IBU_PGZ_X86
  if (len >= 64 && have_clmul()) {
    const size_t k = len & ~(size_t)15;
    crc = crc32_clmul(crc, buf, k);
    buf += k;
    len -= k;
  }
#endif
  while (len) {                                        // zlib's crc32 takes a uInt length
    const size_t k = len > ((size_t)1 << 30) ? (size_t)1 << 30 : len;
    crc = (uint32_t)crc32(crc, buf, (uInt)k);
    buf += k;
    len -= k;
  }
  return crc;
}

struct Chunk {
  size_t start = kNone;                              // candidate bit position (chunk 0: the true position)
  size_t target = kNone;
  bool start_stored = false, target_stored = false;  // the candidate / the target is a stored block's LEN field
  Inflater inf;
  OutBuf<uint16_t> o16;                              // head of a chunk >= 1: symbols (bytes and markers)
  OutBuf<uint8_t> o8;                                // chunk 0, and the tail of a chunk >= 1 once it is marker-free
  Stop stop = R_ERR;
  int oom = 0;
  size_t n_out() const { return o16.n + o8.n; }
  // after acceptance
  size_t out_off = 0;
  std::vector<uint8_t> win_tmp;
  const uint8_t* win = nullptr;
  std::vector<uint32_t> piece_crc;                   // one per segment + one for the open tail
  bool bad_symbol = false;
  std::vector<uint8_t> head8;                        // the symbols of o16 as bytes once the window is known
  // the chunk's bytes are head8[0, o16.n) followed by o8[0, o8.n)
  bool resolve_head(size_t a, size_t b, uint64_t* markers) {   // symbols [a, b) of the head -> bytes
    return a >= b || resolve(o16.at0() + a, b - a, win, head8.data() + a, markers);
  }
  uint32_t crc_range(size_t a, size_t b) {                     // CRC-32 of output bytes [a, b) of this chunk
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    if (a < o16.n) {
      const size_t e = b < o16.n ? b : o16.n;
      crc = crc32_bytes(crc, head8.data() + a, e - a);
      a = e;
    }
    if (a < b) crc = crc32_bytes(crc, o8.at0() + (a - o16.n), b - a);
    return crc;
  }
};

}  // namespace

struct WorkerPool::Impl { Pool pool; explicit Impl(unsigned w) : pool(w) {} };
WorkerPool::WorkerPool(unsigned workers) : p_(new Impl(workers)) {}
WorkerPool::~WorkerPool() {}
void WorkerPool::run(unsigned n, const std::function<void(unsigned)>& fn) { p_->pool.run(n, fn); }

struct RawInflater::Impl {
  Inflater inf;
  OutBuf<uint8_t> ob;
};
RawInflater::RawInflater() : p_(new Impl) {}
RawInflater::~RawInflater() {}
int RawInflater::inflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, uint32_t* crc) {
  Impl& P = *p_;
  P.inf.st = S_BLOCK;
  P.inf.raw = true;
  P.inf.last_block = false;
  P.inf.bp = 0;
  P.inf.segs.clear();
  P.ob.n = 0;
  if (!P.ob.reserve(out_len + 1)) return ENOMEM;
  int oom = 0;
  const Stop st = run<uint8_t>(P.inf, in, in_len, true, kNone, false, P.ob, out_len + 1, 0, &oom);
  if (oom) return ENOMEM;
  if (st != R_END || P.ob.n != out_len || ((P.inf.bp + 7) >> 3) != in_len) return EPROTO;
  if (out_len) memcpy(out, P.ob.at0(), out_len);
  if (crc) *crc = crc32_bytes((uint32_t)crc32(0L, Z_NULL, 0), out, out_len);
  return 0;
}

struct ParallelGunzip::Impl {
  ReadFn inner;
  unsigned threads;
  size_t chunk;
  std::vector<uint8_t> comp;                         // compressed bytes not yet consumed live at comp[comp_off, comp_off + comp_len), kPad zeros behind
  size_t comp_off = 0, comp_len = 0;
  bool inner_eof = false, done = false, failed = false;
  Inflater cur;                                      // the true decoder state; cur.bp is relative to comp[0]
  std::vector<uint8_t> window;                       // last <= 32 KiB of output
  uint32_t run_crc = 0;                              // CRC-32 / length of the open member so far
  uint64_t run_len = 0;
  size_t out_cap;
  std::unique_ptr<Pool> workers;                     // threads - 1 of them, started with the first batch that needs them
  std::vector<Chunk> pool[2];                        // per-chunk buffers and tables; the two sets take turns (see next_batch)
  unsigned flip = 0;
  // A stream in which the search finds nothing (only fixed-Huffman blocks, say) would pay for a futile search in every
  // batch: after a batch without any candidate the next `find_skip` batches are decoded by one thread without searching
  // (1, 2, 4 ... 16 batches), then the search is tried again.
  unsigned find_fail_streak = 0, find_skip = 0;
  // read-ahead: while a batch is searched / decoded / patched, one helper thread reads the next batch's compressed bytes
  // (otherwise 1/7 of the wall time: one thread copying 12 GB out of the page cache) — into the SECOND buffer, behind a
  // gap that is large enough for whatever the current batch leaves unconsumed; at the next top-up the leftover (small)
  // is copied in front of the new bytes and the two buffers swap roles: the bulk of the input is never copied again.
  std::vector<uint8_t> ahead;
  size_t ahead_gap = 0, ahead_len = 0;
  bool ahead_eof = false;
  double ahead_seconds = 0;
  std::future<int> ahead_f;
  int join_ahead() {
    if (!ahead_f.valid()) return 0;
    int rc;
    try { rc = ahead_f.get(); } catch (...) { rc = EIO; }
    if (rc) return rc;
    // leftover <= the comp_len the gap was sized for (bytes are only consumed in between)
    if (comp_len) memcpy(ahead.data() + ahead_gap - comp_len, comp.data() + comp_off, comp_len);
    comp.swap(ahead);
    comp_off = ahead_gap - comp_len;
    comp_len += ahead_len;
    ahead_len = 0;
    if (ahead_eof) inner_eof = true;
    return 0;
  }
  void start_ahead(size_t bytes) {
    if (inner_eof || bytes == 0) return;
    try {
      ahead_gap = comp_len;
      // + 1 MiB: the synchronous top-up behind the swap (the slack the last chunk ran into) must fit without compaction
      if (ahead.size() < ahead_gap + bytes + ((size_t)1 << 20) + kPad) ahead.resize(ahead_gap + bytes + ((size_t)1 << 20) + kPad);
      ahead_len = 0;
      ahead_f = std::async(std::launch::async, [this, bytes]() -> int {
        uint8_t* const dst = ahead.data() + ahead_gap;
        const double h0 = now_s();
        struct Acc { double* d; double t; ~Acc() { *d += now_s() - t; } } acc{&ahead_seconds, h0};
        while (ahead_len < bytes) {
          size_t got = 0;
          const int rc = inner(dst + ahead_len, bytes - ahead_len, &got);
          if (rc) return rc;
          if (got == 0) { ahead_eof = true; break; }
          ahead_len += got;
        }
        return 0;
      });
    } catch (...) {                                    // no thread / no memory: the next top-up reads synchronously
    }
  }
};

ParallelGunzip::ParallelGunzip(ReadFn inner, unsigned threads, size_t chunk_bytes) : p_(new Impl) {
  p_->inner = std::move(inner);
  p_->threads = threads < 1 ? 1 : threads;
  p_->chunk = chunk_bytes < 4096 ? 4096 : chunk_bytes;
  // Output elements per chunk and batch: ten times the compressed bytes.  A chunk that inflates further (long runs: up to
  // 1000 : 1) stops there, in the middle of a block if need be, the chain ends with it and the next batch resumes at that
  // spot — so the buffers of 32 threads stay bounded (<= 20 B per compressed byte and thread) whatever the input is.
  p_->out_cap = p_->chunk * 10 < ((size_t)1 << 20) ? (size_t)1 << 20 : p_->chunk * 10;
  p_->run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
}
ParallelGunzip::~ParallelGunzip() {
  if (p_ && p_->ahead_f.valid()) { try { (void)p_->ahead_f.get(); } catch (...) {} }
}

int ParallelGunzip::next_batch(std::vector<Span>& out, bool* eof) {
  Impl& P = *p_;
  out.clear();
  *eof = false;
  if (P.failed) return EPROTO;
  const size_t kSlack = (size_t)256 << 10;
  while (out.empty()) {
    if (P.done) { *eof = true; return 0; }
    // ---- 0. top up the compressed buffer ---------------------------------------------------------------------------
    double t0 = now_s(), t1;
    const size_t base_byte = P.cur.bp >> 3;
    const size_t want = base_byte + (size_t)P.threads * P.chunk + kSlack;
    {
      const size_t before = P.comp_len;
      const int rc = P.join_ahead();
      if (rc) { P.failed = true; return rc; }
      st_.bytes_in += P.comp_len - before;
      st_.s_join_wait += now_s() - t0;
      st_.s_helper_read = P.ahead_seconds;
    }
    if (P.comp_len < want && !P.inner_eof) {
      if (P.comp.size() < P.comp_off + want + kPad) {
        if (P.comp_off) { memmove(P.comp.data(), P.comp.data() + P.comp_off, P.comp_len); P.comp_off = 0; }
        if (P.comp.size() < want + kPad) P.comp.resize(want + kPad);
      }
      while (P.comp_len < want && !P.inner_eof) {
        size_t got = 0;
        const int rc = P.inner(P.comp.data() + P.comp_off + P.comp_len, want - P.comp_len, &got);
        if (rc) { P.failed = true; return rc; }
        if (got == 0) P.inner_eof = true;
        P.comp_len += got;
        st_.bytes_in += got;
      }
    }
    if (P.comp.size() < P.comp_off + P.comp_len + kPad) P.comp.resize(P.comp_off + P.comp_len + kPad);
    memset(P.comp.data() + P.comp_off + P.comp_len, 0, kPad);
    P.start_ahead((size_t)P.threads * P.chunk);       // uses `inner` until joined at the next top-up; nothing else touches it meanwhile
    const uint8_t* in = P.comp.data() + P.comp_off;
    const size_t in_len = P.comp_len;
    const bool final = P.inner_eof;

    t1 = now_s(); st_.s_read += t1 - t0; t0 = t1;
    // ---- 1. candidates ---------------------------------------------------------------------------------------------
    size_t n = P.threads;
    while (n > 1 && base_byte + (n - 1) * P.chunk + 1024 >= in_len) --n;   // chunks that have at least some bytes
    const bool searching = n > 1 && P.find_skip == 0;
    if (!searching && n > 1) { --P.find_skip; n = 1; }
    if (!P.workers) P.workers.reset(new Pool(P.threads - 1));
    if (P.pool[P.flip].size() < P.threads) P.pool[P.flip] = std::vector<Chunk>(P.threads);
    std::vector<Chunk>& ch = P.pool[P.flip];
    for (size_t j = 0; j < n; ++j) {
      Chunk& c = ch[j];
      c.start = c.target = kNone;
      c.start_stored = c.target_stored = false;
      c.stop = R_ERR;
      c.oom = 0;
      c.o8.n = c.o16.n = 0;
      c.inf.segs.clear();
      c.out_off = 0;
      c.win = nullptr;
      c.piece_crc.clear();
      c.bad_symbol = false;
    }
    ch[0].start = P.cur.bp;
    if (n > 1)
      P.workers->run((unsigned)(n - 1), [&](unsigned i) {
        const size_t j = i + 1;
        size_t from = (base_byte + j * P.chunk) * 8, to = (base_byte + (j + 1) * P.chunk) * 8;
        const size_t last = in_len > 8 ? (in_len - 8) * 8 : 0;   // a header needs some bytes; keep the loads inside the padding
        if (to > last) to = last;
        if (from <= P.cur.bp) from = P.cur.bp + 1;
        try {                                          // a worker thread must not throw (bad_alloc of the scratch tables)
          Inflater scratch;
          ch[j].start = from < to ? find_block(in, in_len, from, to, scratch, &ch[j].start_stored) : kNone;
        } catch (...) {
          ch[j].start = kNone;                         // no candidate: the chunk in front decodes through
        }
      });
    std::vector<size_t> act;                           // chunks that will be decoded
    act.push_back(0);
    for (size_t j = 1; j < n; ++j) {
      if (ch[j].start != kNone) act.push_back(j);
      else st_.candidates_missing++;
    }
    if (searching) {
      if (act.size() == 1) {
        P.find_fail_streak = P.find_fail_streak < 4 ? P.find_fail_streak + 1 : 4;
        P.find_skip = 1u << P.find_fail_streak;
      } else {
        P.find_fail_streak = 0;
      }
    }
    const size_t nominal_end = (base_byte + n * P.chunk) * 8;
    for (size_t a = 0; a < act.size(); ++a) {
      Chunk& c = ch[act[a]];
      c.target = a + 1 < act.size() ? ch[act[a + 1]].start : (final && nominal_end >= in_len * 8 ? kNone : nominal_end);
      c.target_stored = a + 1 < act.size() && ch[act[a + 1]].start_stored;
    }

    t1 = now_s(); st_.s_find += t1 - t0; t0 = t1;
    // ---- 2. decode -------------------------------------------------------------------------------------------------
    {
      Chunk& c0 = ch[0];
      c0.inf = std::move(P.cur);
      if (!c0.o8.reserve(P.chunk * 4)) { P.failed = true; return ENOMEM; }
      memset(c0.o8.mem, 0, kWin - P.window.size());
      if (!P.window.empty()) memcpy(c0.o8.mem + kWin - P.window.size(), P.window.data(), P.window.size());
    }
    const size_t wlen0 = P.window.size();
    P.workers->run((unsigned)act.size(), [&](unsigned a) {
      Chunk& c = ch[act[a]];
      try {
      if (a == 0) {
        c.stop = run<uint8_t>(c.inf, in, in_len, final, c.target, c.target_stored, c.o8, P.out_cap, wlen0, &c.oom);
        return;
      }
      c.inf.st = c.start_stored ? S_STORED_LEN : S_BLOCK;
      c.inf.last_block = false;
      c.inf.bp = c.start;
      if (!c.o16.reserve(P.chunk * 4)) { c.oom = 1; c.stop = R_ERR; return; }
      for (size_t i = 0; i < kWin; ++i) c.o16.mem[i] = (uint16_t)(0x8000u + i);
      c.inf.scanned = c.inf.last_marker = 0;
      c.stop = run<uint16_t>(c.inf, in, in_len, final, c.target, c.target_stored, c.o16, P.out_cap, kWin, &c.oom);
      if (c.stop != R_SWITCH) return;
      // the last 32 KiB are plain bytes: they are the window of everything that follows -> byte mode (faster, no patching)
      if (!c.o8.reserve(P.chunk * 4)) { c.oom = 1; c.stop = R_ERR; return; }
      const uint16_t* tail = c.o16.at0() + c.o16.n - kWin;
      for (size_t i = 0; i < kWin; ++i) c.o8.mem[i] = (uint8_t)tail[i];
      const size_t nseg = c.inf.segs.size(), cap8 = P.out_cap > c.o16.n ? P.out_cap - c.o16.n : 0;
      c.stop = run<uint8_t>(c.inf, in, in_len, final, c.target, c.target_stored, c.o8, cap8, kWin, &c.oom);
      for (size_t g = nseg; g < c.inf.segs.size(); ++g) c.inf.segs[g].out_end += c.o16.n;   // member ends count from the chunk's start
      } catch (...) {                                  // bad_alloc in a table / segment vector: a worker thread must not throw
        c.oom = 1;
        c.stop = R_ERR;
      }
    });

    t1 = now_s(); st_.s_decode += t1 - t0; t0 = t1;
    // ---- 3. chain --------------------------------------------------------------------------------------------------
    std::vector<size_t> ok;
    ok.push_back(0);
    for (size_t a = 1; a < act.size(); ++a) {
      const Chunk& prev = ch[ok.back()];
      const Chunk& c = ch[act[a]];
      if (prev.stop != R_BOUNDARY || prev.inf.bp != c.start || prev.inf.st != (c.start_stored ? S_STORED_LEN : S_BLOCK)) break;
      ok.push_back(act[a]);
    }
    st_.chunks_accepted += ok.size();
    st_.chunks_discarded += act.size() - ok.size();
    for (size_t k : ok)
      if (ch[k].oom) { P.failed = true; return ENOMEM; }
    // An accepted chunk decodes the true stream, so an error in it (only the LAST accepted chunk can carry one: nothing
    // chains behind it) is the stream's: what was decoded in front of the bad spot is delivered like any other batch —
    // a sequential inflate hands those bytes out, too — and the NEXT call reports the error.
    const bool stream_bad = ch[ok.back()].stop == R_ERR;

    // ---- 4. windows, patching, CRC ----------------------------------------------------------------------------------
    // Nothing is copied into one output buffer: a chunk's bytes stay where they were decoded (o8) or are patched into a
    // byte buffer of their own (head8, the symbols in front of the byte-mode switch) and are handed out as pieces.
    size_t total = 0;
    for (size_t k : ok) { ch[k].out_off = total; total += ch[k].n_out(); }
    auto push_window = [](std::vector<uint8_t>& w, const uint8_t* p, size_t n) {   // w = last <= 32 KiB of (w ++ p[0, n))
      if (n >= kWin) { w.assign(p + n - kWin, p + n); return; }
      if (n) w.insert(w.end(), p, p + n);
      if (w.size() > kWin) w.erase(w.begin(), w.begin() + (w.size() - kWin));
    };
    // sequential, 32 KiB per chunk: the window in front of every chunk, then that chunk's own last 32 KiB
    std::vector<uint8_t> runwin = P.window;
    for (size_t a = 0; a < ok.size(); ++a) {
      Chunk& c = ch[ok[a]];
      uint64_t m = 0;
      if (c.o16.n) {
        c.win_tmp.assign(kWin, 0);
        if (!runwin.empty()) memcpy(c.win_tmp.data() + kWin - runwin.size(), runwin.data(), runwin.size());
        c.win = c.win_tmp.data();
        if (c.head8.size() < c.o16.n) c.head8.resize(c.o16.n);
        if (c.o8.n < kWin) {                           // the chunk's last 32 KiB reach into its head: patch that part now
          const size_t need = kWin - c.o8.n, from = c.o16.n > need ? c.o16.n - need : 0;
          if (!c.resolve_head(from, c.o16.n, &m)) c.bad_symbol = true;
          push_window(runwin, c.head8.data() + from, c.o16.n - from);
        }
      }
      if (c.o8.n) push_window(runwin, c.o8.at0(), c.o8.n);
      st_.marker_symbols += m;
    }
    t1 = now_s(); st_.s_windows += t1 - t0; t0 = t1;
    std::vector<uint64_t> markers(ok.size(), 0);
    P.workers->run((unsigned)ok.size(), [&](unsigned a) {
      Chunk& c = ch[ok[a]];
      try {
      const size_t nn = c.n_out();
      if (c.o16.n) {                                   // the rest of the head (its tail is done if the windows pass needed it)
        const size_t need = c.o8.n < kWin ? kWin - c.o8.n : 0, upto = c.o16.n > need ? c.o16.n - need : 0;
        if (!c.resolve_head(0, upto, &markers[a])) c.bad_symbol = true;
      }
      size_t from = 0;
      for (size_t g = 0; g <= c.inf.segs.size(); ++g) {
        const size_t to = g < c.inf.segs.size() ? c.inf.segs[g].out_end : nn;
        c.piece_crc.push_back(c.crc_range(from, to));
        from = to;
      }
      } catch (...) {
        c.oom = 1;
      }
    });
    for (size_t k : ok)
      if (ch[k].oom) { P.failed = true; return ENOMEM; }
    for (size_t a = 0; a < ok.size(); ++a) st_.marker_symbols += markers[a];
    // A member whose CRC-32 / ISIZE does not match: a sequential inflate has handed its bytes out before it sees the
    // trailer, so they go out here as well — everything up to the end of that member — and the error follows.
    size_t deliver = total;
    bool crc_bad = false;
    for (size_t k : ok) {
      Chunk& c = ch[k];
      if (c.bad_symbol) { P.failed = true; return EPROTO; }
      size_t from = 0;
      for (size_t g = 0; g <= c.inf.segs.size() && !crc_bad; ++g) {
        const size_t to = g < c.inf.segs.size() ? c.inf.segs[g].out_end : c.n_out();
        P.run_crc = (uint32_t)crc32_combine(P.run_crc, c.piece_crc[g], (z_off_t)(to - from));
        P.run_len += to - from;
        if (g < c.inf.segs.size()) {                   // a member ended here
          if (P.run_crc != c.inf.segs[g].crc || (uint32_t)P.run_len != c.inf.segs[g].isize) {
            crc_bad = true;
            deliver = c.out_off + to;
          }
          P.run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
          P.run_len = 0;
        }
        from = to;
      }
      if (crc_bad) break;
    }

    t1 = now_s(); st_.s_patch_crc += t1 - t0; t0 = t1;
    // ---- 5. carry over ---------------------------------------------------------------------------------------------
    Chunk& lastc = ch[ok.back()];
    if (lastc.stop == R_END) P.done = true;
    if (lastc.stop == R_IN && final) { P.failed = true; return EPROTO; }
    if (stream_bad || crc_bad) {
      P.failed = true;                                 // sticky: every later call returns EPROTO
      if (deliver == 0) return EPROTO;
    }
    P.window.swap(runwin);
    size_t left = deliver;                             // == total unless a member failed its CRC
    for (size_t k : ok) {
      Chunk& c = ch[k];
      const size_t h = c.o16.n < left ? c.o16.n : left;
      if (h) out.push_back(Span{c.head8.data(), h});
      left -= h;
      const size_t t = c.o8.n < left ? c.o8.n : left;
      if (t) out.push_back(Span{c.o8.at0(), t});
      left -= t;
    }
    // The next batch decodes into the other set of buffers — but only when THIS one handed pieces out: the set the caller
    // may still be reading (the batch before, `pieces stay valid until the call after the next one`: pgzip.hpp) is the other
    // one, and an iteration that delivered nothing (4 MiB of empty members, empty stored / sync-flush blocks) must come back
    // to the set it has just used, not decode over the caller's.
    if (!out.empty()) P.flip ^= 1;
    P.cur = std::move(lastc.inf);
    P.cur.segs.clear();
    const size_t shift = P.cur.bp >> 3;
    if (shift) {                                       // consumed bytes are dropped by moving the start, not the bytes
      const size_t drop = shift < P.comp_len ? shift : P.comp_len;
      P.comp_off += drop;
      P.comp_len -= drop;
      P.cur.bp -= shift * 8;
    }
    st_.batches++;
    st_.bytes_out += total;
    st_.s_carry += now_s() - t0;
    if (stream_bad || crc_bad) return 0;               // the bytes in front of the bad spot; the error comes with the next call
    if (total == 0 && !P.done && lastc.stop == R_IN && !final) continue;  // needs more input: the top-up reads it
    if (total == 0 && !P.done && shift == 0 && lastc.stop != R_IN) { P.failed = true; return EPROTO; }  // no progress: cannot happen
  }
  return 0;
}

}  // namespace pgz
}  // namespace ibu
