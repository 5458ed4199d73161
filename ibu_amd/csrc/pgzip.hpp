// pgzip.hpp — parallel inflate of an ORDINARY gzip stream (one or many members, no index, no BGZF framing).
//
// Why: BASELINE configs[4] is "1e9 records gzip-compressed input (niffler path): host inflate -> pinned ring -> GPU
// decode".  niffler / flate2 (reference: src/io/reader.rs:345-352) inflate one stream on one thread, ~0.45 GB/s of
// records, 2 400x below the HBM-resident rate; the GPU and PCIe idle.  A deflate stream has no entry points, but its
// blocks can be FOUND (a dynamic-Huffman block header is self-checking) and decoded before the 32 KiB of history in
// front of them is known, if unresolved back-references are kept as markers and patched afterwards:
//
//   batch = `threads` chunks of `chunk_bytes` compressed bytes
//   1. find   (parallel)   chunk j >= 1: first position in its range that is the header of a non-final dynamic block
//                          (complete code-length / literal-length / distance codes, an end-of-block code) or the LEN
//                          field of a non-final stored block that is followed by another valid block header (incompressible
//                          input, sync-flush seams); fixed-Huffman blocks are not searched for (nothing to check)
//   2. decode (parallel)   chunk 0 from the TRUE position with the true window, to bytes;
//                          chunk j >= 1 from its candidate to 16-bit symbols: < 256 a byte, >= 0x8000 "byte (v - 0x8000)
//                          of the 32 KiB window in front of this chunk" — until 32 KiB in a row hold no marker, from
//                          there on to bytes; each chunk stops at the block boundary that is the next chunk's candidate
//                          (or overshoots it)
//   3. chain  (sequential) chunk j+1 is accepted only if chunk j ended EXACTLY at its candidate: chunk 0 decodes the
//                          true stream, so by induction every accepted chunk started at a true block boundary.  The batch
//                          ends at the first break; the next batch starts there (progress is guaranteed by chunk 0).
//   4. patch  (seq + par)  last 32 KiB of every accepted chunk in order (the windows), then all bodies in parallel,
//                          CRC-32 per (chunk, member) piece, combined with crc32_combine and checked against every
//                          member trailer together with ISIZE.
//
// Output bytes and error class are those of the sequential zlib path (GzSource in host_io.cpp) for every stream zlib accepts
// and for truncated / bit-damaged ones (tests/test_pgzip.py): same bytes, EPROTO for a corrupt / truncated stream.
// ONE known divergence, on crafted input only: the distance check is looser than zlib's at the start of a member.  The
// 32 KiB history is carried across gzip member boundaries (chunk 0 runs with the window of whatever preceded it, marker-mode
// chunks always allow 32 KiB), so a back-reference that reaches in front of the CURRENT member's first byte — which zlib
// rejects as "invalid distance too far back" — resolves against the previous member's bytes (or zeros) here; if the
// crafted member's CRC-32 / ISIZE then still match, this decoder accepts what the sequential one refuses.  No compressor
// emits such a stream.  (The fix — bytes-since-member-start in the Inflater, min(window, that) as the distance bound, markers
// that resolve in front of a member start rejected when patching — is left undone: ADVICE r02 rated it low and VERDICT r02
// closed further work on this file.)  The decoder is suspendable in the middle of a block (output cap per chunk, end of the
// bytes read so far), so memory stays bounded for any input.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <memory>
#include <vector>

namespace ibu {
namespace pgz {

typedef std::function<int(uint8_t* dst, size_t cap, size_t* got)> ReadFn;  // 0 or errno; *got == 0 at EOF

struct Stats {
  uint64_t batches = 0, chunks_accepted = 0, chunks_discarded = 0, candidates_missing = 0;
  uint64_t bytes_in = 0, bytes_out = 0, marker_symbols = 0;
  double s_read = 0, s_find = 0, s_decode = 0, s_windows = 0, s_patch_crc = 0, s_carry = 0;  // wall seconds per phase
  double s_join_wait = 0, s_helper_read = 0;         // of s_read: waiting for the read-ahead helper; the helper's own read time
};

struct Span { const uint8_t* p; size_t n; };  // a piece of a batch's output (the decoder's own chunk buffers: no copy into one buffer)

// One bare deflate stream with nothing in front of it (a BGZF block: <= 64 KiB, its sizes known from the container)
// -> `out_len` bytes, with the same symbol loop and CRC as the parallel decoder (about twice zlib's inflate, and the CRC
// by carry-less multiplication).  `in` must be readable for in_len + 512 bytes.  0, ENOMEM, or EPROTO when the stream is
// invalid, does not end exactly at in_len, or does not produce exactly out_len bytes.  One instance per thread.
class RawInflater {
 public:
  RawInflater();
  ~RawInflater();
  int inflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, uint32_t* crc);

 private:
  struct Impl;
  std::unique_ptr<Impl> p_;
};

// Worker threads that live as long as their owner (the pool the parallel decoder runs its phases on; BgzfSource uses one
// too).  run(n, fn) calls fn(0) ... fn(n-1) on the workers and the calling thread and returns when all are done; fn must
// not throw.  Workers that cannot be started do not exist: the caller runs their share.
class WorkerPool {
 public:
  explicit WorkerPool(unsigned workers);
  ~WorkerPool();
  void run(unsigned n, const std::function<void(unsigned)>& fn);

 private:
  struct Impl;
  std::unique_ptr<Impl> p_;
};

class ParallelGunzip {
 public:
  // threads: chunks decoded at once (>= 1); chunk_bytes: compressed bytes per chunk
  ParallelGunzip(ReadFn inner, unsigned threads, size_t chunk_bytes);
  ~ParallelGunzip();
  // Decodes the next batch; `out` (replaced) lists its bytes in order as pieces of the decoder's chunk buffers.  The pieces
  // stay valid until the call AFTER the next one (two sets of chunk buffers take turns), so a caller may consume batch k
  // while batch k + 1 is being decoded.  Returns 0 or errno (EPROTO: corrupt / truncated stream).  An empty `out` with
  // *eof set is the clean end of the stream.
  int next_batch(std::vector<Span>& out, bool* eof);
  const Stats& stats() const { return st_; }

 private:
  struct Impl;
  std::unique_ptr<Impl> p_;
  Stats st_;
};

}  // namespace pgz
}  // namespace ibu
