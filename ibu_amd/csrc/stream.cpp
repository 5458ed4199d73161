// stream.cpp — record streams between files / mmap / gzip and the device, through a ring of
// pinned (hipHostMalloc) staging slots.  Host work (page-cache memcpy, pread, inflate) for
// slot k+1 overlaps the H2D copy of slot k (copy stream) and the kernel on slot k-1 (compute
// stream); the two streams are chained per slot with events, never with a device-wide sync.
//
//   file/mmap/gz --host threads--> pinned[slot] --copy_stream H2D--> dev[slot] --stream--> kernel
//
// ONE pipeline feeds the device: the pull stream (ibu_stream_*, below) — a producer thread that fills pinned slots and queues
// their H2D copies, and a consumer that takes device-resident batches in order.  ibu_mmap_process_device and
// ibu_reader_process_device are that consumer with one of the two built-in processors as the loop body.
//
// These are the device-backed forms of load_to_vec (reader.rs:510-535), Writer::write_batch
// (writer.rs:315-351), MmapReader::process_parallel (mmap.rs:286-332, ONE shard of its static
// split per call = per GPU) and the streaming Reader (reader.rs:279-306, incl. the gzip path
// of reader.rs:345-352).
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "host_io.hpp"
#include "kernels.h"
#include "pgzip.hpp"

using namespace ibu;

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Split [0,total) over up to `threads` host threads; fn(offset, len) returns 0 or errno.
template <class F>
int parallel_bytes(size_t total, uint32_t threads, F fn) {
  if (threads < 1) threads = 1;
  const size_t min_chunk = 1u << 20;
  size_t parts = (total + min_chunk - 1) / min_chunk;
  if (parts > threads) parts = threads;
  if (parts <= 1) return total ? fn((size_t)0, total) : 0;
  if (parts > 64) parts = 64;
  int rc[64];
  for (size_t i = 0; i < parts; ++i) rc[i] = 0;
  const size_t per = (total / parts + 4095) & ~(size_t)4095;
  run_pieces((unsigned)parts, [&](unsigned i) {  // never throws (common.hpp): no exception crosses the C ABI
    const size_t off = (size_t)i * per;
    if (off >= total) return;
    rc[i] = fn(off, off + per < total ? per : total - off);
  });
  for (size_t i = 0; i < parts; ++i)
    if (rc[i]) return rc[i];
  return 0;
}

int pread_all(int fd, uint8_t* dst, size_t len, off_t off) {
  while (len) {
    ssize_t k = ::pread(fd, dst, len, off);
    if (k < 0) {
      if (errno == EINTR) continue;
      return errno;
    }
    if (k == 0) return EIO;  // file shrank underneath us
    dst += k;
    off += k;
    len -= (size_t)k;
  }
  return 0;
}

struct KernelClock {  // sums hipEvent spans of the per-slot kernels
  std::vector<hipEvent_t> a, b;
  std::vector<char> live;
  double ms = 0;
  int32_t init(uint32_t slots) {
    a.resize(slots);
    b.resize(slots);
    live.assign(slots, 0);
    for (uint32_t i = 0; i < slots; ++i) {
      IBU_HIP(hipEventCreate(&a[i]));
      IBU_HIP(hipEventCreate(&b[i]));
    }
    return IBU_OK;
  }
  void harvest(uint32_t s) {
    if (!live[s]) return;
    float t = 0;
    if (hipEventSynchronize(b[s]) == hipSuccess && hipEventElapsedTime(&t, a[s], b[s]) == hipSuccess) ms += t;
    live[s] = 0;
  }
  ~KernelClock() {
    for (auto e : a) (void)hipEventDestroy(e);
    for (auto e : b) (void)hipEventDestroy(e);
  }
};

int32_t drain(ibu_ctx* ctx, int32_t rc) {  // leave no copy or kernel in flight over ring memory
  (void)hipStreamSynchronize(ctx->copy_stream);
  (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

uint32_t feeder_threads(const ibu_ring_config_t* cfg) { return cfg && cfg->feeder_threads ? cfg->feeder_threads : 4; }

struct DeviceProc {  // the device-side ParallelProcessor applied to each staged slot
  ibu_ctx* ctx;
  int32_t kind;
  uint32_t bc_len, umi_len;
  ibu_decode_sink_t sink{};
  // the columns of a DECODE sink hold cap_records rows: a batch that does not fit is refused BEFORE anything is launched
  int32_t fits(size_t n, size_t row0) const {
    if (kind != IBU_PROC_DECODE || row0 + n <= sink.cap_records) return IBU_OK;
    return set_error(IBU_ERR_INVALID_ARG, row0 + n, sink.cap_records, 0,
                     "Invalid argument: decode sink holds %zu records, the stream has at least %zu", sink.cap_records, row0 + n);
  }
  int32_t launch(const uint8_t* d_slot, size_t n, size_t row0) {
    if (kind == IBU_PROC_REDUCE) {
      IBU_HIP(launch_reduce(ctx->cfg, d_slot, n, ctx->d_acc, ctx->stream));
    } else {
      IBU_HIP(launch_decode(ctx->cfg, d_slot, n, bc_len, umi_len,
                            sink.d_bc_ascii ? sink.d_bc_ascii + row0 * bc_len : nullptr,
                            sink.d_umi_ascii ? sink.d_umi_ascii + row0 * umi_len : nullptr,
                            sink.d_index ? sink.d_index + row0 : nullptr, ctx->stream));
    }
    return IBU_OK;
  }
};

int32_t make_proc(ibu_ctx* ctx, int32_t proc, const ibu_header_t& h, void* sink, DeviceProc* out) {
  if (!sink) return err_arg("sink is NULL");
  out->ctx = ctx;
  out->kind = proc;
  out->bc_len = h.bc_len;
  out->umi_len = h.umi_len;
  if (proc == IBU_PROC_DECODE) out->sink = *static_cast<ibu_decode_sink_t*>(sink);
  else if (proc != IBU_PROC_REDUCE) return err_arg("unknown device processor");
  return IBU_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// ring management
// ------------------------------------------------------------------------------------------
void ibu::ring_release(ibu_ctx* ctx) {
  Ring& r = ctx->ring;
  for (auto p : r.pinned)
    if (p) (void)hipHostFree(p);
  for (auto p : r.dev)
    if (p) (void)hipFree(p);
  for (auto e : r.copied) (void)hipEventDestroy(e);
  for (auto e : r.consumed) (void)hipEventDestroy(e);
  r = Ring();
}

int32_t ibu::ring_ensure(ibu_ctx* ctx, const ibu_ring_config_t* cfg, bool need_dev) {
  uint32_t slots = cfg && cfg->slots ? cfg->slots : 4;
  if (slots < 2) slots = 2;
  size_t slot_records = cfg && cfg->slot_records ? cfg->slot_records : 4u * IBU_BATCH_SIZE;
  slot_records = (slot_records + 127) & ~(size_t)127;  // whole kernel tiles, 16-B aligned column offsets
  const size_t slot_bytes = slot_records * IBU_RECORD_SIZE;
  Ring& r = ctx->ring;
  if (ctx->ring_lent) return err_arg("the context's ring is lent to an open ibu_stream_t: close the stream first (or use a second context)");
  if (r.slots == slots && r.slot_bytes == slot_bytes && (!need_dev || !r.dev.empty())) return IBU_OK;
  (void)hipStreamSynchronize(ctx->copy_stream);
  (void)hipStreamSynchronize(ctx->stream);
  ring_release(ctx);
  r.pinned.assign(slots, nullptr);
  r.copied.resize(slots);
  r.consumed.resize(slots);
  for (uint32_t i = 0; i < slots; ++i) {
    IBU_HIP(hipEventCreateWithFlags(&r.copied[i], hipEventDisableTiming));
    IBU_HIP(hipEventCreateWithFlags(&r.consumed[i], hipEventDisableTiming));
  }
  {
    // Option "numa": the pinned slots on the node the device hangs off.  Under a preferred-node policy of this thread the
    // allocation follows it (hipHostMallocNumaUser); a kernel that refuses the policy (a container's seccomp profile) leaves
    // the runtime's own choice, as before.  Either way ring.node says where the pages are, if the kernel will tell.
    PreferNode prefer(ctx->numa_mode ? ctx->place.node : -1);
    const unsigned flags = prefer.active() ? hipHostMallocNumaUser : hipHostMallocDefault;
    for (uint32_t i = 0; i < slots; ++i)
      IBU_HIP(hipHostMalloc(reinterpret_cast<void**>(&r.pinned[i]), slot_bytes, flags));
    r.placed = prefer.active();
  }
  r.node = node_of_range(r.pinned[0], slot_bytes);
  if (need_dev) {
    r.dev.assign(slots, nullptr);
    for (uint32_t i = 0; i < slots; ++i) IBU_HIP(ctx_malloc(ctx, reinterpret_cast<void**>(&r.dev[i]), slot_bytes));
  }
  r.slots = slots;
  r.slot_bytes = slot_bytes;
  return IBU_OK;
}

// ------------------------------------------------------------------------------------------
// load_to_vec, device form
// ------------------------------------------------------------------------------------------
extern "C" int32_t ibu_load_to_device(ibu_ctx_t* ctx, const char* path, const ibu_ring_config_t* cfg,
                                      ibu_header_t* header, void** d_records, size_t cap_records, size_t* n,
                                      ibu_stream_stats_t* stats) {
  if (!ctx || !path || !header || !d_records || !n) return err_arg("NULL argument");
  IBU_HIP(hipSetDevice(ctx->device));
  RunOnNode on_node(feed_place(ctx));   // for the length of the call this thread and the pread threads it starts run on the device's node (option "numa")
  const double t0 = now_s();
  int fd = -1;
  size_t num = 0;
  int32_t rc = open_plain_file(path, &fd, header, &num);
  if (rc) return rc;
  bool owned = false;
  if (*d_records == nullptr) {
    rc = ctx_alloc(ctx, num * IBU_RECORD_SIZE, d_records);   // (placement-probed under option "alloc_probe_tries")
    if (rc) {
      close(fd);
      return rc;
    }
    owned = true;
  } else if (num > cap_records) {
    close(fd);
    return err_arg("device buffer too small for the file");
  }
  rc = ring_ensure(ctx, cfg, false);
  Ring& r = ctx->ring;
  const size_t slot_records = r.slot_bytes / IBU_RECORD_SIZE;
  if (stats) memset(stats, 0, sizeof *stats);
  for (size_t done = 0, k = 0; rc == IBU_OK && done < num; ++k) {
    const uint32_t s = (uint32_t)(k % r.slots);
    const size_t nb = num - done < slot_records ? num - done : slot_records;
    const size_t bytes = nb * IBU_RECORD_SIZE;
    hipError_t e = hipEventSynchronize(r.copied[s]);  // slot's previous H2D has left the pinned buffer
    if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; }
    const off_t base = (off_t)(IBU_HEADER_SIZE + done * IBU_RECORD_SIZE);
    uint8_t* dst = r.pinned[s];
    int err = parallel_bytes(bytes, feeder_threads(cfg), [&](size_t off, size_t len) {
      return pread_all(fd, dst + off, len, base + (off_t)off);
    });
    if (err) { rc = err_io(err, "read records"); break; }
    e = hipMemcpyAsync(static_cast<uint8_t*>(*d_records) + done * IBU_RECORD_SIZE, dst, bytes, hipMemcpyHostToDevice,
                       ctx->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(r.copied[s], ctx->copy_stream);
    if (e != hipSuccess) { rc = hip_fail(e, "H2D"); break; }
    if (stats) { stats->bytes_h2d += bytes; stats->batches += 1; }
    done += nb;
  }
  close(fd);
  hipError_t e = hipStreamSynchronize(ctx->copy_stream);
  if (rc == IBU_OK && e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
  if (rc != IBU_OK) {
    if (owned) { (void)hipFree(*d_records); *d_records = nullptr; }
    return rc;
  }
  *n = num;
  if (stats) { stats->records = num; stats->seconds_total = now_s() - t0; }
  return IBU_OK;
}

// ------------------------------------------------------------------------------------------
// load_to_vec of a BGZF file, inflated on the device (k_inflate.hip): the COMPRESSED bytes cross the link
// ------------------------------------------------------------------------------------------
// The result is what ibu_load_to_device gives for the gunzipped file (load_to_vec, reader.rs:510-535: header read and validated,
// (length - 32) % 24 != 0 -> InvalidMapSize).  The file is mapped; a thread walks its block headers (ibu_bgzf_scan: no inflating; large
// files in eight pieces side by side) and inflates the blocks that hold the 32 header bytes on the host, while the calling thread
// already sends the file through the pinned ring into a device buffer; the device then inflates every block straight to its place in
// the records (the launch policy: further down).  A block is accepted exactly as the host
// decoder accepts it; anything else — a member that is not a BGZF block, a file that ends inside one, a block that does not inflate
// to its announced length and CRC — is IBU_ERR_NIFFLER, as from the Reader.
extern "C" int32_t ibu_load_bgzf_to_device(ibu_ctx_t* ctx, const char* path, const ibu_ring_config_t* cfg, ibu_header_t* header,
                                           void** d_records, size_t cap_records, size_t* n, ibu_stream_stats_t* stats) {
  return ibu_load_bgzf_shard_to_device(ctx, path, cfg, 0, 1, header, d_records, cap_records, n, nullptr, stats);
}
// Shard `shard` of `n_shards` of the file's records (the split of process_parallel, mmap.rs:297-307): the blocks that lie wholly inside
// the shard's bytes are copied and inflated on the device, the (at most two) blocks that straddle its ends are inflated on the host and
// their part copied — as the blocks holding the header always are.  Every device of a node loads its own range of the same file.
extern "C" int32_t ibu_load_bgzf_shard_to_device(ibu_ctx_t* ctx, const char* path, const ibu_ring_config_t* cfg, size_t shard, size_t n_shards,
                                                 ibu_header_t* header, void** d_records, size_t cap_records, size_t* n, uint64_t* first_record,
                                                 ibu_stream_stats_t* stats) {
  if (!ctx || !path || !header || !d_records || !n) return err_arg("NULL argument");
  if (n_shards == 0 || shard >= n_shards) return err_arg("shard out of range");
  IBU_HIP(hipSetDevice(ctx->device));
  RunOnNode on_node(feed_place(ctx));
  const double t0 = now_s();
  if (stats) memset(stats, 0, sizeof *stats);
  int fd = ::open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return err_io(errno, path);
  struct stat sb;
  if (fstat(fd, &sb)) { const int e = errno; close(fd); return err_io(e, "metadata"); }
  const size_t size = (size_t)sb.st_size;
  if (size == 0) { close(fd); return err_io(0, "read header"); }
  void* mp = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  const int map_errno = errno;
  close(fd);
  if (mp == MAP_FAILED) return err_io(map_errno, "mmap");
  struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } unmap{mp, size};
  (void)madvise(mp, size, MADV_SEQUENTIAL);
  const uint8_t* map = static_cast<const uint8_t*>(mp);
  double t_mark = now_s(), t_ph[4] = {0, 0, 0, 0};        // (IBU_TRACE_SORT=1: where the call's time went)
  auto lap = [&](int k) { const double t = now_s(); t_ph[k] += t - t_mark; t_mark = t; };

  // The WALK over the block headers (and the blocks that hold the 32 header bytes, inflated on the host) runs on a thread of its own
  // while this one already copies the file to the device: the copies need nothing but the file's size.  (In line, the walk's 12.5 ms
  // of page faults stood in front of a call of 1e8 records that takes 88.)
  std::vector<ibu_inflate_block_t> B;
  uint64_t total = 0;
  uint8_t head[IBU_HEADER_SIZE + 65536];
  size_t lead = 0, lead_bytes = 0;
  int32_t walk_rc = IBU_OK;
  bool walked_in_pieces = false;
  ibu_error_detail_t walk_detail{};
  std::atomic<bool> walked{false};
  // The walk, in pieces side by side.  A block's start cannot be computed without the blocks in front of it, but it can be GUESSED: every
  // piece but the first looks for the 16 bytes a bgzip header begins with (1f 8b 08 04 .. 06 00 'B' 'C' 02 00) at or behind its first
  // byte and walks the chain from there to the end of its piece.  The guesses are then checked: piece i's chain must END exactly where
  // piece i + 1's began — where it does not (the signature inside compressed data, an unusual extra field), or anything at all is off,
  // the plain walk from byte 0 below decides, errors included.  (184 k blocks of a 6 GB file: 60 ms of page faults in one thread.)
  auto walk_pieces = [&]() -> bool {
    const size_t T = 8;
    if (size < (T << 22)) return false;                    // (small files: the plain walk)
    struct Piece { size_t begin = 0, end = 0; bool ok = false; uint64_t out = 0; std::vector<ibu_inflate_block_t> blocks; };
    std::vector<Piece> pc(T);
    auto run = [&](size_t i) {
      Piece& P = pc[i];
      try {
        const size_t lo = size / T * i, hi = i + 1 == T ? size : size / T * (i + 1);
        size_t pos = lo;
        if (i) {                                           // the first header-like spot at or behind lo
          const uint8_t sig_a[4] = {0x1f, 0x8b, 0x08, 0x04}, sig_b[6] = {0x06, 0x00, 'B', 'C', 0x02, 0x00};
          for (;; ++pos) {
            if (pos + 18 > size || pos >= hi) return;      // none in this piece: give up (the plain walk decides)
            if (memcmp(map + pos, sig_a, 4) == 0 && memcmp(map + pos + 10, sig_b, 6) == 0) break;
          }
        }
        P.begin = pos;
        std::vector<ibu_inflate_block_t> part(1 << 14);
        while (pos < hi) {
          size_t nb = 0, consumed = 0, cap = part.size();
          uint64_t ob = 0;
          if (ibu_bgzf_scan(map + pos, size - pos, 1, part.data(), cap, &nb, &consumed, &ob) != IBU_OK || consumed == 0) return;
          size_t keep = 0, bytes = 0;                      // only the blocks that START inside the piece
          uint64_t outb = 0;
          for (; keep < nb; ++keep) {
            const size_t start = pos + (size_t)part[keep].comp_offset - 18;   // (bgzip's header: 18 bytes; checked again when the pieces are joined)
            if (start >= hi) break;
            bytes = (size_t)part[keep].comp_offset + part[keep].comp_len + 8;
            part[keep].comp_offset += pos;
            part[keep].out_offset += (int64_t)P.out;
            outb = (uint64_t)(part[keep].out_offset - (int64_t)P.out) + part[keep].out_len;
          }
          P.blocks.insert(P.blocks.end(), part.begin(), part.begin() + (ptrdiff_t)keep);
          P.out += outb;
          pos += bytes;
          if (keep < nb || keep == 0) break;
        }
        P.end = pos;
        P.ok = true;
      } catch (...) {}
    };
    {
      std::vector<std::thread> th;
      try { for (size_t i = 1; i < T; ++i) th.emplace_back(run, i); } catch (...) {}
      const size_t started = th.size();
      run(0);
      for (auto& t : th) t.join();
      if (started != T - 1) return false;
    }
    size_t nblocks = 0;
    for (size_t i = 0; i < T; ++i) {
      if (!pc[i].ok || (i == 0 && pc[i].begin != 0) || (i && pc[i].begin != pc[i - 1].end)) return false;
      nblocks += pc[i].blocks.size();
    }
    if (pc[T - 1].end != size) return false;
    B.reserve(nblocks);
    for (size_t i = 0; i < T; ++i) {
      for (ibu_inflate_block_t& b : pc[i].blocks) b.out_offset += (int64_t)total;
      B.insert(B.end(), pc[i].blocks.begin(), pc[i].blocks.end());
      total += pc[i].out;
    }
    return true;
  };
  auto walk = [&]() -> int32_t {
    try {
      bool have = false;
      try { have = walk_pieces(); } catch (...) { have = false; }
      if (!have) { B.clear(); total = 0; }
      walked_in_pieces = have;
      std::vector<ibu_inflate_block_t> part(have ? 1 : 1 << 16);
      for (size_t pos = have ? size : 0; pos < size;) {    // 1. the blocks (a cut-off or foreign member: IBU_ERR_NIFFLER from the walk)
        size_t nb = 0, consumed = 0;
        uint64_t ob = 0;
        const int32_t rc = ibu_bgzf_scan(map + pos, size - pos, 1, part.data(), part.size(), &nb, &consumed, &ob);
        for (size_t i = 0; i < nb; ++i) {
          part[i].comp_offset += pos;
          part[i].out_offset += (int64_t)total;
        }
        B.insert(B.end(), part.begin(), part.begin() + (ptrdiff_t)nb);
        if (rc) return rc;
        if (consumed == 0) return err_niffler("the stream ends inside a BGZF block");
        pos += consumed;
        total += ob;
      }
      pgz::RawInflater raw;                                // 2. the header: the leading blocks, inflated here
      std::vector<uint8_t> in;
      while (lead_bytes < IBU_HEADER_SIZE && lead < B.size()) {
        const ibu_inflate_block_t& b = B[lead];
        if (b.out_len) {
          in.assign(map + b.comp_offset, map + b.comp_offset + b.comp_len);
          in.resize(b.comp_len + 512, 0);                  // the decoder may read (not use) a few bytes behind the stream
          uint32_t crc = 0;
          const int e = raw.inflate(in.data(), b.comp_len, head + lead_bytes, b.out_len, &crc);
          if (e == ENOMEM) return err_io(ENOMEM, "inflate");
          if (e || crc != b.crc32) return err_niffler("a BGZF block does not inflate to its announced length and CRC-32");
        }
        lead_bytes += b.out_len;
        ++lead;
      }
    } catch (...) { return caught_io("ibu_load_bgzf_to_device"); }
    if (lead_bytes < IBU_HEADER_SIZE) return err_io(0, "read header");
    memcpy(header, head, IBU_HEADER_SIZE);
    const int32_t rc = ibu_header_validate(header);
    if (rc) return rc;
    if ((total - IBU_HEADER_SIZE) % IBU_RECORD_SIZE != 0) return err_map_size();
    return IBU_OK;
  };
  std::thread walker;
  auto run_walk = [&] {
    walk_rc = walk();
    if (walk_rc) walk_detail = tls_error();                // (the detail lives in the walker's thread: the caller gets a copy)
    walked.store(true, std::memory_order_release);
  };
  try { walker = std::thread(run_walk); } catch (...) { run_walk(); }   // no thread to be had: in line
  struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_walker{walker};

  bool owned = false;
  constexpr int kStreams = 3;
  hipStream_t* ks = ctx->inflate_streams;
  uint64_t* d_ready = nullptr;                             // (a launch that runs ahead of its input: see below)
  bool ahead = false;
  constexpr size_t kInflateMarks = 8;
  auto fail = [&](int32_t code) {
    if (walker.joinable()) walker.join();
    if (ahead && d_ready)                                  // its waves must not wait for bytes that will not come
      __atomic_store_n(d_ready, ~0ull, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(ctx->copy_stream);
    for (int k = 0; k < kStreams; ++k)
      if (ks[k]) (void)hipStreamSynchronize(ks[k]);
    if (owned) { (void)hipFree(*d_records); *d_records = nullptr; }
    return code;
  };
  // the staging on the device — the compressed file; the descriptors and status words behind it once their number is known — is the
  // context's and grows only (freeing 1.2 GB and allocating it again cost a call of 1e8 records 12 of its 104 ms)
  auto stage = [&](size_t need) -> int32_t {
    if (need <= ctx->inflate_stage_bytes) return IBU_OK;
    void* p = nullptr;
    hipError_t e = ctx_malloc(ctx, &p, need);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    if (ctx->d_inflate_stage) {                            // (the bytes copied so far move along)
      e = hipMemcpyAsync(p, ctx->d_inflate_stage, ctx->inflate_stage_bytes, hipMemcpyDeviceToDevice, ctx->copy_stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
      (void)hipFree(ctx->d_inflate_stage);
      if (e != hipSuccess) { (void)hipFree(p); ctx->d_inflate_stage = nullptr; ctx->inflate_stage_bytes = 0; return hip_fail(e, "hipMemcpy"); }
    }
    ctx->d_inflate_stage = p;
    ctx->inflate_stage_bytes = need;
    return IBU_OK;
  };
  // (one shard of one: room for the whole file and the descriptors of a file of ordinary 64 KiB blocks right away, so that the copies
  // can start at once and nothing is allocated a second time in the usual case)
  int32_t rc = n_shards == 1 ? stage(((size + kInflatePad + 255) & ~(size_t)255) + 40 * (size / 8192 + 64)) : IBU_OK;
  if (!rc) rc = ensure_sort_scratch(ctx, 16);
  if (!rc) rc = ring_ensure(ctx, cfg, false);
  hipError_t e = hipSuccess;
  for (int k = 0; k < kStreams && e == hipSuccess && !rc; ++k)
    if (!ks[k]) e = hipStreamCreateWithFlags(&ks[k], hipStreamNonBlocking);
  if (!rc && e != hipSuccess) rc = hip_fail(e, "hipStreamCreate");
  if (rc) return fail(rc);
  Ring& r = ctx->ring;
  lap(0);

  // What the walk's result allows, once it is there: the shard's range, the destination, the descriptors on the device
  size_t num = 0, nrest = 0, dev_first = 0;                // device blocks: B[dev_first, dev_first + nrest)
  size_t cbeg = 0, cend = size;                            // the file bytes that go to the device (one shard of one: all, so that the copies
  uint64_t rec_first = 0;                                  // can start before the walk is done)
  uint8_t* d_out = nullptr;
  InflateBlockDesc* d_desc = nullptr;
  uint32_t *d_status = nullptr, *d_first_bad = nullptr;
  uint8_t* d_tables = nullptr;                             // the lanes' tables of a launch in the decoder's scratch form, behind the status words
  size_t tables_room = 0;
  const uint32_t none = 0xFFFFFFFFu;
  bool prepared = false;
  auto prepare = [&]() -> int32_t {
    if (walker.joinable()) walker.join();
    if (walk_rc) { tls_error() = walk_detail; return walk_rc; }
    const size_t num_all = (size_t)((total - IBU_HEADER_SIZE) / IBU_RECORD_SIZE);
    size_t rs = 0, re = 0;
    int32_t prc = ibu_shard_range(num_all, n_shards, shard, &rs, &re);
    if (prc) return prc;
    num = re - rs;
    rec_first = rs;
    const uint64_t lo = IBU_HEADER_SIZE + (uint64_t)IBU_RECORD_SIZE * rs, hi = IBU_HEADER_SIZE + (uint64_t)IBU_RECORD_SIZE * re;   // the shard's bytes
    if (*d_records == nullptr) {
      const int32_t arc = ctx_alloc(ctx, num * IBU_RECORD_SIZE, d_records);
      if (arc) return arc;
      owned = true;
    } else if (num > cap_records) {
      return set_error(IBU_ERR_INVALID_ARG, num, cap_records, 0, "Invalid argument: device buffer too small for the shard (%zu records, room for %zu)", num, cap_records);
    }
    d_out = static_cast<uint8_t*>(*d_records);
    // the blocks wholly inside [lo, hi): the device's; what straddles an end (and the header's blocks): inflated here
    size_t dev_end = lead;
    dev_first = lead;
    while (dev_first < B.size() && (uint64_t)B[dev_first].out_offset < lo) ++dev_first;
    dev_end = dev_first;
    while (dev_end < B.size() && (uint64_t)B[dev_end].out_offset + B[dev_end].out_len <= hi) ++dev_end;
    nrest = dev_end - dev_first;
    hipError_t pe = hipSuccess;
    auto put = [&](const uint8_t* bytes, uint64_t at, uint64_t len) {   // bytes [at, at + len) of the stream, as far as they are the shard's
      const uint64_t a = at < lo ? lo : at, z = at + len > hi ? hi : at + len;
      if (a < z && pe == hipSuccess) pe = hipMemcpy(d_out + (a - lo), bytes + (a - at), z - a, hipMemcpyHostToDevice);
    };
    put(head, 0, lead_bytes);                              // the records behind the header in the blocks inflated for it
    {
      pgz::RawInflater raw;
      std::vector<uint8_t> in, outb(65536);
      const size_t edge[2] = {dev_first > lead ? dev_first - 1 : (size_t)-1, dev_end < B.size() ? dev_end : (size_t)-1};
      for (int k = 0; k < 2 && num; ++k) {
        const size_t i = edge[k];
        if (i == (size_t)-1 || i < lead || (k == 1 && edge[0] == i)) continue;
        const ibu_inflate_block_t& b = B[i];
        if (!b.out_len || (uint64_t)b.out_offset >= hi || (uint64_t)b.out_offset + b.out_len <= lo) continue;
        in.assign(map + b.comp_offset, map + b.comp_offset + b.comp_len);
        in.resize(b.comp_len + 512, 0);
        uint32_t crc = 0;
        const int ie = raw.inflate(in.data(), b.comp_len, outb.data(), b.out_len, &crc);
        if (ie == ENOMEM) return err_io(ENOMEM, "inflate");
        if (ie || crc != b.crc32) return err_niffler("a BGZF block does not inflate to its announced length and CRC-32");
        put(outb.data(), (uint64_t)b.out_offset, b.out_len);
      }
    }
    if (pe != hipSuccess) return hip_fail(pe, "hipMemcpy");
    if (n_shards > 1) {                                    // only the device blocks' bytes cross the link
      cbeg = nrest ? (size_t)B[dev_first].comp_offset : 0;
      cend = nrest ? (size_t)(B[dev_end - 1].comp_offset + B[dev_end - 1].comp_len) : 0;
    }
    const size_t comp_room = (cend - cbeg + kInflatePad + 255) & ~(size_t)255;
    const size_t desc_room = (nrest * sizeof(InflateBlockDesc) + 255) & ~(size_t)255;
    const size_t status_room = (4 * nrest + 16 + 255) & ~(size_t)255;
    tables_room = nrest > (ctx->inflate_one_launch ? ctx->inflate_one_launch : (size_t)ctx->cfg.cus * 3 * 64)
                      ? inflate_scratch_bytes(ctx->cfg, nrest, 2) : 256;   // (the short form needs none)
    const int32_t src = stage(comp_room + desc_room + status_room + 256 + tables_room);
    if (src) return src;
    d_desc = reinterpret_cast<InflateBlockDesc*>(static_cast<uint8_t*>(ctx->d_inflate_stage) + comp_room);
    d_status = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(d_desc) + desc_room);
    d_first_bad = d_status + nrest;
    d_tables = reinterpret_cast<uint8_t*>(d_status) + status_room + 256;
    if (!ctx->h_inflate_marks) {
      // (coherent whatever HIP_HOST_COHERENT says: the device must see the host's stores while its kernel runs)
      hipError_t he = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_inflate_marks), kInflateMarks * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent);
      if (he != hipSuccess) { ctx->h_inflate_marks = nullptr; return hip_fail(he, "hipHostMalloc"); }
    }
    d_ready = ctx->h_inflate_marks;                        // (pinned host memory: the device reads it over the link)
    for (size_t i = dev_first; i < dev_end; ++i) {         // relative to the shard's records / to the bytes on the device
      B[i].out_offset -= (int64_t)lo;
      B[i].comp_offset -= cbeg;
    }
    if (nrest) pe = hipMemcpy(d_desc, B.data() + dev_first, nrest * sizeof(InflateBlockDesc), hipMemcpyHostToDevice);
    if (pe == hipSuccess) pe = hipMemcpy(d_first_bad, &none, 4, hipMemcpyHostToDevice);
    __atomic_store_n(d_ready, 0ull, __ATOMIC_RELEASE);
    if (pe != hipSuccess) return hip_fail(pe, "hipMemcpy");
    prepared = true;
    return IBU_OK;
  };

  // The file's bytes to the device through the pinned ring; the blocks inflated where their records belong.  At most one round of the
  // decoder's short form (three waves of 64 blocks per CU: 49 152 blocks, 3 GB of records): ONE launch behind the last copy — a wave
  // takes its ~45 ms whatever the launch's size, so the call ends that long after its last byte has arrived either way.  More: ONE launch
  // as well, but AHEAD of the copies — as soon as the walk's results are on the device — in the decoder's other form (tables in scratch,
  // eight waves per CU): its waves take the blocks in file order and each waits until the copy stream has said that its blocks are there
  // (`d_ready`, written behind every piece; k_inflate.hip), so the device inflates at the rate the bytes come in.  (Launches behind the
  // copies instead — of 16 Ki or 32 Ki blocks on three streams, or of everything that had arrived once the one before was done — ran one
  // after the other, each for its waves' 45-75 ms: 5e8 records 0.343 / 0.306 / 0.228 s.)
  const size_t kOneLaunch = ctx->inflate_one_launch ? ctx->inflate_one_launch : (size_t)ctx->cfg.cus * 3 * 64;
  size_t up = 0, launches = 0;                             // up: file bytes [cbeg, up) are on their way
  uint32_t last_slot = 0;
  std::vector<size_t> piece_end;                           // piece k of the copies ends at this file byte
  auto publish = [&](size_t done_upto) {                   // the host has SEEN the copies up to this file byte complete: the launch may use them
    if (ahead) __atomic_store_n(d_ready, (uint64_t)(done_upto - cbeg), __ATOMIC_RELEASE);
  };
  auto launch_ready = [&](bool all) -> int32_t {
    if (launches || nrest == 0) return IBU_OK;
    const bool streamed = nrest > kOneLaunch && d_ready;
    if (!streamed && !all) return IBU_OK;
    hipStream_t q = ks[0];
    hipError_t le = hipSuccess;
    if (streamed) {
      ahead = true;
      le = launch_inflate_blocks(ctx->cfg, ctx->d_inflate_stage, d_desc, nrest, d_out, d_status, d_first_bad, d_tables, tables_room, q, 2, d_ready,
                                   cend - cbeg);
    } else {
      le = hipStreamWaitEvent(q, r.copied[last_slot], 0);
      if (le == hipSuccess)
        le = launch_inflate_blocks(ctx->cfg, ctx->d_inflate_stage, d_desc, nrest, d_out, d_status, d_first_bad, d_tables, tables_room, q,
                                   nrest > (size_t)ctx->cfg.cus * 3 * 64 ? 2 : 0);
    }
    if (le != hipSuccess) return hip_fail(le, "inflate");
    ++launches;
    return IBU_OK;
  };
  if (n_shards > 1) {                                      // a shard's bytes are known only after the walk
    rc = prepare();
    if (rc) return fail(rc);
  }
  up = cbeg;
  for (size_t k = 0; up < cend; ++k) {
    const uint32_t sl = (uint32_t)(k % r.slots);
    const size_t len = cend - up < r.slot_bytes ? cend - up : r.slot_bytes;
    if (ctx->load_piece_delay_ms) std::this_thread::sleep_for(std::chrono::milliseconds(ctx->load_piece_delay_ms));
    e = hipEventSynchronize(r.copied[sl]);
    if (e != hipSuccess) return fail(hip_fail(e, "hipEventSynchronize"));
    if (k >= r.slots) publish(piece_end[k - r.slots]);     // (this slot's previous piece has landed: so has everything in front of it)
    uint8_t* dst = r.pinned[sl];
    const uint8_t* src = map + up;
    parallel_bytes(len, feeder_threads(cfg), [&](size_t off, size_t l) { memcpy(dst + off, src + off, l); return 0; });
    e = hipMemcpyAsync(static_cast<uint8_t*>(ctx->d_inflate_stage) + (up - cbeg), dst, len, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(r.copied[sl], ctx->copy_stream);
    if (e != hipSuccess) return fail(hip_fail(e, "H2D"));
    up += len;
    last_slot = sl;
    if (stats) { stats->bytes_h2d += len; stats->batches += 1; }
    try { piece_end.push_back(up); } catch (...) { return fail(caught_io("ibu_load_bgzf_to_device")); }
    if (!prepared && walked.load(std::memory_order_acquire)) {
      rc = prepare();
      if (rc) return fail(rc);
    }
    if (prepared) {
      rc = launch_ready(false);
      if (rc) return fail(rc);
    }
  }
  lap(1);
  if (!prepared) {
    rc = prepare();
    if (rc) return fail(rc);
  }
  rc = launch_ready(true);
  if (rc) return fail(rc);
  lap(2);
  uint32_t first_bad = none;
  e = hipStreamSynchronize(ctx->copy_stream);
  if (e == hipSuccess) publish(cend);                      // every byte is there
  for (int k = 0; k < kStreams && e == hipSuccess; ++k) e = hipStreamSynchronize(ks[k]);
  if (e == hipSuccess) e = hipMemcpy(&first_bad, d_first_bad, 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(hip_fail(e, "ibu_load_bgzf_to_device"));
  if (first_bad != none && ahead) {
    // Waves of the launch that ran ahead give up after ~4 s without their blocks (status 3): a slow source, not a bad file.  Everything
    // is on the device now: those blocks — from the first of them on — go through a plain launch.  A block that was REFUSED stays refused.
    try {
      std::vector<uint32_t> stv(nrest);
      e = hipMemcpy(stv.data(), d_status, 4 * nrest, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return fail(hip_fail(e, "hipMemcpy"));
      size_t late = nrest;
      bool refused = false;
      for (size_t i = 0; i < nrest; ++i) {
        if (stv[i] == 3) { if (late == nrest) late = i; }
        else if (stv[i]) refused = true;
      }
      if (!refused && late < nrest) {
        if (trace_sort()) fprintf(stderr, "ibu load_bgzf: the bytes of blocks %zu ... came later than the waves waited: inflating them now\n", late);
        e = hipMemcpy(d_first_bad, &none, 4, hipMemcpyHostToDevice);
        ahead = false;                                      // (fail() has nothing to release any more)
        for (size_t at = late; at < nrest && e == hipSuccess;) {
          const size_t cnt = nrest - at < (size_t)ctx->cfg.cus * 8 * 64 ? nrest - at : (size_t)ctx->cfg.cus * 8 * 64;
          e = launch_inflate_blocks(ctx->cfg, ctx->d_inflate_stage, d_desc + at, cnt, d_out, d_status + at, d_first_bad, d_tables, tables_room, ks[0],
                                    cnt > (size_t)ctx->cfg.cus * 3 * 64 ? 2 : 0);
          at += cnt;
          ++launches;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ks[0]);
        if (e == hipSuccess) e = hipMemcpy(&first_bad, d_first_bad, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(hip_fail(e, "ibu_load_bgzf_to_device"));
      }
    } catch (...) { return fail(caught_io("ibu_load_bgzf_to_device")); }
  }
  if (first_bad != none) {
    if (trace_sort()) {
      uint32_t st1 = 0;
      (void)hipMemcpy(&st1, d_status + first_bad, 4, hipMemcpyDeviceToHost);
      fprintf(stderr, "ibu load_bgzf: block %u of the device's %zu refused (status %u: 1 not a deflate stream of these sizes, 2 CRC-32, 3 its bytes never arrived)\n",
              first_bad, nrest, st1);
    }
    return fail(err_niffler("a BGZF block does not inflate to its announced length and CRC-32"));
  }
  lap(3);
  if (trace_sort())
    fprintf(stderr, "ibu load_bgzf: %zu blocks (walked %s), %zu launches; ms: staging %.2f, copies (the walk beside them) and early launches %.2f, walk's results to the "
            "device + last launches %.2f, waiting for them %.2f\n", B.size(), walked_in_pieces ? "in 8 pieces side by side" : "in one go", launches, 1e3 * t_ph[0], 1e3 * t_ph[1], 1e3 * t_ph[2], 1e3 * t_ph[3]);
  *n = num;
  if (first_record) *first_record = rec_first;
  if (stats) { stats->records = num; stats->seconds_total = now_s() - t0; stats->numa_node = feed_place(ctx).node; stats->ring_node = ctx->ring.node; }
  return IBU_OK;
}

// ------------------------------------------------------------------------------------------
// Writer::write_batch, device form
// ------------------------------------------------------------------------------------------
extern "C" int32_t ibu_writer_write_batch_device(ibu_writer_t* w, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                                 const void* d_records, size_t n, ibu_stream_stats_t* stats) {
  return ibu_writer_write_batch_device_on(w, ctx, cfg, d_records, n, nullptr, stats);
}
extern "C" int32_t ibu_writer_write_batch_device_on(ibu_writer_t* w, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                                    const void* d_records, size_t n, void* producer_stream,
                                                    ibu_stream_stats_t* stats) {
  if (!w || !ctx || (!d_records && n)) return err_arg("NULL argument");
  IBU_HIP(hipSetDevice(ctx->device));
  RunOnNode on_node(feed_place(ctx));   // the copies pinned ring -> writer buffer / file on the device's node (option "numa")
  const double t0 = now_s();
  if (stats) memset(stats, 0, sizeof *stats);
  int32_t rc = ring_ensure(ctx, cfg, false);
  if (rc) return rc;
  Ring& r = ctx->ring;
  const size_t slot_records = r.slot_bytes / IBU_RECORD_SIZE;
  const size_t nchunks = (n + slot_records - 1) / slot_records;
  const uint8_t* src = static_cast<const uint8_t*>(d_records);
  // the records were produced on `producer_stream` (NULL: the context's own stream): order the copy stream behind it
  IBU_HIP(hipEventRecord(r.consumed[0], pick_stream(ctx, producer_stream)));
  IBU_HIP(hipStreamWaitEvent(ctx->copy_stream, r.consumed[0], 0));
  auto issue = [&](size_t c) -> int32_t {
    const uint32_t s = (uint32_t)(c % r.slots);
    const size_t row = c * slot_records;
    const size_t nb = n - row < slot_records ? n - row : slot_records;
    IBU_HIP(hipMemcpyAsync(r.pinned[s], src + row * IBU_RECORD_SIZE, nb * IBU_RECORD_SIZE, hipMemcpyDeviceToHost,
                           ctx->copy_stream));
    IBU_HIP(hipEventRecord(r.copied[s], ctx->copy_stream));
    return IBU_OK;
  };
  for (size_t c = 0; c < nchunks && c < r.slots && rc == IBU_OK; ++c) rc = issue(c);
  for (size_t c = 0; c < nchunks && rc == IBU_OK; ++c) {
    const uint32_t s = (uint32_t)(c % r.slots);
    const size_t row = c * slot_records;
    const size_t nb = n - row < slot_records ? n - row : slot_records;
    hipError_t e = hipEventSynchronize(r.copied[s]);
    if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; }
    rc = writer_write_bytes(w, r.pinned[s], nb * IBU_RECORD_SIZE);  // buffered / direct rule of writer.rs:321-351
    if (rc) break;
    if (stats) { stats->bytes_d2h += nb * IBU_RECORD_SIZE; stats->batches += 1; }
    if (c + r.slots < nchunks) rc = issue(c + r.slots);
  }
  if (rc) return drain(ctx, rc);
  if (stats) { stats->records = n; stats->seconds_total = now_s() - t0; }
  return IBU_OK;
}

// ------------------------------------------------------------------------------------------
// The pull stream: Reader::read_batch + Iterator (reader.rs:218-242, :279-306) and the per-batch loop of process_parallel
// (mmap.rs:312-320) with the batch in HBM
// ------------------------------------------------------------------------------------------
// A producer thread owns the source, the pinned slots and the copy stream; the consumer (the caller's thread) owns the order in
// which batches are taken and the streams that read them.  Slot s goes FREE -> FILLING -> (H2D queued) READY -> (next) HELD ->
// (release: `consumed[s]` recorded on the caller's stream) RELEASED -> (producer waits for that event) FILLING ...  The producer
// takes ANY slot the consumer does not hold, so batches held for long (at most slots - 1) never stall the others.
struct ibu_stream {
  ibu_ctx* ctx = nullptr;
  ibu_header_t header{};
  ibu_reader_t* rd = nullptr;        // source: a borrowed Reader ...
  const ibu_mmap_t* m = nullptr;     // ... or records [start, end) of a map
  size_t start = 0, end = 0;
  uint32_t feeders = 4;
  size_t slot_records = 0;
  uint64_t first0 = 0;               // number of the stream's first record (mmap: start; reader: 0)
  double t0 = 0;
  std::mutex mu;
  std::condition_variable cv;
  enum : uint8_t { FREE, FILLING, READY, HELD, RELEASED };
  struct Slot { uint8_t state = FREE; size_t n = 0; uint64_t first = 0, seq = 0; };   // seq: when it was released (oldest refilled first)
  uint64_t release_seq = 0;
  std::vector<Slot> slot;
  std::deque<uint32_t> ready;        // READY slots in stream order
  uint32_t held = 0;
  bool stop = false, done = false, started = false;
  int32_t rc = IBU_OK;               // the source's error, delivered after the batches in front of it
  ibu_error_detail_t detail{};
  ibu_stream_stats_t stats{};
  std::vector<uint8_t> stage;        // Reader sources with slots smaller than one refill: the refill being handed out piecewise
  size_t stage_pos = 0, stage_len = 0;   // records
  bool src_eof = false;              // the Reader source reported its end
  std::thread producer;
};

namespace {

// One slot's worth of a Reader source into `dst`: *filled records to deliver (possibly > 0 together with an error: the batch in
// front of a truncation), *eof = the source has ended.
// The stream delivers EXACTLY the records the reference's iterator yields, also in front of an error.  The reference refills
// IBU_DEFAULT_BUFFER_SIZE (= 49 152 records) at a time and a source that ends inside a record loses its whole final refill
// (reader.rs:232-237, quirk Q8); a record may therefore only be handed out once the refill it belongs to has arrived whole.
// So the stream reads in whole refills, counted from where it took over (the reader's own buffer empty = a refill boundary):
//   a slot of at least one refill asks the source for floor(room / refill) refills in one call, straight into the pinned slot
//     (parallel preads of a plain file, the inflate threads' own copies; no detour through the reader's buffer) — every batch
//     ends on a refill boundary;
//   a smaller slot (test rings) goes through a one-refill staging buffer.
int32_t fill_from_reader(ibu_stream* s, uint8_t* dst, size_t* filled, bool* eof) {
  ibu_reader_t* rd = s->rd;
  const size_t cap = s->slot_records;
  constexpr size_t kRefill = IBU_DEFAULT_BUFFER_SIZE, kRefillRecords = kRefill / IBU_RECORD_SIZE;
  size_t n = 0;
  *filled = 0;
  for (;;) {
    const ibu_record_t* recs;
    size_t have = 0;
    ibu_reader_buffered(rd, &recs, &have);
    if (have == 0) break;            // records the caller had pulled into the reader's buffer before the stream took over go first
    const size_t take = have < cap - n ? have : cap - n;
    memcpy(dst + n * IBU_RECORD_SIZE, recs, take * IBU_RECORD_SIZE);
    ibu_reader_consume(rd, take);
    n += take;
    if (n == cap) { *filled = n; return IBU_OK; }
  }
  if (s->stage_pos < s->stage_len) {                   // small slots: the rest of the staged refill
    const size_t take = s->stage_len - s->stage_pos < cap - n ? s->stage_len - s->stage_pos : cap - n;
    memcpy(dst + n * IBU_RECORD_SIZE, s->stage.data() + s->stage_pos * IBU_RECORD_SIZE, take * IBU_RECORD_SIZE);
    s->stage_pos += take;
    *filled = n + take;
    *eof = s->src_eof && s->stage_pos == s->stage_len;
    return IBU_OK;
  }
  if (s->src_eof) { *filled = n; *eof = true; return IBU_OK; }
  const size_t room = cap - n;
  if (room >= kRefillRecords) {
    const size_t ask = room / kRefillRecords * kRefill;
    size_t got = 0;
    const int32_t rc = reader_read_direct(rd, dst + n * IBU_RECORD_SIZE, ask, &got, &s->src_eof);
    if (rc) {                                          // a stream that ends inside a record, or a source error (reader.rs:225-237): `got` = the
      *filled = n + got / kRefill * kRefillRecords;    // complete record bytes in front of it — their WHOLE refills go out, as the reference's
      *eof = true;                                     // iterator has yielded them by then; the refill under way is lost with the error
      return rc;
    }
    *filled = n + got / IBU_RECORD_SIZE;
    *eof = s->src_eof;               // a short read is the end of the source: this batch is the last
    return IBU_OK;
  }
  if (n) { *filled = n; return IBU_OK; }               // buffered records left less than a refill of room: a short batch
  try {
    if (s->stage.size() < kRefill) s->stage.resize(kRefill);
  } catch (...) {
    *eof = true;
    return caught_io("ibu_stream: staging buffer");
  }
  size_t got = 0;
  const int32_t rc = reader_read_direct(rd, s->stage.data(), kRefill, &got, &s->src_eof);
  if (rc) { *eof = true; return rc; }                  // truncated: the final refill is dropped whole
  s->stage_len = got / IBU_RECORD_SIZE;
  s->stage_pos = s->stage_len < cap ? s->stage_len : cap;
  memcpy(dst, s->stage.data(), s->stage_pos * IBU_RECORD_SIZE);
  *filled = s->stage_pos;
  *eof = s->src_eof && s->stage_pos == s->stage_len;
  return IBU_OK;
}

void stream_produce(ibu_stream* s) {
  ibu_ctx* ctx = s->ctx;
  Ring& r = ctx->ring;
  (void)pthread_setname_np(pthread_self(), "ibu-feed");
  RunOnNode on_node(feed_place(ctx));   // this thread and every thread it starts (feeders, inflate workers) on the device's node
  {
    std::lock_guard<std::mutex> g(s->mu);
    s->started = true;               // name and affinity are in place: ibu_stream_open_* returns only now (what ibu_ctx_numa and a
    s->cv.notify_all();              // look at /proc/self/task say is true from the first moment the caller holds the handle)
  }
  int32_t rc = IBU_OK;
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) rc = hip_fail(e, "hipSetDevice");
  const uint8_t* map = s->m ? static_cast<const uint8_t*>(ibu_mmap_base(s->m)) + IBU_HEADER_SIZE : nullptr;
  uint64_t delivered = 0;
  size_t row = s->start;
  bool eof = s->m ? s->start >= s->end : false;
  while (rc == IBU_OK && !eof) {
    // any slot the consumer does not hold will do (never-used ones first, then the one released longest ago): with slots - 1 batches
    // held the one slot left keeps the stream moving — filling in ring order would wait for a HELD slot while a free one sat idle
    uint32_t si = 0;
    bool released = false;
    {
      std::unique_lock<std::mutex> lk(s->mu);
      int pick = -1;
      s->cv.wait(lk, [&] {
        if (s->stop) return true;
        pick = -1;
        for (uint32_t i = 0; i < r.slots && pick < 0; ++i)
          if (s->slot[i].state == ibu_stream::FREE) pick = (int)i;
        if (pick < 0)
          for (uint32_t i = 0; i < r.slots; ++i)
            if (s->slot[i].state == ibu_stream::RELEASED && (pick < 0 || s->slot[i].seq < s->slot[pick].seq)) pick = (int)i;
        return pick >= 0;
      });
      if (s->stop) break;
      si = (uint32_t)pick;
      released = s->slot[si].state == ibu_stream::RELEASED;
      s->slot[si].state = ibu_stream::FILLING;
    }
    if (released) {                  // the consumer's work on the slot's previous batch (and so its H2D) is done
      e = hipEventSynchronize(r.consumed[si]);
      if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; }
    }
    size_t n = 0;
    int32_t src_rc = IBU_OK;
    if (s->m) {
      n = s->end - row < s->slot_records ? s->end - row : s->slot_records;
      const uint8_t* srcp = map + row * IBU_RECORD_SIZE;
      uint8_t* dst = r.pinned[si];
      parallel_bytes(n * IBU_RECORD_SIZE, s->feeders, [&](size_t off, size_t len) {
        memcpy(dst + off, srcp + off, len);  // page-cache / page-fault side of the reference's hot loop
        return 0;
      });
      row += n;
      eof = row >= s->end;
    } else {
      src_rc = fill_from_reader(s, r.pinned[si], &n, &eof);
    }
    if (n) {
      const size_t bytes = n * IBU_RECORD_SIZE;
      e = hipMemcpyAsync(r.dev[si], r.pinned[si], bytes, hipMemcpyHostToDevice, ctx->copy_stream);
      if (e == hipSuccess) e = hipEventRecord(r.copied[si], ctx->copy_stream);
      if (e != hipSuccess) { rc = hip_fail(e, "H2D"); break; }
      std::lock_guard<std::mutex> g(s->mu);
      s->slot[si].state = ibu_stream::READY;
      s->slot[si].n = n;
      s->slot[si].first = s->first0 + delivered;
      s->ready.push_back(si);
      s->stats.records += n;
      s->stats.bytes_h2d += bytes;
      s->stats.batches += 1;
      delivered += n;
      s->cv.notify_all();
    } else {
      std::lock_guard<std::mutex> g(s->mu);
      s->slot[si].state = ibu_stream::FREE;     // nothing came (the end, or an error with no batch in front of it)
    }
    if (src_rc) rc = src_rc;
  }
  std::lock_guard<std::mutex> g(s->mu);
  s->rc = rc;
  if (rc) s->detail = tls_error();   // the detail lives in THIS thread's slot: next() copies it into its caller's
  s->done = true;
  s->cv.notify_all();
}

int32_t stream_open(ibu_ctx* ctx, const ibu_ring_config_t* cfg, ibu_stream* s) {
  IBU_HIP(hipSetDevice(ctx->device));
  int32_t rc = ring_ensure(ctx, cfg, true);
  if (rc) return rc;
  Ring& r = ctx->ring;
  s->ctx = ctx;
  s->feeders = feeder_threads(cfg);
  s->slot_records = r.slot_bytes / IBU_RECORD_SIZE;
  s->t0 = now_s();
  s->stats.numa_node = feed_place(ctx).node;
  s->stats.ring_node = r.node;
  try {
    s->slot.assign(r.slots, ibu_stream::Slot());
    ctx->ring_lent = s;
    s->producer = std::thread(stream_produce, s);
  } catch (...) {
    ctx->ring_lent = nullptr;
    return caught_io("ibu_stream_open");
  }
  std::unique_lock<std::mutex> lk(s->mu);
  s->cv.wait(lk, [&] { return s->started; });
  return IBU_OK;
}

// The consumer side of next(): the oldest READY slot, the caller's stream ordered behind its copy.  *n == 0: end of stream.
int32_t stream_take(ibu_stream* s, hipStream_t st, uint32_t* slot_out, size_t* n, uint64_t* first) {
  Ring& r = s->ctx->ring;
  uint32_t si = 0;
  {
    std::unique_lock<std::mutex> lk(s->mu);
    for (;;) {
      if (!s->ready.empty()) break;
      if (s->done) {
        *n = 0;
        if (s->rc) { tls_error() = s->detail; return s->rc; }
        return IBU_OK;
      }
      if (s->held >= r.slots) return err_arg("every ring slot is held: release a batch before asking for the next");
      s->cv.wait(lk);
    }
    si = s->ready.front();
    s->ready.pop_front();
    s->slot[si].state = ibu_stream::HELD;
    ++s->held;
    *n = s->slot[si].n;
    *first = s->slot[si].first;
  }
  IBU_HIP(hipStreamWaitEvent(st, r.copied[si], 0));
  *slot_out = si;
  return IBU_OK;
}

int32_t stream_give_back(ibu_stream* s, uint32_t si, hipStream_t st) {
  Ring& r = s->ctx->ring;
  hipError_t e = hipEventRecord(r.consumed[si], st);
  {
    std::lock_guard<std::mutex> g(s->mu);
    s->slot[si].state = ibu_stream::RELEASED;   // (even when the record failed: the stream must be able to end)
    s->slot[si].seq = ++s->release_seq;
    --s->held;
    s->cv.notify_all();
  }
  if (e != hipSuccess) return hip_fail(e, "hipEventRecord");
  return IBU_OK;
}

void stream_shutdown(ibu_stream* s) {
  ibu_ctx* ctx = s->ctx;
  {
    std::lock_guard<std::mutex> g(s->mu);
    s->stop = true;
    s->cv.notify_all();
  }
  if (s->producer.joinable()) s->producer.join();
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->copy_stream);
  (void)hipStreamSynchronize(ctx->stream);
  for (uint32_t i = 0; i < s->slot.size(); ++i)   // work the caller queued on its own streams before releasing
    if (s->slot[i].state == ibu_stream::RELEASED) (void)hipEventSynchronize(ctx->ring.consumed[i]);
  ctx->ring_lent = nullptr;
}

}  // namespace

extern "C" int32_t ibu_stream_open_reader(ibu_reader_t* r, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg, ibu_stream_t** out) {
  if (!r || !ctx || !out) return err_arg("NULL argument");
  *out = nullptr;
  ibu_stream* s = new (std::nothrow) ibu_stream;
  if (!s) return err_io(ENOMEM, "ibu_stream_open_reader");
  s->rd = r;
  ibu_reader_header(r, &s->header);
  const int32_t rc = stream_open(ctx, cfg, s);
  if (rc) { delete s; return rc; }
  *out = s;
  return IBU_OK;
}

extern "C" int32_t ibu_stream_open_mmap(const ibu_mmap_t* m, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg, size_t shard,
                                        size_t n_shards, ibu_stream_t** out) {
  if (!m || !ctx || !out) return err_arg("NULL argument");
  *out = nullptr;
  size_t start = 0, end = 0;
  int32_t rc = ibu_shard_range(ibu_mmap_len(m), n_shards, shard, &start, &end);  // mmap.rs:297-307
  if (rc) return rc;
  ibu_stream* s = new (std::nothrow) ibu_stream;
  if (!s) return err_io(ENOMEM, "ibu_stream_open_mmap");
  s->m = m;
  s->start = start;
  s->end = end;
  s->first0 = start;
  ibu_mmap_header(m, &s->header);
  rc = stream_open(ctx, cfg, s);
  if (rc) { delete s; return rc; }
  *out = s;
  return IBU_OK;
}

extern "C" int32_t ibu_stream_header(const ibu_stream_t* s, ibu_header_t* out) {
  if (!s || !out) return err_arg("NULL argument");
  *out = s->header;
  return IBU_OK;
}

extern "C" int32_t ibu_stream_next(ibu_stream_t* s, void* stream, const void** d_records, size_t* n, uint64_t* first_index) {
  if (!s || !d_records || !n) return err_arg("NULL argument");
  *d_records = nullptr;
  *n = 0;
  if (!s->ctx) return err_arg("the stream's context has been destroyed");
  IBU_HIP(hipSetDevice(s->ctx->device));
  uint32_t si = 0;
  uint64_t first = 0;
  const int32_t rc = stream_take(s, pick_stream(s->ctx, stream), &si, n, &first);
  if (rc || *n == 0) return rc;
  *d_records = s->ctx->ring.dev[si];
  if (first_index) *first_index = first;
  return IBU_OK;
}

extern "C" int32_t ibu_stream_release(ibu_stream_t* s, const void* d_records, void* stream) {
  if (!s || !d_records) return err_arg("NULL argument");
  if (!s->ctx) return err_arg("the stream's context has been destroyed");
  IBU_HIP(hipSetDevice(s->ctx->device));
  Ring& r = s->ctx->ring;
  uint32_t si = r.slots;
  {
    std::lock_guard<std::mutex> g(s->mu);
    for (uint32_t i = 0; i < r.slots; ++i)
      if (r.dev[i] == d_records && s->slot[i].state == ibu_stream::HELD) si = i;
  }
  if (si == r.slots) return err_arg("not a batch this stream handed out and still holds");
  return stream_give_back(s, si, pick_stream(s->ctx, stream));
}

extern "C" int32_t ibu_stream_stats(const ibu_stream_t* s, ibu_stream_stats_t* out) {
  if (!s || !out) return err_arg("NULL argument");
  ibu_stream* m = const_cast<ibu_stream*>(s);
  std::lock_guard<std::mutex> g(m->mu);
  *out = m->stats;
  out->seconds_total = now_s() - m->t0;
  return IBU_OK;
}

extern "C" void ibu_stream_close(ibu_stream_t* s) {
  if (!s) return;
  if (s->ctx) stream_shutdown(s);    // (an orphan — its context was destroyed first — has been shut down already)
  delete s;
}
// ibu_ctx_destroy under an open stream: the producer is stopped and joined and the ring given back BEFORE the context frees it; the
// handle stays valid for ibu_stream_close (every other call on it reports the destroyed context).
void ibu::stream_orphan(ibu_ctx* ctx) {
  ibu_stream* s = static_cast<ibu_stream*>(ctx->ring_lent);
  if (!s) return;
  stream_shutdown(s);
  s->ctx = nullptr;
}

// ------------------------------------------------------------------------------------------
// The two built-in device processors over the pull stream: process_parallel (mmap.rs:286-332, one shard of the static split)
// and the streaming Reader (reader.rs:279-306, :345-352), device forms
// ------------------------------------------------------------------------------------------
namespace {
int32_t run_processor(ibu_stream* s, int32_t proc, void* sink, ibu_stream_stats_t* stats) {
  ibu_ctx* ctx = s->ctx;
  Ring& r = ctx->ring;
  DeviceProc dp;
  int32_t rc = make_proc(ctx, proc, s->header, sink, &dp);
  if (rc) return rc;
  KernelClock kc;
  rc = kc.init(r.slots);
  if (rc) return rc;
  if (proc == IBU_PROC_REDUCE) IBU_HIP(hipMemsetAsync(ctx->d_acc, 0, kReduceAccBytes, ctx->stream));
  for (;;) {
    uint32_t si = 0;
    size_t n = 0;
    uint64_t first = 0;
    rc = stream_take(s, ctx->stream, &si, &n, &first);
    if (rc || n == 0) break;
    const size_t row0 = (size_t)(first - s->first0);
    rc = dp.fits(n, row0);           // a gzip / BGZF / xz / zstd stream does not announce its length: every batch is checked
    if (rc == IBU_OK) {
      kc.harvest(si);                // the slot's previous kernel finished before the producer refilled it: no wait
      hipError_t e = hipEventRecord(kc.a[si], ctx->stream);
      if (e == hipSuccess) {
        rc = dp.launch(r.dev[si], n, row0);
        if (rc == IBU_OK) e = hipEventRecord(kc.b[si], ctx->stream);
      }
      if (rc == IBU_OK && e != hipSuccess) rc = hip_fail(e, "hipEventRecord");
      if (rc == IBU_OK) kc.live[si] = 1;
    }
    const int32_t rel = stream_give_back(s, si, ctx->stream);
    if (rc == IBU_OK) rc = rel;
    if (rc) break;
  }
  if (rc) return rc;
  if (proc == IBU_PROC_REDUCE) rc = ibu_reduce_fetch(ctx, ctx->stream, static_cast<ibu_reduce_result_t*>(sink));
  else if (hipError_t e = hipStreamSynchronize(ctx->stream); e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
  if (rc) return rc;
  for (uint32_t i = 0; i < r.slots; ++i) kc.harvest(i);
  if (stats) {
    std::lock_guard<std::mutex> g(s->mu);
    *stats = s->stats;
    stats->seconds_kernel = kc.ms * 1e-3;
  }
  return IBU_OK;
}
// open -> run -> close, with the error (and its detail) of the first failing step
int32_t process_stream(ibu_stream* s, int32_t open_rc, int32_t proc, void* sink, ibu_stream_stats_t* stats, double t0) {
  if (open_rc) return open_rc;
  const int32_t rc = run_processor(s, proc, sink, stats);
  const ibu_error_detail_t keep = tls_error();
  ibu_stream_close(s);               // drains the copy stream and the context's stream: nothing is in flight over ring memory
  if (rc) { tls_error() = keep; return rc; }
  if (stats) stats->seconds_total = now_s() - t0;
  return IBU_OK;
}
}  // namespace

extern "C" int32_t ibu_mmap_process_device(const ibu_mmap_t* m, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                           int32_t proc, size_t shard, size_t n_shards, void* sink,
                                           ibu_stream_stats_t* stats) {
  if (!m || !ctx) return err_arg("NULL argument");
  const double t0 = now_s();
  if (stats) memset(stats, 0, sizeof *stats);
  size_t start = 0, end = 0;
  int32_t rc = ibu_shard_range(ibu_mmap_len(m), n_shards, shard, &start, &end);  // mmap.rs:297-307
  if (rc) return rc;
  ibu_header_t h;
  ibu_mmap_header(m, &h);
  DeviceProc dp;
  rc = make_proc(ctx, proc, h, sink, &dp);
  if (rc) return rc;
  rc = dp.fits(end - start, 0);   // the shard's size is known up front: refused before any work starts
  if (rc) return rc;
  ibu_stream_t* s = nullptr;
  rc = ibu_stream_open_mmap(m, ctx, cfg, shard, n_shards, &s);
  return process_stream(s, rc, proc, sink, stats, t0);
}

// ------------------------------------------------------------------------------------------
// MmapReader::process_parallel across DEVICES in one call (mmap.rs:286-332 with a GPU per worker)
// ------------------------------------------------------------------------------------------
// The reference's loop: n workers, worker i owns shard i of the static split (mmap.rs:297-307), workers are joined in spawn
// order and the first Err in that order is the call's result (quirk Q12).  Here a worker is one host thread driving one
// context (= one device): ibu_mmap_process_device(m, ctx_i, cfg, proc, i, n, sink_i).  No data-path collective exists: the
// only cross-device value is the reduce processor's {count, 3 wrapping sums, 3 XORs}, seven words per device, added on the
// host by the calling thread once the workers are joined (latency-bound, the xGMI links play no part).
// A worker's error detail lives in ITS thread's slot; the winner's is copied into the caller's.
namespace {
int32_t process_contexts(const ibu_mmap_t* m, ibu_ctx_t* const* ctxs, size_t n, const ibu_ring_config_t* cfg, int32_t proc,
                         void* sinks, ibu_reduce_result_t* total, ibu_stream_stats_t* stats) {
  if (proc != IBU_PROC_REDUCE && proc != IBU_PROC_DECODE) return err_arg("unknown device processor");
  if (proc == IBU_PROC_DECODE && !sinks) return err_arg("IBU_PROC_DECODE needs one ibu_decode_sink_t per device");
  std::vector<int32_t> rc;
  std::vector<ibu_error_detail_t> detail;
  std::vector<ibu_reduce_result_t> part;
  try {
    rc.assign(n, IBU_OK);
    detail.resize(n);
    part.resize(n);
  } catch (...) {
    return caught_io("ibu_mmap_process_devices");
  }
  for (auto& r : part) memset(&r, 0, sizeof r);
  run_pieces((unsigned)n, [&](unsigned i) {     // never throws; a thread that cannot start runs on the caller (common.hpp)
    void* sink = proc == IBU_PROC_REDUCE ? static_cast<void*>(&part[i]) : static_cast<void*>(static_cast<ibu_decode_sink_t*>(sinks) + i);
    rc[i] = ibu_mmap_process_device(m, ctxs[i], cfg, proc, i, n, sink, stats ? stats + i : nullptr);
    if (rc[i] != IBU_OK) detail[i] = tls_error();
  });
  for (size_t i = 0; i < n; ++i)
    if (rc[i] != IBU_OK) {                      // first error in worker order (mmap.rs:326-328)
      tls_error() = detail[i];
      return rc[i];
    }
  ibu_reduce_result_t t;
  memset(&t, 0, sizeof t);
  if (proc == IBU_PROC_REDUCE) {
    for (size_t i = 0; i < n; ++i) {
      t.count += part[i].count;
      for (int f = 0; f < 3; ++f) { t.sum[f] += part[i].sum[f]; t.xor_[f] ^= part[i].xor_[f]; }   // wrapping (mod 2^64), as the device adds
    }
    if (sinks) memcpy(sinks, part.data(), n * sizeof(ibu_reduce_result_t));
  } else {
    t.count = ibu_mmap_len(m);                  // every shard decoded: the whole map
  }
  if (total) *total = t;
  return IBU_OK;
}
}  // namespace

extern "C" int32_t ibu_mmap_process_contexts(const ibu_mmap_t* m, ibu_ctx_t* const* ctxs, size_t n_ctxs, const ibu_ring_config_t* cfg,
                                             int32_t proc, void* sinks, ibu_reduce_result_t* total, ibu_stream_stats_t* stats) {
  if (!m || !ctxs || n_ctxs == 0) return err_arg("NULL argument or no context");
  if (n_ctxs > 1024) return err_arg("more than 1024 contexts");
  for (size_t i = 0; i < n_ctxs; ++i) {
    if (!ctxs[i]) return err_arg("a context is NULL");
    for (size_t j = 0; j < i; ++j)
      if (ctxs[j] == ctxs[i]) return err_arg("the same context twice (a context serves one host thread; create two on one device instead)");
  }
  return process_contexts(m, ctxs, n_ctxs, cfg, proc, sinks, total, stats);
}

extern "C" int32_t ibu_mmap_process_devices(const ibu_mmap_t* m, const int32_t* devices, size_t n_devices, const ibu_ring_config_t* cfg,
                                            int32_t proc, void* sinks, ibu_reduce_result_t* total, ibu_stream_stats_t* stats) {
  if (!m) return err_arg("NULL argument");
  std::vector<int32_t> all;
  std::vector<ibu_ctx_t*> ctxs;
  try {
    if (n_devices == 0) {                       // "0 = all of them", as num_threads == 0 means all cores (mmap.rs:292-296)
      int32_t c = 0;
      int32_t rc = ibu_device_count(&c);
      if (rc) return rc;
      if (c <= 0) return set_error(IBU_ERR_NO_DEVICE, 0, 0, 0, "no HIP device visible");
      for (int32_t d = 0; d < c; ++d) all.push_back(d);
      devices = all.data();
      n_devices = all.size();
    } else if (!devices) {
      return err_arg("devices is NULL");
    }
    if (n_devices > 1024) return err_arg("more than 1024 devices");
    ctxs.assign(n_devices, nullptr);
  } catch (...) {
    return caught_io("ibu_mmap_process_devices");
  }
  // The contexts of this form live for ONE call, so their rings are allocated and pinned inside it: 4 x 96 MiB (the default
  // of a kept context) costs 70 ms per context — more than streaming an eighth of a 24 GB file — and pinning serialises in
  // the driver.  Without a caller's ring configuration the one-shot form uses 3 x 24 MiB slots (1 Mi records, the
  // reference's BATCH_SIZE, mmap.rs:284): 18 ms, 54.5 GB/s against 56.1 (profiles/README.md r03_y).
  ibu_ring_config_t one_shot;
  memset(&one_shot, 0, sizeof one_shot);
  one_shot.slots = 3;
  one_shot.slot_records = IBU_BATCH_SIZE;
  if (!cfg) cfg = &one_shot;
  int32_t rc = IBU_OK;
  for (size_t i = 0; i < n_devices && rc == IBU_OK; ++i) rc = ibu_ctx_create(devices[i], &ctxs[i]);   // in order: the first bad ordinal is the error
  if (rc == IBU_OK) rc = process_contexts(m, ctxs.data(), n_devices, cfg, proc, sinks, total, stats);
  const ibu_error_detail_t keep = tls_error();
  for (ibu_ctx_t* c : ctxs) ibu_ctx_destroy(c);
  if (rc != IBU_OK) tls_error() = keep;
  return rc;
}

// ------------------------------------------------------------------------------------------
// streaming Reader (plain / gzip), device form
// ------------------------------------------------------------------------------------------
namespace {
// A Reader over a BGZF FILE nothing has been read from: the processors need not pull it through the Reader's host inflate (0.4 G records/s
// on 16 CPUs) — the file is loaded range by range with its compressed bytes crossing the link and its blocks inflated on the device
// (ibu_load_bgzf_shard_to_device: ranges of about 6 GB of records, so that any file fits), and the processor runs over each range.
// *handled = false: the file is not what that load takes (a foreign member, a cut, a length that is no whole number of records ...) —
// nothing has been touched, and the Reader's own path delivers what it delivers for such a file, error and all.
int32_t process_bgzf_file(ibu_ctx* ctx, const char* path, const ibu_ring_config_t* cfg, const ibu_header_t& want, DeviceProc& dp, int32_t proc, void* sink,
                          ibu_stream_stats_t* stats, uint64_t* records, bool* handled) {
  *handled = false;
  struct stat sb;
  if (stat(path, &sb) != 0) return IBU_OK;
  // ranges of ~6.4 GB of records where the file compresses to half (BGZF of 16/12 records: 0.50) — every range costs its launch's
  // waves' 45-75 ms once more, and a file that compresses better just gets larger ranges
  const size_t K = (size_t)((double)sb.st_size / (ctx->bgzf_range_bytes_opt ? (double)ctx->bgzf_range_bytes_opt : 3.2e9)) + 1;
  ibu_header_t h;
  size_t n0 = 0;
  uint64_t first = 0;
  ibu_stream_stats_t st{};
  const ibu_error_detail_t keep = tls_error();
  // The ranges land in a buffer the CONTEXT keeps (it grows only; option "release_staging" frees it): allocating and freeing 2.4 GB
  // around every call cost a call of 1e8 records 80 of its 155 ms.  First use, or a larger range than ever: the load allocates, and
  // the context adopts what it allocated.
  auto cap = [&] { return ctx->bgzf_range_bytes / IBU_RECORD_SIZE; };
  int32_t rc = IBU_ERR_INVALID_ARG;
  if (cap()) {
    void* p = ctx->d_bgzf_range;
    rc = ibu_load_bgzf_shard_to_device(ctx, path, cfg, 0, K, &h, &p, cap(), &n0, &first, &st);
    if (rc == IBU_ERR_INVALID_ARG && tls_error().b == cap() && tls_error().a > cap()) {   // too small for this file's ranges
      (void)hipFree(ctx->d_bgzf_range);
      ctx->d_bgzf_range = nullptr;
      ctx->bgzf_range_bytes = 0;
    }
  }
  if (!cap()) {
    const int probe = ctx->cfg.alloc_probe_tries;        // (no placement probing for it: the records only pass through)
    ctx->cfg.alloc_probe_tries = 1;
    void* p = nullptr;
    rc = ibu_load_bgzf_shard_to_device(ctx, path, cfg, 0, K, &h, &p, 0, &n0, &first, &st);
    ctx->cfg.alloc_probe_tries = probe;
    if (rc == IBU_OK) { ctx->d_bgzf_range = p; ctx->bgzf_range_bytes = (n0 ? n0 : 1) * IBU_RECORD_SIZE; }
  }
  if (rc || memcmp(&h, &want, sizeof h) != 0) {          // not for this path: as if it had not been tried
    tls_error() = keep;
    return IBU_OK;
  }
  *handled = true;
  auto done = [&](int32_t code) {
    (void)hipStreamSynchronize(ctx->stream);
    return code;
  };
  if (K > 1 && cap() < n0 + K) {                         // (a later range has at most K - 1 records more than the first)
    void* big = nullptr;
    hipError_t e = ctx_malloc(ctx, &big, (n0 + K) * IBU_RECORD_SIZE);
    if (e == hipSuccess) e = hipMemcpy(big, ctx->d_bgzf_range, n0 * IBU_RECORD_SIZE, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { if (big) (void)hipFree(big); return done(hip_fail(e, "hipMalloc")); }
    (void)hipFree(ctx->d_bgzf_range);
    ctx->d_bgzf_range = big;
    ctx->bgzf_range_bytes = (n0 + K) * IBU_RECORD_SIZE;
  }
  uint64_t total = 0;
  if (proc == IBU_PROC_REDUCE) {
    hipError_t e = hipMemsetAsync(ctx->d_acc, 0, kReduceAccBytes, ctx->stream);
    if (e != hipSuccess) return done(hip_fail(e, "hipMemsetAsync"));
  }
  for (size_t i = 0; i < K; ++i) {
    size_t n = n0;
    if (i) {
      void* p = ctx->d_bgzf_range;
      rc = ibu_load_bgzf_shard_to_device(ctx, path, cfg, i, K, &h, &p, cap(), &n, &first, &st);
      if (rc) return done(rc);
    }
    if (stats) { stats->bytes_h2d += st.bytes_h2d; stats->batches += st.batches; }
    rc = dp.fits(n, (size_t)first);
    if (rc) return done(rc);
    if (n) {
      rc = dp.launch(static_cast<const uint8_t*>(ctx->d_bgzf_range), n, (size_t)first);
      if (rc) return done(rc);
    }
    total += n;
    hipError_t e = hipStreamSynchronize(ctx->stream);      // (the next range is loaded over these records)
    if (e != hipSuccess) return done(hip_fail(e, "hipStreamSynchronize"));
  }
  if (proc == IBU_PROC_REDUCE) {
    rc = ibu_reduce_fetch(ctx, ctx->stream, static_cast<ibu_reduce_result_t*>(sink));
    if (rc) return done(rc);
  }
  *records = total;
  if (stats) { stats->records = total; stats->numa_node = st.numa_node; stats->ring_node = st.ring_node; }
  return done(IBU_OK);
}
}  // namespace

extern "C" int32_t ibu_reader_process_device(ibu_reader_t* rd, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                             int32_t proc, void* sink, ibu_stream_stats_t* stats) {
  if (!rd || !ctx) return err_arg("NULL argument");
  const double t0 = now_s();
  if (stats) memset(stats, 0, sizeof *stats);
  ibu_header_t h;
  ibu_reader_header(rd, &h);
  DeviceProc dp;
  int32_t rc = make_proc(ctx, proc, h, sink, &dp);   // argument errors before the producer thread exists
  if (rc) return rc;
  if (const char* bp = ctx->bgzf_device ? reader_bgzf_path_if_untouched(rd) : nullptr) {
    bool handled = false;
    uint64_t records = 0;
    IBU_HIP(hipSetDevice(ctx->device));
    rc = process_bgzf_file(ctx, bp, cfg, h, dp, proc, sink, stats, &records, &handled);
    if (handled) {
      if (rc == IBU_OK) {
        reader_set_drained(rd, records);
        if (stats) stats->seconds_total = now_s() - t0;
      }
      return rc;
    }
    if (stats) memset(stats, 0, sizeof *stats);
  }
  ibu_stream_t* s = nullptr;
  rc = ibu_stream_open_reader(rd, ctx, cfg, &s);
  return process_stream(s, rc, proc, sink, stats, t0);
}
