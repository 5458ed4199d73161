// stream.cpp — record streams between files / mmap / gzip and the device, through a ring of
// pinned (hipHostMalloc) staging slots.  Host work (page-cache memcpy, pread, inflate) for
// slot k+1 overlaps the H2D copy of slot k (copy stream) and the kernel on slot k-1 (compute
// stream); the two streams are chained per slot with events, never with a device-wide sync.
//
//   file/mmap/gz --host threads--> pinned[slot] --copy_stream H2D--> dev[slot] --stream--> kernel
//
// These are the device-backed forms of load_to_vec (reader.rs:510-535), Writer::write_batch
// (writer.rs:315-351), MmapReader::process_parallel (mmap.rs:286-332, ONE shard of its static
// split per call = per GPU) and the streaming Reader (reader.rs:279-306, incl. the gzip path
// of reader.rs:345-352).
#include <errno.h>
#include <unistd.h>

#include <chrono>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "host_io.hpp"

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

// Push one filled pinned slot to the device and run the processor on it.
int32_t submit_slot(ibu_ctx* ctx, KernelClock& kc, DeviceProc& dp, uint32_t s, size_t n, size_t row0,
                    ibu_stream_stats_t* stats) {
  Ring& r = ctx->ring;
  const size_t bytes = n * IBU_RECORD_SIZE;
  IBU_HIP(hipMemcpyAsync(r.dev[s], r.pinned[s], bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  IBU_HIP(hipEventRecord(r.copied[s], ctx->copy_stream));
  IBU_HIP(hipStreamWaitEvent(ctx->stream, r.copied[s], 0));
  IBU_HIP(hipEventRecord(kc.a[s], ctx->stream));
  int32_t rc = dp.launch(r.dev[s], n, row0);
  if (rc) return rc;
  IBU_HIP(hipEventRecord(kc.b[s], ctx->stream));
  IBU_HIP(hipEventRecord(r.consumed[s], ctx->stream));
  kc.live[s] = 1;
  if (stats) {
    stats->bytes_h2d += bytes;
    stats->records += n;
    stats->batches += 1;
  }
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
  for (uint32_t i = 0; i < slots; ++i)
    IBU_HIP(hipHostMalloc(reinterpret_cast<void**>(&r.pinned[i]), slot_bytes, hipHostMallocDefault));
  if (need_dev) {
    r.dev.assign(slots, nullptr);
    for (uint32_t i = 0; i < slots; ++i) IBU_HIP(hipMalloc(reinterpret_cast<void**>(&r.dev[i]), slot_bytes));
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
// MmapReader::process_parallel, device form (one shard of the static split)
// ------------------------------------------------------------------------------------------
extern "C" int32_t ibu_mmap_process_device(const ibu_mmap_t* m, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                           int32_t proc, size_t shard, size_t n_shards, void* sink,
                                           ibu_stream_stats_t* stats) {
  if (!m || !ctx) return err_arg("NULL argument");
  IBU_HIP(hipSetDevice(ctx->device));
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
  rc = dp.fits(end - start, 0);   // the shard's size is known up front
  if (rc) return rc;
  rc = ring_ensure(ctx, cfg, true);
  if (rc) return rc;
  Ring& r = ctx->ring;
  KernelClock kc;
  rc = kc.init(r.slots);
  if (rc) return rc;
  if (proc == IBU_PROC_REDUCE) IBU_HIP(hipMemsetAsync(ctx->d_acc, 0, kReduceAccBytes, ctx->stream));
  const size_t slot_records = r.slot_bytes / IBU_RECORD_SIZE;
  const uint8_t* base = static_cast<const uint8_t*>(ibu_mmap_base(m)) + IBU_HEADER_SIZE;
  size_t k = 0;
  for (size_t row = start; row < end && rc == IBU_OK; row += slot_records, ++k) {
    const uint32_t s = (uint32_t)(k % r.slots);
    const size_t nb = end - row < slot_records ? end - row : slot_records;
    hipError_t e = hipEventSynchronize(r.consumed[s]);  // the kernel that read dev[s] (and so its H2D) is done
    if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; }
    kc.harvest(s);
    const uint8_t* srcp = base + row * IBU_RECORD_SIZE;
    uint8_t* dst = r.pinned[s];
    parallel_bytes(nb * IBU_RECORD_SIZE, feeder_threads(cfg), [&](size_t off, size_t len) {
      memcpy(dst + off, srcp + off, len);  // page-cache / page-fault side of the reference's hot loop
      return 0;
    });
    rc = submit_slot(ctx, kc, dp, s, nb, row - start, stats);
  }
  if (rc) return drain(ctx, rc);
  if (proc == IBU_PROC_REDUCE) rc = ibu_reduce_fetch(ctx, ctx->stream, static_cast<ibu_reduce_result_t*>(sink));
  else if (hipError_t e = hipStreamSynchronize(ctx->stream); e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
  if (rc) return drain(ctx, rc);
  for (uint32_t s = 0; s < r.slots; ++s) kc.harvest(s);
  if (stats) { stats->seconds_kernel = kc.ms * 1e-3; stats->seconds_total = now_s() - t0; }
  return IBU_OK;
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
extern "C" int32_t ibu_reader_process_device(ibu_reader_t* rd, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                             int32_t proc, void* sink, ibu_stream_stats_t* stats) {
  if (!rd || !ctx) return err_arg("NULL argument");
  IBU_HIP(hipSetDevice(ctx->device));
  const double t0 = now_s();
  if (stats) memset(stats, 0, sizeof *stats);
  ibu_header_t h;
  ibu_reader_header(rd, &h);
  DeviceProc dp;
  int32_t rc = make_proc(ctx, proc, h, sink, &dp);
  if (rc) return rc;
  rc = ring_ensure(ctx, cfg, true);
  if (rc) return rc;
  Ring& r = ctx->ring;
  KernelClock kc;
  rc = kc.init(r.slots);
  if (rc) return rc;
  if (proc == IBU_PROC_REDUCE) IBU_HIP(hipMemsetAsync(ctx->d_acc, 0, kReduceAccBytes, ctx->stream));
  const size_t slot_records = r.slot_bytes / IBU_RECORD_SIZE;
  size_t total = 0, k = 0;
  bool eof = false;
  while (!eof && rc == IBU_OK) {
    const uint32_t s = (uint32_t)(k % r.slots);
    hipError_t e = hipEventSynchronize(r.consumed[s]);
    if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; }
    kc.harvest(s);
    size_t filled = 0;
    while (filled < slot_records) {
      const ibu_record_t* recs;
      size_t have = 0;
      ibu_reader_buffered(rd, &recs, &have);
      if (have == 0) {
        // the source fills the pinned slot directly (parallel preads of a plain file, the inflate threads' own copies):
        // no detour through the reader's 1.18 MB buffer; the truncation rule of reader.rs:232-237 applies to the slot
        size_t got = 0;
        rc = reader_read_direct(rd, r.pinned[s] + filled * IBU_RECORD_SIZE, (slot_records - filled) * IBU_RECORD_SIZE, &got, &eof);
        if (rc) { eof = true; break; }
        filled += got / IBU_RECORD_SIZE;
        if (eof) break;
        continue;
      }
      // records the caller had already pulled into the reader's buffer (read_batch / next before this call) go first
      const size_t take = have < slot_records - filled ? have : slot_records - filled;
      memcpy(r.pinned[s] + filled * IBU_RECORD_SIZE, recs, take * IBU_RECORD_SIZE);
      ibu_reader_consume(rd, take);
      filled += take;
    }
    if (rc) break;
    if (filled) {
      rc = dp.fits(filled, total);  // a gzip / BGZF / xz / zstd stream does not announce its length: check every batch
      if (rc) break;
      rc = submit_slot(ctx, kc, dp, s, filled, total, stats);
      total += filled;
      ++k;
    }
  }
  if (rc) return drain(ctx, rc);
  if (proc == IBU_PROC_REDUCE) rc = ibu_reduce_fetch(ctx, ctx->stream, static_cast<ibu_reduce_result_t*>(sink));
  else if (hipError_t e = hipStreamSynchronize(ctx->stream); e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
  if (rc) return drain(ctx, rc);
  for (uint32_t s = 0; s < r.slots; ++s) kc.harvest(s);
  if (stats) { stats->seconds_kernel = kc.ms * 1e-3; stats->seconds_total = now_s() - t0; }
  return IBU_OK;
}
