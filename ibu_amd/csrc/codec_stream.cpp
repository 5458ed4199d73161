// codec_stream.cpp — host <-> host codec pipelines: the 2-bit codec applied to data that starts and ends in
// host memory or in a file, which is how the reference's consumers hold it (README.md:38-47: sequences ->
// 2-bit u64s -> Record -> Writer; and back: Reader/MmapReader -> Record -> sequences).
//
//   decode: file (mmap) --memcpy--> pinned AoS --H2D--> K2 decode --D2H--> pinned columns --memcpy--> caller
//   encode: caller columns --memcpy--> pinned columns --H2D--> K3 encode --D2H--> pinned AoS --> Writer
//
// Three HIP streams per context (H2D, compute, D2H) chained per slot with events; the host threads copy batch
// k+1 in and batch k-1 out while batch k is on the device.  PCIe is the bound (24 B/record one way, 36 B the
// other at 16/12), not the kernels.
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "host_io.hpp"

using namespace ibu;

namespace {

constexpr uint32_t kMaxCols = IBU_MAX_SEQ_LEN + IBU_MAX_SEQ_LEN + 8;  // bytes per record on the column side

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// memcpy split over a few threads (page-cache / page-fault side of the host work)
void par_memcpy(uint8_t* dst, const uint8_t* src, size_t bytes, uint32_t threads) {
  const size_t min_chunk = (size_t)4 << 20;
  size_t parts = bytes / min_chunk;
  if (parts > threads) parts = threads;
  if (parts <= 1) {
    if (bytes) memcpy(dst, src, bytes);
    return;
  }
  const size_t per = ((bytes / parts) + 4095) & ~(size_t)4095;
  run_pieces((unsigned)parts, [=](unsigned i) {  // never throws (common.hpp)
    const size_t off = (size_t)i * per;
    if (off >= bytes) return;
    memcpy(dst + off, src + off, off + per < bytes ? per : bytes - off);
  });
}

uint32_t feeders(const ibu_ring_config_t* cfg) { return cfg && cfg->feeder_threads ? cfg->feeder_threads : 4; }

int32_t codec_ring_ensure(ibu_ctx* ctx, const ibu_ring_config_t* cfg) {
  uint32_t slots = cfg && cfg->slots ? cfg->slots : 4;
  if (slots < 2) slots = 2;
  size_t slot_records = cfg && cfg->slot_records ? cfg->slot_records : (size_t)IBU_BATCH_SIZE;
  slot_records = (slot_records + 127) & ~(size_t)127;  // whole kernel tiles; keeps every column offset 16-B aligned
  CodecRing& r = ctx->cring;
  if (r.slots == slots && r.slot_records == slot_records) return IBU_OK;
  (void)hipStreamSynchronize(ctx->copy_stream);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->d2h_stream);
  codec_ring_release(ctx);
  r.h_aos.assign(slots, nullptr); r.h_col.assign(slots, nullptr);
  r.d_aos.assign(slots, nullptr); r.d_col.assign(slots, nullptr);
  r.h_status.assign(slots, nullptr); r.d_status.assign(slots, nullptr);
  r.up.resize(slots); r.done.resize(slots); r.down.resize(slots);
  r.slots = slots;  // set early so a failure below is cleaned up by codec_ring_release
  r.slot_records = slot_records;
  r.events = 0;
  for (uint32_t i = 0; i < slots; ++i) {
    IBU_HIP(hipEventCreateWithFlags(&r.up[i], hipEventDisableTiming));
    IBU_HIP(hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming));
    IBU_HIP(hipEventCreateWithFlags(&r.down[i], hipEventDisableTiming));
    r.events = i + 1;
  }
  PreferNode prefer(feed_place(ctx).node);   // option "numa": the pinned halves on the device's node (stream.cpp: ring_ensure)
  const unsigned hflags = prefer.active() ? hipHostMallocNumaUser : hipHostMallocDefault;
  for (uint32_t i = 0; i < slots; ++i) {
    IBU_HIP(hipHostMalloc(reinterpret_cast<void**>(&r.h_aos[i]), slot_records * IBU_RECORD_SIZE, hflags));
    IBU_HIP(hipHostMalloc(reinterpret_cast<void**>(&r.h_col[i]), slot_records * kMaxCols, hflags));
    IBU_HIP(hipHostMalloc(reinterpret_cast<void**>(&r.h_status[i]), 2 * sizeof(uint64_t), hipHostMallocDefault));
    IBU_HIP(ctx_malloc(ctx, reinterpret_cast<void**>(&r.d_aos[i]), slot_records * IBU_RECORD_SIZE));
    IBU_HIP(ctx_malloc(ctx, reinterpret_cast<void**>(&r.d_col[i]), slot_records * kMaxCols));
    IBU_HIP(ctx_malloc(ctx, reinterpret_cast<void**>(&r.d_status[i]), 2 * sizeof(uint64_t)));
  }
  return IBU_OK;
}

int32_t drain3(ibu_ctx* ctx, int32_t rc) {  // leave nothing in flight over ring memory
  (void)hipStreamSynchronize(ctx->copy_stream);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->d2h_stream);
  return rc;
}

// Two host threads per call: the caller's thread fills and submits batch k, a collector thread delivers batch
// k - depth (waits for its D2H, copies it out / hands it to the writer).  `submitted` / `delivered` are the only
// shared state; the first error on either side stops both.
struct Handoff {
  std::mutex m;
  std::condition_variable cv;
  size_t submitted = 0, delivered = 0;
  int32_t err = IBU_OK;
  bool producer_done = false;
  void publish() { { std::lock_guard<std::mutex> g(m); ++submitted; } cv.notify_all(); }
  void finish_producing(int32_t rc) { { std::lock_guard<std::mutex> g(m); producer_done = true; if (rc && !err) err = rc; } cv.notify_all(); }
  // producer: slot of batch k is free once batch k - slots has been delivered
  int32_t wait_slot(size_t k, size_t slots) {
    std::unique_lock<std::mutex> g(m);
    cv.wait(g, [&] { return err || k < slots || delivered + slots > k; });
    return err;
  }
  // collector: next batch to deliver, or false when everything submitted has been delivered and no more will come
  bool next(size_t* k) {
    std::unique_lock<std::mutex> g(m);
    cv.wait(g, [&] { return err || delivered < submitted || producer_done; });
    if (err || delivered >= submitted) return false;
    *k = delivered;
    return true;
  }
  void done(int32_t rc) { { std::lock_guard<std::mutex> g(m); if (rc && !err) err = rc; else if (!rc) ++delivered; } cv.notify_all(); }
};

struct ColLayout {  // where the three columns of a slot live inside its column buffer
  size_t bc, umi, idx, bytes;
  ColLayout(size_t slot_records, uint32_t bc_len, uint32_t umi_len)
      : bc(0), umi(slot_records * bc_len), idx(slot_records * (bc_len + umi_len)), bytes(slot_records * (bc_len + umi_len + 8)) {}
};

}  // namespace

void ibu::codec_ring_release(ibu_ctx* ctx) {
  CodecRing& r = ctx->cring;
  for (auto p : r.h_aos) if (p) (void)hipHostFree(p);
  for (auto p : r.h_col) if (p) (void)hipHostFree(p);
  for (auto p : r.h_status) if (p) (void)hipHostFree(p);
  for (auto p : r.d_aos) if (p) (void)hipFree(p);
  for (auto p : r.d_col) if (p) (void)hipFree(p);
  for (auto p : r.d_status) if (p) (void)hipFree(p);
  for (uint32_t i = 0; i < r.events; ++i) {
    (void)hipEventDestroy(r.up[i]);
    (void)hipEventDestroy(r.done[i]);
    (void)hipEventDestroy(r.down[i]);
  }
  r = CodecRing();
}

// ------------------------------------------------------------------------------------------------------------
// file -> ASCII columns in host memory (one shard of the static split per call / GPU)
// ------------------------------------------------------------------------------------------------------------
extern "C" int32_t ibu_mmap_decode_to_host(const ibu_mmap_t* m, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg, size_t shard,
                                           size_t n_shards, uint8_t* h_bc_ascii, uint8_t* h_umi_ascii, uint64_t* h_index,
                                           ibu_stream_stats_t* stats) {
  if (!m || !ctx) return err_arg("NULL argument");
  IBU_HIP(hipSetDevice(ctx->device));
  RunOnNode on_node(feed_place(ctx));   // this thread, the collector and the copy threads on the device's node for the length of the call (option "numa")
  const double t0 = now_s();
  if (stats) memset(stats, 0, sizeof *stats);
  size_t start = 0, end = 0;
  int32_t rc = ibu_shard_range(ibu_mmap_len(m), n_shards, shard, &start, &end);  // mmap.rs:297-307
  if (rc) return rc;
  ibu_header_t h;
  ibu_mmap_header(m, &h);
  rc = codec_ring_ensure(ctx, cfg);
  if (rc) return rc;
  CodecRing& r = ctx->cring;
  const ColLayout L(r.slot_records, h.bc_len, h.umi_len);
  const uint8_t* base = static_cast<const uint8_t*>(ibu_mmap_base(m)) + IBU_HEADER_SIZE;
  const uint32_t nf = feeders(cfg);
  const size_t total = end - start;
  const size_t nb = (total + r.slot_records - 1) / r.slot_records;

  auto rows_of = [&](size_t k) { return k + 1 < nb ? r.slot_records : total - k * r.slot_records; };
  auto collect = [&](size_t k) -> int32_t {  // batch k's columns: pinned -> caller memory
    const uint32_t s = (uint32_t)(k % r.slots);
    IBU_HIP(hipEventSynchronize(r.down[s]));
    const size_t rows = rows_of(k), row0 = k * r.slot_records;
    if (h_bc_ascii) par_memcpy(h_bc_ascii + row0 * h.bc_len, r.h_col[s] + L.bc, rows * h.bc_len, nf);
    if (h_umi_ascii) par_memcpy(h_umi_ascii + row0 * h.umi_len, r.h_col[s] + L.umi, rows * h.umi_len, nf);
    if (h_index) par_memcpy(reinterpret_cast<uint8_t*>(h_index + row0), r.h_col[s] + L.idx, rows * 8, nf);
    return IBU_OK;
  };

  Handoff ho;
  ibu_error_detail_t collector_detail;
  memset(&collector_detail, 0, sizeof collector_detail);
  std::thread collector;
  try {
  collector = std::thread([&]() {
    (void)hipSetDevice(ctx->device);
    size_t k;
    while (ho.next(&k)) {
      const int32_t e = collect(k);
      if (e) ibu_last_error(&collector_detail);  // the error record is per thread: carry it over
      ho.done(e);
    }
  });
  } catch (...) {  // std::system_error (EAGAIN under a pids cgroup) / bad_alloc: nothing is in flight yet
    return caught_io("cannot start the collector thread");
  }
  for (size_t k = 0; k < nb && rc == IBU_OK; ++k) {
    const uint32_t s = (uint32_t)(k % r.slots);
    rc = ho.wait_slot(k, r.slots);  // slot s is free: batch k - slots has been delivered
    if (rc) break;
    const size_t rows = rows_of(k);
    par_memcpy(r.h_aos[s], base + (start + k * r.slot_records) * IBU_RECORD_SIZE, rows * IBU_RECORD_SIZE, nf);
    hipError_t e = hipMemcpyAsync(r.d_aos[s], r.h_aos[s], rows * IBU_RECORD_SIZE, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(r.up[s], ctx->copy_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, r.up[s], 0);
    if (e == hipSuccess)
      e = launch_decode(ctx->cfg, r.d_aos[s], rows, h.bc_len, h.umi_len, h_bc_ascii ? r.d_col[s] + L.bc : nullptr,
                        h_umi_ascii ? r.d_col[s] + L.umi : nullptr,
                        h_index ? reinterpret_cast<uint64_t*>(r.d_col[s] + L.idx) : nullptr, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(r.done[s], ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->d2h_stream, r.done[s], 0);
    // one D2H per requested column (they are not adjacent when a column is skipped or the batch is short)
    if (e == hipSuccess && h_bc_ascii)
      e = hipMemcpyAsync(r.h_col[s] + L.bc, r.d_col[s] + L.bc, rows * h.bc_len, hipMemcpyDeviceToHost, ctx->d2h_stream);
    if (e == hipSuccess && h_umi_ascii)
      e = hipMemcpyAsync(r.h_col[s] + L.umi, r.d_col[s] + L.umi, rows * h.umi_len, hipMemcpyDeviceToHost, ctx->d2h_stream);
    if (e == hipSuccess && h_index)
      e = hipMemcpyAsync(r.h_col[s] + L.idx, r.d_col[s] + L.idx, rows * 8, hipMemcpyDeviceToHost, ctx->d2h_stream);
    if (e == hipSuccess) e = hipEventRecord(r.down[s], ctx->d2h_stream);
    if (e != hipSuccess) { rc = hip_fail(e, "decode pipeline"); break; }
    if (stats) {
      stats->bytes_h2d += rows * IBU_RECORD_SIZE;
      stats->bytes_d2h += rows * ((h_bc_ascii ? h.bc_len : 0) + (h_umi_ascii ? h.umi_len : 0) + (h_index ? 8 : 0));
      stats->batches += 1;
    }
    ho.publish();
  }
  ho.finish_producing(rc);
  collector.join();
  if (collector_detail.code)  // the collector failed first (the producer only saw its flag): report ITS error from this thread
    rc = set_error(collector_detail.code, collector_detail.a, collector_detail.b, collector_detail.os_errno, "%s", collector_detail.message);
  else if (rc == IBU_OK && ho.err)
    rc = ho.err;
  if (rc) return drain3(ctx, rc);
  if (stats) { stats->records = total; stats->seconds_total = now_s() - t0; }
  return IBU_OK;
}

// ------------------------------------------------------------------------------------------------------------
// ASCII columns in host memory -> records -> Writer
// ------------------------------------------------------------------------------------------------------------
extern "C" int32_t ibu_writer_write_ascii_batch(ibu_writer_t* w, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                                const uint8_t* h_bc_ascii, const uint8_t* h_umi_ascii, const uint64_t* h_index,
                                                uint64_t first_index, size_t n, uint32_t bc_len, uint32_t umi_len,
                                                ibu_stream_stats_t* stats) {
  if (!w || !ctx || ((!h_bc_ascii || !h_umi_ascii) && n)) return err_arg("NULL argument");
  if (bc_len == 0 || bc_len > IBU_MAX_SEQ_LEN) return err_bc_len(bc_len);
  if (umi_len == 0 || umi_len > IBU_MAX_SEQ_LEN) return err_umi_len(umi_len);
  IBU_HIP(hipSetDevice(ctx->device));
  RunOnNode on_node(feed_place(ctx));
  const double t0 = now_s();
  if (stats) memset(stats, 0, sizeof *stats);
  if (n == 0) return IBU_OK;
  int32_t rc = codec_ring_ensure(ctx, cfg);
  if (rc) return rc;
  CodecRing& r = ctx->cring;
  const ColLayout L(r.slot_records, bc_len, umi_len);
  const uint32_t nf = feeders(cfg);
  const size_t nb = (n + r.slot_records - 1) / r.slot_records;
  auto rows_of = [&](size_t k) { return k + 1 < nb ? r.slot_records : n - k * r.slot_records; };
  uint64_t first_bad = ~0ull, n_bad = 0;

  auto collect = [&](size_t k) -> int32_t {  // batch k's records: pinned -> writer, unless a base was invalid
    const uint32_t s = (uint32_t)(k % r.slots);
    IBU_HIP(hipEventSynchronize(r.down[s]));
    const size_t rows = rows_of(k);
    if (r.h_status[s][1]) {  // rows of this batch held a byte outside ACGTacgt
      if (n_bad == 0) first_bad = k * r.slot_records + r.h_status[s][0];
      n_bad += r.h_status[s][1];
    }
    if (n_bad) return IBU_OK;  // nothing from the first bad batch on reaches the writer
    return writer_write_bytes(w, r.h_aos[s], rows * IBU_RECORD_SIZE);  // buffered / direct rule of writer.rs:321-351
  };

  Handoff ho;
  ibu_error_detail_t collector_detail;
  memset(&collector_detail, 0, sizeof collector_detail);
  std::thread collector;
  try {
  collector = std::thread([&]() {  // delivers batches to the writer while the caller's thread stages the next ones
    (void)hipSetDevice(ctx->device);
    size_t k;
    while (ho.next(&k)) {
      const int32_t e = collect(k);
      if (e) ibu_last_error(&collector_detail);
      ho.done(e);
    }
  });
  } catch (...) {  // std::system_error (EAGAIN under a pids cgroup) / bad_alloc: nothing is in flight yet
    return caught_io("cannot start the collector thread");
  }
  for (size_t k = 0; k < nb && rc == IBU_OK; ++k) {
    const uint32_t s = (uint32_t)(k % r.slots);
    rc = ho.wait_slot(k, r.slots);
    if (rc) break;
    const size_t rows = rows_of(k), row0 = k * r.slot_records;
    par_memcpy(r.h_col[s] + L.bc, h_bc_ascii + row0 * bc_len, rows * bc_len, nf);
    par_memcpy(r.h_col[s] + L.umi, h_umi_ascii + row0 * umi_len, rows * umi_len, nf);
    if (h_index) par_memcpy(r.h_col[s] + L.idx, reinterpret_cast<const uint8_t*>(h_index + row0), rows * 8, nf);
    hipError_t e = hipMemcpyAsync(r.d_col[s] + L.bc, r.h_col[s] + L.bc, rows * bc_len, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(r.d_col[s] + L.umi, r.h_col[s] + L.umi, rows * umi_len, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess && h_index)
      e = hipMemcpyAsync(r.d_col[s] + L.idx, r.h_col[s] + L.idx, rows * 8, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(r.up[s], ctx->copy_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, r.up[s], 0);
    if (e == hipSuccess) e = launch_fill2(r.d_status[s], ~0ull, 0, ctx->stream);
    if (e == hipSuccess)
      e = launch_encode(ctx->cfg, r.d_col[s] + L.bc, r.d_col[s] + L.umi,
                        h_index ? reinterpret_cast<const uint64_t*>(r.d_col[s] + L.idx) : nullptr, first_index + row0, rows, bc_len,
                        umi_len, r.d_aos[s], r.d_status[s], ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(r.done[s], ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->d2h_stream, r.done[s], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(r.h_aos[s], r.d_aos[s], rows * IBU_RECORD_SIZE, hipMemcpyDeviceToHost, ctx->d2h_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r.h_status[s], r.d_status[s], 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->d2h_stream);
    if (e == hipSuccess) e = hipEventRecord(r.down[s], ctx->d2h_stream);
    if (e != hipSuccess) { rc = hip_fail(e, "encode pipeline"); break; }
    if (stats) {
      stats->bytes_h2d += rows * (bc_len + umi_len + (h_index ? 8 : 0));
      stats->bytes_d2h += rows * IBU_RECORD_SIZE;
      stats->batches += 1;
    }
    ho.publish();
  }
  ho.finish_producing(rc);
  collector.join();
  if (collector_detail.code)  // the collector failed first (the producer only saw its flag): report ITS error from this thread
    rc = set_error(collector_detail.code, collector_detail.a, collector_detail.b, collector_detail.os_errno, "%s", collector_detail.message);
  else if (rc == IBU_OK && ho.err)
    rc = ho.err;
  if (rc) return drain3(ctx, rc);
  if (stats) { stats->records = n; stats->seconds_total = now_s() - t0; }
  if (n_bad)
    return set_error(IBU_ERR_INVALID_BASE, first_bad, n_bad, 0,
                     "Invalid base: %llu row(s) hold a byte outside ACGTacgt, first at row %llu; rows from that batch on were not written",
                     (unsigned long long)n_bad, (unsigned long long)first_bad);
  return IBU_OK;
}
