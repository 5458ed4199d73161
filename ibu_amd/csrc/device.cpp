// device.cpp — device context and the kernel entry points of the C ABI (include/ibu_hip.h).
//
// Every function here fails loudly (IBU_ERR_NO_DEVICE / IBU_ERR_HIP) when there is no gfx950
// device; nothing falls back to host arithmetic.  Launch functions are asynchronous, allocate
// nothing and never synchronise, so callers may capture them into hipGraphs.
#include <stdlib.h>
#include <stddef.h>
#include <string.h>

#include <chrono>

#include <cstddef>
#include "ctx.hpp"

using namespace ibu;

namespace {

int32_t check_ctx(const ibu_ctx* ctx) {
  if (!ctx) return err_arg("ctx is NULL");
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  return IBU_OK;
}
int32_t check_lens(uint32_t bc_len, uint32_t umi_len) {  // header.rs:180-185
  if (bc_len == 0 || bc_len > 32) return err_bc_len(bc_len);
  if (umi_len == 0 || umi_len > 32) return err_umi_len(umi_len);
  return IBU_OK;
}
bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

}  // namespace

extern "C" int32_t ibu_device_count(int32_t* n) {
  if (!n) return err_arg("n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return hip_fail(e, "hipGetDeviceCount");
  }
  *n = c;
  return IBU_OK;
}

extern "C" int32_t ibu_ctx_create(int32_t device, ibu_ctx_t** out) {
  if (!out) return err_arg("out is NULL");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
  if (device < 0 || device >= count)
    return set_error(IBU_ERR_NO_DEVICE, 0, 0, 0, "device %d not present (%d visible)", device, count);
  IBU_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  IBU_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(IBU_ERR_NO_DEVICE, 0, 0, 0, "device %d is %s; this library carries gfx950 code objects only",
                     device, prop.gcnArchName);
  ibu_ctx* ctx = new ibu_ctx;
  ctx->device = device;
  ctx->cfg.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // blocks_per_cu keeps LaunchCfg's measured default; ibu_ctx_set_option overrides it
  static const int trace_rows_env = [] { const char* v = getenv("IBU_TRACE_ROWS"); return (v && *v && *v != '0') ? 1 : 0; }();   // read once
  ctx->cfg.trace_rows = trace_rows_env;
  // Where the device hangs off the host (numa.hpp): feeds the ring's placement and the feeder threads' affinity (option "numa").
  // IBU_SYSFS_ROOT: a test tree instead of /sys.  Failure of any step leaves node -1: nothing is pinned, as before round 5.
  if (hipDeviceGetPCIBusId(ctx->pci_bus_id, (int)sizeof ctx->pci_bus_id, device) == hipSuccess)
    numa_lookup(getenv("IBU_SYSFS_ROOT"), ctx->pci_bus_id, nullptr, &ctx->place);
  else
    (void)hipGetLastError();
  hipError_t rc = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
  if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking);
  if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&ctx->d_status), 2 * sizeof(uint64_t));
  if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&ctx->d_acc), kReduceAccBytes);
  if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void**>(&ctx->d_flag), 16);
  if (rc == hipSuccess) rc = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pinned), 16 * sizeof(uint64_t), hipHostMallocDefault);
  if (rc == hipSuccess) rc = hipMemsetAsync(ctx->d_acc, 0, kReduceAccBytes, ctx->stream);
  if (rc == hipSuccess) rc = launch_fill2(ctx->d_status, ~0ull, 0, ctx->stream);
  if (rc == hipSuccess) rc = hipStreamSynchronize(ctx->stream);
  if (rc != hipSuccess) {
    ibu_ctx_destroy(ctx);
    return hip_fail(rc, "ibu_ctx_create");
  }
  *out = ctx;
  return IBU_OK;
}

extern "C" void ibu_ctx_destroy(ibu_ctx_t* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->ring_lent) stream_orphan(ctx);   // a stream still open on this context: stop its producer before the ring goes
  if (ctx->loser_free.joinable()) ctx->loser_free.join();
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
  ring_release(ctx);
  codec_ring_release(ctx);
  for (hipStream_t ps : ctx->pull_streams) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
  for (hipEvent_t pe : ctx->pull_events) (void)hipEventDestroy(pe);
  if (ctx->d_sort_scratch) (void)hipFree(ctx->d_sort_scratch);
  if (ctx->d_runs_scratch) (void)hipFree(ctx->d_runs_scratch);
  if (ctx->d_status) (void)hipFree(ctx->d_status);
  if (ctx->d_acc) (void)hipFree(ctx->d_acc);
  if (ctx->d_flag) (void)hipFree(ctx->d_flag);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->h_part) (void)hipHostFree(ctx->h_part);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  for (hipStream_t q : ctx->inflate_streams)
    if (q) (void)hipStreamDestroy(q);
  if (ctx->d_inflate_stage) (void)hipFree(ctx->d_inflate_stage);
  if (ctx->d_bgzf_range) (void)hipFree(ctx->d_bgzf_range);
  if (ctx->h_inflate_marks) (void)hipHostFree(ctx->h_inflate_marks);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
  delete ctx;
}
extern "C" int32_t ibu_ctx_set_option(ibu_ctx_t* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return err_arg("ctx or key is NULL");
  if (strcmp(key, "blocks_per_cu") == 0) {
    if (value < 1 || value > 8) return err_arg("blocks_per_cu must be 1..8");
    ctx->cfg.blocks_per_cu = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "sort_variant") == 0) {
    if (value < 0 || value >= sort_num_variants()) return err_arg("sort_variant out of range");
    ctx->cfg.sort_variant = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "sort_guess") == 0) {
    if (value < 0 || value > (1ll << 31)) return err_arg("sort_guess out of range");
    ctx->cfg.sort_guess = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "sort_hybrid") == 0) {
    if (value < 0 || value > 2) return err_arg("sort_hybrid must be 0, 1 or 2");
    ctx->cfg.sort_hybrid = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "sort_idx64") == 0) {
    if (value != 0 && value != 1) return err_arg("sort_idx64 must be 0 or 1");
    ctx->cfg.sort_idx64 = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "sort_compact") == 0) {
    if (value < 0 || value > sort_num_compact_variants()) return err_arg("sort_compact out of range");
    ctx->cfg.sort_compact = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "alloc_probe_tries") == 0) {
    if (value < 0 || value > IBU_ALLOC_PROBE_MAX) return err_arg("alloc_probe_tries must be 0 (auto) .. 16");
    ctx->cfg.alloc_probe_tries = (int)value;
    if (ctx->loser_free.joinable()) ctx->loser_free.join();   // (documented: setting the option waits for candidates still being freed)
    return IBU_OK;
  }
  if (strcmp(key, "inflate_one_launch") == 0) {          // a test knob: the block count from which a BGZF load launches its decoder ahead of the copies
    if (value < 0 || value > (int64_t)ctx->cfg.cus * 3 * 64) return err_arg("inflate_one_launch must be 0 (default) .. one round of the decoder's short form");
    ctx->inflate_one_launch = (size_t)value;
    return IBU_OK;
  }
  if (strcmp(key, "bgzf_range_bytes") == 0) {            // a test knob: the ranges of ibu_reader_process_device's BGZF path
    if (value < 0) return err_arg("bgzf_range_bytes must be >= 0");
    ctx->bgzf_range_bytes_opt = (size_t)value;
    return IBU_OK;
  }
  if (strcmp(key, "bgzf_device") == 0) {
    if (value != 0 && value != 1) return err_arg("bgzf_device must be 0 or 1");
    ctx->bgzf_device = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "load_piece_delay_ms") == 0) {         // a test knob: a slow source for the BGZF loads
    if (value < 0 || value > 10000) return err_arg("load_piece_delay_ms must be 0 .. 10000");
    ctx->load_piece_delay_ms = (uint32_t)value;
    return IBU_OK;
  }
  if (strcmp(key, "release_staging") == 0) {             // one-shot: the device staging ibu_load_bgzf_*_to_device keeps (the compressed file's size) goes back now
    if (value != 1) return err_arg("release_staging must be 1");
    IBU_HIP(hipSetDevice(ctx->device));
    for (hipStream_t q : ctx->inflate_streams)
      if (q) IBU_HIP(hipStreamSynchronize(q));
    if (ctx->d_inflate_stage) IBU_HIP(hipFree(ctx->d_inflate_stage));
    ctx->d_inflate_stage = nullptr;
    ctx->inflate_stage_bytes = 0;
    if (ctx->d_bgzf_range) IBU_HIP(hipFree(ctx->d_bgzf_range));
    ctx->d_bgzf_range = nullptr;
    ctx->bgzf_range_bytes = 0;
    return IBU_OK;
  }
  if (strcmp(key, "sort_pull_streams") == 0) {
    if (value != 0 && value != 1) return err_arg("sort_pull_streams must be 0 or 1");
    ctx->force_pull_streams = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "peer_access") == 0) {
    if (value != 0 && value != 1) return err_arg("peer_access must be 0 or 1");
    ctx->peer_access = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "numa") == 0) {
    if (value != 0 && value != 1) return err_arg("numa must be 0 (off) or 1 (auto)");
    ctx->numa_mode = (int)value;
    return IBU_OK;
  }
  if (strcmp(key, "trace_rows") == 0) {
    ctx->cfg.trace_rows = value != 0;
    return IBU_OK;
  }
  if (strcmp(key, "base_order") == 0) {
    if (value != IBU_BASE_ORDER_LSB_FIRST && value != IBU_BASE_ORDER_MSB_FIRST) return err_arg("base_order must be 0 (LSB first) or 1 (MSB first)");
    ctx->cfg.base_order = (uint32_t)value;
    return IBU_OK;
  }
  return err_arg("unknown option key");
}
extern "C" int32_t ibu_ctx_device(const ibu_ctx_t* ctx) { return ctx ? ctx->device : -1; }
extern "C" int32_t ibu_ctx_numa(const ibu_ctx_t* ctx, ibu_numa_info_t* out) {
  if (!ctx || !out) return err_arg("ctx or out is NULL");
  memset(out, 0, sizeof *out);
  out->mode = ctx->numa_mode;
  out->node = ctx->place.node;
  out->usable_cpus = ctx->place.ncpus;
  out->ring_node = ctx->ring.slots ? ctx->ring.node : -1;
  out->ring_placed = ctx->ring.slots && ctx->ring.placed ? 1 : 0;
  snprintf(out->pci_bus_id, sizeof out->pci_bus_id, "%s", ctx->pci_bus_id);
  snprintf(out->cpulist, sizeof out->cpulist, "%s", ctx->place.cpulist);
  return IBU_OK;
}
extern "C" void* ibu_ctx_stream(const ibu_ctx_t* ctx) { return ctx ? ctx->stream : nullptr; }
extern "C" int32_t ibu_ctx_synchronize(ibu_ctx_t* ctx, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  IBU_HIP(hipStreamSynchronize(pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_device_alloc(ibu_ctx_t* ctx, size_t bytes, void** d_ptr) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (!d_ptr) return err_arg("d_ptr is NULL");
  return ibu::ctx_alloc(ctx, bytes, d_ptr);
}
// Long-lived device memory the library allocates for a caller: with the context option "alloc_probe_tries" > 1, arrays of at least
// 256 MiB go through the placement probing below (smaller ones are not worth the candidates' time: the ring's slots, scratch).
// Option "alloc_probe_tries": how many candidates an allocation of `bytes` draws.  Auto (0) probes only what is worth it and
// cannot hurt: at least 1 GiB (the arrays whose placement the streaming kernels feel; smaller ones are ring slots and scratch
// tables) and at least three candidates fitting the free memory at once (so that the candidates never push a caller out of memory).
static int32_t alloc_probed_impl(ibu_ctx_t* ctx, size_t bytes, uint32_t tries, void** d_ptr, ibu_alloc_probe_t* report, double alloc_budget_ms,
                                 double* slow_ms);
static uint32_t probe_tries_for(const ibu_ctx* ctx, size_t bytes) {
  const int t = ctx->cfg.alloc_probe_tries;
  if (t == 1) return 1;
  if (t > 1) return bytes >= ((size_t)256 << 20) ? (uint32_t)t : 1u;
  if (bytes < ((size_t)1 << 30)) return 1;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 1; }
  const size_t fit = free_b / bytes;
  return fit < 3 ? 1u : (fit < 4 ? (uint32_t)fit : 4u);
}
hipError_t ibu::ctx_malloc(ibu_ctx* ctx, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  if (e == hipErrorOutOfMemory && ctx->loser_free.joinable()) {
    (void)hipGetLastError();
    ctx->loser_free.join();
    e = hipMalloc(p, bytes ? bytes : 16);
    if (trace_sort()) fprintf(stderr, "ibu alloc: %zu bytes did not fit while candidates were being freed: waited for them, %s\n", bytes, e == hipSuccess ? "fits now" : "still does not fit");
  }
  return e;
}
int32_t ibu::ctx_alloc(ibu_ctx* ctx, size_t bytes, void** d_ptr) {
  const uint32_t tries = probe_tries_for(ctx, bytes);
  if (tries > 1) {
    ibu_alloc_probe_t rep;
    // auto: a candidate may take 10 ms + 2 ms per GB to allocate (the runtime's usual 0.1-1 ms with a wide margin); beyond that the
    // driver is busy handing out cleared memory and drawing more candidates would multiply the wait, not the choice
    const double budget = ctx->cfg.alloc_probe_tries == 0 ? 10.0 + 2.0 * ((double)bytes / 1e9) : 0.0;
    double slow = 0;
    const int32_t rc = alloc_probed_impl(ctx, bytes, tries, d_ptr, &rep, budget, &slow);
    if (rc == IBU_OK && trace_sort()) {
      if (rep.tries > 1) {
        fprintf(stderr, "ibu alloc: %zu bytes probed: %u candidates, kept #%u (ms:", bytes, rep.tries, rep.chosen);
        for (uint32_t k = 0; k < rep.tries; ++k) fprintf(stderr, " %.3f", rep.ms[k]);
        fprintf(stderr, ")%s\n", slow > 0 ? "; the drawing stopped at a slow allocation" : "");
      } else {
        fprintf(stderr, "ibu alloc: %zu bytes not probed: the allocation took %.1f ms (the driver is handing out memory slowly)\n", bytes, slow);
      }
    }
    return rc;
  }
  IBU_HIP(ctx_malloc(ctx, d_ptr, bytes));
  return IBU_OK;
}
// Placement probing behind the ABI (round 3; bench.py did this in Python in round 2).  On this part the rate of a streaming
// kernel depends on WHERE the driver put an array's physical pages (the same kernel runs 9.4 ... 11.3 ms from one allocation to
// the next; the rate belongs to the physical region and an allocation keeps it for as long as it lives: profiles/README.md
// r02_ag).  A caller that keeps an array resident can therefore choose: allocate `tries` candidates — all held at once, so that
// the allocator has to hand out different pages —, stream a write and a read over each (the write-only generator kernel and the
// read-only reduce kernel over the whole range, timed with events on the context's stream, second run of two), keep the
// fastest, free the rest.  Allocation stops quietly at the first candidate that does not fit.  The contents are unspecified.
// alloc_budget_ms > 0 (the auto mode): a candidate whose hipMalloc took longer than that ends the drawing — what costs in probing is
// not the measuring (0.8 ms per GB and candidate) but the driver handing out memory slowly (right after large frees it clears VRAM:
// seconds for 24 GB, profiles/README.md r05_h), and every further candidate would pay that again.  *slow_ms: that allocation's time.
static int32_t alloc_probed_impl(ibu_ctx_t* ctx, size_t bytes, uint32_t tries, void** d_ptr, ibu_alloc_probe_t* report, double alloc_budget_ms,
                                 double* slow_ms) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (!d_ptr) return err_arg("d_ptr is NULL");
  if (slow_ms) *slow_ms = 0;
  if (report) memset(report, 0, sizeof *report);
  if (tries < 1) tries = 1;
  if (tries > IBU_ALLOC_PROBE_MAX) tries = IBU_ALLOC_PROBE_MAX;
  const size_t nrec = bytes / IBU_RECORD_SIZE;
  if (nrec < 4096) tries = 1;                   // nothing to measure on a few kilobytes
  void* cand[IBU_ALLOC_PROBE_MAX] = {nullptr};
  float ms[IBU_ALLOC_PROBE_MAX] = {0};
  uint32_t got = 0;
  for (; got < tries; ++got) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = got == 0 ? ctx_malloc(ctx, &cand[got], bytes) : hipMalloc(&cand[got], bytes ? bytes : 16);   // (further candidates: what fits now)
    if (e != hipSuccess) {
      (void)hipGetLastError();
      cand[got] = nullptr;
      if (got == 0) return hip_fail(e, "hipMalloc");
      break;                                    // the candidates that exist are the field
    }
    const double ms_alloc = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (trace_sort() && tries > 1) fprintf(stderr, "ibu alloc: candidate %u of %zu bytes: hipMalloc %.2f ms\n", got, bytes, ms_alloc);
    if (alloc_budget_ms > 0 && ms_alloc > alloc_budget_ms) {   // the driver is slow right now: no further candidates
      if (slow_ms) *slow_ms = ms_alloc;
      ++got;
      break;
    }
  }
  uint32_t best = 0;
  if (got > 1) {
    // The read half of the probe is the reduce kernel; it accumulates into a PRIVATE scratch accumulator, never into ctx->d_acc:
    // reset; {load_to_device; reduce} x N; fetch must add up over all N files whether or not the loads were probed.
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint64_t* d_probe_acc = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_probe_acc), kReduceAccBytes);
    if (e == hipSuccess) e = hipMemsetAsync(d_probe_acc, 0, kReduceAccBytes, ctx->stream);
    for (uint32_t k = 0; k < got && e == hipSuccess; ++k)
      for (int run = 0; run < 2 && e == hipSuccess; ++run) {   // the first run touches the pages' translations
        e = hipEventRecord(e0, ctx->stream);
        if (e == hipSuccess) e = launch_generate(ctx->cfg, 0x1B0, 0, nrec, 32, 32, cand[k], ctx->stream);
        if (e == hipSuccess) e = launch_reduce(ctx->cfg, cand[k], nrec, d_probe_acc, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms[k], e0, e1);
      }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_probe_acc) (void)hipFree(d_probe_acc);
    if (e != hipSuccess) {
      for (uint32_t k = 0; k < got; ++k) (void)hipFree(cand[k]);
      return hip_fail(e, "ibu_device_alloc_probed");
    }
    for (uint32_t k = 1; k < got; ++k)
      if (ms[k] < ms[best]) best = k;
    // The candidates not kept go back on a helper thread: hipFree of a block kernels have touched takes tens of milliseconds per
    // gigabyte-sized block in a busy process (three losers of 2.4 GB: 100-200 ms, four times the load they were probed for), and
    // nothing the caller does next depends on it.  One helper at a time per context; the next probing allocation and
    // ibu_ctx_destroy join it.
    if (ctx->loser_free.joinable()) ctx->loser_free.join();
    std::vector<void*> losers;
    bool deferred = false;
    try {
      for (uint32_t k = 0; k < got; ++k)
        if (k != best) losers.push_back(cand[k]);
      const int dev = ctx->device;
      const bool trace = trace_sort();
      ctx->loser_free = std::thread([losers, dev, trace] {
        const auto tf = std::chrono::steady_clock::now();
        (void)hipSetDevice(dev);
        for (void* q : losers) (void)hipFree(q);
        if (trace) fprintf(stderr, "ibu alloc: freed the %zu candidates not kept in %.2f ms (helper thread)\n", losers.size(),
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf).count());
      });
      deferred = true;
    } catch (...) {
    }
    if (!deferred)
      for (uint32_t k = 0; k < got; ++k)
        if (k != best) (void)hipFree(cand[k]);
  }
  *d_ptr = cand[best];
  if (report) {
    report->tries = got;
    report->chosen = best;
    for (uint32_t k = 0; k < got; ++k) report->ms[k] = ms[k];
  }
  return IBU_OK;
}
extern "C" int32_t ibu_device_alloc_probed(ibu_ctx_t* ctx, size_t bytes, uint32_t tries, void** d_ptr, ibu_alloc_probe_t* report) {
  return alloc_probed_impl(ctx, bytes, tries, d_ptr, report, 0, nullptr);
}
extern "C" int32_t ibu_device_free(ibu_ctx_t* ctx, void* d_ptr) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  IBU_HIP(hipFree(d_ptr));
  return IBU_OK;
}
extern "C" int32_t ibu_memcpy_h2d(ibu_ctx_t* ctx, void* d_dst, const void* h_src, size_t bytes, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (bytes) IBU_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_memcpy_d2h(ibu_ctx_t* ctx, void* h_dst, const void* d_src, size_t bytes, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (bytes) IBU_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, pick_stream(ctx, stream)));
  return IBU_OK;
}

// ---- K1 / K1' ------------------------------------------------------------------------------
extern "C" int32_t ibu_deserialize(ibu_ctx_t* ctx, const void* d_records, size_t n, uint64_t* d_barcode,
                                   uint64_t* d_umi, uint64_t* d_index, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (n == 0) return IBU_OK;
  if (!d_records || !d_barcode || !d_umi || !d_index) return err_arg("NULL device pointer");
  if (!aligned8(d_records) || !aligned8(d_barcode) || !aligned8(d_umi) || !aligned8(d_index))
    return err_arg("u64 data must be 8-byte aligned");
  IBU_HIP(launch_deserialize(ctx->cfg, d_records, n, d_barcode, d_umi, d_index, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_serialize(ibu_ctx_t* ctx, const uint64_t* d_barcode, const uint64_t* d_umi,
                                 const uint64_t* d_index, size_t n, void* d_records, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (n == 0) return IBU_OK;
  if (!d_records || !d_barcode || !d_umi || !d_index) return err_arg("NULL device pointer");
  if (!aligned8(d_records) || !aligned8(d_barcode) || !aligned8(d_umi) || !aligned8(d_index))
    return err_arg("u64 data must be 8-byte aligned");
  IBU_HIP(launch_serialize(ctx->cfg, d_barcode, d_umi, d_index, n, d_records, pick_stream(ctx, stream)));
  return IBU_OK;
}

// ---- column codec ----------------------------------------------------------------------------
extern "C" int32_t ibu_unpack_2bit(ibu_ctx_t* ctx, const uint64_t* d_codes, size_t n, uint32_t len, uint8_t* d_ascii,
                                   void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (len == 0 || len > 32) return err_seq_len(len);
  if (n == 0) return IBU_OK;
  if (!d_codes || !d_ascii) return err_arg("NULL device pointer");
  if (!aligned8(d_codes)) return err_arg("u64 data must be 8-byte aligned");
  IBU_HIP(launch_unpack(ctx->cfg, d_codes, n, len, d_ascii, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_pack_2bit(ibu_ctx_t* ctx, const uint8_t* d_ascii, size_t n, uint32_t len, uint64_t* d_codes,
                                 void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (len == 0 || len > 32) return err_seq_len(len);
  if (n == 0) return IBU_OK;
  if (!d_codes || !d_ascii) return err_arg("NULL device pointer");
  if (!aligned8(d_codes)) return err_arg("u64 data must be 8-byte aligned");
  IBU_HIP(launch_pack(ctx->cfg, d_ascii, n, len, d_codes, ctx->d_status, pick_stream(ctx, stream)));
  return IBU_OK;
}

// ---- K2 / K3 -----------------------------------------------------------------------------------
extern "C" int32_t ibu_decode_ascii(ibu_ctx_t* ctx, const void* d_records, size_t n, uint32_t bc_len, uint32_t umi_len,
                                    uint8_t* d_bc_ascii, uint8_t* d_umi_ascii, uint64_t* d_index, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  rc = check_lens(bc_len, umi_len);
  if (rc) return rc;
  if (n == 0) return IBU_OK;
  if (!d_records) return err_arg("d_records is NULL");
  if (!aligned8(d_records) || !aligned8(d_index)) return err_arg("u64 data must be 8-byte aligned");
  IBU_HIP(launch_decode(ctx->cfg, d_records, n, bc_len, umi_len, d_bc_ascii, d_umi_ascii, d_index,
                        pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_encode_ascii(ibu_ctx_t* ctx, const uint8_t* d_bc_ascii, const uint8_t* d_umi_ascii,
                                    const uint64_t* d_index, uint64_t first_index, size_t n, uint32_t bc_len,
                                    uint32_t umi_len, void* d_records, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  rc = check_lens(bc_len, umi_len);
  if (rc) return rc;
  if (n == 0) return IBU_OK;
  if (!d_records || !d_bc_ascii || !d_umi_ascii) return err_arg("NULL device pointer");
  if (!aligned8(d_records) || !aligned8(d_index)) return err_arg("u64 data must be 8-byte aligned");
  IBU_HIP(launch_encode(ctx->cfg, d_bc_ascii, d_umi_ascii, d_index, first_index, n, bc_len, umi_len, d_records,
                        ctx->d_status, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_codec_status(ibu_ctx_t* ctx, void* stream, uint64_t* first_bad_record, uint64_t* n_bad_records) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  hipStream_t st = pick_stream(ctx, stream);
  IBU_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_status, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  IBU_HIP(hipStreamSynchronize(st));
  const uint64_t first = ctx->h_pinned[0], nbad = ctx->h_pinned[1];
  if (first_bad_record) *first_bad_record = first;
  if (n_bad_records) *n_bad_records = nbad;
  if (nbad == 0) return IBU_OK;
  IBU_HIP(launch_fill2(ctx->d_status, ~0ull, 0, st));  // re-arm
  return set_error(IBU_ERR_INVALID_BASE, first, nbad, 0,
                   "Invalid base: %llu record(s) hold a byte outside ACGTacgt, first at record %llu",
                   (unsigned long long)nbad, (unsigned long long)first);
}

// ---- K4 ---------------------------------------------------------------------------------------------
extern "C" int32_t ibu_reduce_reset(ibu_ctx_t* ctx, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  IBU_HIP(hipMemsetAsync(ctx->d_acc, 0, kReduceAccBytes, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_reduce(ibu_ctx_t* ctx, const void* d_records, size_t n, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (n == 0) return IBU_OK;
  if (!d_records || !aligned8(d_records)) return err_arg("d_records must be non-NULL and 8-byte aligned");
  IBU_HIP(launch_reduce(ctx->cfg, d_records, n, ctx->d_acc, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_reduce_fetch(ibu_ctx_t* ctx, void* stream, ibu_reduce_result_t* out) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (!out) return err_arg("out is NULL");
  hipStream_t st = pick_stream(ctx, stream);
  IBU_HIP(launch_reduce_fold(ctx->d_acc, st));
  IBU_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_acc, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  IBU_HIP(hipStreamSynchronize(st));
  out->count = ctx->h_pinned[0];
  for (int k = 0; k < 3; ++k) {
    out->sum[k] = ctx->h_pinned[1 + k];
    out->xor_[k] = ctx->h_pinned[4 + k];
  }
  return IBU_OK;
}

// ---- synthetic records, sortedness, sort ---------------------------------------------------------------
extern "C" int32_t ibu_generate(ibu_ctx_t* ctx, uint64_t seed, uint64_t first, size_t n, uint32_t bc_len,
                                uint32_t umi_len, void* d_records, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  rc = check_lens(bc_len, umi_len);
  if (rc) return rc;
  if (n == 0) return IBU_OK;
  if (!d_records || !aligned8(d_records)) return err_arg("d_records must be non-NULL and 8-byte aligned");
  IBU_HIP(launch_generate(ctx->cfg, seed, first, n, bc_len, umi_len, d_records, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_device_copy(ibu_ctx_t* ctx, void* d_dst, const void* d_src, size_t bytes, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (bytes == 0) return IBU_OK;
  if (!d_dst || !d_src) return err_arg("NULL device pointer");
  const uintptr_t a = reinterpret_cast<uintptr_t>(d_dst), b = reinterpret_cast<uintptr_t>(d_src);
  if (a < b + bytes && b < a + bytes) return err_arg("source and destination overlap");
  IBU_HIP(launch_copy(ctx->cfg, d_src, d_dst, bytes, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_lower_bound_records(ibu_ctx_t* ctx, const void* d_sorted_records, size_t n, const void* d_keys, size_t k,
                                           uint64_t* d_pos, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (k == 0) return IBU_OK;
  if (k > (1u << 20)) return err_arg("at most 2^20 keys per call");
  if (!d_keys || !d_pos || !aligned8(d_keys) || !aligned8(d_pos)) return err_arg("d_keys / d_pos must be non-NULL and 8-byte aligned");
  if (n && (!d_sorted_records || !aligned8(d_sorted_records))) return err_arg("d_sorted_records must be non-NULL and 8-byte aligned");
  IBU_HIP(launch_lower_bound(d_sorted_records, n, d_keys, k, d_pos, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_is_sorted(ibu_ctx_t* ctx, const void* d_records, size_t n, void* stream, int32_t* sorted) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (!sorted) return err_arg("sorted is NULL");
  *sorted = 1;
  if (n < 2) return IBU_OK;
  if (!d_records || !aligned8(d_records)) return err_arg("d_records must be non-NULL and 8-byte aligned");
  hipStream_t st = pick_stream(ctx, stream);
  IBU_HIP(hipMemsetAsync(ctx->d_flag, 0, 4, st));
  IBU_HIP(launch_sorted_check(ctx->cfg, d_records, n, ctx->d_flag, st));
  IBU_HIP(hipMemcpyAsync(ctx->h_pinned + 8, ctx->d_flag, 4, hipMemcpyDeviceToHost, st));
  IBU_HIP(hipStreamSynchronize(st));
  *sorted = (*reinterpret_cast<uint32_t*>(ctx->h_pinned + 8)) == 0;
  return IBU_OK;
}
// `a == b` on two device-resident record slices, with the position (Record: PartialEq / Eq, record.rs:58).
extern "C" int32_t ibu_records_first_mismatch(ibu_ctx_t* ctx, const void* d_a, const void* d_b, size_t n, uint64_t* first,
                                              void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (!first) return err_arg("first is NULL");
  *first = n;
  if (n == 0) return IBU_OK;
  if (!d_a || !d_b || !aligned8(d_a) || !aligned8(d_b)) return err_arg("d_a / d_b must be non-NULL and 8-byte aligned");
  if (n > (~(size_t)0) / 3) return err_arg("n too large");
  hipStream_t st = pick_stream(ctx, stream);
  uint64_t* slot = reinterpret_cast<uint64_t*>(ctx->d_flag) + 1;   // d_flag: 16 bytes; [0] is the sortedness flag
  IBU_HIP(hipMemsetAsync(slot, 0xFF, 8, st));
  IBU_HIP(launch_mismatch(ctx->cfg, d_a, d_b, 3 * n, slot, st));
  IBU_HIP(hipMemcpyAsync(ctx->h_pinned + 10, slot, 8, hipMemcpyDeviceToHost, st));
  IBU_HIP(hipStreamSynchronize(st));
  const uint64_t w = ctx->h_pinned[10];
  if (w != ~0ull) *first = w / 3;
  return IBU_OK;
}
// ---- compacted keys: census, plan, records <-> 12-byte elements (the exchange format of the multi-GPU sort) ----------
static_assert(sizeof(ibu_key_plan_t) == sizeof(ibu::CompactPlan) && offsetof(ibu_key_plan_t, base) == offsetof(ibu::CompactPlan, base) &&
                  offsetof(ibu_key_plan_t, k) == offsetof(ibu::CompactPlan, k),
              "ibu_key_plan_t is the kernels' CompactPlan");
static_assert(sizeof(ibu_inflate_block_t) == sizeof(InflateBlockDesc) && offsetof(ibu_inflate_block_t, crc32) == offsetof(InflateBlockDesc, crc) &&
              offsetof(ibu_inflate_block_t, out_offset) == offsetof(InflateBlockDesc, ooff) && IBU_INFLATE_PAD == kInflatePad,
              "ibu_inflate_block_t is the kernel's descriptor");
extern "C" int32_t ibu_inflate_blocks_device(ibu_ctx_t* ctx, const void* d_comp, const ibu_inflate_block_t* d_blocks, size_t n, void* d_out,
                                             uint32_t* d_status, uint32_t* d_first_bad, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (n == 0) return IBU_OK;
  if (!d_comp || !d_blocks || !d_out || !d_status || !d_first_bad) return err_arg("NULL argument");
  if (n >= (1ull << 31)) return err_arg("fewer than 2^31 blocks per call");
  rc = ensure_sort_scratch(ctx, inflate_scratch_bytes(ctx->cfg, n));   // (the lanes' symbol tables: the context's scratch, as the sort's)
  if (rc) return rc;
  IBU_HIP(launch_inflate_blocks(ctx->cfg, d_comp, reinterpret_cast<const InflateBlockDesc*>(d_blocks), n, d_out, d_status, d_first_bad,
                                ctx->d_sort_scratch, ctx->sort_scratch_bytes, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_records_census(ibu_ctx_t* ctx, const void* d_records, size_t n, uint64_t out[8], void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (!out) return err_arg("out is NULL");
  if (n && (!d_records || !aligned8(d_records))) return err_arg("d_records must be non-NULL and 8-byte aligned");
  rc = ensure_sort_scratch(ctx, 4096);   // the census slots (sort.hip: kCensusBytes)
  if (rc) return rc;
  hipStream_t st = pick_stream(ctx, stream);
  uint64_t* d_c = static_cast<uint64_t*>(ctx->d_sort_scratch);
  IBU_HIP(launch_records_census(ctx->cfg, d_records, n, d_c, st));
  IBU_HIP(hipMemcpyAsync(out, d_c, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  IBU_HIP(hipStreamSynchronize(st));
  return IBU_OK;
}
extern "C" int32_t ibu_key_plan_init(const uint64_t or_words[3], const uint64_t and_words[3], ibu_key_plan_t* plan) {
  if (!or_words || !and_words || !plan) return err_arg("NULL argument");
  compact_plan_init(or_words, and_words, reinterpret_cast<CompactPlan*>(plan));
  return IBU_OK;
}
static int32_t check_plan(const ibu_key_plan_t* plan) {
  if (!plan) return err_arg("plan is NULL");
  if (plan->k > 12) return err_arg("more than 12 key bytes vary: these records do not fit 12-byte elements");
  return IBU_OK;
}
extern "C" int32_t ibu_records_compact(ibu_ctx_t* ctx, const ibu_key_plan_t* plan, const void* d_records, size_t n, void* d_elems,
                                       void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if ((rc = check_plan(plan)) != 0) return rc;
  if (n == 0) return IBU_OK;
  if (!d_records || !aligned8(d_records)) return err_arg("d_records must be non-NULL and 8-byte aligned");
  if (!d_elems || (reinterpret_cast<uintptr_t>(d_elems) & 3u)) return err_arg("d_elems must be non-NULL and 4-byte aligned");
  IBU_HIP(launch_compact(ctx->cfg, *reinterpret_cast<const CompactPlan*>(plan), d_records, n, d_elems, pick_stream(ctx, stream)));
  return IBU_OK;
}
extern "C" int32_t ibu_records_expand(ibu_ctx_t* ctx, const ibu_key_plan_t* plan, const void* d_elems, size_t n, void* d_records,
                                      void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if ((rc = check_plan(plan)) != 0) return rc;
  if (n == 0) return IBU_OK;
  if (!d_records || !aligned8(d_records)) return err_arg("d_records must be non-NULL and 8-byte aligned");
  if (!d_elems || (reinterpret_cast<uintptr_t>(d_elems) & 3u)) return err_arg("d_elems must be non-NULL and 4-byte aligned");
  IBU_HIP(launch_expand(ctx->cfg, *reinterpret_cast<const CompactPlan*>(plan), d_elems, n, d_records, pick_stream(ctx, stream)));
  return IBU_OK;
}
// BarcodeAnalyzer (parallel.rs:72-98) on sorted device records.
extern "C" int32_t ibu_barcode_counts(ibu_ctx_t* ctx, const void* d_sorted_records, size_t n, uint64_t* d_barcodes,
                                      uint64_t* d_counts, uint64_t* d_unique_umis, size_t cap, size_t* n_barcodes,
                                      size_t* n_barcode_umi_pairs, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (!n_barcodes) return err_arg("n_barcodes is NULL");
  *n_barcodes = 0;
  if (n_barcode_umi_pairs) *n_barcode_umi_pairs = 0;
  if (n == 0) return IBU_OK;
  if (!d_sorted_records || !aligned8(d_sorted_records)) return err_arg("d_sorted_records must be non-NULL and 8-byte aligned");
  if (n >= (1ull << 40)) return err_arg("barcode_counts handles fewer than 2^40 records per call");
  hipStream_t st = pick_stream(ctx, stream);
  rc = ensure_sort_scratch(ctx, runs_scratch_bytes(n));
  if (rc) return rc;
  const bool size_query = !d_barcodes && !d_counts && cap == 0;
  // (with outputs to fill, the count pass keeps every segment's first few run heads: the emit pass then reads the records again only
  // where runs are short — k_aggregate.hip)
  IBU_HIP(launch_runs_count(ctx->cfg, d_sorted_records, n, ctx->d_sort_scratch, ctx->sort_scratch_bytes, !size_query, st));
  IBU_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_sort_scratch, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  IBU_HIP(hipStreamSynchronize(st));
  const uint64_t* tot = reinterpret_cast<const uint64_t*>(ctx->h_pinned);
  const uint64_t runs = tot[0], pairs = tot[1];
  *n_barcodes = runs;
  if (n_barcode_umi_pairs) *n_barcode_umi_pairs = pairs;
  if (size_query) return IBU_OK;
  if (!d_barcodes || !d_counts) return err_arg("d_barcodes / d_counts are NULL");
  if (runs > cap) return err_arg("output capacity is smaller than the number of distinct barcodes (see *n_barcodes)");
  const size_t need = runs_emit_scratch_bytes(runs);
  if (need > ctx->runs_scratch_bytes) {
    if (ctx->d_runs_scratch) IBU_HIP(hipFree(ctx->d_runs_scratch));
    ctx->d_runs_scratch = nullptr;
    ctx->runs_scratch_bytes = 0;
    IBU_HIP(ctx_malloc(ctx, &ctx->d_runs_scratch, need));
    ctx->runs_scratch_bytes = need;
  }
  IBU_HIP(launch_runs_emit(ctx->cfg, d_sorted_records, n, ctx->d_sort_scratch, true, ctx->d_runs_scratch, runs, pairs, d_barcodes,
                           d_counts, d_unique_umis, st));
  return IBU_OK;
}
extern "C" int32_t ibu_sort_records(ibu_ctx_t* ctx, void* d_records, void* d_tmp, size_t n, void* stream) {
  int32_t rc = check_ctx(ctx);
  if (rc) return rc;
  if (n < 2) return IBU_OK;
  if (!d_records || !d_tmp || !aligned8(d_records) || !aligned8(d_tmp))
    return err_arg("d_records / d_tmp must be non-NULL and 8-byte aligned");
  if (n >= (1ull << 40)) return err_arg("sort_records handles fewer than 2^40 records per call");  // an argument limit, not a HIP error
  rc = ensure_sort_scratch(ctx, sort_scratch_bytes(ctx->cfg, n));
  if (rc) return rc;
  IBU_HIP(launch_sort_records(ctx->cfg, d_records, d_tmp, n, ctx->d_sort_scratch, ctx->sort_scratch_bytes,
                              pick_stream(ctx, stream)));
  return IBU_OK;
}
