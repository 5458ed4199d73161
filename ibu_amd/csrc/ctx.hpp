// ctx.hpp — the device context behind ibu_ctx_t (shared by device.cpp and stream.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

#include "common.hpp"
#include "kernels.h"
#include "numa.hpp"

namespace ibu {

inline int32_t hip_fail(hipError_t e, const char* what) {
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
    return set_error(IBU_ERR_NO_DEVICE, (uint64_t)e, 0, 0, "no usable HIP device (%s): %s", what, hipGetErrorString(e));
  return set_error(IBU_ERR_HIP, (uint64_t)e, 0, 0, "HIP error in %s: %s", what, hipGetErrorString(e));
}
#define IBU_HIP(call)                                  \
  do {                                                 \
    hipError_t e__ = (call);                           \
    if (e__ != hipSuccess) return hip_fail(e__, #call); \
  } while (0)

// Pinned-host + device staging ring used by the record-stream entry points.
struct Ring {
  uint32_t slots = 0;
  size_t slot_bytes = 0;
  std::vector<uint8_t*> pinned;  // hipHostMalloc
  std::vector<uint8_t*> dev;     // hipMalloc, same size (H2D landing zone / D2H source)
  std::vector<hipEvent_t> copied;    // H2D (or D2H) of the slot finished
  std::vector<hipEvent_t> consumed;  // kernel that read the slot finished
  int node = -1;                     // NUMA node the pinned slots' pages are on, as the kernel reports it (-1: it would not say)
  bool placed = false;               // the slots were allocated under a preferred-node policy (option "numa" on and accepted)
};

// Staging ring of the host <-> host codec pipelines (codec_stream.cpp): per slot one AoS side (24 B/record) and
// one column side (up to 32 + 32 + 8 B/record), each as a pinned host buffer and a device buffer.
struct CodecRing {
  uint32_t slots = 0;
  size_t slot_records = 0;
  std::vector<uint8_t*> h_aos, h_col, d_aos, d_col;
  std::vector<uint64_t*> h_status, d_status;  // [first_bad, n_bad] per slot: pinned copy / device slot
  uint32_t events = 0;                        // events created so far (partial construction is released safely)
  std::vector<hipEvent_t> up, done, down;     // H2D finished / kernel finished / D2H finished
};

}  // namespace ibu

struct ibu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;   // compute
  hipStream_t copy_stream = nullptr;
  hipStream_t d2h_stream = nullptr;  // results travel back while the next batch travels in (PCIe is full duplex)
  ibu::LaunchCfg cfg;
  uint64_t* d_status = nullptr;  // [first_bad_record, n_bad_records]
  uint64_t* d_acc = nullptr;     // [count, sum0..2, xor0..2, pad]
  uint32_t* d_flag = nullptr;    // sortedness flag
  uint64_t* h_pinned = nullptr;  // 16 x u64 of pinned host memory for small read-backs
  void* d_sort_scratch = nullptr;
  size_t sort_scratch_bytes = 0;
  void* d_runs_scratch = nullptr;  // per-run starts / pair ranks of ibu_barcode_counts (grows only)
  size_t runs_scratch_bytes = 0;
  ibu::Ring ring;
  ibu::CodecRing cring;
  void* ring_lent = nullptr;       // the open ibu_stream_t that holds `ring` (its producer thread fills the slots): every other ring user is refused meanwhile
  void* d_bgzf_range = nullptr;    // ibu_reader_process_device of a BGZF file: the records of the range being processed (grows only)
  size_t bgzf_range_bytes = 0;
  void* d_inflate_stage = nullptr; // ibu_load_bgzf_to_device: the compressed file, the block descriptors and their status words on the device (grows only)
  size_t inflate_stage_bytes = 0;
  size_t bgzf_range_bytes_opt = 0; // option "bgzf_range_bytes" (a test knob): compressed bytes per range of that path (0: 3.2 GB)
  int bgzf_device = 1;             // option "bgzf_device": ibu_reader_process_device of a BGZF file reads the file itself and inflates on the device (1, default) or goes through the Reader's host inflate (0)
  uint32_t load_piece_delay_ms = 0; // option "load_piece_delay_ms" (a test knob): ibu_load_bgzf_*_to_device sleeps that long before every piece it copies — a slow disk
  size_t inflate_one_launch = 0;   // option "inflate_one_launch": files of more blocks than this get the launch that runs ahead of the copies (0: one round of the short form)
  uint64_t* h_inflate_marks = nullptr; // pinned: [0] = the compressed bytes whose copies the host has seen complete — what a launch that runs ahead of its input looks at
  hipStream_t inflate_streams[3] = {nullptr, nullptr, nullptr};   // ... and the streams its launches go out on (created on first use, kept)
  hipStream_t side_stream = nullptr;   // the multi-GPU sort's shared prefix estimate runs here, beside the pulls (created on first use, kept)
  uint64_t* h_part = nullptr;      // pinned, u64[264]: a partition pass's range starts and census words land here early (multi_sort.cpp)
  std::thread loser_free;          // placement probing: the candidates not kept are freed off the caller's path (hipFree of a touched
                                   // gigabyte-sized block has been seen to take 35-65 ms: profiles/README.md r05_t); joined by the next
                                   // probing allocation and by ibu_ctx_destroy
  std::vector<hipStream_t> pull_streams;   // the multi-GPU sort's pulls: one stream per peer link in use at once (created on demand, multi_sort.cpp)
  std::vector<hipEvent_t> pull_events;
  int force_pull_streams = 0;      // option "sort_pull_streams" (test knob): 1 = pieces of same-device peers travel on their own pull streams too
  int peer_access = 1;             // option "peer_access": 0 = never enable direct peer access for this context's pulls (the runtime stages the copies)
  int numa_mode = 1;               // option "numa": 1 = auto (feed threads and the pinned ring on the device's node), 0 = off
  char pci_bus_id[32] = {0};       // "0000:c1:00.0"
  ibu::NumaPlace place;            // the device's node and its CPUs (node -1 / ncpus 0: unknown -> nothing is pinned)
};

namespace ibu {
int32_t ring_ensure(ibu_ctx* ctx, const ibu_ring_config_t* cfg, bool need_dev);
void ring_release(ibu_ctx* ctx);
void stream_orphan(ibu_ctx* ctx);   // stream.cpp: shut down the open ibu_stream_t that holds the ring (ibu_ctx_destroy)
void codec_ring_release(ibu_ctx* ctx);
int32_t ctx_alloc(ibu_ctx* ctx, size_t bytes, void** d_ptr);   // device.cpp: hipMalloc, or the probed form under option "alloc_probe_tries"
// The context's sort scratch (census slots, histograms, digit side stream): grows only; the one allocation a launch path may make.
// hipMalloc for the library's own device memory.  The candidates a placement probe did not keep are freed on a helper thread
// (ctx->loser_free): until it has finished their memory is still taken, and an allocation that fails for want of memory while it
// runs waits for it and tries once more — a caller never sees an out-of-memory error the synchronous free would not have given.
hipError_t ctx_malloc(ibu_ctx* ctx, void** p, size_t bytes);   // device.cpp

inline int32_t ensure_sort_scratch(ibu_ctx* ctx, size_t need) {
  if (need > ctx->sort_scratch_bytes) {
    if (ctx->d_sort_scratch) IBU_HIP(hipFree(ctx->d_sort_scratch));
    ctx->d_sort_scratch = nullptr;
    ctx->sort_scratch_bytes = 0;
    const int32_t rc = ctx_alloc(ctx, need, &ctx->d_sort_scratch);   // (placement-probed under option "alloc_probe_tries": 1e9 records' side stream is 1.75 GB the passes stream through)
    if (rc) { ctx->d_sort_scratch = nullptr; return rc; }
    ctx->sort_scratch_bytes = need;
  }
  return IBU_OK;
}
// Where the host side of this context's feed belongs (option "numa"): the device's node, or nowhere in particular.
inline const NumaPlace& feed_place(const ibu_ctx* ctx) {
  static const NumaPlace nowhere;
  return ctx->numa_mode ? ctx->place : nowhere;
}
inline hipStream_t pick_stream(const ibu_ctx* ctx, void* stream) {
  return stream ? static_cast<hipStream_t>(stream) : ctx->stream;
}
}  // namespace ibu
