// ctx.hpp — the device context behind ibu_ctx_t (shared by device.cpp and stream.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "common.hpp"
#include "kernels.h"

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
};

namespace ibu {
int32_t ring_ensure(ibu_ctx* ctx, const ibu_ring_config_t* cfg, bool need_dev);
void ring_release(ibu_ctx* ctx);
void codec_ring_release(ibu_ctx* ctx);
// The context's sort scratch (census slots, histograms, digit side stream): grows only; the one allocation a launch path may make.
inline int32_t ensure_sort_scratch(ibu_ctx* ctx, size_t need) {
  if (need > ctx->sort_scratch_bytes) {
    if (ctx->d_sort_scratch) IBU_HIP(hipFree(ctx->d_sort_scratch));
    ctx->d_sort_scratch = nullptr;
    ctx->sort_scratch_bytes = 0;
    IBU_HIP(hipMalloc(&ctx->d_sort_scratch, need));
    ctx->sort_scratch_bytes = need;
  }
  return IBU_OK;
}
int32_t ctx_alloc(ibu_ctx* ctx, size_t bytes, void** d_ptr);   // device.cpp: hipMalloc, or the probed form under option "alloc_probe_tries"
inline hipStream_t pick_stream(const ibu_ctx* ctx, void* stream) {
  return stream ? static_cast<hipStream_t>(stream) : ctx->stream;
}
}  // namespace ibu
