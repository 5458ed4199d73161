// multi_sort.cpp — ibu_sort_records_contexts: the records of several shards, one per context (= per GPU), sorted GLOBALLY by
// (barcode, umi, index) — the order derive(Ord) gives Record (src/constructs/record.rs:58) and the header's sorted flag
// promises (header.rs:111-113) — behind ONE call of the C ABI, the way ibu_mmap_process_contexts is the one-call form of
// process_parallel.  Sample sort: every shard is sorted where it lives, evenly spaced samples of all shards
// pick n - 1 splitters, every shard is cut at the splitters (binary search on the device), the pieces travel to their
// owners device to device (hipMemcpyPeerAsync: over xGMI between GPUs, a plain copy inside one), and every owner sorts what
// it received.  Shard i ends up with the i-th contiguous range of the global order.  A host thread per context, like
// process_contexts (stream.cpp); the phases are separated by joins, which is all the cross-device ordering there is.
//
// The one-process-per-GPU form of the same algorithm is ibu_amd/sharding.py (torch.distributed: all-gather of the samples,
// all-to-all of 12-byte compacted keys); this form needs no collective library and ships the same 12-byte elements whenever
// the keys of all shards allow it.  Neither has run
// on more than one distinct GPU yet (no multi-GPU box in any round's budget): unmeasured.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "ctx.hpp"

using namespace ibu;

namespace {
constexpr size_t kRec = IBU_RECORD_SIZE;
struct Rec { uint64_t b, u, x; };
inline bool rec_lt(const Rec& a, const Rec& c) { return a.b != c.b ? a.b < c.b : (a.u != c.u ? a.u < c.u : a.x < c.x); }

// fn(i) for every context on its own host thread; first error in context order (with its detail) is the call's
template <class F>
int32_t on_every_context(size_t n, F&& fn) {
  std::vector<int32_t> rc;
  std::vector<ibu_error_detail_t> detail;
  try {
    rc.assign(n, IBU_OK);
    detail.resize(n);
  } catch (...) {
    return caught_io("ibu_sort_records_contexts");
  }
  run_pieces((unsigned)n, [&](unsigned i) {
    rc[i] = fn((size_t)i);
    if (rc[i] != IBU_OK) detail[i] = tls_error();
  });
  for (size_t i = 0; i < n; ++i)
    if (rc[i] != IBU_OK) {
      tls_error() = detail[i];
      return rc[i];
    }
  return IBU_OK;
}
}  // namespace

extern "C" int32_t ibu_sort_records_contexts(ibu_ctx_t* const* ctxs, size_t n_ctxs, ibu_sort_shard_t* shards) {
  if (!ctxs || !shards || n_ctxs == 0) return err_arg("NULL argument or no context");
  if (n_ctxs > 1024) return err_arg("more than 1024 contexts");
  const size_t W = n_ctxs;
  for (size_t i = 0; i < W; ++i) {
    if (!ctxs[i]) return err_arg("a context is NULL");
    for (size_t j = 0; j < i; ++j)
      if (ctxs[j] == ctxs[i]) return err_arg("the same context twice (a context serves one host thread; create two on one device instead)");
    const ibu_sort_shard_t& s = shards[i];
    if (s.n > s.capacity) return err_arg("a shard holds more records than its capacity");
    if (s.capacity && (!s.d_records || !s.d_tmp)) return err_arg("a shard's d_records / d_tmp is NULL");
    if ((reinterpret_cast<uintptr_t>(s.d_records) | reinterpret_cast<uintptr_t>(s.d_tmp)) & 7u) return err_arg("d_records / d_tmp must be 8-byte aligned");
    if (W > 1 && s.capacity < W + 1) return err_arg("a shard's capacity must be at least the number of shards + 1 (the splitters are staged in d_tmp)");
  }
  // 1. every shard sorted where it lives
  int32_t rc = on_every_context(W, [&](size_t i) -> int32_t {
    int32_t r = ibu_sort_records(ctxs[i], shards[i].d_records, shards[i].d_tmp, shards[i].n, nullptr);
    return r ? r : ibu_ctx_synchronize(ctxs[i], nullptr);
  });
  if (rc || W == 1) return rc;

  try {
    // 2. samples: up to 64 W evenly spaced records of every shard (one strided copy each), sorted on the host; splitter k = the
    //    sample at k / W of them
    const size_t per = 64 * W;
    std::vector<std::vector<Rec>> samp(W);
    rc = on_every_context(W, [&](size_t i) -> int32_t {
      const size_t n = shards[i].n, take = n < per ? n : per;
      try { samp[i].resize(take); } catch (...) { return caught_io("ibu_sort_records_contexts"); }
      if (!take) return IBU_OK;
      IBU_HIP(hipSetDevice(ctxs[i]->device));
      const size_t stride = n / take;                          // >= 1
      IBU_HIP(hipMemcpy2DAsync(samp[i].data(), kRec, shards[i].d_records, stride * kRec, kRec, take, hipMemcpyDeviceToHost, ctxs[i]->stream));
      IBU_HIP(hipStreamSynchronize(ctxs[i]->stream));
      return IBU_OK;
    });
    if (rc) return rc;
    std::vector<Rec> all;
    for (auto& v : samp) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end(), rec_lt);
    std::vector<Rec> split(W - 1);
    for (size_t k = 1; k < W; ++k)
      split[k - 1] = all.empty() ? Rec{~0ull, ~0ull, ~0ull} : all[std::min(all.size() - 1, k * all.size() / W)];

    // 3. every shard cut at the splitters: bound[i][k] = first record of shard i that is >= splitter k (device binary search;
    //    keys and positions staged in the shard's scratch)
    std::vector<std::vector<uint64_t>> bound(W, std::vector<uint64_t>(W + 1, 0));
    rc = on_every_context(W, [&](size_t i) -> int32_t {
      uint8_t* t = static_cast<uint8_t*>(shards[i].d_tmp);
      uint64_t* d_pos = reinterpret_cast<uint64_t*>(t + kRec * (W - 1));
      int32_t r = ibu_memcpy_h2d(ctxs[i], t, split.data(), kRec * (W - 1), nullptr);
      if (!r) r = ibu_lower_bound_records(ctxs[i], shards[i].d_records, shards[i].n, t, W - 1, d_pos, nullptr);
      if (!r) r = ibu_memcpy_d2h(ctxs[i], bound[i].data() + 1, d_pos, 8 * (W - 1), nullptr);
      if (!r) r = ibu_ctx_synchronize(ctxs[i], nullptr);
      bound[i][0] = 0;
      bound[i][W] = shards[i].n;
      return r;
    });
    if (rc) return rc;
    // 4. who receives how much, and where each piece lands in its owner's scratch (pieces in shard order)
    std::vector<size_t> n_out(W, 0);
    std::vector<std::vector<size_t>> land(W, std::vector<size_t>(W, 0));   // land[j][i]: record offset of shard i's piece in owner j
    for (size_t j = 0; j < W; ++j) {
      for (size_t i = 0; i < W; ++i) {
        land[j][i] = n_out[j];
        n_out[j] += (size_t)(bound[i][j + 1] - bound[i][j]);
      }
      if (n_out[j] > shards[j].capacity)
        return set_error(IBU_ERR_INVALID_ARG, n_out[j], shards[j].capacity, 0,
                         "shard %zu would receive %zu records, its capacity is %zu (the shards are sorted locally, nothing was moved)", j,
                         n_out[j], shards[j].capacity);
    }
    // 5. the exchange.  When at most 12 key bytes vary over ALL shards (the census words of every shard combined: one plan for
    //    everybody) the records travel as 12-byte elements — half the bytes on the links (ibu_records_compact / _expand, the
    //    exchange format of ibu_amd/sharding.py): every shard compacts itself into the lower half of its scratch, every owner
    //    pulls its pieces into the upper half, expands them over its records and sorts.  Otherwise 24-byte records travel into
    //    the owner's scratch and are copied over its records before the sort.
    ibu_key_plan_t plan;
    memset(&plan, 0, sizeof plan);
    bool compact = ctxs[0]->cfg.sort_compact != 0;
    if (compact) {
      std::vector<std::array<uint64_t, 8>> words(W);
      rc = on_every_context(W, [&](size_t i) -> int32_t { return ibu_records_census(ctxs[i], shards[i].d_records, shards[i].n, words[i].data(), nullptr); });
      if (rc) return rc;
      uint64_t o[3] = {0, 0, 0}, a[3] = {~0ull, ~0ull, ~0ull};
      for (auto& w : words)
        for (int f = 0; f < 3; ++f) { o[f] |= w[f]; a[f] &= w[3 + f]; }
      rc = ibu_key_plan_init(o, a, &plan);
      if (rc) return rc;
      compact = plan.k <= 12;
    }
    const size_t wire = compact ? 12 : kRec;                  // bytes per record on the links
    if (compact) {
      rc = on_every_context(W, [&](size_t i) -> int32_t {
        int32_t r = ibu_records_compact(ctxs[i], &plan, shards[i].d_records, shards[i].n, shards[i].d_tmp, nullptr);
        return r ? r : ibu_ctx_synchronize(ctxs[i], nullptr);
      });
      if (rc) return rc;
    }
    rc = on_every_context(W, [&](size_t j) -> int32_t {
      IBU_HIP(hipSetDevice(ctxs[j]->device));
      uint8_t* t = static_cast<uint8_t*>(shards[j].d_tmp) + (compact ? 12 * shards[j].capacity : 0);
      for (size_t i = 0; i < W; ++i) {
        const size_t cnt = (size_t)(bound[i][j + 1] - bound[i][j]);
        if (!cnt) continue;
        if (ctxs[i]->device != ctxs[j]->device) {             // direct xGMI access where the topology has it; without, the copy is staged
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, ctxs[j]->device, ctxs[i]->device) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(ctxs[i]->device, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // not fatal: see above
            else if (pe == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
          } else {
            (void)hipGetLastError();
          }
        }
        const uint8_t* src = static_cast<const uint8_t*>(compact ? shards[i].d_tmp : shards[i].d_records) + wire * bound[i][j];
        IBU_HIP(hipMemcpyPeerAsync(t + wire * land[j][i], ctxs[j]->device, src, ctxs[i]->device, wire * cnt, ctxs[j]->stream));
      }
      IBU_HIP(hipStreamSynchronize(ctxs[j]->stream));
      return IBU_OK;
    });
    if (rc) return rc;                                        // (joined: nobody overwrites what a peer is still reading)
    rc = on_every_context(W, [&](size_t j) -> int32_t {
      int32_t r = IBU_OK;
      if (n_out[j]) {
        uint8_t* t = static_cast<uint8_t*>(shards[j].d_tmp);
        r = compact ? ibu_records_expand(ctxs[j], &plan, t + 12 * shards[j].capacity, n_out[j], shards[j].d_records, nullptr)
                    : ibu_device_copy(ctxs[j], shards[j].d_records, t, kRec * n_out[j], nullptr);
      }
      if (!r) r = ibu_sort_records(ctxs[j], shards[j].d_records, shards[j].d_tmp, n_out[j], nullptr);
      if (!r) r = ibu_ctx_synchronize(ctxs[j], nullptr);
      return r;
    });
    if (rc) return rc;
    if (getenv("IBU_TRACE_SORT")) fprintf(stderr, "ibu sort: contexts=%zu exchange=%zu bytes per record\n", W, wire);
    for (size_t j = 0; j < W; ++j) shards[j].n = n_out[j];
  } catch (...) {
    return caught_io("ibu_sort_records_contexts");
  }
  return IBU_OK;
}
