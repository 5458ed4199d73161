// multi_sort.cpp — ibu_sort_records_contexts: the records of several shards, one per context (= per GPU), sorted GLOBALLY by
// (barcode, umi, index) — the order derive(Ord) gives Record (src/constructs/record.rs:58) and the header's sorted flag
// promises (header.rs:111-113) — behind ONE call of the C ABI, the way ibu_mmap_process_contexts is the one-call form of
// process_parallel.  Sample sort: samples of all shards pick n - 1 splitters, every record travels to the
// owner of the range between two splitters device to device (hipMemcpyPeerAsync: over xGMI between GPUs, a plain copy inside
// one), and shard i ends up with the i-th contiguous range of the global order.  A host thread per context, like
// process_contexts (stream.cpp).  Ordering across devices (partition-first forms): the two points at which the HOST needs every
// shard's answer — the samples (to pick the splitters) and the range counts of the partition pass (to cut the owners' pieces; the
// sizes of the peer copies are host arguments) — are joins; from there on the devices order themselves: an owner's sort is
// queued behind its own pulls on its stream and behind the "pulled" events of the owners that read its scratch
// (hipStreamWaitEvent across devices), and the call joins once more at the end.  Round 4 joined after the exchange as well: no
// owner sorted until the slowest pull of ANY owner had finished.
//
// Three forms.  PARTITION FIRST ON ELEMENTS (round 4; keys of which at most 11 bytes vary over ALL shards — 16/12 records with
// indices below 2^32 —, up to 256 shards, 16-byte aligned buffers): nothing is sorted before the exchange.  A shard is compacted to
// 12-byte elements (one plan for all shards from the combined census words — of SAMPLE ranges when the shards are large, checked
// against the exact census the partition pass takes on its way), every element gets the number of its key RANGE (256 of them, cut by
// sampled splitters) in its free top byte, one ordinary element pass on that byte puts the elements in range order (sort.hip:
// launch_partition_elems), the exact counts of that pass say which consecutive ranges an owner gets, the owners pull their
// pieces — 12 bytes per record on the links — and sort them straight into records (launch_sort_elems: the sort's passes without
// its census and compress steps).  Every record is sorted ONCE; round 3 sorted every shard, exchanged, and sorted every owner's
// pieces again (one-GPU rehearsal at 1e9 records: 0.097 s then, 0.048 now).  PARTITION FIRST ON RECORDS (any other key, any 8-byte
// aligned buffer, up to 256 shards): the same scheme on 24-byte records — the range goes into the digit side stream, one 24-byte
// pass whose digit comes from that stream orders the records by range in the shard's scratch, the owners pull their pieces over
// their own (dead) records and sort once; no census (full-range (32,32) records: 0.091 s; (32,12): 0.073).  SORT FIRST (the
// round-3 form: more than 256 shards, or sort_compact = 0 on ctxs[0]): shards sorted where they live, cut at the splitters by
// binary search, 24-byte records (or 12-byte elements when at most 12 bytes vary) exchanged, owners sort again.
//
// The one-process-per-GPU form of the same algorithm is ibu_amd/sharding.py (torch.distributed: all-gather of the samples,
// all-to-all of 12-byte compacted keys).  Neither has run on more than one distinct GPU yet (no multi-GPU box in any round's
// budget): unmeasured, and this entry point is EXPERIMENTAL until it has.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
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

// Direct xGMI access where the topology has it (current device = the puller's); already enabled, or not possible (the runtime
// stages the copy then): neither is an error.  Peer access stays enabled for the rest of the process.
void enable_peer(const ibu_ctx* puller, int peer) {
  const int self = puller->device;
  int can = 0;
  if (puller->peer_access && self != peer && hipDeviceCanAccessPeer(&can, self, peer) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(peer, 0);
  (void)hipGetLastError();
}

// xGMI is point to point: a GPU has one link to each peer, and copies queued on ONE stream run one after the other — one link busy,
// six idle.  An owner's pulls therefore go out on one stream PER PEER DEVICE (up to kPullStreams, created on demand and kept by the
// context), so that the links carry their pieces at the same time; the context's own stream then waits for all of them.  Pieces
// that live on the owner's own device are device-local copies and stay on the context's stream.
constexpr size_t kPullStreams = 7;
// (option "sort_pull_streams" = 1, a test knob: same-device peers — a rehearsal on one GPU — get their own streams too, by shard number,
// so that the fork / join of the pull streams runs where there is only one device)
int32_t pull_stream_for(ibu_ctx* c, int peer_device, size_t peer_shard, hipStream_t* out) {
  if (peer_device == c->device && !c->force_pull_streams) { *out = c->stream; return IBU_OK; }
  const size_t slot = (c->force_pull_streams ? peer_shard : (size_t)peer_device) % kPullStreams;
  while (c->pull_streams.size() <= slot) {
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    IBU_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipStreamDestroy(st); return hip_fail(e, "hipEventCreateWithFlags"); }
    try { c->pull_streams.push_back(st); c->pull_events.push_back(ev); }
    catch (...) { (void)hipStreamDestroy(st); (void)hipEventDestroy(ev); return caught_io("ibu_sort_records_contexts"); }
  }
  *out = c->pull_streams[slot];
  return IBU_OK;
}
// the context's stream continues only when every pull stream has delivered
int32_t join_pull_streams(ibu_ctx* c) {
  for (size_t k = 0; k < c->pull_streams.size(); ++k) {
    IBU_HIP(hipEventRecord(c->pull_events[k], c->pull_streams[k]));
    IBU_HIP(hipStreamWaitEvent(c->stream, c->pull_events[k], 0));
  }
  return IBU_OK;
}
// ... and a pull stream starts only behind what the context's stream has queued so far (the owner's own earlier work on the
// buffers the pulls write)
int32_t fork_pull_streams(ibu_ctx* c, hipEvent_t scratch_event) {
  if (c->pull_streams.empty()) return IBU_OK;
  IBU_HIP(hipEventRecord(scratch_event, c->stream));
  for (hipStream_t ps : c->pull_streams) IBU_HIP(hipStreamWaitEvent(ps, scratch_event, 0));
  return IBU_OK;
}

// Evenly spaced samples of every shard, IN PROPORTION to its size (a shard of 1e3 records beside one of 1e9 does not get the
// same say: ADVICE r03), `budget` in all — one strided copy per shard.
int32_t sample_one(ibu_ctx_t* ctx, const ibu_sort_shard_t& shard, size_t total, size_t budget, std::vector<Rec>& out) {
  const size_t n = shard.n;
  size_t want = total ? (size_t)(((long double)budget * n) / total) + 1 : 0;
  if (want > n) want = n;
  if (!want) { out.clear(); return IBU_OK; }
  // one strided copy: the stride is an integer, so the COUNT is what gives way — take = n / stride samples, centred, reach
  // within half a stride of both ends.  (Round 4 fixed the count and floored the stride: 513 samples at stride 5 of 3001
  // records never saw the top 15 % of a sorted shard, and the last owner of 33 received 5.6 shares.)
  const size_t stride = (n + want - 1) / want;             // >= 1
  const size_t take = n / stride;                          // 1 .. want
  const size_t first = (n - take * stride) / 2 + stride / 2;
  try { out.resize(take); } catch (...) { return caught_io("ibu_sort_records_contexts"); }
  IBU_HIP(hipSetDevice(ctx->device));
  IBU_HIP(hipMemcpy2DAsync(out.data(), kRec, static_cast<const uint8_t*>(shard.d_records) + kRec * first, stride * kRec, kRec, take,
                           hipMemcpyDeviceToHost, ctx->stream));
  IBU_HIP(hipStreamSynchronize(ctx->stream));
  return IBU_OK;
}
int32_t sample_shards(ibu_ctx_t* const* ctxs, const ibu_sort_shard_t* shards, size_t W, size_t budget, std::vector<std::vector<Rec>>& samp) {
  size_t total = 0;
  for (size_t i = 0; i < W; ++i) total += shards[i].n;
  return on_every_context(W, [&](size_t i) -> int32_t { return sample_one(ctxs[i], shards[i], total, budget, samp[i]); });
}
// splitter k = the pooled sample at k / W (W: owners, or the 256 fine ranges of the partition-first form)
void pick_splitters(std::vector<std::vector<Rec>>& samp, size_t W, std::vector<Rec>& split) {
  std::vector<Rec> all;
  for (auto& v : samp) all.insert(all.end(), v.begin(), v.end());
  std::sort(all.begin(), all.end(), rec_lt);
  split.resize(W - 1);
  for (size_t k = 1; k < W; ++k)
    split[k - 1] = all.empty() ? Rec{~0ull, ~0ull, ~0ull} : all[std::min(all.size() - 1, k * all.size() / W)];
}
size_t sample_budget(size_t W) {
  const size_t want = 512 * W;                                // ~512 samples per owner: its share is known to a few percent
  return want < 16384 ? 16384 : (want > (1u << 19) ? (1u << 19) : want);
}

constexpr size_t kPartitionFirstMaxShards = 32;
constexpr int32_t kLandingFailed = -77;   // internal: a shard would receive more than its capacity (detail set); the partition-first forms fall back on it

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// who receives how much, and where each piece lands at its owner (pieces in shard order); a shard without the room fails the call
int32_t plan_landing(const ibu_sort_shard_t* shards, size_t W, const std::vector<std::vector<uint64_t>>& bound, std::vector<size_t>& n_out,
                     std::vector<std::vector<size_t>>& land, const char* state) {
  for (size_t j = 0; j < W; ++j) {
    for (size_t i = 0; i < W; ++i) {
      land[j][i] = n_out[j];
      n_out[j] += (size_t)(bound[i][j + 1] - bound[i][j]);
    }
    if (n_out[j] > shards[j].capacity)
      return set_error(IBU_ERR_INVALID_ARG, n_out[j], shards[j].capacity, 0, "shard %zu would receive %zu records, its capacity is %zu (%s)", j, n_out[j],
                       shards[j].capacity, state);
  }
  return IBU_OK;
}

// Which of the 256 key ranges an owner gets: consecutive ones, up to the point nearest to its share of the records (and not past its
// capacity while an earlier cut avoids that; the last owner takes what is left).  fine[i][f]: first element of range f in shard i
// (fine[i][256] = its count); bound[i][j]: first element of owner j's piece in shard i.
void cut_owners(const ibu_sort_shard_t* shards, size_t W, const std::vector<std::vector<uint64_t>>& fine, size_t total,
                std::vector<std::vector<uint64_t>>& bound) {
  constexpr size_t F = 256;
  std::vector<uint64_t> g(F, 0);
  for (size_t i = 0; i < W; ++i)
    for (size_t f = 0; f < F; ++f) g[f] += fine[i][f + 1] - fine[i][f];
  std::vector<size_t> cut(W + 1, F);
  cut[0] = 0;
  uint64_t cum = 0;
  for (size_t j = 0, f = 0; j + 1 < W; ++j) {
    const long double target = (long double)total * (j + 1) / W;
    uint64_t load = 0;
    while (f < F && load + g[f] <= shards[j].capacity && (long double)cum + (long double)g[f] / 2 <= target) {
      load += g[f];
      cum += g[f];
      ++f;
    }
    cut[j + 1] = f;
  }
  for (size_t i = 0; i < W; ++i)
    for (size_t j = 0; j <= W; ++j) bound[i][j] = fine[i][cut[j]];
}

// Events of one call, destroyed on every way out.
struct EventSet {
  std::vector<hipEvent_t> ev;
  std::vector<int> dev;
  explicit EventSet(size_t n) : ev(n, nullptr), dev(n, 0) {}
  ~EventSet() {
    for (size_t i = 0; i < ev.size(); ++i)
      if (ev[i]) { (void)hipSetDevice(dev[i]); (void)hipEventDestroy(ev[i]); }
  }
};
// A shard's partition pass with its counts handed over early (kernels.h: PartitionEarly): `launch(early)` queues the pass on the
// context's stream; `done` is recorded behind it; the thread waits only until the counts (and census words) have arrived in pinned
// memory — the pass's scatter kernel, a third of its time, is still running when this returns.
template <class Launch>
int32_t partition_with_early_counts(ibu_ctx* c, Launch launch, hipEvent_t* done, int* done_dev, uint64_t* starts_out /*[256]*/, uint64_t* words_out /*[8] or null*/) {
  if (!c->h_part) IBU_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_part), 264 * sizeof(uint64_t), hipHostMallocDefault));
  hipEvent_t ready = nullptr;
  IBU_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  PartitionEarly early{c->h_part, c->h_part + 256, ready};
  hipError_t e = launch(&early);
  if (e == hipSuccess) e = hipEventCreateWithFlags(done, hipEventDisableTiming);
  if (e == hipSuccess) { *done_dev = c->device; e = hipEventRecord(*done, c->stream); }
  if (e == hipSuccess) e = hipEventSynchronize(ready);
  (void)hipEventDestroy(ready);
  if (e != hipSuccess) return hip_fail(e, "partition pass");
  memcpy(starts_out, c->h_part, 256 * sizeof(uint64_t));
  if (words_out) memcpy(words_out, c->h_part + 256, 8 * sizeof(uint64_t));
  return IBU_OK;
}

// The exchange and the owners' sorts without a join between them.  pull(j, i, cnt, first, land, stream): queue owner j's copy of `cnt`
// units of shard i, from unit `first` of its partitioned scratch to unit `land` at the owner, on `stream` (one per peer device).
// sort(j): queue owner j's sort on ctxs[j]->stream (may synchronise that stream itself).  What orders them:
//   - owner j's sort follows its own pulls in its stream;
//   - the sort overwrites shard j's scratch, which the OTHER owners read: it waits (hipStreamWaitEvent, across devices) for the
//     `pulled` event of every owner that takes a piece of shard j.  The events are recorded by an enqueue-only round of the
//     host threads (no device is waited for) before any sort is queued — an event that has not been recorded yet would not be
//     waited for.
// One join at the end.  *t_enqueue_ms / *t_total_ms: for the trace.
template <class Pull, class Sort>
int32_t exchange_then_sort(ibu_ctx_t* const* ctxs, size_t W, const std::vector<std::vector<uint64_t>>& bound, const std::vector<std::vector<size_t>>& land,
                           const std::vector<hipEvent_t>& partitioned /*shard i's partition pass has finished (null: nothing was queued)*/,
                           Pull pull, Sort sort, double* t_enqueue_ms) {
  std::vector<hipEvent_t> pulled(W, nullptr);
  const double t0 = now_ms();
  int32_t rc = on_every_context(W, [&](size_t j) -> int32_t {
    IBU_HIP(hipSetDevice(ctxs[j]->device));
    IBU_HIP(hipEventCreateWithFlags(&pulled[j], hipEventDisableTiming));
    for (size_t i = 0; i < W; ++i) {                            // streams for the peers this owner pulls from, before anything is queued on them
      hipStream_t ps;
      if (bound[i][j + 1] > bound[i][j]) { const int32_t r = pull_stream_for(ctxs[j], ctxs[i]->device, i, &ps); if (r) return r; }
    }
    int32_t r = fork_pull_streams(ctxs[j], pulled[j]);          // (the event is free until it is recorded for good below)
    if (r) return r;
    for (size_t i = 0; i < W; ++i) {
      const size_t cnt = (size_t)(bound[i][j + 1] - bound[i][j]);
      if (!cnt) continue;
      enable_peer(ctxs[j], ctxs[i]->device);
      hipStream_t ps;
      r = pull_stream_for(ctxs[j], ctxs[i]->device, i, &ps);
      if (r) return r;
      // shard i's scatter may still be running (its counts came back early): the pull waits for it on the device
      if (partitioned[i] && ps != ctxs[i]->stream) IBU_HIP(hipStreamWaitEvent(ps, partitioned[i], 0));
      r = pull(j, i, cnt, (size_t)bound[i][j], land[j][i], ps);
      if (r) return r;
    }
    r = join_pull_streams(ctxs[j]);
    if (r) return r;
    IBU_HIP(hipEventRecord(pulled[j], ctxs[j]->stream));
    return IBU_OK;
  });
  *t_enqueue_ms = now_ms() - t0;
  if (rc == IBU_OK)
    rc = on_every_context(W, [&](size_t j) -> int32_t {
      IBU_HIP(hipSetDevice(ctxs[j]->device));
      for (size_t k = 0; k < W; ++k)
        if (k != j && bound[j][k + 1] > bound[j][k]) IBU_HIP(hipStreamWaitEvent(ctxs[j]->stream, pulled[k], 0));   // owner k reads shard j's scratch
      const int32_t r = sort(j);
      if (r) return r;
      IBU_HIP(hipStreamSynchronize(ctxs[j]->stream));
      return IBU_OK;
    });
  if (rc != IBU_OK)                                            // leave nothing in flight over the shards (their contents are unspecified now)
    for (size_t j = 0; j < W; ++j) { (void)hipSetDevice(ctxs[j]->device); (void)hipStreamSynchronize(ctxs[j]->stream); }
  for (size_t j = 0; j < W; ++j)
    if (pulled[j]) { (void)hipSetDevice(ctxs[j]->device); (void)hipEventDestroy(pulled[j]); }
  return rc;
}

// ---- PARTITION FIRST -------------------------------------------------------------------------------------------------------
// `guessed`: the plan comes from SAMPLE censuses (three ranges of every shard); the partition pass then accumulates the exact census of
// every record on the way, and *covered says afterwards whether the guess was the truth — the same bytes vary.  If it was not
// (false), nothing has been exchanged and no record touched: the caller runs the call again with the exact plan it now has.
// `split`: the 255 splitters; empty = sample the shards and pick them, kept for a second run (a plan miss re-runs only the partition pass).
int32_t sort_partition_first(ibu_ctx_t* const* ctxs, size_t W, ibu_sort_shard_t* shards, const CompactPlan& plan, size_t total, bool guessed,
                             std::vector<std::array<uint64_t, 8>>* exact_words, bool* covered, std::vector<Rec>& split) {
  const bool trace = trace_sort();
  double t_mark = now_ms(), t_phase[5] = {0, 0, 0, 0, 0};
  auto lap = [&](int k) { const double t = now_ms(); t_phase[k] = t - t_mark; t_mark = t; };
  // 1. 255 splitters from samples of the (unsorted) shards cut the key space into 256 FINE ranges — more than there are owners:
  //    the samples only have to make the ranges small, the exact counts of the partition pass then say which consecutive ranges
  //    an owner gets (a range is ~1/256 of the records: the owners' loads differ by about that much, not by the sampling noise
  //    of W - 1 splitters drawn from unsorted data)
  constexpr size_t F = 256;
  int32_t rc = IBU_OK;
  if (split.empty()) {
    std::vector<std::vector<Rec>> samp(W);
    rc = sample_shards(ctxs, shards, W, sample_budget(W), samp);
    if (rc) return rc;
    pick_splitters(samp, F, split);
  }
  lap(0);
  // 2. every shard: records -> elements stamped with their range (one kernel) -> range order in the upper half of its scratch;
  //    fine[i][f] = first element of range f.
  //    The largest shard also says how many prefix passes a sort of `total` records like its own wants (it is a sample of them).
  size_t big = 0;
  for (size_t i = 1; i < W; ++i)
    if (shards[i].n > shards[big].n) big = i;
  uint32_t prefix_passes = 0;
  std::vector<std::vector<uint64_t>> fine(W, std::vector<uint64_t>(F + 1, 0));
  EventSet partitioned(W);                                    // shard i's partition pass has finished (its counts arrive earlier)
  const size_t split_bytes = (kRec + 12) * (F - 1) + 512;
  rc = on_every_context(W, [&](size_t i) -> int32_t {
    ibu_ctx_t* c = ctxs[i];
    const size_t n = shards[i].n;
    if (!n) return IBU_OK;
    IBU_HIP(hipSetDevice(c->device));
    const size_t need = sort_scratch_bytes(c->cfg, n);
    int32_t r = ensure_sort_scratch(c, need + split_bytes);   // the splitters ride behind the sort's own scratch
    if (r) return r;
    hipStream_t st = c->stream;
    uint8_t* t = static_cast<uint8_t*>(shards[i].d_tmp);
    uint8_t* d_split_recs = static_cast<uint8_t*>(c->d_sort_scratch) + ((need + 255) & ~(size_t)255);
    uint8_t* d_split_elems = d_split_recs + ((kRec * (F - 1) + 15) & ~(size_t)15);
    if (i == big) IBU_HIP(launch_estimate_prefix(c->cfg, shards[i].d_records, n, total, t, plan, &prefix_passes, st));   // (scratch: the head of d_tmp)
    IBU_HIP(hipMemcpyAsync(d_split_recs, split.data(), kRec * (F - 1), hipMemcpyHostToDevice, st));
    IBU_HIP(launch_compact(c->cfg, plan, d_split_recs, F - 1, d_split_elems, st));
    const uint64_t *d_starts = nullptr, *d_words = nullptr;
    int32_t pr = partition_with_early_counts(
        c,
        [&](const PartitionEarly* early) {
          return launch_partition_elems(c->cfg, plan, shards[i].d_records, t, n, d_split_elems, (uint32_t)(F - 1), t + 12 * shards[i].capacity, c->d_sort_scratch,
                                        need, &d_starts, guessed ? &d_words : nullptr, st, early);
        },
        &partitioned.ev[i], &partitioned.dev[i], fine[i].data(), guessed ? (*exact_words)[i].data() : nullptr);
    if (pr) return pr;
    fine[i][F] = n;
    return IBU_OK;
  });
  if (rc) return rc;
  lap(1);
  if (guessed) {                                              // was the guess the truth?  (a byte that varies in a sample varies; the other way round is the question)
    uint64_t o[3] = {0, 0, 0}, a[3] = {~0ull, ~0ull, ~0ull};
    for (size_t i = 0; i < W; ++i)
      if (shards[i].n)
        for (int f = 0; f < 3; ++f) { o[f] |= (*exact_words)[i][f]; a[f] &= (*exact_words)[i][3 + f]; }
    CompactPlan exact;
    compact_plan_init(o, a, &exact);
    *covered = exact.k == plan.k && memcmp(exact.csel, plan.csel, sizeof exact.csel) == 0 && memcmp(exact.base, plan.base, sizeof exact.base) == 0;
    if (!*covered) {
      if (trace) fprintf(stderr, "ibu sort: contexts=%zu the sampled plan missed a varying byte: again with the exact census\n", W);
      return IBU_OK;
    }
  }
  std::vector<std::vector<uint64_t>> bound(W, std::vector<uint64_t>(W + 1, 0));
  cut_owners(shards, W, fine, total, bound);
  std::vector<size_t> n_out(W, 0);
  std::vector<std::vector<size_t>> land(W, std::vector<size_t>(W, 0));   // land[j][i]: element offset of shard i's piece at owner j
  rc = plan_landing(shards, W, bound, n_out, land, "no shard's records were touched");
  if (rc) return kLandingFailed;
  lap(2);
  // 3. the exchange — every owner pulls its pieces into the LOWER half of its scratch (its own unpartitioned elements: dead) — and
  // 4. every owner sorts what it received, elements -> records, queued behind its pulls and behind the pulls that read ITS upper half
  double t_enq = 0;
  rc = exchange_then_sort(
      ctxs, W, bound, land, partitioned.ev,
      [&](size_t j, size_t i, size_t cnt, size_t first, size_t at, hipStream_t ps) -> int32_t {
        const uint8_t* src = static_cast<const uint8_t*>(shards[i].d_tmp) + 12 * (shards[i].capacity + first);
        IBU_HIP(hipMemcpyPeerAsync(static_cast<uint8_t*>(shards[j].d_tmp) + 12 * at, ctxs[j]->device, src, ctxs[i]->device, 12 * cnt, ps));
        return IBU_OK;
      },
      [&](size_t j) -> int32_t {
        ibu_ctx_t* c = ctxs[j];
        if (!n_out[j]) return IBU_OK;
        const size_t need = sort_scratch_bytes(c->cfg, n_out[j]);
        int32_t r = ensure_sort_scratch(c, need);
        if (r) return r;
        IBU_HIP(launch_sort_elems(c->cfg, plan, shards[j].d_records, shards[j].d_tmp, n_out[j], prefix_passes, c->d_sort_scratch, c->sort_scratch_bytes, c->stream));
        return IBU_OK;
      },
      &t_enq);
  if (rc) return rc;
  lap(3);
  if (trace)
    fprintf(stderr, "ibu sort: contexts=%zu exchange=12 bytes per record (partition first, prefix_passes=%u; host joins: samples, range counts (handed over before the partition's scatter), end; ms: samples %.2f, "
            "partition %.2f, plan %.2f, exchange+sort %.2f of which enqueueing the pulls %.2f)\n", W, prefix_passes, t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_enq);
  for (size_t j = 0; j < W; ++j) shards[j].n = n_out[j];
  return IBU_OK;
}

// ---- PARTITION FIRST on 24-byte records (any key) -------------------------------------------------------------------------------
// The same scheme without the elements: keys of more than 11 varying bytes (full-range (32,32) records: 20), or buffers the element
// kernels cannot take.  A record's key range goes into the digit side stream, one 24-byte pass of the sort puts the records in range
// order in the shard's scratch, the owners pull their pieces over their own (dead) records and sort them once.  The census is taken by the
// partition pass on its way (the stamp kernel reads every record anyway) and shared: no owner runs one of its own.
int32_t sort_partition_first_records(ibu_ctx_t* const* ctxs, size_t W, ibu_sort_shard_t* shards, size_t total, std::vector<Rec>& split) {
  const bool trace = trace_sort();
  double t_mark = now_ms(), t_phase[4] = {0, 0, 0, 0};
  auto lap = [&](int k) { const double t = now_ms(); t_phase[k] = t - t_mark; t_mark = t; };
  constexpr size_t F = 256;
  int32_t rc = IBU_OK;
  if (split.empty()) {                                          // (the caller has them already when it sampled for the element form's census)
    std::vector<std::vector<Rec>> samp(W);
    rc = sample_shards(ctxs, shards, W, sample_budget(W), samp);
    if (rc) return rc;
    pick_splitters(samp, F, split);
  }
  lap(0);
  std::vector<std::vector<uint64_t>> fine(W, std::vector<uint64_t>(F + 1, 0));
  std::vector<std::array<uint64_t, 8>> words(W);
  EventSet partitioned(W);
  const size_t split_bytes = kRec * (F - 1) + 512;
  size_t big = 0;                                               // the biggest shard: the prefix estimate samples it (below)
  for (size_t i = 1; i < W; ++i)
    if (shards[i].n > shards[big].n) big = i;
  const size_t est_tables = sort_prefix_estimate_tables(ctxs[big]->cfg, shards[big].n);
  rc = on_every_context(W, [&](size_t i) -> int32_t {
    ibu_ctx_t* c = ctxs[i];
    const size_t n = shards[i].n;
    if (!n) return IBU_OK;
    IBU_HIP(hipSetDevice(c->device));
    const size_t need = sort_scratch_bytes(c->cfg, n);
    int32_t r = ensure_sort_scratch(c, std::max(need + split_bytes, i == big ? est_tables : (size_t)0));   // the splitters ride behind the sort's own scratch
    if (r) return r;
    hipStream_t st = c->stream;
    uint8_t* d_split_recs = static_cast<uint8_t*>(c->d_sort_scratch) + ((need + 255) & ~(size_t)255);
    IBU_HIP(hipMemcpyAsync(d_split_recs, split.data(), kRec * (F - 1), hipMemcpyHostToDevice, st));
    const uint64_t *d_starts = nullptr, *d_words = nullptr;
    int32_t pr = partition_with_early_counts(
        c,
        [&](const PartitionEarly* early) {
          return launch_partition_records(c->cfg, shards[i].d_records, n, d_split_recs, (uint32_t)(F - 1), shards[i].d_tmp, c->d_sort_scratch, need, &d_starts, &d_words,
                                          st, early);
        },
        &partitioned.ev[i], &partitioned.dev[i], fine[i].data(), words[i].data());   // (the exact OR / AND words of this shard's records, taken on the way)
    if (pr) return pr;
    fine[i][F] = n;
    return IBU_OK;
  });
  if (rc) return rc;
  lap(1);
  // ONE census for everybody: the words of all shards combined say which key bytes vary anywhere; every owner's sort takes them as
  // given and runs no census pass of its own (round 4: eight private ones, 24 B/record read again)
  uint64_t all_words[6] = {0, 0, 0, ~0ull, ~0ull, ~0ull};
  for (size_t i = 0; i < W; ++i)
    if (shards[i].n)
      for (int f = 0; f < 3; ++f) { all_words[f] |= words[i][f]; all_words[3 + f] &= words[i][3 + f]; }
  std::vector<std::vector<uint64_t>> bound(W, std::vector<uint64_t>(W + 1, 0));
  cut_owners(shards, W, fine, total, bound);
  std::vector<size_t> n_out(W, 0);
  std::vector<std::vector<size_t>> land(W, std::vector<size_t>(W, 0));
  rc = plan_landing(shards, W, bound, n_out, land, "no shard's records were touched");
  if (rc) return kLandingFailed;
  // ... and ONE prefix estimate (the prefix + finish path of the owners' sorts: how many leading key bytes leave short runs), as the
  // element form has had: taken on the biggest shard's partitioned records — a sample of every key range — for `total` records, tables
  // in that context's sort scratch (idle between its partition pass and its owner's sort), on a stream of its own behind the partition
  // pass, by a thread of its own: the pulls are queued meanwhile, and an owner asks for the result when it queues its sort, by which
  // time its pulls are still under way.  (Round 4 and before: W private estimates, each a host round trip at the head of an owner's sort.)
  std::shared_future<int> shared_prefix;
  if (shards[big].n >= 2) {
    try {
      shared_prefix = std::async(std::launch::async, [&, big]() -> int {
        ibu_ctx_t* c = ctxs[big];
        int P = -1;
        if (hipSetDevice(c->device) != hipSuccess) return -1;
        if (!c->side_stream && hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) { c->side_stream = nullptr; return -1; }   // (creating a stream stalls the threads that are queueing the pulls: once per context)
        hipStream_t es = c->side_stream;
        hipError_t e = partitioned.ev[big] ? hipStreamWaitEvent(es, partitioned.ev[big], 0) : hipSuccess;
        if (e == hipSuccess)
          e = launch_sort_records(c->cfg, shards[big].d_tmp, c->d_sort_scratch, shards[big].n, c->d_sort_scratch, c->sort_scratch_bytes, es, all_words, -1, &P, total);
        (void)hipStreamSynchronize(es);
        if (e != hipSuccess) { (void)hipGetLastError(); return -1; }   // (every owner estimates for itself then)
        return P;
      }).share();
    } catch (...) { return caught_io("ibu_sort_records_contexts"); }
  }
  lap(2);
  // the exchange — every owner pulls its pieces from the shards' scratch over its own records (partitioned into its scratch: dead) —
  // and the owners' sorts, each behind its own pulls and behind the pulls that read ITS scratch
  double t_enq = 0;
  rc = exchange_then_sort(
      ctxs, W, bound, land, partitioned.ev,
      [&](size_t j, size_t i, size_t cnt, size_t first, size_t at, hipStream_t ps) -> int32_t {
        const uint8_t* src = static_cast<const uint8_t*>(shards[i].d_tmp) + kRec * first;
        IBU_HIP(hipMemcpyPeerAsync(static_cast<uint8_t*>(shards[j].d_records) + kRec * at, ctxs[j]->device, src, ctxs[i]->device, kRec * cnt, ps));
        return IBU_OK;
      },
      [&](size_t j) -> int32_t {
        ibu_ctx_t* c = ctxs[j];
        const int known_prefix = shared_prefix.valid() ? std::shared_future<int>(shared_prefix).get() : -1;   // (first: the estimate's tables live in a context's sort scratch)
        if (n_out[j] < 2) return IBU_OK;
        int32_t r = ensure_sort_scratch(c, sort_scratch_bytes(c->cfg, n_out[j]));
        if (r) return r;
        IBU_HIP(launch_sort_records(c->cfg, shards[j].d_records, shards[j].d_tmp, n_out[j], c->d_sort_scratch, c->sort_scratch_bytes, c->stream, all_words, known_prefix));
        return IBU_OK;
      },
      &t_enq);
  if (rc) return rc;
  lap(3);
  if (trace)
    fprintf(stderr, "ibu sort: contexts=%zu exchange=24 bytes per record (partition first, one prefix estimate for all owners: %d; host joins: samples, range counts (handed over before the partition's scatter), end; ms: samples %.2f, partition %.2f, "
            "plan %.2f, exchange+sort %.2f of which enqueueing the pulls %.2f)\n", W, shared_prefix.valid() ? shared_prefix.get() : -1, t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_enq);
  for (size_t j = 0; j < W; ++j) shards[j].n = n_out[j];
  return IBU_OK;
}

// ---- SORT FIRST (round 3) -----------------------------------------------------------------------------------------------------
int32_t sort_sort_first(ibu_ctx_t* const* ctxs, size_t W, ibu_sort_shard_t* shards, const CompactPlan& census_plan, bool have_plan) {
  // 1. every shard sorted where it lives
  int32_t rc = on_every_context(W, [&](size_t i) -> int32_t {
    int32_t r = ibu_sort_records(ctxs[i], shards[i].d_records, shards[i].d_tmp, shards[i].n, nullptr);
    return r ? r : ibu_ctx_synchronize(ctxs[i], nullptr);
  });
  if (rc) return rc;
  // 2. samples (of sorted shards: exact local quantiles) -> splitters
  std::vector<std::vector<Rec>> samp(W);
  rc = sample_shards(ctxs, shards, W, sample_budget(W), samp);
  if (rc) return rc;
  std::vector<Rec> split;
  pick_splitters(samp, W, split);
  // 3. every shard cut at the splitters: bound[i][k] = first record of shard i that is >= splitter k (device binary search;
  //    keys and positions staged in the shard's scratch: 32 (W - 1) bytes, which the argument check guarantees)
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
  std::vector<size_t> n_out(W, 0);
  std::vector<std::vector<size_t>> land(W, std::vector<size_t>(W, 0));   // land[j][i]: record offset of shard i's piece in owner j
  rc = plan_landing(shards, W, bound, n_out, land, "the shards are sorted locally, nothing was moved");
  if (rc) return rc;
  // 4. the exchange.  Exactly 12 varying bytes over all shards: 12-byte elements (every shard compacts itself into the lower half
  //    of its scratch, every owner pulls its pieces into the upper half and expands them over its records); otherwise 24-byte
  //    records travel into the owner's scratch and are copied over its records.  Then every owner sorts.
  const bool compact = have_plan && census_plan.k <= 12;
  const ibu_key_plan_t* plan = reinterpret_cast<const ibu_key_plan_t*>(&census_plan);
  const size_t wire = compact ? 12 : kRec;                    // bytes per record on the links
  if (compact) {
    rc = on_every_context(W, [&](size_t i) -> int32_t {
      int32_t r = ibu_records_compact(ctxs[i], plan, shards[i].d_records, shards[i].n, shards[i].d_tmp, nullptr);
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
      enable_peer(ctxs[j], ctxs[i]->device);
      hipStream_t ps;                                           // one stream per peer device: the links carry their pieces at the same time
      int32_t r = pull_stream_for(ctxs[j], ctxs[i]->device, i, &ps);   // (everything queued on the context's stream so far has completed: joined above)
      if (r) return r;
      const uint8_t* src = static_cast<const uint8_t*>(compact ? shards[i].d_tmp : shards[i].d_records) + wire * bound[i][j];
      IBU_HIP(hipMemcpyPeerAsync(t + wire * land[j][i], ctxs[j]->device, src, ctxs[i]->device, wire * cnt, ps));
    }
    int32_t r = join_pull_streams(ctxs[j]);
    if (r) return r;
    IBU_HIP(hipStreamSynchronize(ctxs[j]->stream));
    return IBU_OK;
  });
  if (rc) return rc;                                          // (joined: nobody overwrites what a peer is still reading)
  rc = on_every_context(W, [&](size_t j) -> int32_t {
    int32_t r = IBU_OK;
    if (n_out[j]) {
      uint8_t* t = static_cast<uint8_t*>(shards[j].d_tmp);
      r = compact ? ibu_records_expand(ctxs[j], plan, t + 12 * shards[j].capacity, n_out[j], shards[j].d_records, nullptr)
                  : ibu_device_copy(ctxs[j], shards[j].d_records, t, kRec * n_out[j], nullptr);
    }
    if (!r) r = ibu_sort_records(ctxs[j], shards[j].d_records, shards[j].d_tmp, n_out[j], nullptr);
    if (!r) r = ibu_ctx_synchronize(ctxs[j], nullptr);
    return r;
  });
  if (rc) return rc;
  if (trace_sort()) fprintf(stderr, "ibu sort: contexts=%zu exchange=%zu bytes per record (sort first)\n", W, wire);
  for (size_t j = 0; j < W; ++j) shards[j].n = n_out[j];
  return IBU_OK;
}
}  // namespace

extern "C" int32_t ibu_sort_records_contexts(ibu_ctx_t* const* ctxs, size_t n_ctxs, ibu_sort_shard_t* shards) {
  if (!ctxs || !shards || n_ctxs == 0) return err_arg("NULL argument or no context");
  if (n_ctxs > 1024) return err_arg("more than 1024 contexts");
  const size_t W = n_ctxs;
  size_t total = 0;
  bool aligned = true;
  for (size_t i = 0; i < W; ++i) {
    if (!ctxs[i]) return err_arg("a context is NULL");
    for (size_t j = 0; j < i; ++j)
      if (ctxs[j] == ctxs[i]) return err_arg("the same context twice (a context serves one host thread; create two on one device instead)");
    const ibu_sort_shard_t& s = shards[i];
    if (s.n > s.capacity) return err_arg("a shard holds more records than its capacity");
    if (s.capacity && (!s.d_records || !s.d_tmp)) return err_arg("a shard's d_records / d_tmp is NULL");
    if ((reinterpret_cast<uintptr_t>(s.d_records) | reinterpret_cast<uintptr_t>(s.d_tmp)) & 7u) return err_arg("d_records / d_tmp must be 8-byte aligned");
    // the splitters (24 bytes each) and their positions (8 bytes each) are staged in d_tmp: 32 (n_ctxs - 1) bytes of it
    if (W > 1 && (s.capacity < W + 1 || kRec * s.capacity < 32 * (W - 1)))
      return err_arg("a shard's capacity must be at least the number of shards + 1, and 24 x capacity at least 32 x (shards - 1) bytes (the splitters and their positions are staged in d_tmp)");
    total += s.n;
    aligned = aligned && sort_elems_supported(ctxs[i]->cfg, s.d_records, s.d_tmp, s.capacity);
  }
  if (W == 1) {
    const int32_t r = ibu_sort_records(ctxs[0], shards[0].d_records, shards[0].d_tmp, shards[0].n, nullptr);
    return r ? r : ibu_ctx_synchronize(ctxs[0], nullptr);
  }
  try {
    // one plan for everybody: the census words of every shard combined.  First from SAMPLES (three ranges of every shard: microseconds):
    // if that plan qualifies for the partition-first form, the partition pass takes the exact census on its way (24 B/record less to
    // read than a census pass of its own) and the guess is checked afterwards; a miss costs the partition pass it wasted.
    CompactPlan plan;
    memset(&plan, 0, sizeof plan);
    bool have_plan = false;
    std::vector<std::array<uint64_t, 8>> words(W);
    auto combine = [&](CompactPlan* out) {
      uint64_t o[3] = {0, 0, 0}, a[3] = {~0ull, ~0ull, ~0ull};
      for (size_t i = 0; i < W; ++i)
        if (shards[i].n)
          for (int f = 0; f < 3; ++f) { o[f] |= words[i][f]; a[f] &= words[i][3 + f]; }
      compact_plan_init(o, a, out);
    };
    // PARTITION FIRST while an owner's share is many of the 256 fine ranges (the owners are cut at range boundaries: loads are
    // quantised to total / 256 — 3 % of a share with 8 shards, 12 % with 32; beyond that the sort-first form cuts finer), elements
    // if the keys allow it, else records.  A cut that does not fit a shard's capacity falls back to the sort-first form as
    // well: nothing has moved at that point (the partitioned copies sit in the scratch arrays).
    if (ctxs[0]->cfg.sort_compact != 0 && total > 0 && W <= kPartitionFirstMaxShards) {
      int32_t rc = IBU_OK;
      bool tried_elements = false;
      std::vector<std::vector<Rec>> samp(W);
      std::vector<Rec> split;                                   // the 255 splitters of the 256 fine ranges; empty: the form samples for itself
      if (aligned) {
        bool all_exact = true;
        std::vector<char> was_exact(W, 1);
        rc = on_every_context(W, [&](size_t i) -> int32_t {
          ibu_ctx_t* c = ctxs[i];
          if (!shards[i].n) return IBU_OK;
          IBU_HIP(hipSetDevice(c->device));
          int32_t r = ensure_sort_scratch(c, 4096);
          if (r) return r;
          bool ex = true;
          uint64_t* d_c = static_cast<uint64_t*>(c->d_sort_scratch);
          IBU_HIP(launch_records_census_sample(c->cfg, shards[i].d_records, shards[i].n, d_c, &ex, c->stream));
          IBU_HIP(hipMemcpyAsync(words[i].data(), d_c, 64, hipMemcpyDeviceToHost, c->stream));
          was_exact[i] = ex ? 1 : 0;
          return sample_one(c, shards[i], total, sample_budget(W), samp[i]);   // (synchronises: the census words have arrived too) — one round of threads for both
        });
        if (rc) return rc;
        for (size_t i = 0; i < W; ++i) all_exact = all_exact && was_exact[i];
        combine(&plan);
        pick_splitters(samp, 256, split);                       // sampled once: a plan miss re-runs the partition pass only
        if (plan.k <= 11) {
          bool covered = true;
          tried_elements = true;
          rc = sort_partition_first(ctxs, W, shards, plan, total, !all_exact, &words, &covered, split);
          if (rc == IBU_OK && !covered) {
            combine(&plan);                                     // `words` now holds every shard's exact census
            tried_elements = plan.k <= 11;
            if (tried_elements) rc = sort_partition_first(ctxs, W, shards, plan, total, false, &words, &covered, split);
          }
        }
      }
      if (!tried_elements) rc = sort_partition_first_records(ctxs, W, shards, total, split);   // more than 11 varying key bytes, or buffers the element kernels cannot take
      if (rc != kLandingFailed) return rc;
      if (trace_sort()) fprintf(stderr, "ibu sort: contexts=%zu the range cut does not fit a shard's capacity: falling back to the sort-first form\n", W);
      return sort_sort_first(ctxs, W, shards, plan, false);
    }
    // SORT FIRST: told to (sort_compact = 0 on ctxs[0]), more than 32 shards, or nothing to sort
    if (ctxs[0]->cfg.sort_compact != 0 && total > 0) {
      int32_t rc = on_every_context(W, [&](size_t i) -> int32_t { return ibu_records_census(ctxs[i], shards[i].d_records, shards[i].n, words[i].data(), nullptr); });
      if (rc) return rc;
      combine(&plan);
      have_plan = true;
    }
    return sort_sort_first(ctxs, W, shards, plan, have_plan);
  } catch (...) {
    return caught_io("ibu_sort_records_contexts");
  }
}
