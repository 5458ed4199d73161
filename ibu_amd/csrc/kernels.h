// kernels.h — launch interface between the C-ABI layer (device.cpp) and the k_*.hip kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace ibu {

struct LaunchCfg {
  int cus = 256;           // hipDeviceProp_t::multiProcessorCount
  int blocks_per_cu = 7;   // persistent grid = cus * blocks_per_cu workgroups of 256 threads (7 beats 8: profiles/r01_c)
  int sort_variant = 0;    // tile shape / write-out mode of the radix passes (sort.hip kSweep; A/B knob)
  int sort_guess = 1;      // compact-key sorts compress on a SAMPLED guess of the varying bytes while the exact census runs in the same pass: 0 = never, 1 = from 2^17 records (the default), k = from k records
  int sort_compact = 1;    // compact-key passes (12-byte elements when at most 12 key bytes vary): 0 = never, k = tile shape kCompact[k-1]
  int sort_hybrid = 1;     // wide keys (more than 16 varying bytes): LSD over the top P varying bytes only, then ONE finishing pass that completes every run of equal prefix in LDS (sort.hip, ibu_k_sort_finish): 0 = never, 1 = when at least three passes are saved, 2 = whenever one is (tests)
  int sort_idx64 = 0;      // test knob: 1 = the sort indexes with 64 bits at any size (the kernels inputs of 2^32 records and more take)
  int alloc_probe_tries = 0; // placement probing inside the library, for resident arrays the library allocates (ibu_device_alloc, ibu_load_to_device's destination, the sort scratch): 0 = auto (>= 1 GiB and at least three candidates fit: up to four), 1 = plain hipMalloc, k = allocations of at least 256 MiB draw k candidates and keep the fastest (device.cpp: ibu_device_alloc_probed)
  int trace_rows = 0;      // tests: one stderr line per launch saying how many rows took the tiled / the tail kernel (kcommon.hpp: split_rows)
  uint32_t base_order = 0; // bit order of the 2-bit codec: 0 = base i at bits [2i,2i+1] (default), 1 = first base most significant
};

// Persistent grids must be exactly resident: a workgroup that has to wait for a slot runs its
// whole share alone at the end (measured: decode 5.07 -> 5.6 TB/s once fixed).  Ask the runtime
// how many 256-thread blocks of this kernel fit on a CU (registers, LDS), cap by
// cfg.blocks_per_cu, remember the answer per kernel instantiation.
template <int BLOCK, class K>
static inline int resident_blocks(const LaunchCfg& cfg, K kernel, size_t dyn_lds, std::atomic<int>* cache) {
  int cached = cache->load(std::memory_order_relaxed);  // contexts of several host threads may race here: same value either way
  if (cached <= 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, BLOCK, dyn_lds) != hipSuccess || nb <= 0)
      nb = 1024 / BLOCK;
    cache->store(nb, std::memory_order_relaxed);
    cached = nb;
  }
  int cap = cfg.blocks_per_cu * 256 / BLOCK;  // blocks_per_cu is stated in 256-thread units (4 waves)
  if (cap < 1) cap = 1;
  return cached < cap ? cached : cap;
}

// All launchers are asynchronous on `st`, allocate nothing and never synchronise.
hipError_t launch_decode(const LaunchCfg&, const void* recs, size_t n, uint32_t bc_len, uint32_t umi_len,
                         uint8_t* bc, uint8_t* umi, uint64_t* idx, hipStream_t st);
hipError_t launch_encode(const LaunchCfg&, const uint8_t* bc, const uint8_t* umi, const uint64_t* idx,
                         uint64_t first_index, size_t n, uint32_t bc_len, uint32_t umi_len, void* recs,
                         uint64_t* status, hipStream_t st);
hipError_t launch_deserialize(const LaunchCfg&, const void* recs, size_t n, uint64_t* bc, uint64_t* umi,
                              uint64_t* idx, hipStream_t st);
hipError_t launch_serialize(const LaunchCfg&, const uint64_t* bc, const uint64_t* umi, const uint64_t* idx,
                            size_t n, void* recs, hipStream_t st);
hipError_t launch_unpack(const LaunchCfg&, const uint64_t* codes, size_t n, uint32_t len, uint8_t* out,
                         hipStream_t st);
hipError_t launch_pack(const LaunchCfg&, const uint8_t* in, size_t n, uint32_t len, uint64_t* codes,
                       uint64_t* status, hipStream_t st);
static constexpr int kReduceSlots = 64;   // acc: kReduceSlots x 8 u64 (slot 0 after launch_reduce_fold: count, 3 sums, 3 XORs)
static constexpr size_t kReduceAccBytes = (size_t)kReduceSlots * 8 * sizeof(uint64_t);
hipError_t launch_reduce(const LaunchCfg&, const void* recs, size_t n, uint64_t* acc, hipStream_t st);
hipError_t launch_reduce_fold(uint64_t* acc, hipStream_t st);
hipError_t launch_generate(const LaunchCfg&, uint64_t seed, uint64_t first, size_t n, uint32_t bc_len,
                           uint32_t umi_len, void* recs, hipStream_t st);
hipError_t launch_copy(const LaunchCfg&, const void* src, void* dst, size_t bytes, hipStream_t st);
hipError_t launch_sorted_check(const LaunchCfg&, const void* recs, size_t n, uint32_t* flag, hipStream_t st);
hipError_t launch_mismatch(const LaunchCfg&, const void* a, const void* b, size_t nwords, uint64_t* first, hipStream_t st);
hipError_t launch_fill2(uint64_t* p, uint64_t v0, uint64_t v1, hipStream_t st);

// sort.hip
bool trace_sort();   // IBU_TRACE_SORT set to anything but "" / "0" (read once): one stderr line per sort / probed allocation saying what was chosen
hipError_t launch_sort_records(const LaunchCfg&, void* recs, void* tmp, size_t n, void* scratch,
                               size_t scratch_bytes, hipStream_t st, const uint64_t* known_words = nullptr /*u64[6]: OR x 3, AND x 3 of a superset: no census pass*/,
                               int known_prefix = -1 /*>= 0: the 24-byte path's prefix length, estimated elsewhere (0 = all passes)*/,
                               int* only_estimate = nullptr /*non-null: only estimate that prefix length for n_scale records like these*/,
                               size_t n_scale = 0);
size_t sort_scratch_bytes(const LaunchCfg&, size_t n);
size_t sort_prefix_estimate_tables(const LaunchCfg&, size_t n);   // bytes of `tmp` the only_estimate form of launch_sort_records writes (0: none)
int sort_num_variants();
int sort_num_compact_variants();
hipError_t launch_lower_bound(const void* recs, size_t n, const void* keys, size_t k, uint64_t* pos, hipStream_t st);
// compacted keys (sort.hip): the varying bytes of a set of records as 12-byte elements.  Layout = ibu_key_plan_t (ibu_hip.h).
struct CompactPlan {
  uint32_t csel[4][3];   // compress: element word w = OR over the fields f of perm(f.hi, f.lo, csel[w][f])
  uint32_t xsel[6][2];   // expand: record dword d (= 2 f + half) = base | perm(e.w1, e.w0, xsel[d][0]) | perm(e.w3 or 0, e.w2, xsel[d][1])
  uint32_t k;            // varying bytes (the element bytes above them are zero): 12-byte elements while k <= 12, 16-byte ones (inside the sort) while k <= 16
  uint32_t index_bytes;  // how many of them are index bytes (the least significant element bytes)
  uint64_t base[3];      // each field with its varying bytes cleared (the AND words)
};
void compact_plan_init(const uint64_t or_words[3], const uint64_t and_words[3], CompactPlan* pl);
hipError_t launch_records_census(const LaunchCfg&, const void* recs, size_t n, uint64_t* d_census /*u64[8]*/, hipStream_t st);
hipError_t launch_compact(const LaunchCfg&, const CompactPlan& pl, const void* recs, size_t n, void* elems, hipStream_t st);
hipError_t launch_expand(const LaunchCfg&, const CompactPlan& pl, const void* elems, size_t n, void* recs, hipStream_t st);
// The multi-GPU sort on 12-byte elements (sort.hip, used by multi_sort.cpp): how many prefix passes a sort of n_scale records like
// these n wants (synchronises st); records -> 12-byte elements at `elems`, stamped with their key range among the splitters, -> range order at `out`
// (d_starts: u64[256] in `scratch`, the first element of every range); received elements -> sorted records.
hipError_t launch_estimate_prefix(const LaunchCfg&, const void* recs, size_t n, size_t n_scale, void* tmp, const CompactPlan& pl,
                                  uint32_t* prefix_passes, hipStream_t st);
// What the host needs from a partition pass — the range starts (u64[256]) and, where taken, the census words (u64[8]) — exists as soon
// as the pass's small scan kernel has run, BEFORE its scatter kernel (a third of the pass's time) has: with `early` the launcher copies
// both into pinned host memory right there and records `ready` behind the copies, so that the caller can plan the exchange while the
// scatter still runs.
struct PartitionEarly {
  uint64_t* h_starts;   // pinned, u64[256]
  uint64_t* h_words;    // pinned, u64[8] (written only when the census is taken)
  hipEvent_t ready;
};
hipError_t launch_partition_elems(const LaunchCfg&, const CompactPlan& pl, const void* recs, void* elems, size_t n, const void* d_split,
                                  uint32_t nsplit, void* out, void* scratch, size_t scratch_bytes, const uint64_t** d_starts,
                                  const uint64_t** d_census /*nullable: the exact census words of these records, accumulated on the way*/, hipStream_t st,
                                  const PartitionEarly* early = nullptr);
hipError_t launch_records_census_sample(const LaunchCfg&, const void* recs, size_t n, uint64_t* d_census /*u64[8 x 64]*/, bool* exact, hipStream_t st);
hipError_t launch_partition_records(const LaunchCfg&, const void* recs, size_t n, const void* d_split /*24-byte records*/, uint32_t nsplit, void* out,
                                    void* scratch, size_t scratch_bytes, const uint64_t** d_starts,
                                    const uint64_t** d_census /*nullable: the exact OR / AND words of these records, accumulated on the way*/, hipStream_t st,
                                    const PartitionEarly* early = nullptr);   // the same on 24-byte records (any key)
bool sort_elems_supported(const LaunchCfg&, const void* recs, const void* tmp, size_t capacity);
hipError_t launch_sort_elems(const LaunchCfg&, const CompactPlan& pl, void* recs, void* tmp, size_t n, uint32_t prefix_passes, void* scratch,
                             size_t scratch_bytes, hipStream_t st);
// per-barcode run-length aggregation of sorted records (k_aggregate.hip)
size_t runs_scratch_bytes(size_t n);
hipError_t launch_runs_count(const LaunchCfg&, const void* recs, size_t n, void* scratch, size_t scratch_bytes, bool keep_heads, hipStream_t st);
size_t runs_emit_scratch_bytes(uint64_t n_runs);
hipError_t launch_runs_emit(const LaunchCfg&, const void* recs, size_t n, const void* scratch, bool from_stash, void* run_scratch, uint64_t n_runs,
                            uint64_t n_pairs, uint64_t* barcodes, uint64_t* counts, uint64_t* uniq, hipStream_t st);

// DEFLATE blocks inflated on the device, one wave per block (k_inflate.hip).  The compressed bytes must be readable 2 KiB past the
// last block (kInflatePad); block i's output goes to d_out_base + ooff (signed: a block that begins in front of the window a batch
// keeps lands in the headroom before it).  d_status[i]: 0 good, 1 not a valid deflate stream of these sizes, 2 CRC-32 mismatch;
// *d_first_bad (the caller sets it to 0xFFFFFFFF): the lowest bad block.
struct InflateBlockDesc { uint64_t coff; int64_t ooff; uint32_t clen, isize, crc, reserved; };
constexpr size_t kInflatePad = 2048;
// form 0: by size; 2: the lanes' tables in scratch whatever the size (k_inflate.hip)
size_t inflate_scratch_bytes(const LaunchCfg&, size_t nblocks, int form = 0);   // the lanes' tables (44 KB per workgroup of the grid; 16 bytes for the LDS form)
hipError_t launch_inflate_blocks(const LaunchCfg&, const void* d_comp, const InflateBlockDesc* d_blocks, size_t nblocks, void* d_out_base,
                                 uint32_t* d_status, uint32_t* d_first_bad, void* scratch, size_t scratch_bytes, hipStream_t st, int form = 0,
                                 const uint64_t* d_ready = nullptr /*the launch runs ahead of its input: compressed bytes arrived so far (status 3: never came)*/,
                                 uint64_t ready_total = 0);

}  // namespace ibu
