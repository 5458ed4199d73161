// stream_pull — the reference's BarcodeAnalyzer (a ParallelProcessor: HashMap<barcode, count> per batch, merged in
// on_batch_complete; parallel.rs:72-98) as a CALLER-SIDE processor over the pull-style device stream: the file (plain, gzip, BGZF
// ... — Reader::from_path sniffs it, reader.rs:345-352) is pulled one device-resident batch at a time; each batch is copied out
// of its ring slot, the slot is given back (its refill overlaps what follows), the copy is sorted on the device and run-length
// counted per barcode (ibu_sort_records + ibu_barcode_counts), and the per-batch tables are merged on the host.
//   stream_pull IN [--top K] [--slot-records N] [--json]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "ibu.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: stream_pull IN [--top K] [--slot-records N] [--json]\n"); return 2; }
  size_t top = 5, slot_records = 0;
  bool json = false;
  for (int i = 2; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--top") && i + 1 < argc) top = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--slot-records") && i + 1 < argc) slot_records = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--json")) json = true;
  }
  try {
    using namespace ibu;
    device::Context ctx(0);
    Reader rd = Reader::from_path(argv[1]);
    RingConfig ring{};
    ring.slots = 3;
    ring.slot_records = (uint32_t)slot_records;                    // 0: the library's default (96 MiB slots)
    const size_t cap = slot_records ? (slot_records + 127) / 128 * 128 : 4u * (size_t)IBU_BATCH_SIZE;
    device::DeviceBuffer work(ctx, cap * RECORD_SIZE), tmp(ctx, cap * RECORD_SIZE);
    std::unordered_map<uint64_t, std::pair<uint64_t, uint64_t>> table;   // barcode -> (records, distinct UMIs seen per batch, summed)
    uint64_t records = 0, batches = 0;
    const double t0 = now();
    {
      DeviceStream s = rd.device_stream(ctx, &ring);
      while (auto b = s.next()) {
        ctx.copy(work.ptr(), b->d_records, b->n * RECORD_SIZE);      // queued on the context's stream, behind the batch's H2D
        const size_t n = b->n;
        b->release();                                                // the slot refills as soon as the copy has run
        ctx.sort_records(work.ptr(), tmp.ptr(), n);
        for (auto& [barcode, count, umis] : ctx.barcode_counts(work.ptr(), n)) {
          auto& e = table[barcode];
          e.first += count;
          e.second += umis;
        }
        records += n;
        ++batches;
      }
    }
    const double t1 = now();
    std::vector<std::pair<uint64_t, std::pair<uint64_t, uint64_t>>> rows(table.begin(), table.end());
    std::sort(rows.begin(), rows.end(), [](auto& a, auto& b) { return a.second.first != b.second.first ? a.second.first > b.second.first : a.first < b.first; });
    if (json) {
      std::printf("{\"records\": %llu, \"batches\": %llu, \"barcodes\": %zu, \"seconds\": %.4f, \"top\": [", (unsigned long long)records,
                  (unsigned long long)batches, rows.size(), t1 - t0);
      for (size_t i = 0; i < rows.size() && i < top; ++i)
        std::printf("%s[%llu, %llu]", i ? ", " : "", (unsigned long long)rows[i].first, (unsigned long long)rows[i].second.first);
      std::printf("]}\n");
    } else {
      std::printf("%llu records in %llu batches, %zu barcodes, %.3f s\n", (unsigned long long)records, (unsigned long long)batches, rows.size(), t1 - t0);
      for (size_t i = 0; i < rows.size() && i < top; ++i)
        std::printf("  barcode %#llx: %llu records\n", (unsigned long long)rows[i].first, (unsigned long long)rows[i].second.first);
    }
  } catch (const ibu::IbuError& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
