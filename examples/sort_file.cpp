// sort_file — what the header's sorted flag is for (header.rs:111-113): read an IBU file, sort its records by
// (barcode, umi, index) on the GPU, write them back under a header with the flag set, and print the per-barcode
// summary the reference's BarcodeAnalyzer example computes (parallel.rs:72-98).
//   sort_file IN OUT [--top K] [--contexts N]
// --contexts N: the same through N contexts — context i on GPU i % (GPUs of the box), shard i of the static split
// (mmap.rs:297-307) — and ONE call of the multi-GPU sort (ibu_sort_records_contexts): shard i comes back as the i-th range of
// the global order and the ranges are written one after the other (the Writer::ingest pattern, writer.rs:477-482).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "ibu.hpp"

static double now();
// the multi-context form: shards of the mapped file on N contexts, one global sort, ranges written in order
static int sort_over_contexts(const char* in, const char* out_path, size_t nctx) {
  using namespace ibu;
  MmapReader m{std::string(in)};
  const size_t n = m.len();
  const int ndev = std::max(1, device::device_count());
  std::vector<std::unique_ptr<device::Context>> ctxs;
  std::vector<std::unique_ptr<device::DeviceBuffer>> bufs;
  std::vector<device::Context*> raw;
  std::vector<ibu_sort_shard_t> shards;
  const double t0 = now();
  for (size_t i = 0; i < nctx; ++i) {
    ctxs.emplace_back(new device::Context((int)(i % (size_t)ndev)));
    raw.push_back(ctxs.back().get());
    const auto [s, e] = shard_range(n, nctx, i);
    const size_t cap = (n / nctx) * 5 / 4 + nctx + 64;          // headroom for an uneven split
    bufs.emplace_back(new device::DeviceBuffer(*ctxs.back(), cap * RECORD_SIZE));
    bufs.emplace_back(new device::DeviceBuffer(*ctxs.back(), cap * RECORD_SIZE));
    if (e > s) ctxs.back()->upload(bufs[2 * i]->ptr(), m.slice(s, e).ptr, (e - s) * RECORD_SIZE);
    shards.push_back({bufs[2 * i]->ptr(), bufs[2 * i + 1]->ptr(), e - s, cap});
  }
  const double t1 = now();
  device::Context::sort_records_contexts(raw, shards);
  const double t2 = now();
  Header h = m.header();
  h.set_sorted();
  Writer w = Writer::from_path(out_path, h);
  for (size_t i = 0; i < nctx; ++i) w.write_batch_device(*ctxs[i], shards[i].d_records, shards[i].n);
  w.finish();
  const double t3 = now();
  std::printf("%zu records over %zu contexts on %d GPU(s): load %.3fs, sort %.3fs, write %.3fs; shard sizes", n, nctx, ndev, t1 - t0, t2 - t1, t3 - t2);
  for (auto& sh : shards) std::printf(" %zu", sh.n);
  std::printf("\n");
  bufs.clear();                                               // before the contexts they belong to
  return 0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: sort_file IN OUT [--top K] [--contexts N]\n"); return 2; }
  size_t top = 5, nctx = 0;
  for (int i = 3; i + 1 < argc; ++i) {
    if (!std::strcmp(argv[i], "--top")) top = std::strtoull(argv[i + 1], nullptr, 10);
    if (!std::strcmp(argv[i], "--contexts")) nctx = std::strtoull(argv[i + 1], nullptr, 10);
  }
  try {
    if (nctx > 1) return sort_over_contexts(argv[1], argv[2], nctx);
    ibu::device::Context ctx(0);
    ctx.set_option("alloc_probe_tries", 4);                     // resident arrays of 256 MiB and more choose their placement (ibu_hip.h)
    const double t0 = now();
    auto [h, d_recs, n] = ctx.load_to_device(argv[1]);          // load_to_vec, device form
    const double t1 = now();
    if (!h.sorted()) {                                          // a file that says it is sorted is taken at its word...
      ibu::device::DeviceBuffer tmp(ctx, n * ibu::RECORD_SIZE);
      ctx.sort_records(d_recs, tmp.ptr(), n);
    }
    if (n && !ctx.is_sorted(d_recs, n)) throw std::runtime_error("input claims to be sorted but is not");  // ...and checked
    const double t2 = now();
    auto per_barcode = ctx.barcode_counts(d_recs, n);           // (barcode, records, distinct UMIs), ascending barcode
    const double t3 = now();
    ibu::Header out = h;
    out.set_sorted();
    {
      ibu::Writer w = ibu::Writer::from_path(argv[2], out);
      w.write_batch_device(ctx, d_recs, n);
      w.finish();
    }
    const double t4 = now();
    ctx.free(d_recs);
    std::printf("%zu records, %zu barcodes: load %.3fs, sort %.3fs, aggregate %.3fs, write %.3fs\n", n, per_barcode.size(), t1 - t0,
                t2 - t1, t3 - t2, t4 - t3);
    std::partial_sort(per_barcode.begin(), per_barcode.begin() + std::min(top, per_barcode.size()), per_barcode.end(),
                      [](const auto& a, const auto& b) { return std::get<1>(a) > std::get<1>(b); });
    for (size_t k = 0; k < std::min(top, per_barcode.size()); ++k)
      std::printf("  barcode 0x%llx: %llu records, %llu distinct UMIs\n", (unsigned long long)std::get<0>(per_barcode[k]),
                  (unsigned long long)std::get<1>(per_barcode[k]), (unsigned long long)std::get<2>(per_barcode[k]));
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
