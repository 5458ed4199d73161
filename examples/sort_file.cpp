// sort_file — what the header's sorted flag is for (header.rs:111-113): read an IBU file, sort its records by
// (barcode, umi, index) on the GPU, write them back under a header with the flag set, and print the per-barcode
// summary the reference's BarcodeAnalyzer example computes (parallel.rs:72-98).
//   sort_file IN OUT [--top K]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "ibu.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: sort_file IN OUT [--top K]\n"); return 2; }
  size_t top = 5;
  for (int i = 3; i + 1 < argc; ++i) if (!std::strcmp(argv[i], "--top")) top = std::strtoull(argv[i + 1], nullptr, 10);
  try {
    ibu::device::Context ctx(0);
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
