// random — the reference's examples/random.rs restated on the C++ mirror: write `--records` million records with
// barcode in [0, --barcodes), index in [0, --max-index) and a full-range u64 UMI (NOT masked to umi_len — the
// reference does the same) under a validated (--bc-len, --umi-len) header.  The reference draws from
// rand::SmallRng, whose stream cannot be reproduced without that crate; this one uses splitmix64 (--seed, default
// from the clock), so files differ from the reference's but the CLI, the value ranges and the file format do not.
//   random PATH [--records 1.0] [--barcodes 1000] [--max-index 10000] [--bc-len 16] [--umi-len 12] [--seed S]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "ibu.hpp"

static uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint64_t below(uint64_t& s, uint64_t bound) {  // unbiased enough for test data: 128-bit multiply-shift
  return bound ? (uint64_t)(((unsigned __int128)splitmix64(s) * bound) >> 64) : 0;
}

int main(int argc, char** argv) {
  std::string path;
  double records = 1.0;
  uint64_t barcodes = 1000, max_index = 10000, seed = 0;
  uint32_t bc_len = 16, umi_len = 12;
  bool have_seed = false;
  for (int i = 1; i < argc; ++i) {
    auto val = [&](const char* flag) -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
      return argv[++i];
    };
    if (!std::strcmp(argv[i], "--records")) records = std::atof(val("--records"));
    else if (!std::strcmp(argv[i], "--barcodes")) barcodes = std::strtoull(val("--barcodes"), nullptr, 10);
    else if (!std::strcmp(argv[i], "--max-index")) max_index = std::strtoull(val("--max-index"), nullptr, 10);
    else if (!std::strcmp(argv[i], "--bc-len")) bc_len = (uint32_t)std::atoi(val("--bc-len"));
    else if (!std::strcmp(argv[i], "--umi-len")) umi_len = (uint32_t)std::atoi(val("--umi-len"));
    else if (!std::strcmp(argv[i], "--seed")) { seed = std::strtoull(val("--seed"), nullptr, 0); have_seed = true; }
    else path = argv[i];
  }
  if (path.empty()) { std::fprintf(stderr, "usage: random PATH [--records M] [--barcodes N] [--max-index N] [--bc-len L] [--umi-len L] [--seed S]\n"); return 2; }
  if (!have_seed) seed = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  try {
    ibu::Header header(bc_len, umi_len);
    header.validate();
    ibu::Writer writer = ibu::Writer::from_path(path, header);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t n = (size_t)(records * 1e6);
    for (size_t i = 0; i < n; ++i) {
      const uint64_t barcode = below(seed, barcodes), index = below(seed, max_index), umi = splitmix64(seed);
      writer.write_record(ibu::Record(barcode, umi, index));
    }
    writer.finish();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "Finished generating %zu records\nElapsed time: %.3fs\nBandwidth: %.2f GB/s\n", n, el,
                 (ibu::HEADER_SIZE + n * ibu::RECORD_SIZE) / el / 1e9);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
