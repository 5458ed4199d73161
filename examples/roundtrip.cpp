// roundtrip — the workload of the reference's examples/roundtrip.rs (BASELINE configs[0]) on the C++ mirror:
// write N records (i % 1e6, 31 i % 1e6, i) one write_record at a time under a sorted (16,12) header, stream
// them back through Reader with an XOR checksum, then load_to_vec.  With --device the same file is also
// loaded with load_to_device and reduced / decoded on the GPU, and the results are compared.
//   roundtrip [N=1000000] [--device] [--json] [--dir DIR]
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "ibu.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  uint64_t n = 1000000;
  bool dev = false, json = false;
  std::string dir = ".";
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--device")) dev = true;
    else if (!std::strcmp(argv[i], "--json")) json = true;
    else if (!std::strcmp(argv[i], "--dir") && i + 1 < argc) dir = argv[++i];
    else n = std::strtoull(argv[i], nullptr, 10);
  }
  const std::string path = dir + "/test_roundtrip_" + std::to_string(getpid()) + ".ibu";
  try {
    ibu::Header header(16, 12);
    header.set_sorted();
    double t0 = now();
    {
      ibu::Writer w = ibu::Writer::from_path(path, header);
      for (uint64_t i = 0; i < n; ++i) w.write_record(ibu::Record(i % 1000000, (i * 31) % 1000000, i));
      w.finish();
    }
    const double t_write = now() - t0;

    t0 = now();
    uint64_t read = 0, checksum = 0, sums[3] = {0, 0, 0};
    {
      ibu::Reader r = ibu::Reader::from_path(path);
      const ibu::Header h = r.header();
      if (h.bc_len != 16 || h.umi_len != 12 || !h.sorted()) throw std::runtime_error("header mismatch");
      while (auto rec = r.next()) {
        ++read;
        checksum ^= rec->barcode ^ rec->umi ^ rec->index;
        sums[0] += rec->barcode; sums[1] += rec->umi; sums[2] += rec->index;
      }
    }
    const double t_read = now() - t0;
    if (read != n) throw std::runtime_error("record count mismatch");

    t0 = now();
    auto [h2, recs] = ibu::load_to_vec(path);
    const double t_load = now() - t0;
    if (recs.size() != n) throw std::runtime_error("load_to_vec count mismatch");

    double t_dev = 0, t_dev_kernel = 0;
    bool dev_ok = true;
    if (dev) {
      ibu::device::Context ctx(0);
      t0 = now();
      ibu::StreamStats st{};
      auto [hd, dptr, dn] = ctx.load_to_device(path, nullptr, &st);
      auto red = ctx.reduce(dptr, dn);
      t_dev = now() - t0;
      dev_ok = dn == n && red.count == n && red.sum[0] == sums[0] && red.sum[1] == sums[1] && red.sum[2] == sums[2] &&
               (red.xor_[0] ^ red.xor_[1] ^ red.xor_[2]) == checksum && hd == header;
      // decode on device, spot-check the first and last rows against the 2-bit table (record.rs:19-27)
      ibu::device::DeviceBuffer bc(ctx, n * 16), umi(ctx, n * 12), idx(ctx, n * 8);
      double k0 = now();
      ctx.decode_ascii(dptr, n, hd, bc.as<uint8_t>(), umi.as<uint8_t>(), idx.as<uint64_t>());
      ctx.synchronize();
      t_dev_kernel = now() - k0;
      auto first = bc.download<uint8_t>(16);
      for (int b = 0; b < 16; ++b) dev_ok = dev_ok && first[b] == "ACGT"[(recs[0].barcode >> (2 * b)) & 3];
      ctx.free(dptr);
      if (!dev_ok) throw std::runtime_error("device results differ from the host stream");
    }
    unlink(path.c_str());
    const double mb = 24.0 * n / 1e9;
    if (json) {
      std::printf("{\"records\": %llu, \"write_Mrec_s\": %.2f, \"read_Mrec_s\": %.2f, \"load_to_vec_Mrec_s\": %.2f, "
                  "\"write_GBps\": %.3f, \"read_GBps\": %.3f, \"load_GBps\": %.3f, \"checksum\": \"0x%016llX\", "
                  "\"sums\": [%llu, %llu, %llu], \"device\": %s, \"device_load_reduce_s\": %.4f, \"device_decode_s\": %.5f}\n",
                  (unsigned long long)n, n / t_write / 1e6, n / t_read / 1e6, n / t_load / 1e6, mb / t_write, mb / t_read, mb / t_load,
                  (unsigned long long)checksum, (unsigned long long)sums[0], (unsigned long long)sums[1], (unsigned long long)sums[2],
                  dev ? "true" : "false", t_dev, t_dev_kernel);
    } else {
      std::printf("IBU Roundtrip Test\n==================\nRecords: %llu\nFile size: ~%.2f GB\n\n", (unsigned long long)n, mb);
      std::printf("Write:       %.2fs  %.2f M records/s  %.2f GB/s\n", t_write, n / t_write / 1e6, mb / t_write);
      std::printf("Read:        %.2fs  %.2f M records/s  %.2f GB/s\n", t_read, n / t_read / 1e6, mb / t_read);
      std::printf("Direct load: %.2fs  %.2f M records/s  %.2f GB/s\n", t_load, n / t_load / 1e6, mb / t_load);
      std::printf("Checksum: 0x%016llX  records read: %llu\n", (unsigned long long)checksum, (unsigned long long)read);
      if (dev) std::printf("Device: load_to_device + reduce %.3fs (matches host stream), decode kernel %.2f ms\n", t_dev, t_dev_kernel * 1e3);
    }
  } catch (const std::exception& e) {
    unlink(path.c_str());
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
