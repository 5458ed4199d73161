// parallel — the workload of the reference's examples/parallel.rs on the C++ mirror: write N records
// (i % 1e6, 31 i % 1e6, i), then MmapReader::process_parallel(Processor, 0) where Processor sums the three
// fields locally and folds them into a mutex-guarded global in on_batch_complete.  With --device the same
// map is streamed through the pinned ring to the GPU's reduce processor and must give the same three sums.
//   parallel [N=10000000] [--threads T] [--device] [--json] [--dir DIR]
#include <unistd.h>

#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

#include "ibu.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Processor {  // Clone = copy: local counts are per worker, the global is shared
  std::array<uint64_t, 3> local{0, 0, 0};
  struct Global { std::mutex m; std::array<uint64_t, 3> v{0, 0, 0}; };
  std::shared_ptr<Global> global = std::make_shared<Global>();
  void process_record(const ibu::Record& r) { local[0] += r.barcode; local[1] += r.umi; local[2] += r.index; }
  void on_batch_complete() {
    std::lock_guard<std::mutex> g(global->m);
    for (int k = 0; k < 3; ++k) global->v[k] += local[k];
    local = {0, 0, 0};
  }
  std::array<uint64_t, 3> final_counts() const { std::lock_guard<std::mutex> g(global->m); return global->v; }
};

int main(int argc, char** argv) {
  uint64_t n = 10000000;
  size_t threads = 0;
  bool dev = false, json = false;
  std::string dir = ".";
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--device")) dev = true;
    else if (!std::strcmp(argv[i], "--json")) json = true;
    else if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) threads = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--dir") && i + 1 < argc) dir = argv[++i];
    else n = std::strtoull(argv[i], nullptr, 10);
  }
  const std::string path = dir + "/test_parallel_" + std::to_string(getpid()) + ".ibu";
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

    ibu::MmapReader reader(path);
    Processor proc;
    t0 = now();
    reader.process_parallel(proc, threads);
    const double t_proc = now() - t0;
    t0 = now();
    Processor again;
    reader.process_parallel(again, threads);  // second pass: pages are resident
    const double t_proc2 = now() - t0;
    const auto c = proc.final_counts();
    if (again.final_counts() != c) throw std::runtime_error("two passes disagree");

    double t_dev = 0, t_kernel = 0;
    if (dev) {
      ibu::device::Context ctx(0);
      t0 = now();
      auto [red, st] = reader.process_device_reduce(ctx);
      t_dev = now() - t0;
      t_kernel = st.seconds_kernel;
      if (red.count != n || red.sum[0] != c[0] || red.sum[1] != c[1] || red.sum[2] != c[2])
        throw std::runtime_error("device sums differ from process_parallel");
      // ... and the reference's own shape, process_parallel(proc, 0), with a GPU per worker: ONE call over every visible device
      // (ibu_mmap_process_devices: a host thread + context per device, shard i of the static split, partials added on the host)
      auto [total, per_device] = reader.process_devices_reduce();
      if (total.count != n || total.sum[0] != c[0] || total.sum[1] != c[1] || total.sum[2] != c[2] || per_device.empty())
        throw std::runtime_error("multi-device sums differ from process_parallel");
    }
    unlink(path.c_str());
    if (json) {
      std::printf("{\"records\": %llu, \"threads_arg\": %zu, \"write_Mrec_s\": %.2f, \"process_parallel_Mrec_s\": %.2f, "
                  "\"process_parallel_warm_Mrec_s\": %.2f, \"sums\": [%llu, %llu, %llu], \"device\": %s, \"device_e2e_Mrec_s\": %.2f, "
                  "\"device_kernel_s\": %.5f}\n",
                  (unsigned long long)n, threads, n / t_write / 1e6, n / t_proc / 1e6, n / t_proc2 / 1e6, (unsigned long long)c[0],
                  (unsigned long long)c[1], (unsigned long long)c[2], dev ? "true" : "false", dev ? n / t_dev / 1e6 : 0.0, t_kernel);
    } else {
      std::printf("Records: %llu  (file ~%.2f GB)\nWrite: %.2fs (%.2f M records/s)\n", (unsigned long long)n, 24.0 * n / 1e9, t_write, n / t_write / 1e6);
      std::printf("Number of records processed: [%llu, %llu, %llu]\n", (unsigned long long)c[0], (unsigned long long)c[1], (unsigned long long)c[2]);
      std::printf("Processing duration: %.5fs first pass, %.5fs warm (%.1f M records/s)\n", t_proc, t_proc2, n / t_proc2 / 1e6);
      if (dev) std::printf("Device: %.5fs end to end (%.1f M records/s), kernels %.5fs — same sums\n", t_dev, n / t_dev / 1e6, t_kernel);
    }
  } catch (const std::exception& e) {
    unlink(path.c_str());
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
