#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const char* path = argv[1];
  int mode = atoi(argv[2]);
  int T = atoi(argv[3]);
  size_t total = (size_t)atof(argv[4]);
  size_t chunk = 96u << 20;
  uint8_t* src = (uint8_t*)malloc(chunk);
  memset(src, 0x5a, chunk);
  int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
  double t0 = now();
  if (mode == 0) {
    for (size_t off = 0; off < total; off += chunk) { size_t n = std::min(chunk, total - off); size_t d = 0; while (d < n) { ssize_t k = write(fd, src + d, n - d); if (k < 0) return 1; d += k; } }
  } else if (mode == 1) {
    for (size_t off = 0; off < total; off += chunk) {
      size_t n = std::min(chunk, total - off);
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t a = n * t / T, b = n * (t + 1) / T; while (a < b) { ssize_t k = pwrite(fd, src + a, b - a, off + a); if (k < 0) abort(); a += k; } });
      for (auto& x : th) x.join();
    }
  } else {
    if (mode == 3) { if (posix_fallocate(fd, 0, total)) perror("fallocate"); } else if (ftruncate(fd, total)) return 2;
    for (size_t off = 0; off < total; off += chunk) {
      size_t n = std::min(chunk, total - off);
      uint8_t* m = (uint8_t*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
      if (m == MAP_FAILED) { perror("mmap"); return 3; }
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t a = n * t / T, b = n * (t + 1) / T; memcpy(m + a, src + a, b - a); });
      for (auto& x : th) x.join();
      munmap(m, n);
    }
  }
  double t1 = now();
  close(fd);
  printf("mode %d T %d: %.2f GB/s\n", mode, T, total / (t1 - t0) / 1e9);
  unlink(path);
}
