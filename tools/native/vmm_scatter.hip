// vmm_scatter.hip — measurement only (not part of libibu_hip.so).
//
// tools/vram_map.py shows that the write rate of a streaming kernel is a property of the PHYSICAL pages it writes, and that
// the rare chunks the driver has to piece together from leftovers are the fastest to write.  This program asks whether
// that can be produced on purpose with HIP's virtual-memory API: the same 4 GiB virtual range is backed by
//   A  hipMalloc
//   B  handles of the minimum granularity, created in order, mapped in order
//   C  the same handles mapped in a shuffled order
//   D  a random subset of 8x as many handles (the others stay allocated: the pages are spread over 8x the span)
//   E  handles of 1 GiB
// and a write-only fill, a copy from a hipMalloc'ed source and a read-only sum are timed on each (HIP events, best of 5).
//   hipcc --offload-arch=gfx950 -O3 tools/native/vmm_scatter.hip -o /tmp/vmm_scatter && /tmp/vmm_scatter [GiB] [handle KiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <random>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

__global__ void __launch_bounds__(256, 7) k_fill(u32x4* __restrict__ dst, uint64_t chunks) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < chunks; s += stride) {
    u32x4 v = {(unsigned)s, 1u, 2u, 3u};
    __builtin_nontemporal_store(v, dst + s);
  }
}
__global__ void __launch_bounds__(256, 7) k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t chunks) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < chunks; s += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + s), dst + s);
}
__global__ void __launch_bounds__(256, 7) k_sum(const u32x4* __restrict__ src, uint64_t chunks, unsigned* out) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  unsigned acc = 0;
  for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < chunks; s += stride) {
    u32x4 v = __builtin_nontemporal_load(src + s);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *out = acc;   // practically never: keeps the loads alive
}

static hipEvent_t e0, e1;
template <class F>
static double best_ms(F&& launch) {
  double best = 1e30;
  for (int r = 0; r < 6; ++r) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms < best) best = ms;
  }
  return best;
}

static void measure(const char* tag, void* p, const void* src, size_t bytes, unsigned* d_out) {
  const uint64_t chunks = bytes / 16;
  const int grid = 256 * 7;
  double f = best_ms([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, (u32x4*)p, chunks); });
  double c = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)p, chunks); });
  double s = best_ms([&] { hipLaunchKernelGGL(k_sum, dim3(grid), dim3(256), 0, 0, (const u32x4*)p, chunks, d_out); });
  CK(hipGetLastError());
  printf("{\"case\": \"%s\", \"GiB\": %.1f, \"fill_GBps\": %.0f, \"copy_GBps\": %.0f, \"sum_GBps\": %.0f}\n", tag, bytes / 1073741824.0,
         bytes / f / 1e6, 2.0 * bytes / c / 1e6, bytes / s / 1e6);
  fflush(stdout);
}

struct Mapped {
  void* va = nullptr;
  size_t bytes = 0;
};
static Mapped map_handles(const std::vector<hipMemGenericAllocationHandle_t>& hs, size_t hbytes) {
  Mapped m;
  m.bytes = hs.size() * hbytes;
  CK(hipMemAddressReserve(&m.va, m.bytes, 0, nullptr, 0));
  for (size_t i = 0; i < hs.size(); ++i) CK(hipMemMap((char*)m.va + i * hbytes, hbytes, 0, hs[i], 0));
  hipMemAccessDesc d{};
  d.location.type = hipMemLocationTypeDevice;
  d.location.id = 0;
  d.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(m.va, m.bytes, &d, 1));
  return m;
}
static void unmap(Mapped& m) {
  CK(hipDeviceSynchronize());
  CK(hipMemUnmap(m.va, m.bytes));
  CK(hipMemAddressFree(m.va, m.bytes));
  m.va = nullptr;
}

// mode "map": per handle size, G groups of 1 GiB each (all held until the end), fill / sum per group; then the reuse test:
// free a hipMalloc'ed block and back a new range with handles (do they inherit its speed?), and the other way round.
static double fill_rate(void* p, size_t bytes) {
  const uint64_t chunks = bytes / 16;
  double f = best_ms([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 7), dim3(256), 0, 0, (u32x4*)p, chunks); });
  return bytes / f / 1e6;
}
static double sum_rate(void* p, size_t bytes, unsigned* d_out) {
  const uint64_t chunks = bytes / 16;
  double f = best_ms([&] { hipLaunchKernelGGL(k_sum, dim3(256 * 7), dim3(256), 0, 0, (const u32x4*)p, chunks, d_out); });
  return bytes / f / 1e6;
}
static int run_map(int groups, hipMemAllocationProp& prop, unsigned* d_out) {
  const size_t gbytes = 1ull << 30;
  std::vector<std::pair<Mapped, std::vector<hipMemGenericAllocationHandle_t>>> held;
  const size_t sizes[] = {2ull << 20, 32ull << 20, 256ull << 20, 1ull << 30};
  for (size_t hb : sizes) {
    printf("{\"handle_bytes\": %zu, \"fill_sum_GBps_per_GiB_group\": [", hb);
    for (int g = 0; g < groups; ++g) {
      std::vector<hipMemGenericAllocationHandle_t> hs(gbytes / hb);
      for (auto& h : hs) CK(hipMemCreate(&h, hb, &prop, 0));
      Mapped m = map_handles(hs, hb);
      printf("%s[%.0f, %.0f]", g ? ", " : "", fill_rate(m.va, gbytes), sum_rate(m.va, gbytes, d_out));
      fflush(stdout);
      held.emplace_back(m, std::move(hs));
    }
    printf("]}\n");
  }
  printf("{\"hipMalloc_1GiB\": [");
  std::vector<void*> plain;
  for (int g = 0; g < groups; ++g) {
    void* p;
    CK(hipMalloc(&p, gbytes));
    plain.push_back(p);
    printf("%s[%.0f, %.0f]", g ? ", " : "", fill_rate(p, gbytes), sum_rate(p, gbytes, d_out));
  }
  printf("]}\n");
  // reuse test at 8 GiB
  const size_t big = 8ull << 30, hb = 32ull << 20;
  void* x;
  CK(hipMalloc(&x, big));
  const double x_fill = fill_rate(x, big);
  std::vector<hipMemGenericAllocationHandle_t> hs(big / hb);
  for (auto& h : hs) CK(hipMemCreate(&h, hb, &prop, 0));
  Mapped m = map_handles(hs, hb);
  const double b_fill = fill_rate(m.va, big);
  CK(hipDeviceSynchronize());
  CK(hipFree(x));                                   // X's pages go back to the driver ...
  std::vector<hipMemGenericAllocationHandle_t> hs2(big / hb);
  for (auto& h : hs2) CK(hipMemCreate(&h, hb, &prop, 0));   // ... and most likely come out again here
  Mapped m2 = map_handles(hs2, hb);
  const double b2_fill = fill_rate(m2.va, big);
  unmap(m);
  for (auto h : hs) CK(hipMemRelease(h));          // the first handle set's pages go back ...
  void* z;
  CK(hipMalloc(&z, big));                          // ... and most likely come out again here
  const double z_fill = fill_rate(z, big);
  printf("{\"reuse_test_fill_GBps\": {\"hipMalloc X\": %.0f, \"handles while X is held\": %.0f, \"handles after X was freed\": %.0f, "
         "\"hipMalloc Z after the first handles were released\": %.0f}}\n", x_fill, b_fill, b2_fill, z_fill);
  CK(hipDeviceSynchronize());
  return 0;   // process exit returns everything
}

// mode "fine": per handle size, a 2 GiB hipMalloc block is timed and freed, handles of that size are created at once (most
// likely out of the same pages) and mapped in order, then shuffled: does scatter below 2 MiB change the write rate?
static int run_fine(hipMemAllocationProp& prop, unsigned* d_out) {
  const size_t bytes = 2ull << 30;
  std::mt19937_64 rng(777);
  const size_t sizes[] = {2048u << 10, 1024u << 10, 512u << 10, 256u << 10, 128u << 10};
  for (size_t hb : sizes) {
    void* x;
    CK(hipMalloc(&x, bytes));
    const double xf = fill_rate(x, bytes), xs = sum_rate(x, bytes, d_out);
    CK(hipDeviceSynchronize());
    CK(hipFree(x));
    std::vector<hipMemGenericAllocationHandle_t> hs(bytes / hb);
    for (auto& h : hs) CK(hipMemCreate(&h, hb, &prop, 0));
    Mapped m = map_handles(hs, hb);
    const double bf = fill_rate(m.va, bytes), bs = sum_rate(m.va, bytes, d_out);
    unmap(m);
    std::shuffle(hs.begin(), hs.end(), rng);
    Mapped m2 = map_handles(hs, hb);
    const double cf = fill_rate(m2.va, bytes), cs = sum_rate(m2.va, bytes, d_out);
    unmap(m2);
    for (auto h : hs) CK(hipMemRelease(h));
    printf("{\"handle_KiB\": %zu, \"fill_sum_GBps\": {\"hipMalloc\": [%.0f, %.0f], \"handles in order\": [%.0f, %.0f], \"handles shuffled\": [%.0f, %.0f]}}\n",
           hb >> 10, xf, xs, bf, bs, cf, cs);
    fflush(stdout);
  }
  return 0;
}

// mode "select U H": U units of 1 GiB, each of H-KiB handles, created one after the other and probed with a fill; then ranges of
// R GiB are built from the fastest, the slowest and the first-created units and timed with fill / copy / a three-stream write.
__global__ void __launch_bounds__(256, 7) k_w3(u32x4* __restrict__ a, u32x4* __restrict__ b, u32x4* __restrict__ c, const u32x4* __restrict__ src,
                                              uint64_t steps) {
  // decode's traffic shape per step and lane: 24 B read, 16 + 12 + 8 B written (here 32 R : 16 + 16 + 16 W in whole chunks)
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < steps; s += stride) {
    u32x4 v = __builtin_nontemporal_load(src + 2 * s), w = __builtin_nontemporal_load(src + 2 * s + 1);
    __builtin_nontemporal_store(v, a + s);
    __builtin_nontemporal_store(w, b + s);
    __builtin_nontemporal_store(v ^ w, c + s);
  }
}
static int run_select(int units, size_t hb, int range_gib, hipMemAllocationProp& prop, unsigned* d_out) {
  const size_t ub = 1ull << 30, per = ub / hb;
  struct Unit { std::vector<hipMemGenericAllocationHandle_t> hs; double fill; int id; };
  std::vector<Unit> us(units);
  void* probe_va;
  CK(hipMemAddressReserve(&probe_va, ub, 0, nullptr, 0));
  hipMemAccessDesc d{};
  d.location.type = hipMemLocationTypeDevice;
  d.location.id = 0;
  d.flags = hipMemAccessFlagsProtReadWrite;
  printf("{\"unit_fill_GBps\": [");
  for (int u = 0; u < units; ++u) {
    us[u].id = u;
    us[u].hs.resize(per);
    for (auto& h : us[u].hs) CK(hipMemCreate(&h, hb, &prop, 0));
    for (size_t i = 0; i < per; ++i) CK(hipMemMap((char*)probe_va + i * hb, hb, 0, us[u].hs[i], 0));
    CK(hipMemSetAccess(probe_va, ub, &d, 1));
    us[u].fill = fill_rate(probe_va, ub);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(probe_va, ub));
    printf("%s%.0f", u ? ", " : "", us[u].fill);
    fflush(stdout);
  }
  printf("]}\n");
  std::vector<Unit*> by(units);
  for (int u = 0; u < units; ++u) by[u] = &us[u];
  std::stable_sort(by.begin(), by.end(), [](Unit* a, Unit* b) { return a->fill > b->fill; });
  void* src;
  const size_t rb = (size_t)range_gib << 30;
  CK(hipMalloc(&src, rb));
  CK(hipMemset(src, 1, rb));
  auto build = [&](const char* tag, std::vector<Unit*> pick) {
    std::vector<hipMemGenericAllocationHandle_t> hs;
    double mean = 0;
    for (auto* u : pick) { hs.insert(hs.end(), u->hs.begin(), u->hs.end()); mean += u->fill / pick.size(); }
    Mapped m = map_handles(hs, hb);
    const uint64_t steps = rb / 16 / 3;   // three equal output streams inside the range
    u32x4* base = (u32x4*)m.va;
    double w3 = best_ms([&] { hipLaunchKernelGGL(k_w3, dim3(256 * 7), dim3(256), 0, 0, base, base + steps, base + 2 * steps, (const u32x4*)src, steps); });
    printf("{\"range\": \"%s\", \"GiB\": %d, \"mean_unit_fill\": %.0f, \"fill_GBps\": %.0f, \"copy_GBps\": %.0f, \"sum_GBps\": %.0f, \"w3_GBps\": %.0f}\n", tag, range_gib, mean,
           fill_rate(m.va, rb), 0.0, sum_rate(m.va, rb, d_out), steps * 80.0 / w3 / 1e6);
    measure(tag, m.va, src, rb, d_out);
    unmap(m);
  };
  std::vector<Unit*> fast(by.begin(), by.begin() + range_gib), slow(by.end() - range_gib, by.end()), first;
  for (int u = 0; u < range_gib; ++u) first.push_back(&us[u]);
  build("fastest units", fast);
  build("slowest units", slow);
  build("first-created units", first);
  void* plain;
  CK(hipMalloc(&plain, rb));
  {
    const uint64_t steps = rb / 16 / 3;
    u32x4* base = (u32x4*)plain;
    double w3 = best_ms([&] { hipLaunchKernelGGL(k_w3, dim3(256 * 7), dim3(256), 0, 0, base, base + steps, base + 2 * steps, (const u32x4*)src, steps); });
    printf("{\"range\": \"hipMalloc\", \"w3_GBps\": %.0f}\n", steps * 80.0 / w3 / 1e6);
  }
  measure("hipMalloc", plain, src, rb, d_out);
  CK(hipDeviceSynchronize());
  return 0;
}

// mode "vapa G": is the write rate a property of the physical pages or of the virtual address?  P physical sets (2 MiB
// handles, G GiB each) x V reserved virtual ranges; every set is mapped into every range in turn and filled / copied into.
static int run_vapa(int gib, hipMemAllocationProp& prop, unsigned* d_out) {
  const size_t hb = 2ull << 20, bytes = (size_t)gib << 30, n = bytes / hb;
  const int P = 4, V = 4;
  std::vector<std::vector<hipMemGenericAllocationHandle_t>> sets(P, std::vector<hipMemGenericAllocationHandle_t>(n));
  std::vector<void*> spacer;
  for (int p = 0; p < P; ++p) {
    for (auto& h : sets[p]) CK(hipMemCreate(&h, hb, &prop, 0));
    void* sp;                                   // something else between the sets
    CK(hipMalloc(&sp, (size_t)(3 + 2 * p) << 30));
    spacer.push_back(sp);
  }
  void* vas[V];
  for (int v = 0; v < V; ++v) {
    CK(hipMemAddressReserve(&vas[v], bytes + ((size_t)v << 30), 0, nullptr, 0));   // different lengths: different addresses
  }
  void* src;
  CK(hipMalloc(&src, bytes));
  CK(hipMemset(src, 1, bytes));
  hipMemAccessDesc d{};
  d.location.type = hipMemLocationTypeDevice;
  d.location.id = 0;
  d.flags = hipMemAccessFlagsProtReadWrite;
  for (int rep = 0; rep < 2; ++rep)
    for (int p = 0; p < P; ++p)
      for (int v = 0; v < V; ++v) {
        for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)vas[v] + i * hb, hb, 0, sets[p][i], 0));
        CK(hipMemSetAccess(vas[v], bytes, &d, 1));
        const double f = fill_rate(vas[v], bytes);
        const uint64_t chunks = bytes / 16;
        const double c = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 7), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)vas[v], chunks); });
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(vas[v], bytes));
        printf("{\"rep\": %d, \"physical_set\": %d, \"va\": \"%p\", \"fill_GBps\": %.0f, \"copy_GBps\": %.0f}\n", rep, p, vas[v], f, 2.0 * bytes / c / 1e6);
        fflush(stdout);
      }
  return 0;
}

// argv: "map [groups]"  or  pairs "GiB handleKiB" ...; per pair: hipMalloc, VMM (handles in order), VMM (shuffled), hipMalloc again
int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("{\"granularity_min\": %zu, \"granularity_recommended\": %zu}\n", gmin, grec);
  unsigned* d_out;
  CK(hipMalloc(&d_out, 4));
  std::mt19937_64 rng(12345);
  if (argc > 3 && !strcmp(argv[1], "select"))
    return run_select(atoi(argv[2]), (size_t)atoi(argv[3]) << 10, argc > 4 ? atoi(argv[4]) : 24, prop, d_out);
  if (argc > 1 && !strcmp(argv[1], "vapa")) return run_vapa(argc > 2 ? atoi(argv[2]) : 12, prop, d_out);
  if (argc > 1 && !strcmp(argv[1], "fine")) return run_fine(prop, d_out);
  if (argc > 1 && !strcmp(argv[1], "map")) return run_map(argc > 2 ? atoi(argv[2]) : 8, prop, d_out);
  for (int a = 1; a + 1 < argc; a += 2) {
    const size_t bytes = (size_t)atoi(argv[a]) << 30;
    size_t hbytes = (size_t)atoi(argv[a + 1]) << 10;
    hbytes = (hbytes + gmin - 1) / gmin * gmin;
    printf("{\"GiB\": %zu, \"handle_bytes\": %zu}\n", bytes >> 30, hbytes);
    void *src, *plain;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    CK(hipMalloc(&plain, bytes));
    measure("A hipMalloc", plain, src, bytes, d_out);
    const size_t nh = bytes / hbytes;
    std::vector<hipMemGenericAllocationHandle_t> hs(nh);
    for (size_t i = 0; i < nh; ++i) CK(hipMemCreate(&hs[i], hbytes, &prop, 0));
    {
      Mapped m = map_handles(hs, hbytes);
      measure("B handles in order", m.va, src, bytes, d_out);
      measure("A hipMalloc (again)", plain, src, bytes, d_out);
      measure("B handles in order (again)", m.va, src, bytes, d_out);
      unmap(m);
    }
    {
      std::vector<hipMemGenericAllocationHandle_t> sh = hs;
      std::shuffle(sh.begin(), sh.end(), rng);
      Mapped m = map_handles(sh, hbytes);
      measure("C handles shuffled", m.va, src, bytes, d_out);
      unmap(m);
    }
    for (auto h : hs) CK(hipMemRelease(h));
    // the source as VMM memory too: is the read side of a copy affected?
    {
      std::vector<hipMemGenericAllocationHandle_t> h2(nh);
      for (size_t i = 0; i < nh; ++i) CK(hipMemCreate(&h2[i], hbytes, &prop, 0));
      Mapped m = map_handles(h2, hbytes);
      CK(hipMemset(m.va, 1, bytes));
      measure("F hipMalloc dst, VMM src", plain, m.va, bytes, d_out);
      unmap(m);
      for (auto h : h2) CK(hipMemRelease(h));
    }
    CK(hipFree(plain));
    CK(hipFree(src));
  }
  CK(hipFree(d_out));
  return 0;
}
