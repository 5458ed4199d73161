// vmm_check.hip — does a range built from handles of H MiB hold what was written to it?  (measurement / diagnosis only)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_write(uint64_t* p, uint64_t words, uint64_t salt) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull + salt;
}
__global__ void k_verify(const uint64_t* p, uint64_t words, uint64_t salt, unsigned long long* bad, unsigned long long* first) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
    if (p[i] != i * 0x9E3779B97F4A7C15ull + salt) { atomicAdd(bad, 1ull); atomicMin(first, (unsigned long long)i); }
}
int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? atoi(argv[1]) : 4;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  unsigned long long *d_bad, h[2];
  CK(hipMalloc(&d_bad, 16));
  for (int align_mode = 0; align_mode < 2; ++align_mode)
    for (size_t mib : {2, 4, 8, 32, 1024}) {
      const size_t hb = mib << 20, bytes = gib << 30, n = bytes / hb;
      void* va;
      CK(hipMemAddressReserve(&va, bytes, align_mode ? hb : 0, nullptr, 0));
      std::vector<hipMemGenericAllocationHandle_t> hs(n);
      for (size_t i = 0; i < n; ++i) {
        CK(hipMemCreate(&hs[i], hb, &prop, 0));
        CK(hipMemMap((char*)va + i * hb, hb, 0, hs[i], 0));
      }
      hipMemAccessDesc d{};
      d.location.type = hipMemLocationTypeDevice;
      d.location.id = 0;
      d.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, bytes, &d, 1));
      h[0] = 0; h[1] = ~0ull;
      CK(hipMemcpy(d_bad, h, 16, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, (uint64_t*)va, bytes / 8, mib);
      hipLaunchKernelGGL(k_verify, dim3(4096), dim3(256), 0, 0, (const uint64_t*)va, bytes / 8, mib, d_bad, d_bad + 1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost));
      printf("{\"handle_MiB\": %zu, \"reserve_alignment\": %zu, \"va\": \"%p\", \"bad_words\": %llu, \"first_bad_word\": %lld, \"words\": %zu}\n", mib, align_mode ? hb : 0, va,
             h[0], h[0] ? (long long)h[1] : -1ll, bytes / 8);
      fflush(stdout);
      CK(hipMemUnmap(va, bytes));
      for (auto x : hs) CK(hipMemRelease(x));
      CK(hipMemAddressFree(va, bytes));
    }
  return 0;
}
