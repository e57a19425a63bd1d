// Does the HIP virtual-memory API work on this box, and is memory mapped through it as fast to probe at random as
// hipMalloc memory?  Reserve 64 GiB of VA, map 8 x 4 GiB handles one after another (a table growing in place), random
// 128-B line reads over what is mapped, then the same over one hipMalloc of the same size.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/mb/vmm_probe.bin scripts/mb/vmm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(const unsigned long long* base, size_t nlines, unsigned iters, unsigned long long* out) {
  unsigned long long acc = 0;
  unsigned long long x = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
  const int sub = threadIdx.x & 15;
  for (unsigned i = 0; i < iters; ++i) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
    unsigned long long g = __shfl(x, threadIdx.x & 48, 64);           // one line per 16 lanes
    size_t line = (size_t)(((unsigned __int128)g * nlines) >> 64);
    acc += base[line * 16 + sub];
  }
  if (acc == 0x1234567) out[0] = acc;
}

static float time_probe(const void* p, size_t bytes) {
  unsigned long long* out; hipMalloc(&out, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned blocks = 256 * 16, threads = 256, iters = 64;
  probe<<<blocks, threads>>>((const unsigned long long*)p, bytes / 128, 4, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<<<blocks, threads>>>((const unsigned long long*)p, bytes / 128, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double lines = (double)blocks * threads / 16 * iters;
  printf("   %.1f GiB: %.3f ms, %.2f G lines/s, %.0f GB/s\n", bytes / 1073741824.0, ms, lines / ms / 1e6, lines * 128 / ms / 1e6);
  hipFree(out);
  return ms;
}

int main(int argc, char** argv) {
  const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 4) << 30, nchunks = 8, va_size = chunk * nchunks;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity %zu\n", gran);
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, va_size, 0, nullptr, 0));
  printf("reserved %zu GiB at %p\n", va_size >> 30, va);
  std::vector<hipMemGenericAllocationHandle_t> handles;
  hipMemAccessDesc acc = {};
  acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  for (size_t i = 0; i < nchunks; ++i) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char*)va + i * chunk, chunk, 0, h, 0));
    CK(hipMemSetAccess((char*)va + i * chunk, chunk, &acc, 1));
    handles.push_back(h);
    CK(hipMemset((char*)va + i * chunk, 1, chunk));
    if (i == 0 || i == 3 || i == 7) { printf("VMM, %zu chunks mapped\n", i + 1); time_probe(va, (i + 1) * chunk); }
  }
  for (size_t i = 0; i < nchunks; ++i) { CK(hipMemUnmap((char*)va + i * chunk, chunk)); CK(hipMemRelease(handles[i])); }
  CK(hipMemAddressFree(va, va_size));
  void* p;
  CK(hipMalloc(&p, va_size));
  CK(hipMemset(p, 1, va_size));
  printf("hipMalloc\n");
  time_probe(p, chunk); time_probe(p, va_size);
  hipFree(p);
  printf("ok\n");
  return 0;
}
