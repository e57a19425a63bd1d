// Random 128-B line reads (+ optional dependent 256-B row read in the same 4-KiB block) over growing prefixes of one
// large allocation: shows where address translation, not HBM bandwidth, starts to bound a hash-table probe.
//   hipcc --offload-arch=gfx950 -O3 scripts/mb/tlb_probe.hip -o gpurun_out/tlb_probe && gpurun_out/tlb_probe [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

// mode 0: one 128-B line per access (16 lanes x 8 B).  mode 1: + 256-B row in the same 4-KiB block.
// mode 2: + 256-B row in ANOTHER random block.  U accesses in flight per 16-lane group.
template <int U, int MODE>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ base, uint64_t nblocks4k, uint32_t n, uint64_t seed,
                                              uint64_t* __restrict__ out, void* __restrict__ outrows) {
  const int lane = threadIdx.x & 15;
  const uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  uint64_t acc = 0;
  uint64_t v[U]; uint4 r[U];
  const char* p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint32_t i = grp * U + u; if (i >= n) i = n - 1;
    uint64_t h = mix(seed + i);
    uint64_t blk = __umul64hi(h, nblocks4k);
    p[u] = base + blk * 4096;
    v[u] = *(const uint64_t*)(p[u] + lane * 8);
  }
  if (MODE >= 1) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t k = __shfl((int)v[u], (threadIdx.x & 48));   // a data-dependent value, same for the 16 lanes
      const char* q;
      unsigned slot = (unsigned)(k >> 4) & 15u; if (slot == 15) slot = 7;
      if (MODE == 1 || MODE == 3) q = p[u] + 256 + slot * 256;  // a row of this block
      else { uint64_t blk2 = __umul64hi((k | 1) * 0x9E3779B97F4A7C15ULL, nblocks4k); q = base + blk2 * 4096 + 256 + slot * 256; }
      r[u] = *(const uint4*)(q + lane * 16);
    }
    if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < U; ++u) *(uint4*)((char*)outrows + (size_t)(grp * U + u) * 256 + lane * 16) = r[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += r[u].x + r[u].y + r[u].z + r[u].w;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) acc += v[u];
  if (acc == 0x1234567deadbeefULL) out[0] = acc;
}

__global__ void fill(uint64_t* p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = i * 0x9E3779B97F4A7C15ULL;
}

template <int U, int MODE>
float run(const char* base, uint64_t nb4k, uint32_t n, uint64_t* out, int reps) {
  static void* outrows = nullptr; if (!outrows) CK(hipMalloc(&outrows, (size_t)(1u << 20) * 256 + 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t groups = (n + U - 1) / U; uint32_t blocks = (groups * 16 + 255) / 256;
  for (int i = 0; i < 3; ++i) probe<U, MODE><<<blocks, 256>>>(base, nb4k, n, 1000 + i, out, outrows);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) probe<U, MODE><<<blocks, 256>>>(base, nb4k, n, 77777ULL * (i + 5), out, outrows);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  size_t gib = argc > 1 ? atol(argv[1]) : 256;
  size_t bytes = gib << 30;
  char* base; 
  size_t fr, tot; CK(hipMemGetInfo(&fr, &tot)); printf("HBM free %.1f GiB total %.1f GiB\n", fr / 1073741824.0, tot / 1073741824.0);
  while (hipMalloc(&base, bytes) != hipSuccess) { (void)hipGetLastError(); gib -= 8; bytes = gib << 30; }
  printf("allocated %zu GiB at %p\n", gib, base);
  fill<<<4096, 256>>>((uint64_t*)base, bytes / 8); CK(hipDeviceSynchronize());
  uint64_t* out; CK(hipMalloc(&out, 64));
  std::vector<size_t> regions = {1ull << 28, 1ull << 30, 4ull << 30, 16ull << 30, 64ull << 30, 128ull << 30, bytes};
  for (uint32_t n : {131072u, 1048576u}) {
    printf("\n n=%u accesses per launch; us per launch (M accesses/s)\n region      line/U4      line/U8     line/U16   +rowsame/U4  +rowsame/U8  +rowother/U4  find-like/U4\n", n);
    for (size_t rg : regions) {
      if (rg > bytes) continue;
      uint64_t nb = rg / 4096;
      float a = run<4, 0>(base, nb, n, out, 20), b = run<8, 0>(base, nb, n, out, 20), c = run<16, 0>(base, nb, n, out, 20);
      float d = run<4, 1>(base, nb, n, out, 20), e = run<8, 1>(base, nb, n, out, 20), f = run<4, 2>(base, nb, n, out, 20), g = run<4, 3>(base, nb, n, out, 20);
      printf("%6.2f GiB  %7.1f(%5.0f) %7.1f(%5.0f) %7.1f(%5.0f) %7.1f(%5.0f) %7.1f(%5.0f) %7.1f(%5.0f) %7.1f(%5.0f)\n", rg / 1073741824.0,
             a, n / a, b, n / b, c, n / c, d, n / d, e, n / e, f, n / f, g, n / g);
    }
  }
  return 0;
}
