// Random agent-scope atomics on lines of a large allocation: what do bucket-ownership claims / slot CAS cost next to the
// line reads they travel with?    scripts/mb/atomic_probe.bin [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// MODE 0: line read.  1: line read + atomicExch(u32) on word 31 of the line (lane 0 of the group).  2: atomicExch only.
// 3: line read + plain 4-B store to word 31.  4: line read + atomicCAS(u64) on word 3.  5: line + atomicExch on a
// SEPARATE dense tag array (4 B per block).   U accesses per 16-lane group in flight.
template <int U, int MODE>
__global__ __launch_bounds__(256) void probe(char* __restrict__ base, uint64_t nblocks4k, uint32_t n, uint64_t seed, unsigned gen,
                                             uint64_t* __restrict__ out, unsigned* __restrict__ tags) {
  const int lane = threadIdx.x & 15;
  const uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  uint64_t acc = 0, v[U]; unsigned t[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint32_t i = grp * U + u; if (i >= n) i = n - 1;
    uint64_t blk = __umul64hi(mix(seed + i), nblocks4k);
    char* p = base + blk * 4096;
    v[u] = 0; t[u] = 0;
    if (MODE != 2) v[u] = *(const uint64_t*)(p + lane * 8);
    if (lane == 0) {
      if (MODE == 1 || MODE == 2) t[u] = atomicExch((unsigned*)(p + 124), gen);
      if (MODE == 3) *(volatile unsigned*)(p + 124) = gen;
      if (MODE == 4) t[u] = (unsigned)atomicCAS((unsigned long long*)(p + 24), 0x1234ULL + gen, (unsigned long long)gen);
      if (MODE == 5) t[u] = atomicExch(tags + blk, gen);
    }
  }
  if (MODE >= 6) {   // DEPENDENT writes after the line has arrived: 6 = 8-B store into the same line, 7 = 8-B store into another
                     // line of the same 4-KiB block, 8 = atomicCAS(u64) on the same line, 9 = 256-B row store in the block
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t i = grp * U + u; if (i >= n) i = n - 1;
      uint64_t blk = __umul64hi(mix(seed + i), nblocks4k);
      char* p = base + blk * 4096;
      uint64_t k = __shfl((int)v[u], (threadIdx.x & 48)) | ((uint64_t)__shfl((int)(v[u] >> 32), (threadIdx.x & 48)) << 32);
      unsigned slot = (unsigned)(k >> 4) & 7u;
      if (MODE == 6 && lane == 0) *(volatile uint64_t*)(p + 8 * slot) = k + gen;
      if (MODE == 7 && lane == 0) *(volatile uint64_t*)(p + 128 + 8 * slot) = k + gen;
      if (MODE == 8 && lane == 0) t[u] = (unsigned)atomicCAS((unsigned long long*)(p + 8 * slot), k ^ (k >> 60), (unsigned long long)(k + gen));
      if (MODE == 9) *(uint4*)(p + 256 + slot * 256 + lane * 16) = make_uint4((unsigned)k, gen, lane, 0);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) acc += v[u] + t[u];
  if (acc == 0x1234567deadbeefULL) out[0] = acc;
}
__global__ void fill(uint64_t* p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = i * 0x9E3779B97F4A7C15ULL;
}
template <int U, int MODE>
float run(char* base, uint64_t nb4k, uint32_t n, uint64_t* out, unsigned* tags, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t groups = (n + U - 1) / U, blocks = (groups * 16 + 255) / 256;
  for (int i = 0; i < 3; ++i) probe<U, MODE><<<blocks, 256>>>(base, nb4k, n, 1000 + i, 7 + i, out, tags);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) probe<U, MODE><<<blocks, 256>>>(base, nb4k, n, 77777ULL * (i + 5), 100 + i, out, tags);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}
int main(int argc, char** argv) {
  size_t gib = argc > 1 ? atol(argv[1]) : 256, bytes = gib << 30;
  char* base; while (hipMalloc(&base, bytes) != hipSuccess) { (void)hipGetLastError(); gib -= 8; bytes = gib << 30; }
  printf("allocated %zu GiB\n", gib);
  fill<<<4096, 256>>>((uint64_t*)base, bytes / 8); CK(hipDeviceSynchronize());
  uint64_t* out; CK(hipMalloc(&out, 64));
  unsigned* tags; CK(hipMalloc(&tags, (bytes / 4096) * 4)); CK(hipMemset(tags, 0, (bytes / 4096) * 4));
  for (uint32_t n : {157056u, 1048576u}) {
    printf("\n n=%u per launch; us per launch\n region     line   line+xchg   xchg-only  line+store  line+cas64  line+xchg(sep)   [U=4]   | dependent: st-same-line st-other-line cas-same-line row-store\n", n);
    for (size_t rg : {(size_t)1 << 30, (size_t)16 << 30, bytes}) {
      uint64_t nb = rg / 4096;
      float a = run<4, 0>(base, nb, n, out, tags, 20), b = run<4, 1>(base, nb, n, out, tags, 20), c = run<4, 2>(base, nb, n, out, tags, 20);
      float d = run<4, 3>(base, nb, n, out, tags, 20), e = run<4, 4>(base, nb, n, out, tags, 20), f = run<4, 5>(base, nb, n, out, tags, 20);
      float g = run<4, 6>(base, nb, n, out, tags, 20), h = run<4, 7>(base, nb, n, out, tags, 20), i8 = run<4, 8>(base, nb, n, out, tags, 20), i9 = run<4, 9>(base, nb, n, out, tags, 20);
      printf("%6.1f GiB %7.1f %9.1f %11.1f %11.1f %11.1f %11.1f            | %10.1f %10.1f %10.1f %10.1f\n", rg / 1073741824.0, a, b, c, d, e, f, g, h, i8, i9);
    }
  }
  return 0;
}
