// Ablation entry point for kernel tuning (scripts/microbench.py).  Not part of the public ABI.
#include <hip/hip_runtime.h>

#include "../../../include/tfra_mi355x.h"
#include "../tfra_device.h"
#include "../tfra_host.h"

using namespace tfra;

// MODE 0 full find; 1 no probe (row = i mod rows: gather+store only); 2 probe only (1 float out);
// 3 stream zeros to out (no table access); 4 ids -> hash -> first key line load only
template <int MODE, int U>
__global__ __launch_bounds__(256) void find_variant_kernel(TableView v, size_t n, const i64* __restrict__ keys,
                                                           unsigned char* __restrict__ out,
                                                           const unsigned char* __restrict__ defaults) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, grp = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int KPW = 4 * U;
  const size_t base = wave * KPW;
  if (base >= n) return;
  i64 kreg = (lane < KPW && base + lane < n) ? keys[base + lane] : 0;
  i64 key[U]; u64 h[U], b[U]; i64 k0[U]; bool valid[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = u * 4 + grp;
    key[u] = shfl_i64(kreg, j);
    valid[u] = base + j < n;
    b[u] = bucket0(key[u], v.nb, h[u]);
    k0[u] = (MODE == 0 || MODE == 2 || MODE == 4) && valid[u] ? v.keys[b[u] * 16 + sub] : EMPTY_KEY;
  }
  const unsigned char* src[U]; unsigned char* dst[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    size_t i = base + u * 4 + grp;
    i64 row = -1;
    if (MODE == 0 || MODE == 2) row = valid[u] ? probe_find_from<false>(v, key[u], h[u], b[u], k0[u], sub, gshift) : -1;
    if (MODE == 1) row = (i64)(fmix64(i) % (v.nb * SLOTS));
    if (MODE == 4) row = (k0[u] == key[u]) ? 1 : -1;
    src[u] = row >= 0 ? v.rows + (size_t)row * v.row_stride : defaults;
    dst[u] = out + i * (size_t)v.field_bytes;
  }
  if (MODE == 2 || MODE == 4) {
#pragma unroll
    for (int u = 0; u < U; ++u) if (valid[u] && sub == 0) *reinterpret_cast<float*>(dst[u]) = src[u] == defaults ? 0.f : 1.f;
    return;
  }
  for (unsigned off = sub * 16; off < v.field_bytes; off += 256) {
    uint4 tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (valid[u]) tmp[u] = MODE == 3 ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(src[u] + off);
#pragma unroll
    for (int u = 0; u < U; ++u) if (valid[u]) *reinterpret_cast<uint4*>(dst[u] + off) = tmp[u];
  }
}

template <int G, int U>
__global__ __launch_bounds__(256) void find_kernel_copy(TableView v, size_t n, const i64* __restrict__ keys,
                                                   unsigned char* __restrict__ out,
                                                   uint8_t* __restrict__ exists,
                                                   const unsigned char* __restrict__ defaults,
                                                   int full, unsigned field_off) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const int grp = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int KPW = 4 * U;
  const size_t base = wave * KPW;
  if (base >= n) return;
  // Loads are kept UNCONDITIONAL (tail keys are clamped to the last valid index): a load inside an
  // `if (valid)` block gets its own `s_waitcnt vmcnt(0)` and the U probes / U rows would be fetched
  // one latency after the other instead of all in flight (measured 23 us -> 11 us per 131072 keys).
  const size_t last = n - 1;
  i64 kreg = keys[min(base + (size_t)(lane & (KPW - 1)), last)];
  i64 key[U];
  u64 h[U], b[U];
  i64 k0[U];
  size_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = u * 4 + grp;
    key[u] = shfl_i64(kreg, j);
    idx[u] = min(base + j, last);
    b[u] = bucket0(key[u], v.nb, h[u]);
    k0[u] = v.keys[b[u] * 16 + sub];  // U probes in flight
  }
  __builtin_amdgcn_sched_barrier(0);  // keep the U loads ahead of their first use
  const unsigned char* src[U];
  unsigned char* dst[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    i64 row = probe_find_from<false>(v, key[u], h[u], b[u], k0[u], sub, gshift);
    if (exists && sub == 0) exists[idx[u]] = row >= 0;
    src[u] = row >= 0 ? v.rows + (size_t)row * v.row_stride + field_off
                      : defaults + (full ? idx[u] * (size_t)v.field_bytes : 0);
    dst[u] = out + idx[u] * (size_t)v.field_bytes;
  }
  typedef typename Granule<G>::T T;
  for (unsigned off = sub * G; off < v.field_bytes; off += 16 * G) {
    T tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) tmp[u] = *reinterpret_cast<const T*>(src[u] + off);  // U rows in flight
    __builtin_amdgcn_sched_barrier(0);  // ... and do not let the scheduler pair each load with its store
#pragma unroll
    for (int u = 0; u < U; ++u) *reinterpret_cast<T*>(dst[u] + off) = tmp[u];  // clamped tail: same bytes twice
  }
}



// bisect kernel: CLAMP = unconditional loads with clamped tail; BAR = sched_barriers; EX = exists store
template <bool CLAMP, bool BAR, bool EX>
__global__ __launch_bounds__(256) void fk(TableView v, size_t n, const i64* __restrict__ keys, unsigned char* __restrict__ out,
                                          uint8_t* __restrict__ exists, const unsigned char* __restrict__ defaults) {
  constexpr int U = 4;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, grp = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int KPW = 4 * U;
  const size_t base = wave * KPW;
  if (base >= n) return;
  const size_t last = n - 1;
  i64 kreg = CLAMP ? keys[min(base + (size_t)(lane & (KPW - 1)), last)] : ((lane < KPW && base + lane < n) ? keys[base + lane] : 0);
  i64 key[U]; u64 h[U], b[U]; i64 k0[U]; bool valid[U]; size_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = u * 4 + grp;
    key[u] = shfl_i64(kreg, j);
    valid[u] = CLAMP ? true : (base + j < n);
    idx[u] = CLAMP ? min(base + j, last) : base + j;
    b[u] = bucket0(key[u], v.nb, h[u]);
    k0[u] = valid[u] ? v.keys[b[u] * 16 + sub] : EMPTY_KEY;
  }
  if (BAR) __builtin_amdgcn_sched_barrier(0);
  const unsigned char* src[U]; unsigned char* dst[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    i64 row = valid[u] ? probe_find_from<false>(v, key[u], h[u], b[u], k0[u], sub, gshift) : -1;
    if (EX && valid[u] && exists && sub == 0) exists[idx[u]] = row >= 0;
    src[u] = row >= 0 ? v.rows + (size_t)row * v.row_stride : defaults;
    dst[u] = out + idx[u] * (size_t)v.field_bytes;
  }
  for (unsigned off = sub * 16; off < v.field_bytes; off += 256) {
    uint4 tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (valid[u]) tmp[u] = *reinterpret_cast<const uint4*>(src[u] + off);
    if (BAR) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) if (valid[u]) *reinterpret_cast<uint4*>(dst[u] + off) = tmp[u];
  }
}

template <int MODE>
static void launch_mode(int U, TableView v, size_t n, const i64* k, unsigned char* o, const unsigned char* d, hipStream_t s) {
  auto grid = [&](int u) { size_t waves = (n + 4 * u - 1) / (4 * u); return dim3((unsigned)((waves + 3) / 4)); };
  switch (U) {
    case 1: find_variant_kernel<MODE, 1><<<grid(1), 256, 0, s>>>(v, n, k, o, d); break;
    case 2: find_variant_kernel<MODE, 2><<<grid(2), 256, 0, s>>>(v, n, k, o, d); break;
    case 8: find_variant_kernel<MODE, 8><<<grid(8), 256, 0, s>>>(v, n, k, o, d); break;
    default: find_variant_kernel<MODE, 4><<<grid(4), 256, 0, s>>>(v, n, k, o, d); break;
  }
}

extern "C" int tfra_debug_find_variant(tfra_table_t* tp, int mode, int U, size_t n, const int64_t* keys, void* values,
                                       const void* defaults, tfra_stream_t stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  TableView v = t->view_of(t->cur);
  hipStream_t s = (hipStream_t)stream;
  const i64* k = (const i64*)keys;
  unsigned char* o = (unsigned char*)values;
  const unsigned char* d = (const unsigned char*)defaults;
  if (mode == 7 || mode == 8) {  // verbatim production kernel; 8 = with an exists buffer
    size_t waves = (n + 15) / 16;
    static uint8_t* ex = nullptr;
    if (mode == 8 && !ex) (void)hipMalloc((void**)&ex, 1 << 24);
    find_kernel_copy<16, 4><<<dim3((unsigned)((waves + 3) / 4)), 256, 0, s>>>(v, n, k, o, mode == 8 ? ex : nullptr, d, 0, 0);
    return 0;
  }
  if (mode >= 10 && mode < 18) {
    size_t waves = (n + 15) / 16;
    dim3 g((unsigned)((waves + 3) / 4));
    static uint8_t* ex = nullptr;
    if (!ex) (void)hipMalloc((void**)&ex, 1 << 24);
    switch (mode - 10) {
      case 0: fk<false, false, false><<<g, 256, 0, s>>>(v, n, k, o, ex, d); break;
      case 1: fk<true, false, false><<<g, 256, 0, s>>>(v, n, k, o, ex, d); break;
      case 2: fk<false, true, false><<<g, 256, 0, s>>>(v, n, k, o, ex, d); break;
      case 3: fk<true, true, false><<<g, 256, 0, s>>>(v, n, k, o, ex, d); break;
      case 4: fk<false, false, true><<<g, 256, 0, s>>>(v, n, k, o, ex, d); break;
      case 5: fk<true, false, true><<<g, 256, 0, s>>>(v, n, k, o, ex, d); break;
      case 6: fk<false, true, true><<<g, 256, 0, s>>>(v, n, k, o, ex, d); break;
      default: fk<true, true, true><<<g, 256, 0, s>>>(v, n, k, o, ex, d); break;
    }
    return 0;
  }
  switch (mode) {
    case 1: launch_mode<1>(U, v, n, k, o, d, s); break;
    case 2: launch_mode<2>(U, v, n, k, o, d, s); break;
    case 3: launch_mode<3>(U, v, n, k, o, d, s); break;
    case 4: launch_mode<4>(U, v, n, k, o, d, s); break;
    default: launch_mode<0>(U, v, n, k, o, d, s); break;
  }
  return 0;
}
