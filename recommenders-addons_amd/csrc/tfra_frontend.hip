// Front-end helpers that bracket the table ops in the reference:
//   tfra_unique        tf.unique                      PY/dynamic_embedding_ops.py:99
//   tfra_gather_rows   tf.gather(unique_rows, idx)    PY/dynamic_embedding_ops.py:111
//   tfra_segment_sum   unsorted_segment_sum of grads  PY/dynamic_embedding_optimizer.py:177-190
//   tfra_partition     default_partition_fn + dynamic_partition  PY/dynamic_embedding_variable.py:131-197
//   tfra_scatter_rows  dynamic_stitch                 PY/dynamic_embedding_variable.py:157-162
// All stream-ordered, counts stay on the device (no host sync), results deterministic.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <string>

#include "../../include/tfra_mi355x.h"
#include "tfra_device.h"
#include "tfra_host.h"

using namespace tfra;

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return set_error(_e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP,                \
                       std::string(#expr) + ": " + hipGetErrorString(_e));                    \
  } while (0)


namespace {

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// carve helper
struct Carver {
  unsigned char* p;
  size_t used = 0;
  template <class T> T* take(size_t n) {
    T* r = reinterpret_cast<T*>(p + used);
    used += align_up(n * sizeof(T));
    return r;
  }
};

// ------------------------------------ unique --------------------------------------------------
constexpr int UNQ_ITEMS = 4;                   // contiguous ids per thread
constexpr int UNQ_TILE = 256 * UNQ_ITEMS;      // ids per block

__global__ void unq_fill_kernel(i64* hkeys, int* hfirst, size_t cap1) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap1; i += (size_t)gridDim.x * blockDim.x) {
    hkeys[i] = EMPTY_KEY;
    hfirst[i] = 0x7fffffff;
  }
}

// scratch open-addressing set: CAS the id in, remember the smallest input index per distinct id.
// Equal ids of one block are grouped in LDS first and only the group's first position touches the
// global set: a Zipf head id (24 000 repeats in a 131 072-id batch) then costs n/512 global atomics
// on its slot instead of one per repeat (516 us -> tens of us for the whole op).
constexpr int UNQ_INS = 512;       // ids = threads per insert block
constexpr unsigned UNQ_LCAP = 1024;  // LDS group table (2x the block: short chains)
__global__ __launch_bounds__(UNQ_INS) void unq_insert_kernel(size_t n, const i64* __restrict__ ids, i64* hkeys, int* hfirst,
                                                             int* slot_of, size_t cap) {
  __shared__ i64 l_id[UNQ_INS];
  __shared__ unsigned l_own[UNQ_LCAP];   // 0 = free, else (claiming thread + 1)
  __shared__ int l_first[UNQ_LCAP];      // smallest thread index of the group
  __shared__ int l_slot[UNQ_LCAP];       // the group's slot in the global set
  const int t = threadIdx.x;
  const size_t i = (size_t)blockIdx.x * UNQ_INS + t;
  const bool ok = i < n;
  const i64 id = ok ? ids[i] : 0;
  l_id[t] = id;
  for (unsigned q = t; q < UNQ_LCAP; q += UNQ_INS) { l_own[q] = 0; l_first[q] = 0x7fffffff; }
  __syncthreads();
  unsigned h = 0;
  if (ok) {
    h = (unsigned)(fmix64((u64)id) >> 20) & (UNQ_LCAP - 1);
    for (;;) {
      unsigned o = l_own[h];
      if (o == 0) {
        o = atomicCAS(&l_own[h], 0u, (unsigned)t + 1u);
        if (o == 0) break;
      }
      if (l_id[o - 1] == id) break;
      h = (h + 1) & (UNQ_LCAP - 1);
    }
    atomicMin(&l_first[h], t);
  }
  __syncthreads();
  if (ok && l_first[h] == t) {  // first position of its id in this block
    size_t s;
    if (id == EMPTY_KEY) {
      s = cap;  // side slot for the sentinel value itself
    } else {
      s = fmix64((u64)id) & (cap - 1);
      for (;;) {
        i64 cur = load_key_coherent(&hkeys[s]);
        if (cur == id) break;
        if (cur == EMPTY_KEY) {
          i64 old = (i64)atomicCAS((u64*)&hkeys[s], (u64)EMPTY_KEY, (u64)id);
          if (old == EMPTY_KEY || old == id) break;
        }
        s = (s + 1) & (cap - 1);
      }
    }
    atomicMin(&hfirst[s], (int)i);
    l_slot[h] = (int)s;
  }
  __syncthreads();
  if (ok) slot_of[i] = l_slot[h];
}

__device__ __forceinline__ int block_sum_256(int v, int* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void unq_count_kernel(size_t n, const int* __restrict__ hfirst,
                                                        const int* __restrict__ slot_of, int* block_counts) {
  __shared__ int sh[4];
  size_t base = (size_t)blockIdx.x * UNQ_TILE + threadIdx.x * UNQ_ITEMS;
  int c = 0;
#pragma unroll
  for (int k = 0; k < UNQ_ITEMS; ++k) {
    size_t i = base + k;
    if (i < n) c += hfirst[slot_of[i]] == (int)i;
  }
  int t = block_sum_256(c, sh);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = t;
}

// single-block exclusive scan of per-tile counts (in place) + grand total
__global__ __launch_bounds__(1024) void scan_counts_kernel(int* counts, size_t m, i64* total_out) {
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < m; base += 1024) {
    size_t i = base + threadIdx.x;
    int v = i < m ? counts[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    int incl = sh[threadIdx.x];
    if (i < m) counts[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(256) void unq_scatter_kernel(size_t n, const i64* __restrict__ ids,
                                                          const int* __restrict__ hfirst, const int* __restrict__ slot_of,
                                                          const int* __restrict__ block_off, i64* unique_out, int* hrank) {
  __shared__ int wsum[4];
  size_t base = (size_t)blockIdx.x * UNQ_TILE + threadIdx.x * UNQ_ITEMS;
  int flag[UNQ_ITEMS], c = 0;
#pragma unroll
  for (int k = 0; k < UNQ_ITEMS; ++k) {
    size_t i = base + k;
    flag[k] = (i < n) && hfirst[slot_of[i]] == (int)i;
    c += flag[k];
  }
  // exclusive scan of c over the block in thread order (= input order)
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, incl = c;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < w; ++k) woff += wsum[k];
  int pos = block_off[blockIdx.x] + woff + incl - c;
#pragma unroll
  for (int k = 0; k < UNQ_ITEMS; ++k) {
    if (flag[k]) {
      size_t i = base + k;
      unique_out[pos] = ids[i];
      hrank[slot_of[i]] = pos;
      ++pos;
    }
  }
}

__global__ void unq_idx_kernel(size_t n, const int* __restrict__ slot_of, const int* __restrict__ hrank, int* idx_out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx_out[i] = hrank[slot_of[i]];
}

// ---- tfra_unique in three launches (n <= 2^20) ------------------------------------------------------------------
// The six launches above (fill, insert, count, scan, scatter, idx) cost 34 us for 131 072 ids, most of it launch latency.
//   unq2_insert   as unq_insert_kernel, into one of TWO persistent sets; the thread whose compare-and-swap installed a
//                 key appends its slot to the set's used list, and the slots the OTHER set used in the previous call are
//                 emptied on the way (no fill kernel);
//   unq2_scatter  per tile of 1024 ids: first-occurrence flags, tile count published with the call's generation, the
//                 tile's offset = sum of the earlier tiles' counts, read as they appear (decoupled look-back: tile j only
//                 waits for tiles < j, and the hardware dispatches the blocks of a grid in index order, so whatever a
//                 waiting tile needs is already resident or finished — this does NOT rely on the whole grid being
//                 co-resident, other streams may hold CUs), then the scatter of unq_scatter_kernel; the last tile writes the total;
//   unq_idx_kernel.
struct UnqSet {
  i64* hkeys;       // [cap + 1]
  int* hfirst;      // [cap + 1]
  int* hrank;       // [cap + 1]
  unsigned* used;   // [nmax] slots taken by the set's last call
  unsigned* nused;  // their number (zero before its call)
};

__global__ __launch_bounds__(UNQ_INS) void unq2_insert_kernel(size_t n, const i64* __restrict__ ids, UnqSet cur, UnqSet old, int* slot_of,
                                                              size_t cap) {
  __shared__ i64 l_id[UNQ_INS];
  __shared__ unsigned l_own[UNQ_LCAP];
  __shared__ int l_first[UNQ_LCAP];
  __shared__ int l_slot[UNQ_LCAP];
  __shared__ unsigned s_n, s_base;
  const int t = threadIdx.x;
  const size_t i = (size_t)blockIdx.x * UNQ_INS + t;
  const unsigned n_old = *old.nused;
  const bool ok = i < n;
  const i64 id = ok ? ids[i] : 0;
  l_id[t] = id;
  for (unsigned q = t; q < UNQ_LCAP; q += UNQ_INS) { l_own[q] = 0; l_first[q] = 0x7fffffff; }
  if (t == 0) s_n = 0;
  __syncthreads();
  unsigned h = 0;
  if (ok) {
    h = (unsigned)(fmix64((u64)id) >> 20) & (UNQ_LCAP - 1);
    for (;;) {
      unsigned o = l_own[h];
      if (o == 0) {
        o = atomicCAS(&l_own[h], 0u, (unsigned)t + 1u);
        if (o == 0) break;
      }
      if (l_id[o - 1] == id) break;
      h = (h + 1) & (UNQ_LCAP - 1);
    }
    atomicMin(&l_first[h], t);
  }
  __syncthreads();
  bool mine = false;
  unsigned myslot = 0, myidx = 0;
  if (ok && l_first[h] == t) {  // first position of its id in this block
    size_t sl;
    if (id == EMPTY_KEY) {
      sl = cap;  // side slot for the sentinel value itself
      mine = atomicMin(&cur.hfirst[sl], (int)i) == 0x7fffffff;
    } else {
      sl = fmix64((u64)id) & (cap - 1);
      for (;;) {
        const i64 was = (i64)atomicCAS((u64*)&cur.hkeys[sl], (u64)EMPTY_KEY, (u64)id);
        if (was == EMPTY_KEY) { mine = true; break; }
        if (was == id) break;
        sl = (sl + 1) & (cap - 1);
      }
      atomicMin(&cur.hfirst[sl], (int)i);
    }
    l_slot[h] = (int)sl;
    myslot = (unsigned)sl;
    if (mine) myidx = atomicAdd(&s_n, 1u);
  }
  __syncthreads();
  if (ok) slot_of[i] = l_slot[h];
  if (t == 0) s_base = s_n ? atomicAdd(cur.nused, s_n) : 0u;
  __syncthreads();
  if (mine) cur.used[s_base + myidx] = myslot;
  // the other set: empty the slots its last call used
  for (size_t k = (size_t)blockIdx.x * UNQ_INS + t; k < n_old; k += (size_t)gridDim.x * UNQ_INS) {
    const unsigned sl = old.used[k];
    old.hkeys[sl] = EMPTY_KEY;
    old.hfirst[sl] = 0x7fffffff;
  }
}

__global__ __launch_bounds__(256) void unq2_scatter_kernel(size_t n, const i64* __restrict__ ids, UnqSet cur, UnqSet old,
                                                           const int* __restrict__ slot_of, unsigned long long* agg, unsigned gen,
                                                           i64* unique_out, i64* total_out) {
  __shared__ int wsum[4];
  __shared__ int s_off;
  const size_t base = (size_t)blockIdx.x * UNQ_TILE + threadIdx.x * UNQ_ITEMS;
  int flag[UNQ_ITEMS], c = 0;
#pragma unroll
  for (int k = 0; k < UNQ_ITEMS; ++k) {
    const size_t i = base + k;
    flag[k] = (i < n) && cur.hfirst[slot_of[i]] == (int)i;
    c += flag[k];
  }
  // exclusive scan of c over the block in thread order (= input order)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = c;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < w; ++k) woff += wsum[k];
  const int tile_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (threadIdx.x == 0)   // publish this tile's count under the call's generation
    __hip_atomic_store(agg + blockIdx.x, ((unsigned long long)gen << 32) | (unsigned)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // offset = counts of the earlier tiles, read as they appear (wave 0: 64 tiles per sweep)
  if (w == 0) {
    int sum = 0;
    for (unsigned j0 = 0; j0 < blockIdx.x; j0 += 64) {
      const unsigned j = j0 + (unsigned)lane;
      if (j < blockIdx.x) {
        unsigned long long v;
        do { v = __hip_atomic_load(agg + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)(v >> 32) != gen);
        sum += (int)(unsigned)v;
      }
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) s_off = sum;
  }
  __syncthreads();
  int pos = s_off + woff + incl - c;
#pragma unroll
  for (int k = 0; k < UNQ_ITEMS; ++k) {
    if (flag[k]) {
      const size_t i = base + k;
      unique_out[pos] = ids[i];
      cur.hrank[slot_of[i]] = pos;
      ++pos;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    if (total_out) *total_out = (i64)(s_off + tile_total);
    *old.nused = 0;   // the other set is empty again (unq2_insert_kernel of this call emptied it): its next call counts from zero
  }
}

// ------------------------------------ int32 <-> int64 keys ------------------------------------
__global__ __launch_bounds__(256) void widen_keys_kernel(size_t n, const int* __restrict__ in, i64* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (i64)in[i];
}
__global__ __launch_bounds__(256) void narrow_keys_kernel(size_t n, const i64* __restrict__ in, int* __restrict__ out, i64* overflow) {
  unsigned bad = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const i64 k = in[i];
    out[i] = (int)k;
    bad += (i64)(int)k != k;
  }
  if (overflow && bad) atomicAdd(reinterpret_cast<unsigned long long*>(overflow), (unsigned long long)bad);
}

// ------------------------------------ row gather / scatter ------------------------------------
template <int G, bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(size_t n, unsigned row_bytes, const unsigned char* __restrict__ in,
                                                        const int* __restrict__ idx, unsigned char* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const size_t i = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  if (i >= n) return;
  size_t j = (size_t)idx[i];
  const unsigned char* src = in + (SCATTER ? i : j) * (size_t)row_bytes;
  unsigned char* dst = out + (SCATTER ? j : i) * (size_t)row_bytes;
  copy_bytes16<G>(dst, src, row_bytes, sub);
}

// gather of 16-byte-granule rows with U rows of a 16-lane group in flight (the row -> position gather behind a lookup of de-duplicated
// ids moves B rows: one row per group was one dependent idx -> row -> store chain per group, 9-13 us for 131 072 rows of 256 B)
template <int U, bool NT = false>
__global__ __launch_bounds__(256) void gather_rows16_kernel(size_t n, unsigned row_bytes, const unsigned char* __restrict__ in,
                                                            const int* __restrict__ idx, unsigned char* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const size_t base = ((((size_t)blockIdx.x * 256) + threadIdx.x) >> 4) * U;
  if (base >= n) return;
  size_t j[U];
#pragma unroll
  for (int u = 0; u < U; ++u) j[u] = base + u < n ? (size_t)idx[base + u] : 0;
  for (unsigned off = (unsigned)sub * 16u; off < row_bytes; off += 256u) {
    uint4 t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = *reinterpret_cast<const uint4*>(in + j[u] * (size_t)row_bytes + off);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + u < n) {
        if (NT) {
          typedef unsigned v4u __attribute__((ext_vector_type(4)));
          v4u x = {t[u].x, t[u].y, t[u].z, t[u].w};
          __builtin_nontemporal_store(x, reinterpret_cast<v4u*>(out + (base + u) * (size_t)row_bytes + off));
        }
        else *reinterpret_cast<uint4*>(out + (base + u) * (size_t)row_bytes + off) = t[u];
      }
  }
}

template <bool SCATTER>
int move_rows(size_t n, size_t row_bytes, const void* in, const int32_t* idx, void* out, hipStream_t s) {
  if (n == 0) return TFRA_OK;
  if (!in || !idx || !out) return set_error(TFRA_ERR_INVALID, "gather/scatter: null buffer");
  size_t x = row_bytes | (size_t)(uintptr_t)in | (size_t)(uintptr_t)out | 16;
  int g = (int)(x & (~x + 1));
  dim3 grid((unsigned)((n * 16 + 255) / 256)), block(256);
  const unsigned char* i8 = (const unsigned char*)in;
  unsigned char* o8 = (unsigned char*)out;
  unsigned rb = (unsigned)row_bytes;
  if (!SCATTER && g == 16 && n >= 4096) {
    // non-temporal stores: the output (B rows: tens of MB) is not read again by this kernel and does not fit the caches — 9.8 -> 8.5 us for
    // 131 072 rows of 256 B (scripts/mb_gather.py; 8 rows in flight instead of 4: no change)
    gather_rows16_kernel<4, true><<<(unsigned)(((n + 3) / 4 * 16 + 255) / 256), block, 0, s>>>(n, rb, i8, idx, o8);
    HIP_TRY(hipGetLastError());
    return TFRA_OK;
  }
  switch (g) {
    case 16: move_rows_kernel<16, SCATTER><<<grid, block, 0, s>>>(n, rb, i8, idx, o8); break;
    case 8: move_rows_kernel<8, SCATTER><<<grid, block, 0, s>>>(n, rb, i8, idx, o8); break;
    case 4: move_rows_kernel<4, SCATTER><<<grid, block, 0, s>>>(n, rb, i8, idx, o8); break;
    case 2: move_rows_kernel<2, SCATTER><<<grid, block, 0, s>>>(n, rb, i8, idx, o8); break;
    default: move_rows_kernel<1, SCATTER><<<grid, block, 0, s>>>(n, rb, i8, idx, o8); break;
  }
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

// ------------------------------------ segment sum ---------------------------------------------
__global__ void iota_kernel(int* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int)i;
}

__global__ void widen_i32_kernel(const int* __restrict__ in, i64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

// seg_sorted ascending (stable => member indices ascending inside a segment).  One 16-lane group
// per segment walks its members in index order: out[seg,:] = sum_i in[i,:], ONE fp32 add per
// element per member, i.e. exactly the order of a sequential CPU unsorted_segment_sum
// (bit-exact vs the oracle).

__global__ void seg_bounds_kernel(size_t n, const int* __restrict__ seg_sorted, int* seg_start, size_t max_segments) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int s = seg_sorted[p];
  if ((size_t)s >= max_segments) return;
  if (p == 0 || seg_sorted[p - 1] != s) seg_start[s] = (int)p;
  if (p == n - 1 || seg_sorted[p + 1] != s) seg_start[max_segments + s] = (int)p + 1;  // end
}

template <bool VEC4>
__global__ __launch_bounds__(256) void seg_sum_kernel(size_t nseg_cap, const i64* __restrict__ d_nseg, int dim,
                                                      const float* __restrict__ in, const int* __restrict__ member,
                                                      const int* __restrict__ seg_start, float* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const size_t g = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  size_t nseg = (size_t)*d_nseg;
  if (nseg > nseg_cap) nseg = nseg_cap;
  if (g >= nseg) return;
  int b = seg_start[g], e = seg_start[nseg_cap + g];
  float* o = out + g * (size_t)dim;
  if (VEC4) {
    for (int c = sub * 4; c < dim; c += 64) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p = b; p < e; ++p) {
        float4 x = *reinterpret_cast<const float4*>(in + (size_t)member[p] * dim + c);
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
      *reinterpret_cast<float4*>(o + c) = acc;
    }
  } else {
    for (int c = sub; c < dim; c += 16) {
      float acc = 0.f;
      for (int p = b; p < e; ++p) acc += in[(size_t)member[p] * dim + c];
      o[c] = acc;
    }
  }
}

// ------------------------------------ sparse segment combiner -----------------------------------
__global__ void seg64_bounds_kernel(size_t n, const i64* __restrict__ seg, int* start_end, size_t n_rows) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  i64 s = seg[p];
  if (s < 0 || (size_t)s >= n_rows) return;
  if (p == 0 || seg[p - 1] != s) start_end[s] = (int)p;
  if (p == n - 1 || seg[p + 1] != s) start_end[n_rows + s] = (int)p + 1;
}

// one 16-lane group per output row; members in input order, 4 row loads in flight
template <bool VEC4>
__global__ __launch_bounds__(256) void seg_combine_kernel(size_t n_rows, int dim, const float* __restrict__ rows,
                                                          const int* __restrict__ idx, const float* __restrict__ w,
                                                          const int* __restrict__ start_end, int combiner,
                                                          float* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const size_t r = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  if (r >= n_rows) return;
  const int b = start_end[r], e = start_end[n_rows + r];
  float wsum = 0.f;
  for (int p = b; p < e; ++p) { float x = w ? w[p] : 1.f; wsum += combiner == 2 ? x * x : x; }
  float scale = 1.f;
  if (combiner == 1) scale = wsum;
  if (combiner == 2) scale = sqrtf(wsum);
  float* o = out + r * (size_t)dim;
  if (VEC4) {
    for (int c = sub * 4; c < dim; c += 64) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p = b; p < e; ++p) {
        float x = w ? w[p] : 1.f;
        float4 v = *reinterpret_cast<const float4*>(rows + (size_t)idx[p] * dim + c);
        acc.x += v.x * x; acc.y += v.y * x; acc.z += v.z * x; acc.w += v.w * x;
      }
      if (combiner != 0) {
        if (wsum != 0.f) { acc.x /= scale; acc.y /= scale; acc.z /= scale; acc.w /= scale; }
        else acc = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<float4*>(o + c) = acc;
    }
  } else {
    for (int c = sub; c < dim; c += 16) {
      float acc = 0.f;
      for (int p = b; p < e; ++p) acc += rows[(size_t)idx[p] * dim + c] * (w ? w[p] : 1.f);
      if (combiner != 0) acc = (wsum != 0.f) ? acc / scale : 0.f;
      o[c] = acc;
    }
  }
}

// ------------------------------------ partition ------------------------------------------------
__device__ __forceinline__ int owner_of(i64 key, int num, int mode) {
  if (mode == 0) return (int)(key & 0x7fffffff) % num;
  if (mode == 1) { i64 m = key % num; return (int)(m < 0 ? m + num : m); }
  return (int)__umul64hi(fmix64((u64)key), (u64)num);
}

// pass 1: per-tile (256 ids) histogram -> hist[tile][shard]
__global__ __launch_bounds__(256) void part_hist_kernel(size_t n, const i64* __restrict__ d_n, const i64* __restrict__ keys,
                                                        const int* __restrict__ owner, int num, int mode, int* hist) {
  extern __shared__ int cnt[];
  if (d_n) n = (size_t)min((i64)n, max(*d_n, (i64)0));
  for (int s = threadIdx.x; s < num; s += 256) cnt[s] = 0;
  __syncthreads();
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    // caller-computed partitions outside [0, num) are DISCARDED, like the reference's GPU DynamicPartition
    // (T/dynamic_partition_op_test.py:221-283; its CPU kernel raises instead)
    const int o = owner ? owner[i] : owner_of(keys[i], num, mode);
    if ((unsigned)o < (unsigned)num) atomicAdd(&cnt[o], 1);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < num; s += 256) hist[(size_t)blockIdx.x * num + s] = cnt[s];
}

// pass 2 (single block): shard-major exclusive scan over (shard, tile); totals -> d_counts
__global__ __launch_bounds__(1024) void part_scan_kernel(int* hist, size_t tiles, int num, i64* d_counts) {
  __shared__ int sh[1024];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int s = 0; s < num; ++s) {
    long long start = carry;
    __syncthreads();
    for (size_t base = 0; base < tiles; base += 1024) {
      size_t tix = base + threadIdx.x;
      int v = tix < tiles ? hist[tix * num + s] : 0;
      sh[threadIdx.x] = v;
      __syncthreads();
      for (int o = 1; o < 1024; o <<= 1) {
        int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
      }
      int incl = sh[threadIdx.x];
      if (tix < tiles) hist[tix * num + s] = (int)(carry + incl - v);
      __syncthreads();
      if (threadIdx.x == 1023) carry += incl;
      __syncthreads();
    }
    if (threadIdx.x == 0) d_counts[s] = carry - start;
    __syncthreads();
  }
}

// pass 3: stable scatter.  Rank inside the tile = (same-owner ids in earlier waves) + (same-owner
// ids in lower lanes of this wave), found with a ballot per distinct owner present in the wave.
__global__ __launch_bounds__(256) void part_scatter_kernel(size_t n, const i64* __restrict__ d_n, const i64* __restrict__ keys,
                                                           const int* __restrict__ owner, int num, int mode,
                                                           const int* __restrict__ hist, i64* keys_out, int* perm_out) {
  extern __shared__ int wcnt[];  // [4][num]
  if (d_n) n = (size_t)min((i64)n, max(*d_n, (i64)0));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int s = threadIdx.x; s < 4 * num; s += 256) wcnt[s] = 0;
  __syncthreads();
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  bool ok = i < n;
  i64 key = (ok && keys) ? keys[i] : 0;
  int own = ok ? (owner ? owner[i] : owner_of(key, num, mode)) : -1;
  if ((unsigned)own >= (unsigned)num) own = -1;  // out of range: discarded
  ok = own >= 0;
  int rank = 0;
  u64 todo = __ballot(ok);
  while (todo) {
    int leader = __ffsll((unsigned long long)todo) - 1;
    int o = __shfl(own, leader);
    u64 m = __ballot(own == o);
    if (own == o) rank = __popcll(m & ((1ULL << lane) - 1));
    if (lane == leader) wcnt[w * num + o] = __popcll(m);
    todo &= ~m;
  }
  __syncthreads();
  if (ok) {
    int before = 0;
    for (int k = 0; k < w; ++k) before += wcnt[k * num + own];
    size_t pos = (size_t)hist[(size_t)blockIdx.x * num + own] + before + rank;
    if (keys_out) keys_out[pos] = key;
    perm_out[pos] = (int)i;
  }
}

}  // namespace

// tfra_unique for n <= 2^20 ids: three launches over two persistent, self-emptying sets (see unq2_insert_kernel)
static int unique_fast(tfra_workspace* ws, size_t n, const i64* ids, i64* unique_out, int32_t* idx_out, i64* d_num_unique, hipStream_t s) {
  const size_t tiles = (n + UNQ_TILE - 1) / UNQ_TILE;
  if (ws->unq_nmax < n) {
    if (ws->unq_buf) {
      if (hipStreamSynchronize(s) != hipSuccess || hipFree(ws->unq_buf) != hipSuccess) return set_error(TFRA_ERR_HIP, "unique: free failed");
      ws->unq_buf = nullptr; ws->unq_nmax = 0;
    }
    const size_t nmax = std::max<size_t>(n, 4096);
    size_t cap = 1024;
    while (cap < 2 * nmax) cap <<= 1;
    const size_t per = align_up((cap + 1) * 8) + 2 * align_up((cap + 1) * 4) + align_up(nmax * 4);
    const size_t bytes = 256 + 2 * per + align_up(nmax * 4) + align_up(((nmax + UNQ_TILE - 1) / UNQ_TILE) * 8);
    hipError_t e = hipMalloc(&ws->unq_buf, bytes);
    if (e != hipSuccess) { ws->unq_buf = nullptr; return set_error(e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP, "unique: hipMalloc failed"); }
    HIP_TRY(hipMemsetAsync(ws->unq_buf, 0, bytes, s));
    ws->unq_cap = cap; ws->unq_nmax = nmax; ws->unq_parity = 1; ws->unq_gen = 0;
    unsigned char* w = (unsigned char*)ws->unq_buf + 256;
    for (int p = 0; p < 2; ++p) {   // both sets start empty
      unq_fill_kernel<<<(unsigned)std::min<size_t>(2048, (cap + 256) / 256), 256, 0, s>>>((i64*)w, (int*)(w + align_up((cap + 1) * 8)), cap + 1);
      w += per;
    }
  }
  // The two sets, their parity and generation are STATE of the workspace: calls must reach them in one order.  Same stream:
  // stream order.  Another stream than the last call's: wait for that call (the Python layer keeps one workspace per stream,
  // so this only triggers for callers of the C ABI that share a workspace across streams).
  if (ws->unq_stream_set && ws->unq_stream != s) {
    if (!ws->unq_ev) HIP_TRY(hipEventCreateWithFlags(&ws->unq_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ws->unq_ev, ws->unq_stream));
    HIP_TRY(hipStreamWaitEvent(s, ws->unq_ev, 0));
  }
  ws->unq_stream = s; ws->unq_stream_set = true;
  const size_t cap = ws->unq_cap, nmax = ws->unq_nmax;
  const size_t per = align_up((cap + 1) * 8) + 2 * align_up((cap + 1) * 4) + align_up(nmax * 4);
  auto set_of = [&](unsigned p) {
    unsigned char* w = (unsigned char*)ws->unq_buf + 256 + (size_t)p * per;
    UnqSet u;
    u.hkeys = (i64*)w; w += align_up((cap + 1) * 8);
    u.hfirst = (int*)w; w += align_up((cap + 1) * 4);
    u.hrank = (int*)w; w += align_up((cap + 1) * 4);
    u.used = (unsigned*)w;
    u.nused = (unsigned*)ws->unq_buf + p;
    return u;
  };
  const unsigned p = ws->unq_parity ^ 1u;
  const UnqSet cur = set_of(p), old = set_of(p ^ 1u);
  int* slot_of = (int*)((unsigned char*)ws->unq_buf + 256 + 2 * per);
  unsigned long long* agg = (unsigned long long*)((unsigned char*)slot_of + align_up(nmax * 4));
  if (++ws->unq_gen == 0) ws->unq_gen = 1;   // (tile counts of an earlier call carry another generation)
  unq2_insert_kernel<<<(unsigned)((n + UNQ_INS - 1) / UNQ_INS), UNQ_INS, 0, s>>>(n, ids, cur, old, slot_of, cap);
  unq2_scatter_kernel<<<(unsigned)tiles, 256, 0, s>>>(n, ids, cur, old, slot_of, agg, ws->unq_gen, unique_out, d_num_unique);
  unq_idx_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(n, slot_of, cur.hrank, idx_out);
  HIP_TRY(hipGetLastError());
  ws->unq_parity = p;
  return TFRA_OK;
}

extern "C" {

int tfra_workspace_create(int device, tfra_workspace_t** out) {
  if (!out) return set_error(TFRA_ERR_INVALID, "null out");
  tfra_workspace* ws = new tfra_workspace();
  if (device < 0 && hipGetDevice(&device) != hipSuccess) { delete ws; return set_error(TFRA_ERR_HIP, "no HIP device"); }
  ws->device = device;
  *out = ws;
  return TFRA_OK;
}

int tfra_workspace_destroy(tfra_workspace_t* ws) {
  if (!ws) return TFRA_OK;
  (void)hipSetDevice(ws->device);
  if (ws->buf) { (void)hipDeviceSynchronize(); (void)hipFree(ws->buf); }
  if (ws->unq_buf) { (void)hipDeviceSynchronize(); (void)hipFree(ws->unq_buf); }
  if (ws->unq_ev) (void)hipEventDestroy(ws->unq_ev);
  if (ws->h_err) { (void)hipDeviceSynchronize(); (void)hipHostFree(ws->h_err); }
  tfra::destroy_workspace_plan(ws->plan);
  tfra::destroy_workspace_plan(ws->uplan);
  delete ws;
  return TFRA_OK;
}

int tfra_unique(tfra_workspace_t* ws, size_t n, const int64_t* ids, int64_t* unique_out, int32_t* idx_out,
                int64_t* d_num_unique, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || !d_num_unique) return set_error(TFRA_ERR_INVALID, "unique: null argument");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) HIP_TRY(hipSetDevice(ws->device)); }
  if (n == 0) { HIP_TRY(hipMemsetAsync(d_num_unique, 0, sizeof(int64_t), s)); return TFRA_OK; }
  if (!ids || !unique_out || !idx_out) return set_error(TFRA_ERR_INVALID, "unique: null buffer");
  if (n >= (1ULL << 30)) return set_error(TFRA_ERR_INVALID, "unique: more than 2^30 ids per call");
  if (n <= ((size_t)1 << 20)) return unique_fast(ws, n, (const i64*)ids, (i64*)unique_out, idx_out, (i64*)d_num_unique, s);
  size_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  size_t tiles = (n + UNQ_TILE - 1) / UNQ_TILE;
  size_t need = align_up((cap + 1) * 8) + 2 * align_up((cap + 1) * 4) + align_up(n * 4) + align_up(tiles * 4);
  int rc = ws->ensure(need, s);
  if (rc) return rc;
  Carver c{(unsigned char*)ws->buf};
  i64* hkeys = c.take<i64>(cap + 1);
  int* hfirst = c.take<int>(cap + 1);
  int* hrank = c.take<int>(cap + 1);
  int* slot_of = c.take<int>(n);
  int* bcnt = c.take<int>(tiles);
  unq_fill_kernel<<<(unsigned)std::min<size_t>(2048, (cap + 256) / 256), 256, 0, s>>>(hkeys, hfirst, cap + 1);
  unq_insert_kernel<<<(unsigned)((n + UNQ_INS - 1) / UNQ_INS), UNQ_INS, 0, s>>>(n, (const i64*)ids, hkeys, hfirst, slot_of, cap);
  unq_count_kernel<<<(unsigned)tiles, 256, 0, s>>>(n, hfirst, slot_of, bcnt);
  scan_counts_kernel<<<1, 1024, 0, s>>>(bcnt, tiles, (i64*)d_num_unique);
  unq_scatter_kernel<<<(unsigned)tiles, 256, 0, s>>>(n, (const i64*)ids, hfirst, slot_of, bcnt, (i64*)unique_out, hrank);
  unq_idx_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(n, slot_of, hrank, idx_out);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_keys_widen_i32(size_t n, const int32_t* keys_in, int64_t* keys_out, tfra_stream_t stream) {
  if (n == 0) return TFRA_OK;
  if (!keys_in || !keys_out) return set_error(TFRA_ERR_INVALID, "keys_widen_i32: null buffer");
  widen_keys_kernel<<<(unsigned)std::min<size_t>(4096, (n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n, keys_in, (i64*)keys_out);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_keys_narrow_i32(size_t n, const int64_t* keys_in, int32_t* keys_out, int64_t* d_overflow, tfra_stream_t stream) {
  if (d_overflow) HIP_TRY(hipMemsetAsync(d_overflow, 0, sizeof(int64_t), (hipStream_t)stream));
  if (n == 0) return TFRA_OK;
  if (!keys_in || !keys_out) return set_error(TFRA_ERR_INVALID, "keys_narrow_i32: null buffer");
  narrow_keys_kernel<<<(unsigned)std::min<size_t>(4096, (n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n, (const i64*)keys_in, keys_out, (i64*)d_overflow);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_gather_rows(size_t n, size_t row_bytes, const void* rows, const int32_t* idx, void* out, tfra_stream_t stream) {
  return move_rows<false>(n, row_bytes, rows, idx, out, (hipStream_t)stream);
}

int tfra_scatter_rows(size_t n, size_t row_bytes, const void* in, const int32_t* perm, void* out, tfra_stream_t stream) {
  return move_rows<true>(n, row_bytes, in, perm, out, (hipStream_t)stream);
}

int tfra_segment_sum(tfra_workspace_t* ws, size_t n, int dim, const float* in, const int32_t* idx,
                     const int64_t* d_num_segments, size_t max_segments, float* out, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || !d_num_segments || !out || dim <= 0) return set_error(TFRA_ERR_INVALID, "segment_sum: bad argument");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) HIP_TRY(hipSetDevice(ws->device)); }
  if (max_segments == 0) return TFRA_OK;
  if (n == 0) { HIP_TRY(hipMemsetAsync(out, 0, max_segments * (size_t)dim * sizeof(float), s)); return TFRA_OK; }
  if (!in || !idx) return set_error(TFRA_ERR_INVALID, "segment_sum: null buffer");
  if (n >= (1ULL << 31)) return set_error(TFRA_ERR_INVALID, "segment_sum: too many rows");
  unsigned bits = 1;
  while ((1ULL << bits) < max_segments && bits < 32) ++bits;
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tmp_bytes, (const int*)idx, (int*)nullptr, (const int*)nullptr,
                                    (int*)nullptr, n, 0u, bits, s));
  size_t need = 3 * align_up(n * 4) + align_up(2 * max_segments * 4) + align_up(tmp_bytes);
  int rc = ws->ensure(need, s);
  if (rc) return rc;
  Carver c{(unsigned char*)ws->buf};
  int* iota = c.take<int>(n);
  int* seg_sorted = c.take<int>(n);
  int* member = c.take<int>(n);
  int* seg_start = c.take<int>(2 * max_segments);
  void* tmp = c.take<unsigned char>(tmp_bytes);
  iota_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(iota, n);
  HIP_TRY(hipMemsetAsync(seg_start, 0, 2 * max_segments * sizeof(int), s));  // empty segments: start=end=0
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, (const int*)idx, seg_sorted, (const int*)iota, member, n, 0u, bits, s));
  seg_bounds_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(n, seg_sorted, seg_start, max_segments);
  dim3 grid((unsigned)((max_segments * 16 + 255) / 256));
  bool vec4 = (dim % 4 == 0) && (((uintptr_t)in | (uintptr_t)out) % 16 == 0);
  if (vec4) seg_sum_kernel<true><<<grid, 256, 0, s>>>(max_segments, (const i64*)d_num_segments, dim, in, member, seg_start, out);
  else seg_sum_kernel<false><<<grid, 256, 0, s>>>(max_segments, (const i64*)d_num_segments, dim, in, member, seg_start, out);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_sparse_segment_combine(tfra_workspace_t* ws, size_t nnz, int dim, const float* rows, const int32_t* idx,
                                const int64_t* seg, const float* weights, int combiner, size_t n_rows, float* out,
                                tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || !out || dim <= 0 || combiner < 0 || combiner > 2) return set_error(TFRA_ERR_INVALID, "segment_combine: bad argument");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) HIP_TRY(hipSetDevice(ws->device)); }
  if (n_rows == 0) return TFRA_OK;
  if (nnz && (!rows || !idx || !seg)) return set_error(TFRA_ERR_INVALID, "segment_combine: null buffer");
  if (nnz >= (1ULL << 31) || n_rows >= (1ULL << 30)) return set_error(TFRA_ERR_INVALID, "segment_combine: too large");
  int rc = ws->ensure(align_up(2 * n_rows * sizeof(int)), s);
  if (rc) return rc;
  int* se = (int*)ws->buf;
  HIP_TRY(hipMemsetAsync(se, 0, 2 * n_rows * sizeof(int), s));  // empty rows: start = end = 0
  if (nnz) seg64_bounds_kernel<<<(unsigned)((nnz + 255) / 256), 256, 0, s>>>(nnz, (const i64*)seg, se, n_rows);
  dim3 grid((unsigned)((n_rows * 16 + 255) / 256));
  bool vec4 = dim % 4 == 0 && (((uintptr_t)rows | (uintptr_t)out) % 16 == 0);
  if (vec4) seg_combine_kernel<true><<<grid, 256, 0, s>>>(n_rows, dim, rows, idx, weights, se, combiner, out);
  else seg_combine_kernel<false><<<grid, 256, 0, s>>>(n_rows, dim, rows, idx, weights, se, combiner, out);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_partition(tfra_workspace_t* ws, size_t n, const int64_t* d_n, const int64_t* keys, int num_shards, int mode,
                   int64_t* keys_out, int32_t* perm_out, int64_t* d_counts, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || !d_counts || num_shards <= 0 || num_shards > 2048 || mode < 0 || mode > 2)
    return set_error(TFRA_ERR_INVALID, "partition: bad argument (1 <= num_shards <= 2048, mode in 0..2)");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) HIP_TRY(hipSetDevice(ws->device)); }
  if (n == 0) { HIP_TRY(hipMemsetAsync(d_counts, 0, num_shards * sizeof(int64_t), s)); return TFRA_OK; }
  if (!keys || !keys_out || !perm_out) return set_error(TFRA_ERR_INVALID, "partition: null buffer");
  if (n >= (1ULL << 31)) return set_error(TFRA_ERR_INVALID, "partition: too many keys");
  size_t tiles = (n + 255) / 256;
  int rc = ws->ensure(align_up(tiles * num_shards * 4), s);
  if (rc) return rc;
  int* hist = (int*)ws->buf;
  part_hist_kernel<<<(unsigned)tiles, 256, num_shards * sizeof(int), s>>>(n, (const i64*)d_n, (const i64*)keys, nullptr, num_shards, mode, hist);
  part_scan_kernel<<<1, 1024, 0, s>>>(hist, tiles, num_shards, (i64*)d_counts);
  part_scatter_kernel<<<(unsigned)tiles, 256, 4 * num_shards * sizeof(int), s>>>(n, (const i64*)d_n, (const i64*)keys, nullptr, num_shards,
                                                                                  mode, hist, (i64*)keys_out, perm_out);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_partition_by_owner(tfra_workspace_t* ws, size_t n, const int32_t* owner, int num_shards, int32_t* perm_out,
                            int64_t* d_counts, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || !d_counts || num_shards <= 0 || num_shards > 2048)
    return set_error(TFRA_ERR_INVALID, "partition_by_owner: bad argument (1 <= num_shards <= 2048)");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) HIP_TRY(hipSetDevice(ws->device)); }
  if (n == 0) { HIP_TRY(hipMemsetAsync(d_counts, 0, num_shards * sizeof(int64_t), s)); return TFRA_OK; }
  if (!owner || !perm_out) return set_error(TFRA_ERR_INVALID, "partition_by_owner: null buffer");
  if (n >= (1ULL << 31)) return set_error(TFRA_ERR_INVALID, "partition_by_owner: too many keys");
  size_t tiles = (n + 255) / 256;
  int rc = ws->ensure(align_up(tiles * num_shards * 4), s);
  if (rc) return rc;
  int* hist = (int*)ws->buf;
  part_hist_kernel<<<(unsigned)tiles, 256, num_shards * sizeof(int), s>>>(n, nullptr, nullptr, owner, num_shards, 0, hist);
  part_scan_kernel<<<1, 1024, 0, s>>>(hist, tiles, num_shards, (i64*)d_counts);
  part_scatter_kernel<<<(unsigned)tiles, 256, 4 * num_shards * sizeof(int), s>>>(n, nullptr, nullptr, owner, num_shards, 0, hist,
                                                                                  nullptr, perm_out);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_select_lowest(tfra_workspace_t* ws, size_t n, const int64_t* keys, const void* status, int status_dtype,
                       size_t k, int64_t* keys_out, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || (status_dtype != TFRA_I32 && status_dtype != TFRA_I64) || k > n)
    return set_error(TFRA_ERR_INVALID, "select_lowest: bad argument (status int32/int64, k <= n)");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) HIP_TRY(hipSetDevice(ws->device)); }
  if (k == 0) return TFRA_OK;
  if (!keys || !status || !keys_out) return set_error(TFRA_ERR_INVALID, "select_lowest: null buffer");
  if (n >= (1ULL << 31)) return set_error(TFRA_ERR_INVALID, "select_lowest: too many keys");
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tmp_bytes, (const i64*)nullptr, (i64*)nullptr, (const i64*)nullptr,
                                    (i64*)nullptr, n, 0u, 64u, s));
  int rc = ws->ensure(3 * align_up(n * 8) + align_up(tmp_bytes), s);
  if (rc) return rc;
  Carver c{(unsigned char*)ws->buf};
  i64* wide = c.take<i64>(n);
  i64* st_sorted = c.take<i64>(n);
  i64* k_sorted = c.take<i64>(n);
  void* tmp = c.take<unsigned char>(tmp_bytes);
  const i64* st = (const i64*)status;
  if (status_dtype == TFRA_I32) {
    widen_i32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((const int*)status, wide, n);
    st = wide;
  }
  // stable LSD radix sort, ascending, signed: equal statuses keep their input order
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, st, st_sorted, (const i64*)keys, k_sorted, n, 0u, 64u, s));
  HIP_TRY(hipMemcpyAsync(keys_out, k_sorted, k * sizeof(i64), hipMemcpyDeviceToDevice, s));
  return TFRA_OK;
}

}  // extern "C"
