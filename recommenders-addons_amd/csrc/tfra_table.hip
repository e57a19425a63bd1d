// MI355X (gfx950) dynamic-embedding table: kernels + C ABI (include/tfra_mi355x.h).
//
// Replaces, behind TFRA's own op surface, what `gpu::TableWrapper` gets from HierarchicalKV
// (R/kernels/lookup_impl/lookup_table_op_hkv.h:515-756) with results defined by the
// reference's CPU table (R/kernels/cuckoo_hashtable_op.cc, lib/cuckoo/cuckoohash_map.hh).
// Work mapping everywhere: 16 lanes per key (one 128-B bucket line per probe, one 16-B
// granule per lane per row step), 4 keys per wave64, U independent keys in flight per group.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/tfra_mi355x.h"
#include "tfra_device.h"
#include "tfra_host.h"

using namespace tfra;
typedef tfra::AuxInitPod AuxInit;  // elem_bytes = sizeof(V); pattern[f] = aux_init[f] as V, replicated to 32 bits

// =============================== kernels ====================================================

// ---- find (+ fused default fill, + exists) -------------------------------------------------
// (the work of a wave — 16 keys — is find_wave, tfra_device.h; PF1: both home-bucket lines of a key in flight together)
template <int G, int U, bool WT = (G == 16), bool PF1 = false>
__global__ __launch_bounds__(256) void find_kernel(TableView v, size_t n, const i64* __restrict__ keys,
                                                   unsigned char* __restrict__ out,
                                                   uint8_t* __restrict__ exists,
                                                   const unsigned char* __restrict__ defaults,
                                                   int full, unsigned field_off, const long long* __restrict__ d_n = nullptr) {
  if (d_n) {   // tfra_table_find_n: the key count lives on the device (n = the buffers' length)
    const long long dn = *d_n;
    n = dn < 0 ? 0 : min(n, (size_t)dn);
  }
  find_wave<G, U, WT, PF1>(v, n, keys, out, exists, defaults, full, field_off, (blockIdx.x * blockDim.x + threadIdx.x) >> 6);
}

// ---- aux-field initialisation for a newly claimed row --------------------------------------

// WT: write-through stores (eviction path: the row must be in memory before the key is published, see publish_key)
template <bool WT = false>
__device__ __forceinline__ void init_aux_fields(const TableView& v, const AuxInit& ai, i64 row,
                                                int sub, unsigned skip_field) {
  unsigned char* r = row_ptr(v, row);
  for (unsigned f = 0; f < v.n_fields; ++f) {
    if (f == skip_field) continue;
    unsigned pat = f == 0 ? 0u : ai.pattern[(f - 1) & 3];
    unsigned char* p = r + f * v.field_bytes;
    if ((v.field_bytes & 3) == 0) {
      for (unsigned off = sub * 4; off < v.field_bytes; off += 64) {
        if (WT) __hip_atomic_store(reinterpret_cast<unsigned*>(p + off), pat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *reinterpret_cast<unsigned*>(p + off) = pat;
      }
    } else {
      for (unsigned off = sub; off < v.field_bytes; off += 16) {
        const unsigned char b = (unsigned char)(pat >> (8 * (off % ai.elem_bytes)));
        if (WT) __hip_atomic_store(p + off, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p[off] = b;
      }
    }
  }
}

// ---- insert_or_assign, unique-keys fast path (single pass) ---------------------------------
template <int G, int U>
__global__ __launch_bounds__(256) void insert_unique_kernel(TableView v, size_t n,
                                                            const i64* __restrict__ keys,
                                                            const unsigned char* __restrict__ vals,
                                                            const u64* __restrict__ scores,
                                                            unsigned field, AuxInit ai, int strategy,
                                                            u64 epoch, int bounded, uint8_t* __restrict__ deferred) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, grp = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int KPW = 4 * U;
  const size_t base = wave * KPW;
  if (base >= n) return;
  const size_t last = n - 1;
  i64 kreg = keys[min(base + (size_t)(lane & (KPW - 1)), last)];
  int fresh = 0, failed = 0;
  i64 key[U], k0[U], k1[U];
  u64 h[U], b0[U];
  const bool pf1 = bounded > 1;  // table near capacity: both home buckets' lines in flight together
#pragma unroll
  for (int u = 0; u < U; ++u) {  // U first probes in flight (unconditional, tail clamped)
    key[u] = shfl_i64(kreg, u * 4 + grp);
    b0[u] = bucket0(key[u], v.nb, h[u]);
    k0[u] = load_key_coherent(key_line(v, b0[u]) + sub);
    k1[u] = load_key_coherent(key_line(v, pf1 ? bucket1(h[u], b0[u], v.nb) : b0[u]) + sub);  // (same line again: an L2 hit)
  }
  keep_live(k0[0], k0[1], k0[2], k0[3]);
  keep_live(k1[0], k1[1], k1[2], k1[3]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = u * 4 + grp;
    size_t i = base + j;
    if (i < n) {
      bool is_new;
      i64 row = locate_or_claim_from(v, key[u], h[u], b0[u], k0[u], sub, gshift, is_new, bounded, pf1 ? &k1[u] : nullptr);
      if (deferred && sub == 0) deferred[i] = row == NEED_EVICT;
      if (row >= 0) {
        copy_bytes16<G>(row_ptr(v, row) + field * v.field_bytes,
                        vals + i * (size_t)v.field_bytes, v.field_bytes, sub);
        if (is_new && v.n_fields > 1) init_aux_fields(v, ai, row, sub, field);
        update_score(v, row, is_new, strategy, scores ? scores[i] : 1, epoch, sub);
        fresh += (is_new && sub == 0);
      } else if (row != NEED_EVICT) {
        failed += (sub == 0);
      }
    }
  }
  // one size update per wave
  for (int o = 32; o > 0; o >>= 1) { fresh += __shfl_xor(fresh, o); failed += __shfl_xor(failed, o); }
  if (lane == 0) {
    if (fresh) size_add(v, wave, fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

// ---- phase 2 of a bounded-table upsert: keys that found neither themselves nor an empty slot
// replace the minimum-score entry of their two home buckets (runs after phase 1 has completed, so
// no row is being written by an assign while it is evicted).
template <int G>
__global__ __launch_bounds__(256) void insert_evict_kernel(TableView v, size_t n, const i64* __restrict__ keys,
                                                           const unsigned char* __restrict__ vals,
                                                           const u64* __restrict__ scores, unsigned field, AuxInit ai,
                                                           int strategy, u64 epoch, const uint8_t* __restrict__ deferred) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const size_t i = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  int fresh = 0, failed = 0;
  if (i < n && deferred[i]) {
    const i64 key = keys[i];
    const u64 in_score = scores ? scores[i] : 1;
    const bool lru_like = strategy == TFRA_EVICT_LRU || strategy == TFRA_EVICT_EPOCHLRU;
    u64 word = 0;
    bool claimed_empty;
    i64 row = evict_and_lock(v, key, strategy == TFRA_EVICT_EPOCHLFU ? ((epoch << 32) | in_score) : in_score, lru_like, sub,
                             gshift, &word, claimed_empty);
    if (row >= 0) {
      copy_bytes16_wt<G>(row_ptr(v, row) + field * v.field_bytes, vals + i * (size_t)v.field_bytes,
                         v.field_bytes, sub);
      if (v.n_fields > 1) init_aux_fields<true>(v, ai, row, sub, field);
      if (sub == 0) store_wt8(score_word(v, word), 0);  // the slot starts a new life: scores count from zero
      update_score<true>(v, row, true, strategy, in_score, epoch, sub);
      publish_key(v, word, key, sub);
      fresh = (claimed_empty && sub == 0);
    } else if (row == -3) {
      failed = (sub == 0);
    }  // -1: not admitted (its score is below every resident score): silently dropped, like HKV
  }
  for (int o = 32; o > 0; o >>= 1) { fresh += __shfl_xor(fresh, o); failed += __shfl_xor(failed, o); }
  if (lane == 0) {
    if (fresh) size_add(v, i >> 2, fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

// ---- insert_or_assign with duplicates: pass 1 locate/claim + elect the LAST index ----------
template <int U>
__global__ __launch_bounds__(256) void insert_locate_kernel(TableView v, size_t n,
                                                            const i64* __restrict__ keys,
                                                            i64* __restrict__ slot_of, unsigned field,
                                                            AuxInit ai) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, grp = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int KPW = 4 * U;
  const size_t base = wave * KPW;
  if (base >= n) return;
  i64 kreg = (lane < KPW && base + lane < n) ? keys[base + lane] : 0;
  int fresh = 0, failed = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = u * 4 + grp;
    size_t i = base + j;
    i64 key = shfl_i64(kreg, j);
    if (i < n) {
      bool is_new;
      i64 row = locate_or_claim(v, key, sub, gshift, is_new);
      if (row >= 0) {
        if (is_new && v.n_fields > 1) init_aux_fields(v, ai, row, sub, field);
        if (sub == 0) {
          // hot keys (Zipf) repeat thousands of times: only occurrences that can still raise the
          // maximum pay for the contended atomic
          if (__hip_atomic_load(&v.winner[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)i) atomicMax(&v.winner[row], (int)i);
          slot_of[i] = row | (is_new ? (i64)1 << 62 : 0);
        }
        fresh += (is_new && sub == 0);
      } else {
        if (sub == 0) slot_of[i] = -1;
        failed += (sub == 0);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) { fresh += __shfl_xor(fresh, o); failed += __shfl_xor(failed, o); }
  if (lane == 0) {
    if (fresh) size_add(v, wave, fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

// pass 2: only the elected occurrence writes the row (sequential "last writer wins" of
// LaunchTensorsInsert with one thread), then re-arms the election word.
template <int G>
__global__ __launch_bounds__(256) void insert_write_kernel(TableView v, size_t n,
                                                           const unsigned char* __restrict__ vals,
                                                           const u64* __restrict__ scores,
                                                           const i64* __restrict__ slot_of,
                                                           unsigned field, int strategy, u64 epoch) {
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const size_t i = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  if (i >= n) return;
  (void)lane;
  i64 so = slot_of[i];
  if (so < 0) return;
  bool is_new = (so >> 62) & 1;
  i64 row = so & (((i64)1 << 62) - 1);
  if (v.winner[row] != (int)i) {
    // LFU counts every upsert, also the overwritten duplicates
    if (strategy == TFRA_EVICT_LFU) update_score(v, row, false, strategy, scores ? scores[i] : 1, epoch, sub);
    return;
  }
  copy_bytes16<G>(row_ptr(v, row) + field * v.field_bytes,
                  vals + i * (size_t)v.field_bytes, v.field_bytes, sub);
  update_score(v, row, is_new, strategy, scores ? scores[i] : 1, epoch, sub);
}

__global__ void rearm_winner_kernel(TableView v, size_t n, const i64* __restrict__ slot_of) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  i64 so = slot_of[i];
  if (so >= 0) v.winner[so & (((i64)1 << 62) - 1)] = -1;
}

// ---- typed accumulate: row[j] += delta[j], one add per element (ValueArray::operator+=) ----

template <int DT>
__device__ __forceinline__ void row_add(unsigned char* row, const unsigned char* delta, unsigned dim, int sub) {
  for (unsigned j = sub; j < dim; j += 16) {
    if (DT == TFRA_F32) reinterpret_cast<float*>(row)[j] += reinterpret_cast<const float*>(delta)[j];
    else if (DT == TFRA_F64) reinterpret_cast<double*>(row)[j] += reinterpret_cast<const double*>(delta)[j];
    else if (DT == TFRA_I8) reinterpret_cast<signed char*>(row)[j] = (signed char)(reinterpret_cast<signed char*>(row)[j] + reinterpret_cast<const signed char*>(delta)[j]);
    else if (DT == TFRA_I32) reinterpret_cast<unsigned*>(row)[j] += reinterpret_cast<const unsigned*>(delta)[j];
    else if (DT == TFRA_I64) reinterpret_cast<u64*>(row)[j] += reinterpret_cast<const u64*>(delta)[j];
    else if (DT == TFRA_F16) {
      _Float16 a = reinterpret_cast<_Float16*>(row)[j], b = reinterpret_cast<const _Float16*>(delta)[j];
      reinterpret_cast<_Float16*>(row)[j] = (_Float16)((float)a + (float)b);
    } else {
      unsigned short* r = reinterpret_cast<unsigned short*>(row);
      r[j] = f32_to_bf16(bf16_to_f32(r[j]) + bf16_to_f32(reinterpret_cast<const unsigned short*>(delta)[j]));
    }
  }
}

// ---- accum_or_assign.  ROUND >= 0: duplicate-safe mode, processes only the occurrence that is
// currently first-in-line for its key (election word), see host loop. -------------------------
// one (key, values-or-delta, exists) triple: absent & !exists -> insert, present & exists -> row += delta, else nothing
template <int DT, int G>
__device__ __forceinline__ void accum_one(const TableView& v, size_t i, const i64* __restrict__ keys,
                                          const unsigned char* __restrict__ vod, const uint8_t* __restrict__ exists,
                                          const u64* __restrict__ scores, unsigned dim, const AuxInit& ai, int strategy, u64 epoch,
                                          uint8_t* __restrict__ deferred, int bounded_mode, int sub, int gshift, int& fresh,
                                          int& failed) {
  const i64 key = keys[i];
  const bool ex = exists[i] != 0;
  const unsigned char* src = vod + i * (size_t)v.field_bytes;
  if (!ex) {
    bool is_new;
    i64 row;
    if (deferred) {  // bounded table at max_capacity: keys without a free slot evict in phase 2
      u64 h;
      const u64 b0 = bucket0(key, v.nb, h);
      const i64 k0 = load_key_coherent(key_line(v, b0) + sub);
      row = locate_or_claim_from(v, key, h, b0, k0, sub, gshift, is_new, bounded_mode);
      if (sub == 0) deferred[i] = row == NEED_EVICT;
    } else {
      row = locate_or_claim(v, key, sub, gshift, is_new);
    }
    if (row < 0) failed += (sub == 0 && row != NEED_EVICT);
    else if (is_new) {
      copy_bytes16<G>(row_ptr(v, row), src, v.field_bytes, sub);
      if (v.n_fields > 1) init_aux_fields(v, ai, row, sub, 0);
      update_score(v, row, true, strategy, scores ? scores[i] : 1, epoch, sub);
      fresh += (sub == 0);
    }  // present & !exists: dropped
  } else {
    i64 row = probe_find<true>(v, key, sub, gshift);
    if (row >= 0) {
      row_add<DT>(row_ptr(v, row), src, dim, sub);
      update_score(v, row, false, strategy, scores ? scores[i] : 1, epoch, sub);
    }  // absent & exists: dropped
    if (deferred && sub == 0) deferred[i] = 0;
  }
}

template <int DT, int G>
__global__ __launch_bounds__(256) void accum_kernel(TableView v, size_t n, const i64* __restrict__ keys,
                                                    const unsigned char* __restrict__ vod,
                                                    const uint8_t* __restrict__ exists,
                                                    const u64* __restrict__ scores, unsigned dim,
                                                    AuxInit ai, int strategy, u64 epoch,
                                                    uint8_t* __restrict__ deferred, int bounded_mode) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const size_t g = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  const size_t wave = g >> 2;
  int fresh = 0, failed = 0;
  if (g < n) accum_one<DT, G>(v, g, keys, vod, exists, scores, dim, ai, strategy, epoch, deferred, bounded_mode, sub, gshift, fresh, failed);
  for (int o = 32; o > 0; o >>= 1) { fresh += __shfl_xor(fresh, o); failed += __shfl_xor(failed, o); }
  if (lane == 0) {
    if (fresh) size_add(v, wave, fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

// Keys that repeat within one call: the reference applies the triples one after the other in index order
// (LaunchTensorsAccum on one thread; accumrase_fn, cuckoohash_map.hh:619-633), and the outcome of an occurrence depends
// on the ones before it (an insert makes the key present for the next).  The (key, index) pairs arrive sorted by key
// (stable radix sort: indices ascend within a key); the group of a key's FIRST sorted position walks the key's
// occurrences in index order, every step through memory (a key that repeats thousands of times is one long chain —
// exact, not fast: TFRA de-duplicates before accum, PY/dynamic_embedding_variable.py:1377-1378).
template <int DT, int G>
__global__ __launch_bounds__(256) void accum_segments_kernel(TableView v, size_t n, const i64* __restrict__ keys,
                                                             const unsigned char* __restrict__ vod,
                                                             const uint8_t* __restrict__ exists, const u64* __restrict__ scores,
                                                             unsigned dim, AuxInit ai, int strategy, u64 epoch,
                                                             const u64* __restrict__ sorted_keys, const unsigned* __restrict__ sorted_idx) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const size_t p = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  int fresh = 0, failed = 0;
  if (p < n) {
    const u64 k = sorted_keys[p];
    if (p == 0 || sorted_keys[p - 1] != k) {
      for (size_t q = p; q < n && sorted_keys[q] == k; ++q) {
        accum_one<DT, G>(v, (size_t)sorted_idx[q], keys, vod, exists, scores, dim, ai, strategy, epoch, nullptr, 0, sub, gshift, fresh, failed);
        __threadfence_block();   // the next occurrence reads what this one wrote (other lanes of the group, same row)
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) { fresh += __shfl_xor(fresh, o); failed += __shfl_xor(failed, o); }
  if (lane == 0) {
    if (fresh) size_add(v, p >> 2, fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

// ---- erase ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void erase_kernel(TableView v, size_t n, const i64* __restrict__ keys) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const size_t g = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  int gone = 0;
  if (g < n) {
    i64 key = keys[g];
    if (is_reserved_key(key)) {
      if (sub == 0) gone = atomicExch(&v.reserved_present[reserved_index(key)], 0u) != 0;
    } else {
      i64 row = probe_find<true>(v, key, sub, gshift);
      if (row >= 0 && sub == 0) {
        u64 wb; unsigned ws; split_row((u64)row, wb, ws);
        u64 w = wb * 16 + ws;
        // CAS so that duplicate keys in one call decrement the size once
        gone = atomicCAS((u64*)key_word(v, w), (u64)key, (u64)EMPTY_KEY) == (u64)key;
        if (gone && has_scores(v)) *score_word(v, w) = 0;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) gone += __shfl_xor(gone, o);
  if (lane == 0 && gone) size_add(v, g >> 2, -(long long)gone);
}

// ---- clear / fill ---------------------------------------------------------------------------
__global__ void clear_kernel(TableView v, int reset_counters) {
  size_t total = v.nb * 16;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
    *key_word(v, w) = ((w & 15) == 15) ? 0 : EMPTY_KEY;
    if (has_scores(v)) *score_word(v, w) = 0;
  }
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (!reset_counters) return;
  if (t < SIZE_SHARDS) v.size_shards[t * SIZE_SHARD_STRIDE] = 0;
  if (t < NUM_RESERVED) v.reserved_present[t] = 0;
  if (t == 0) { *v.err_count = 0; *const_cast<unsigned*>(v.dense_flag) = 0; }
}

__global__ void iota_u32_kernel(unsigned* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (unsigned)i;
}

__global__ void fill_i32_kernel(int* p, size_t n, int val) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = val;
}

__global__ void size_kernel(TableView v, i64* out) {
  __shared__ long long part[SIZE_SHARDS];
  part[threadIdx.x] = (long long)v.size_shards[threadIdx.x * SIZE_SHARD_STRIDE];
  __syncthreads();
  for (int s = SIZE_SHARDS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = part[0];
}

// size + the device-side density flag of a bounded table at max_capacity (TableView::dense_flag): monotone until clear
__global__ void density_kernel(TableView v, i64* out, unsigned* dense_flag, i64 threshold) {
  __shared__ long long part[SIZE_SHARDS];
  part[threadIdx.x] = (long long)v.size_shards[threadIdx.x * SIZE_SHARD_STRIDE];
  __syncthreads();
  for (int s = SIZE_SHARDS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *out = part[0];
    if (part[0] > threshold) *dense_flag = 1u;
  }
}

// introspection (tests, tools): counts of empty / locked / live key slots and of flagged buckets
__global__ void slot_census_kernel(TableView v, u64* out) {
  u64 e = 0, l = 0, live = 0, f0 = 0, f1 = 0;
  const size_t total = v.nb * 16;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
    const i64 k = *key_word(v, w);
    if ((w & 15) == 15) { f0 += ((u64)k & META_OVF0) != 0; f1 += ((u64)k & META_OVF1) != 0; }
    else if (k == EMPTY_KEY) ++e;
    else if (k == LOCKED_KEY) ++l;
    else ++live;
  }
  if (e) atomicAdd(out + 0, e);
  if (l) atomicAdd(out + 1, l);
  if (live) atomicAdd(out + 2, live);
  if (f0) atomicAdd(out + 3, f0);
  if (f1) atomicAdd(out + 4, f1);
}

// ---- export_batch: slots [offset, offset+n) -> compact (key,row,score) at *counter ----------
// One block = 64 buckets; one returned atomic per block (not per wave) reserves the output run.
template <int G>
__global__ __launch_bounds__(256) void export_kernel(TableView v, u64 first_bucket, u64 last_bucket,
                                                     u64 lo, u64 hi, u64* counter, i64* __restrict__ keys_out,
                                                     unsigned char* __restrict__ vals_out,
                                                     u64* __restrict__ scores_out) {
  __shared__ unsigned cnt[64];
  __shared__ u64 base_s;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const int grp_in_block = threadIdx.x >> 4;  // 0..15
  i64 k[4];
  unsigned live[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    u64 b = first_bucket + (u64)blockIdx.x * 64 + it * 16 + grp_in_block;
    bool ok = b < last_bucket;
    k[it] = ok ? key_line(v, b)[sub] : EMPTY_KEY;
    u64 slot = b * SLOTS + sub;
    bool l = ok && sub < SLOTS && k[it] != EMPTY_KEY && k[it] != LOCKED_KEY && slot >= lo && slot < hi;
    live[it] = (unsigned)(__ballot(l) >> gshift) & 0x7fffu;
    if (sub == 0) cnt[it * 16 + grp_in_block] = __popc(live[it]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int i = 0; i < 64; ++i) { unsigned c = cnt[i]; cnt[i] = run; run += c; }
    base_s = run ? atomicAdd(counter, (u64)run) : 0;
  }
  __syncthreads();
  const u64 base = base_s;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    u64 b = first_bucket + (u64)blockIdx.x * 64 + it * 16 + grp_in_block;
    u64 pos0 = base + cnt[it * 16 + grp_in_block];
    unsigned m = live[it];
    if (m & (1u << sub)) {
      u64 pos = pos0 + __popc(m & ((1u << sub) - 1));
      keys_out[pos] = k[it];
      if (scores_out) scores_out[pos] = has_scores(v) ? score_line(v, b)[sub] : 0;
    }
    if (vals_out) {
      unsigned r = 0;
      while (m) {
        int s = __ffs(m) - 1;
        m &= m - 1;
        copy_bytes16<G>(vals_out + (pos0 + r) * (size_t)v.field_bytes,
                        row_at(v, b, (unsigned)s), v.field_bytes, sub);
        ++r;
      }
    }
  }
}

// the two side rows (keys INT64_MIN, INT64_MIN+1) are slots nb*15 and nb*15+1
__global__ void export_reserved_kernel(TableView v, u64 lo, u64 hi, u64* counter, i64* keys_out,
                                       unsigned char* vals_out, u64* scores_out) {
  for (int r = 0; r < NUM_RESERVED; ++r) {
    u64 slot = v.nb * SLOTS + r;
    if (slot < lo || slot >= hi || !v.reserved_present[r]) continue;
    __shared__ u64 pos_s;
    if (threadIdx.x == 0) pos_s = atomicAdd(counter, 1ULL);
    __syncthreads();
    u64 pos = pos_s;
    if (threadIdx.x == 0) { keys_out[pos] = EMPTY_KEY + r; if (scores_out) scores_out[pos] = ~0ULL; }
    if (vals_out)
      for (unsigned off = threadIdx.x; off < v.field_bytes; off += blockDim.x)
        vals_out[pos * (size_t)v.field_bytes + off] = row_ptr(v, (i64)slot)[off];
    __syncthreads();
  }
}

// ---- rehash (growth): move every live row of `o` into `v` -----------------------------------
__global__ __launch_bounds__(256) void rehash_kernel(TableView o, TableView v) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const u64 b = (((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  int failed = 0;
  if (b < o.nb) {
    i64 k = key_line(o, b)[sub];
    unsigned m = (unsigned)(__ballot(sub < SLOTS && k != EMPTY_KEY && k != LOCKED_KEY) >> gshift) & 0x7fffu;
    while (m) {
      int s = __ffs(m) - 1;
      m &= m - 1;
      i64 key = shfl_i64(k, gshift + s);
      bool is_new;
      i64 row = locate_or_claim(v, key, sub, gshift, is_new);
      if (row < 0) { failed += (sub == 0); continue; }
      copy_bytes16<16>(row_ptr(v, row), row_at(o, b, (unsigned)s), o.row_stride, sub);
      if (has_scores(v) && has_scores(o) && sub == 0)
        score_line(v, (u64)row / SLOTS)[(u64)row % SLOTS] = score_line(o, b)[s];
    }
  }
  if (b == 0) {  // side rows
    for (int r = 0; r < NUM_RESERVED; ++r)
      copy_bytes16<16>(row_ptr(v, (i64)(v.nb * SLOTS + r)), row_ptr(o, (i64)(o.nb * SLOTS + r)), o.row_stride, sub);
  }
  for (int off = 32; off > 0; off >>= 1) failed += __shfl_xor(failed, off);
  if (lane == 0 && failed) atomicAdd(v.err_count, (unsigned)failed);
}

// ---- growth in place (storage mapped into a reserved virtual range): nb -> F * nb buckets, F a power of two ---------
// b0 = mulhi(h_hi, nb) and b1 = mulhi(fmix32(..), nb) are RANGE reductions: with F * nb buckets a key's home bucket is one
// of the F children [F*b, F*b + F) of its old home b.  So every old bucket splits into its children independently of all
// others, top-down (children of [lo, hi) lie in [F*lo, F*hi), beyond every bucket still to be split), without a second
// copy of the table.  A key goes to its new b0 if that is a child of the bucket it sat in, else to its new b1 if that
// is, else (it sat in a chain bucket, or its b1 was the b0+1 substitute: ~nb^-1 of the keys) onto a spill list that is
// re-inserted the general way afterwards.  Children start without overflow flags; split_flags_kernel then sets exactly the
// ones the new placement needs.  Keys are packed from slot 0.
struct SpillBuf {
  i64* keys; u64* scores; unsigned char* rows; unsigned long long* count; u64 cap;
};

__device__ __forceinline__ unsigned split_child(const TableView& nv, i64 k, u64 b, unsigned shift) {
  u64 h;
  const u64 b0 = bucket0(k, nv.nb, h);
  if ((b0 >> shift) == b) return (unsigned)(b0 - (b << shift));
  const u64 b1 = bucket1(h, b0, nv.nb);
  if ((b1 >> shift) == b) return (unsigned)(b1 - (b << shift));
  return 0xffu;
}

__device__ __forceinline__ int nth_set_bit(unsigned m, int n) {   // position of the n-th (0-based) set bit, -1 if fewer
  for (int i = 0; i < n; ++i) m &= m - 1;
  return m ? __ffs(m) - 1 : -1;
}

__global__ __launch_bounds__(256) void split_count_kernel(TableView o, TableView nv, unsigned shift, unsigned long long* count) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  int spilled = 0;
  for (u64 b = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4; b < o.nb; b += ((u64)gridDim.x * blockDim.x) >> 4) {
    const i64 k = key_line(o, b)[sub];
    const bool live = sub < SLOTS && k != EMPTY_KEY && k != LOCKED_KEY;
    spilled += live && split_child(nv, k, b, shift) == 0xffu;
  }
  for (int off = 32; off > 0; off >>= 1) spilled += __shfl_xor(spilled, off);
  if (lane == 0 && spilled) atomicAdd(count, (unsigned long long)spilled);
}

// buckets [lo, hi) of the old numbering; `self`: lo == 0 and bucket 0's first child is bucket 0 itself — the keys that
// stay keep their slots, only the others move
__global__ __launch_bounds__(256) void split_kernel(TableView o, TableView nv, unsigned shift, u64 lo, u64 hi, SpillBuf sp) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const u64 b = lo + ((((u64)blockIdx.x * blockDim.x + threadIdx.x)) >> 4);
  if (b >= hi) return;
  const unsigned F = 1u << shift;
  const bool scored = has_scores(o);
  const i64 k = key_line(o, b)[sub];                       // lane 15: the meta word
  const u64 sc = scored ? score_line(o, b)[sub] : 0;
  const bool live = sub < SLOTS && k != EMPTY_KEY && k != LOCKED_KEY;
  const unsigned child = live ? split_child(nv, k, b, shift) : 0xfeu;
  // spill list first: the rows are still where they were
  unsigned ms = (unsigned)(__ballot(child == 0xffu) >> gshift) & 0x7fffu;
  while (ms) {
    const int s = __ffs(ms) - 1;
    ms &= ms - 1;
    unsigned long long idx = 0;
    if (sub == 0) idx = atomicAdd(sp.count, 1ULL);
    idx = (unsigned long long)shfl_i64((i64)idx, gshift);
    const i64 ks = shfl_i64(k, gshift + s);
    const u64 ssc = (u64)shfl_i64((i64)sc, gshift + s);
    if (idx < sp.cap) {
      if (sub == 0) { sp.keys[idx] = ks; if (scored) sp.scores[idx] = ssc; }
      copy_bytes16<16>(sp.rows + idx * (u64)o.row_stride, row_at(o, b, (unsigned)s), o.row_stride, sub);
    } else if (sub == 0) {
      atomicAdd(nv.err_count, 1u);   // cannot happen: the list was sized by split_count_kernel
    }
  }
  for (unsigned c = 0; c < F; ++c) {
    const unsigned mc = (unsigned)(__ballot(child == c) >> gshift) & 0x7fffu;
    const u64 nb_c = (b << shift) + c;
    if (nb_c == b) {   // bucket 0 onto itself: stay in place
      const i64 kout = sub == 15 ? 0 : ((mc >> sub) & 1u ? k : EMPTY_KEY);
      key_line(nv, nb_c)[sub] = kout;
      continue;
    }
    const int cnt = __popc(mc);
    const int src = sub < SLOTS ? nth_set_bit(mc, sub) : -1;
    const i64 ksrc = shfl_i64(k, gshift + (src < 0 ? 0 : src));
    const u64 ssrc = (u64)shfl_i64((i64)sc, gshift + (src < 0 ? 0 : src));
    key_line(nv, nb_c)[sub] = sub == 15 ? 0 : (src < 0 ? EMPTY_KEY : ksrc);   // flags: split_flags_kernel
    if (scored) score_line(nv, nb_c)[sub] = src < 0 ? 0 : ssrc;
    for (int j = 0; j < cnt; ++j) {
      const int sj = nth_set_bit(mc, j);
      copy_bytes16<16>(row_at(nv, nb_c, (unsigned)j), row_at(o, b, (unsigned)sj), o.row_stride, sub);
    }
  }
}

// After the split every key sits in its new b0 or b1 and the children carry no flags: set exactly the ones searches need —
// OVF0 on the b0 of every key that lives in its b1 (inheriting the parents' flags instead would hand every child the
// overflow history of a bucket that was 92 % full: measured, lookups of absent keys 40x slower on the grown table).
// The spill list is re-inserted afterwards by locate_or_claim, which sets its own flags.
__global__ __launch_bounds__(256) void split_flags_kernel(TableView nv) {
  const int lane = threadIdx.x & 63, sub = lane & 15;
  for (u64 b = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4; b < nv.nb; b += ((u64)gridDim.x * blockDim.x) >> 4) {
    const i64 k = key_line(nv, b)[sub];
    if (sub < SLOTS && k != EMPTY_KEY && k != LOCKED_KEY) {
      u64 h;
      const u64 b0 = bucket0(k, nv.nb, h);
      if (b0 != b) atomicOr(reinterpret_cast<unsigned long long*>(key_line(nv, b0) + 15), (unsigned long long)META_OVF0);
    }
  }
}

__global__ __launch_bounds__(256) void spill_reinsert_kernel(TableView v, SpillBuf sp, u64 n) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  int failed = 0;
  if (i < n) {
    bool is_new;
    const i64 row = locate_or_claim(v, sp.keys[i], sub, gshift, is_new);
    if (row < 0) {
      failed = sub == 0;
    } else {
      copy_bytes16<16>(row_ptr(v, row), sp.rows + i * (u64)v.row_stride, v.row_stride, sub);
      if (has_scores(v) && sub == 0) score_line(v, (u64)row / SLOTS)[(u64)row % SLOTS] = sp.scores[i];
    }
  }
  for (int off = 32; off > 0; off >>= 1) failed += __shfl_xor(failed, off);
  if (lane == 0 && failed) atomicAdd(v.err_count, (unsigned)failed);
}

// =============================== host side ==================================================

namespace tfra {
thread_local std::string g_last_error;
int set_error(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
}  // namespace tfra

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return set_error(_e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP,                \
                       std::string(#expr) + ": " + hipGetErrorString(_e));                    \
  } while (0)

static size_t dtype_size(int dt) {
  switch (dt) {
    case TFRA_F32: case TFRA_I32: return 4;
    case TFRA_F16: case TFRA_BF16: return 2;
    case TFRA_I8: return 1;
    case TFRA_I64: case TFRA_F64: return 8;
    default: return 0;
  }
}

static unsigned short host_f2h(float f) {
  _Float16 h = (_Float16)f;
  unsigned short u;
  memcpy(&u, &h, 2);
  return u;
}
static unsigned short host_f2b(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

static AuxInit make_aux_init(const tfra_table_opts& o) {
  AuxInit ai;
  ai.elem_bytes = (unsigned)dtype_size(o.value_dtype);
  for (int f = 0; f < 4; ++f) {
    float x = o.aux_init[f];
    unsigned pat = 0;
    switch (o.value_dtype) {
      case TFRA_F32: memcpy(&pat, &x, 4); break;
      case TFRA_I32: { int v = (int)x; memcpy(&pat, &v, 4); } break;
      case TFRA_F16: { unsigned short h = host_f2h(x); pat = h | ((unsigned)h << 16); } break;
      case TFRA_BF16: { unsigned short h = host_f2b(x); pat = h | ((unsigned)h << 16); } break;
      case TFRA_I8: { unsigned char c = (unsigned char)(signed char)x; pat = c * 0x01010101u; } break;
      default: pat = 0; break;  // 8-byte types: aux fields start at 0
    }
    ai.pattern[f] = pat;
  }
  return ai;
}

static int granule_of(size_t bytes, const void* a, const void* b) {
  size_t x = bytes | (size_t)(uintptr_t)a | (size_t)(uintptr_t)b | 16;
  int g = (int)(x & (~x + 1));
  return g > 16 ? 16 : g;
}

namespace tfra {

void* Table::dalloc(size_t bytes, hipStream_t s) {
  if (alloc.alloc) return alloc.alloc(alloc.user, 0, bytes, (tfra_stream_t)s);
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  return p;
}
void Table::dfree(void* p, hipStream_t s) {
  if (!p) return;
  if (alloc.alloc) alloc.free(alloc.user, 0, p, (tfra_stream_t)s);
  else (void)hipFree(p);
}

static inline unsigned hdr_bytes(const tfra_table_opts& o) { return o.strategy >= 0 ? 256u : 128u; }

// Tables of TFRA_VMM_THRESHOLD_MB (default 4096) or more live in a reserved virtual range with physical memory mapped
// chunk by chunk, so that they can grow in place; smaller ones (and every table of a caller-supplied allocator) are one
// plain allocation and grow by copying, which costs them nothing.  A negative threshold turns the mapping off.
static long long vmm_threshold_bytes() {   // read at every (rare) storage allocation: tests switch it per table
  const char* e = getenv("TFRA_VMM_THRESHOLD_MB");
  const long long mb = e ? atoll(e) : 4096;
  return mb < 0 ? -1LL : mb * (1LL << 20);
}
// Chunks of ONE size per table (a power of two between 2 MiB and 1 GiB, about the table's first size): on ROCm 7.2
// hipMemSetAccess rejects some mappings whose size differs from their neighbours' (2 MiB then 4 MiB: invalid argument;
// scripts/mb/vmm_probe2.hip), equal-sized chunks were accepted in every trial (200 x 2 MiB ... 8 x 4 GiB).
constexpr size_t VMM_ALIGN = (size_t)2 << 20, VMM_CHUNK_MAX = (size_t)1 << 30;

static int vmm_map_more(Storage* st, size_t need, int device) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t sz = st->chunk_bytes;
  need = (need + sz - 1) / sz * sz;
  if (need > st->va_bytes) return set_error(TFRA_ERR_OOM, "table storage: beyond the reserved address range");
  while (st->mapped < need) {
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, sz, &prop, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_error(TFRA_ERR_OOM, std::string("table storage: hipMemCreate: ") + hipGetErrorString(e)); }
    e = hipMemMap(st->base + st->mapped, sz, 0, h, 0);
    if (e == hipSuccess) e = hipMemSetAccess(st->base + st->mapped, sz, &acc, 1);
    if (e != hipSuccess) {
      (void)hipMemUnmap(st->base + st->mapped, sz); (void)hipMemRelease(h); (void)hipGetLastError();
      return set_error(TFRA_ERR_OOM, std::string("table storage: hipMemMap: ") + hipGetErrorString(e));
    }
    st->chunks.emplace_back(h, sz);
    st->mapped += sz;
  }
  return TFRA_OK;
}

void Table::free_storage(Storage& st, hipStream_t s) {
  if (st.vmm) {
    size_t off = 0;
    for (auto& c : st.chunks) { (void)hipMemUnmap(st.base + off, c.second); (void)hipMemRelease(c.first); off += c.second; }
    if (st.base) (void)hipMemAddressFree(st.base, st.va_bytes);
  } else {
    dfree(st.base, s);
  }
  st = Storage();
}

int Table::alloc_storage(u64 nb, Storage* st, hipStream_t s) {
  *st = Storage();
  st->nb = nb;
  const size_t bstride = (size_t)hdr_bytes(opts) + (size_t)SLOTS * row_stride;
  if (nb >= (1ULL << 32) - 1 || bstride >= (1ULL << 32))   // 32-bit bucket arithmetic on the device (tfra_device.h)
    return set_error(TFRA_ERR_INVALID, "table storage: more than 2^32 - 2 buckets or a bucket block of 4 GiB");
  const size_t bytes = nb * bstride + (size_t)NUM_RESERVED * row_stride;
  const long long thr = vmm_threshold_bytes();
  if (!alloc.alloc && thr >= 0 && bytes >= (size_t)thr) {
    // address range: what the table can ever need — max_capacity, else the whole device
    size_t total = 0, free_b = 0;
    (void)hipMemGetInfo(&free_b, &total);
    size_t want = total ? total : bytes;
    if (opts.max_capacity) want = std::min(want, (size_t)(std::max<u64>(2, opts.max_capacity / SLOTS)) * bstride + (size_t)NUM_RESERVED * row_stride);
    want = std::max(want, bytes);
    size_t chunk = VMM_ALIGN;
    while (chunk < bytes && chunk < VMM_CHUNK_MAX) chunk <<= 1;
    want = (want + chunk - 1) / chunk * chunk + chunk;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, want, VMM_ALIGN, nullptr, 0) == hipSuccess) {
      st->base = (unsigned char*)va; st->vmm = true; st->va_bytes = want; st->chunk_bytes = chunk;
      if (vmm_map_more(st, bytes, device) == TFRA_OK) return TFRA_OK;
      free_storage(*st, s);
      st->nb = nb;
    }
    (void)hipGetLastError();
    g_last_error.clear();   // fall back to one plain allocation
  }
  st->base = (unsigned char*)dalloc(bytes, s);
  if (!st->base) {
    *st = Storage();
    return set_error(TFRA_ERR_OOM, "table storage allocation failed (" + std::to_string(nb) + " buckets x " +
                                       std::to_string(bstride) + " B)");
  }
  return TFRA_OK;
}

TableView Table::view_of(const Storage& st) const {
  TableView v;
  v.base = st.base; v.nb = st.nb;
  v.hdr = hdr_bytes(opts);
  v.bucket_stride = (u64)v.hdr + (u64)SLOTS * row_stride;
  v.field_bytes = field_bytes; v.row_stride = row_stride; v.n_fields = 1 + opts.aux_fields;
  v.reserved_present = reserved_present; v.size_shards = size_shards; v.winner = winner;
  v.err_count = err_count;
  v.dense_flag = d_dense;
  return v;
}

// serialise against work queued on another stream (the reference blocks on a per-table mutex and
// a stream sync per op, R/kernels/hkv_hashtable_op_gpu.cu.cc:192-213; here: event chaining).
int Table::enter(hipStream_t s) {
  // hipSetDevice costs tens of microseconds on ROCm 7.2 — more than the find kernel itself — so
  // it is only issued when the calling thread is on another device.
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device) {
    if (hipSetDevice(device) != hipSuccess) return set_error(TFRA_ERR_HIP, "hipSetDevice failed");
  }
  if (has_last && s != last_stream && !capture_safe) {
    HIP_TRY(hipEventRecord(chain_event, last_stream));
    HIP_TRY(hipStreamWaitEvent(s, chain_event, 0));
  }
  last_stream = s;
  has_last = true;
  return TFRA_OK;
}

int Table::read_size(hipStream_t s, size_t* out) {
  size_kernel<<<1, SIZE_SHARDS, 0, s>>>(view_of(cur), d_scalar);
  HIP_TRY(hipMemcpyAsync(h_scalar, d_scalar, sizeof(i64), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  i64 v = *h_scalar;
  *out = v < 0 ? 0 : (size_t)v;
  size_ub = *out;
  return TFRA_OK;
}

int Table::check_errors(hipStream_t s) {
  unsigned e = 0;
  HIP_TRY(hipMemcpyAsync(h_scalar, err_count, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(&e, h_scalar, sizeof(unsigned));
  if (e) {
    HIP_TRY(hipMemsetAsync(err_count, 0, sizeof(unsigned), s));
    return set_error(TFRA_ERR_FULL, std::to_string(e) + " keys could not be placed: table full at max_capacity (or a write-back plan overflowed)");
  }
  return TFRA_OK;
}

// bucket-owner tags of the ownership-based write-backs: 4 B per bucket, zeroed; rebuilt after a rehash
unsigned* Table::ensure_own_tags(hipStream_t s) {
  if (no_owner_tags) return nullptr;
  if (own_tags && own_tags_nb == cur.nb) return own_tags;
  if (own_tags) { (void)hipStreamSynchronize(s); dfree(own_tags, s); own_tags = nullptr; }
  own_tags = (unsigned*)dalloc(cur.nb * sizeof(unsigned), s);
  if (!own_tags) { g_last_error.clear(); return nullptr; }
  if (hipMemsetAsync(own_tags, 0, cur.nb * sizeof(unsigned), s) != hipSuccess) { dfree(own_tags, s); own_tags = nullptr; return nullptr; }
  own_tags_nb = cur.nb;
  own_gen = 0;
  return own_tags;
}

int Table::ensure_winner(hipStream_t s) {
  size_t need = cur.nb * SLOTS + NUM_RESERVED;
  if (winner && winner_len >= need) return TFRA_OK;
  dfree(winner, s);
  winner = (int*)dalloc(need * sizeof(int), s);
  if (!winner) return set_error(TFRA_ERR_OOM, "winner scratch allocation failed");
  winner_len = need;
  fill_i32_kernel<<<2048, 256, 0, s>>>(winner, need, -1);
  return TFRA_OK;
}

// Hkv flavour at max_capacity (the table cannot grow): *out = the per-key phase-2 flag buffer of a
// two-phase (place, then evict) write; nullptr while the table can still grow or is unbounded.
int Table::bounded_flags(size_t n, hipStream_t s, uint8_t** out) {
  *out = nullptr;
  if (!at_max_capacity()) return TFRA_OK;
  if (evict_flags_cap < n) {
    if (evict_flags) { HIP_TRY(hipStreamSynchronize(s)); dfree(evict_flags, s); }
    evict_flags = (uint8_t*)dalloc(n, s);
    if (!evict_flags) { evict_flags_cap = 0; return set_error(TFRA_ERR_OOM, "eviction flag buffer allocation failed"); }
    evict_flags_cap = n;
  }
  *out = evict_flags;
  return TFRA_OK;
}

int Table::ensure_scratch(size_t bytes, hipStream_t s) {
  if (scratch_bytes >= bytes) return TFRA_OK;
  apply_P = 0;  // the armed cursor area of tfra_table_apply_sparse does not survive a reallocation
  if (scratch) { HIP_TRY(hipStreamSynchronize(s)); dfree(scratch, s); }
  size_t want = std::max(bytes, scratch_bytes * 2);
  scratch = dalloc(want, s);
  if (!scratch) { scratch_bytes = 0; return set_error(TFRA_ERR_OOM, "scratch allocation failed"); }
  scratch_bytes = want;
  return TFRA_OK;
}

// grow to at least min_nb buckets: new arrays, rehash kernel, free the old ones.
int Table::grow(u64 min_nb, hipStream_t s) {
  if (min_nb <= cur.nb) return TFRA_OK;
  if (cur.vmm) {
    int rc = grow_in_place(min_nb, s);
    if (rc != TFRA_ERR_UNSUPPORTED) return rc;
    g_last_error.clear();
  }
  Storage nw;
  int rc = alloc_storage(lattice_nb(min_nb), &nw, s);
  if (rc) return rc;
  Storage old = cur;
  int* old_winner = winner;
  winner = nullptr; winner_len = 0;  // sized per storage; rebuilt lazily
  TableView nv = view_of(nw);
  // fresh key lines; keep counters (rehash moves, it does not insert)
  clear_kernel<<<2048, 256, 0, s>>>(nv, 0);
  u64 groups = old.nb;
  rehash_kernel<<<(unsigned)((groups * 16 + 255) / 256), 256, 0, s>>>(view_of(old), nv);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));  // old arrays are freed below
  free_storage(old, s); dfree(old_winner, s);
  cur = nw;
  n_rehash++;
  return TFRA_OK;
}

// Bucket counts of a bounded table sit on the lattice max_nb / 2^j, so that doubling in place ends exactly at
// max_capacity (a table that reached, say, 60 % of it by copying could neither double nor, past a third of the HBM, copy).
// a bounded (Hkv) table that cannot double any more: eviction takes over
bool Table::at_max_capacity() const {
  return opts.strategy >= 0 && opts.max_capacity && cur.nb * 2 > std::max<u64>(2, opts.max_capacity / SLOTS);
}

u64 Table::lattice_nb(u64 min_nb) const {
  if (!opts.max_capacity) return min_nb;
  const u64 max_nb = std::max<u64>(2, opts.max_capacity / SLOTS);
  if (min_nb >= max_nb) return max_nb;
  unsigned j = 0;
  while ((max_nb >> (j + 1)) >= min_nb && (max_nb >> (j + 1)) >= 2) ++j;
  return max_nb >> j;
}

// nb -> F * nb buckets inside the table's address range (see split_kernel), F the smallest power of two reaching min_nb
// that max_capacity allows.  Peak memory = the new size (+ the spill list); the copying path needs old + new.
int Table::grow_in_place(u64 min_nb, hipStream_t s) {
  const u64 max_nb = opts.max_capacity ? std::max<u64>(2, opts.max_capacity / SLOTS) : ((1ULL << 32) - 2);
  unsigned shift = 1;
  while ((cur.nb << shift) < min_nb && shift < 8) ++shift;
  while (shift > 0 && (cur.nb << shift) > max_nb) --shift;
  if (shift == 0) return set_error(TFRA_ERR_UNSUPPORTED, "in-place growth: no power-of-two factor fits max_capacity");
  const u64 nbn = cur.nb << shift;
  const size_t bstride = (size_t)hdr_bytes(opts) + (size_t)SLOTS * row_stride;
  const size_t side = (size_t)NUM_RESERVED * row_stride, new_bytes = nbn * bstride + side;
  if (nbn >= (1ULL << 32) - 1 || new_bytes > cur.va_bytes) return set_error(TFRA_ERR_UNSUPPORTED, "in-place growth: beyond the address range");
  const size_t mapped_before = cur.mapped;
  const size_t chunks_before = cur.chunks.size();
  int rc = vmm_map_more(&cur, new_bytes, device);
  if (rc) {
    // out of memory part-way: give back the chunks mapped so far (up to nearly the table's own size of HBM would otherwise sit
    // behind the table unused, exactly when memory is short); the caller keeps running denser
    while (cur.chunks.size() > chunks_before) {
      const auto c = cur.chunks.back();
      cur.mapped -= c.second;
      (void)hipMemUnmap(cur.base + cur.mapped, c.second);
      (void)hipMemRelease(c.first);
      cur.chunks.pop_back();
    }
    cur.mapped = mapped_before;
    (void)hipGetLastError();
    return rc;
  }
  Storage nw = cur;    // same range, new bucket count
  nw.nb = nbn;
  const TableView ov = view_of(cur), nv = view_of(nw);
  // how many keys cannot stay with their bucket's children
  unsigned long long* d_count = reinterpret_cast<unsigned long long*>(d_scalar);
  HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
  split_count_kernel<<<2048, 256, 0, s>>>(ov, nv, shift, d_count);
  HIP_TRY(hipMemcpyAsync(h_scalar, d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const u64 nspill = (u64)*reinterpret_cast<unsigned long long*>(h_scalar);
  SpillBuf sp{nullptr, nullptr, nullptr, d_count, nspill};
  if (nspill) {
    sp.keys = (i64*)dalloc(nspill * sizeof(i64), s);
    sp.scores = (u64*)dalloc(nspill * sizeof(u64), s);
    sp.rows = (unsigned char*)dalloc(nspill * (size_t)row_stride, s);
    if (!sp.keys || !sp.scores || !sp.rows) {
      dfree(sp.keys, s); dfree(sp.scores, s); dfree(sp.rows, s);
      return set_error(TFRA_ERR_OOM, "in-place growth: spill list allocation failed");
    }
  }
  HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
  // the side rows move behind the new last bucket (that place is beyond every old bucket)
  HIP_TRY(hipMemcpyAsync(cur.base + nbn * bstride, cur.base + cur.nb * bstride, side, hipMemcpyDeviceToDevice, s));
  const u64 F = 1ULL << shift;
  for (u64 hi = cur.nb; hi > 0;) {
    const u64 lo = hi == 1 ? 0 : (hi + F - 1) / F;   // children of [lo, hi) start at F*lo >= hi
    const u64 groups = hi - lo;
    split_kernel<<<(unsigned)((groups * 16 + 255) / 256), 256, 0, s>>>(ov, nv, shift, lo, hi, sp);
    hi = lo;
  }
  split_flags_kernel<<<4096, 256, 0, s>>>(nv);
  if (nspill) spill_reinsert_kernel<<<(unsigned)((nspill * 16 + 255) / 256), 256, 0, s>>>(nv, sp, nspill);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  dfree(sp.keys, s); dfree(sp.scores, s); dfree(sp.rows, s);
  dfree(winner, s); winner = nullptr; winner_len = 0;   // sized per storage; rebuilt lazily
  cur = nw;
  n_rehash++; n_split++;
  return TFRA_OK;
}

// A table that cannot grow any more (Hkv flavour at max_capacity): every 16th insert-type call starts an
// asynchronous size read (size kernel + 8-B D2H + event, no host wait); `dense` = the last completed read saw
// more than 60 % of the slots in use.  From then on new keys are placed in their two home buckets only (below
// that the chance that both are full is < 1e-3 and a 4-bucket walk keeps every key: the reference never evicts at
// load factor 0.5), so the OVF1 flags stop spreading while they are still rare, and find / insert put BOTH home
// buckets' lines in flight at once.
int Table::poll_density(size_t n, hipStream_t s) {
  if (dense) return TFRA_OK;  // monotone until clear()
  if (size_pending && hipEventQuery(size_event) == hipSuccess) {
    i64 v = *h_size;
    dense = (double)(v < 0 ? 0 : v) > 0.6 * (double)(cur.nb * SLOTS);
    size_pending = false;
    if (dense) return TFRA_OK;
  }
  // The kernels decide from the DEVICE flag, refreshed in stream order before every insert-type call of the
  // transition phase (a 1-block kernel): a caller that queues hundreds of calls ahead of the GPU (a bulk load)
  // would otherwise fill the table to capacity in 4-bucket-walk mode before the host ever sees a size, and every
  // later miss would walk the flags that left behind (measured: find 47 us instead of 17 us on a 10^9-slot table).
  // (a call that could itself carry the table past the mark — a bulk load in one call — runs dense from its start)
  density_kernel<<<1, SIZE_SHARDS, 0, s>>>(view_of(cur), d_scalar + 1, d_dense, (i64)(0.6 * (double)(cur.nb * SLOTS)) - (i64)n);
  if (!size_pending) {
    HIP_TRY(hipMemcpyAsync(h_size, d_scalar + 1, sizeof(i64), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(size_event, s));
    size_pending = true;
  }
  return TFRA_OK;
}

// Called before an op that may insert up to n new keys.  Growth policy:
//   * `size_ub` is a host-side UPPER BOUND of the live-key count (every insert-type call adds its
//     n; exact after a size read).  While size_ub + n <= max_load_factor*slots nothing happens.
//   * Past that soft threshold a steady-state training loop (upserts of resident keys) must not
//     pay a host sync per call: an ASYNC size read (size kernel + 8-B D2H into pinned memory +
//     event) refreshes the bound; the call proceeds optimistically while the bound stays under
//     the hard threshold (92 % of the slots, where first-fit probing still terminates quickly).
//   * When the bound passes the hard threshold, or a completed read shows that the TRUE size is past
//     the soft one, we synchronise and grow if the true size needs it.  An out-of-memory during growth
//     is not an error unless the keys cannot fit at all.
int Table::prepare_insert(size_t n, hipStream_t s) {
  if (capture_safe) return TFRA_OK;  // capacity is the caller's responsibility while capturing
  const double slots = (double)(cur.nb * SLOTS);
  const double soft = opts.max_load_factor * slots, hard = 0.92 * slots;
  if ((double)(size_ub + n) <= soft) { size_ub += n; return TFRA_OK; }
  // at max_capacity (eviction takes over) or after a failed growth there is nothing to decide
  // (bounded tables sit on the lattice max_nb / 2^j: the last doubling lands on max_nb, give or take the rounding)
  const bool can_grow = !growth_blocked && (!opts.max_capacity || cur.nb * 2 <= std::max<u64>(2, opts.max_capacity / SLOTS));
  if (!can_grow) {
    // (the bound stays AT the soft threshold from here on: it is not advanced on this path, and a later, smaller call must
    // not fall back under the threshold and skip the density poll — a table filled by a few big calls and then used with
    // small ones never learned that it was dense)
    size_ub = std::max(size_ub, (size_t)soft);
    return poll_density(n, s);
  }
  bool truly_past_soft = false;   // a completed read saw more live keys than max_load_factor allows: grow now, not at 92 %
  if (size_pending && hipEventQuery(size_event) == hipSuccess) {
    i64 v = *h_size;
    size_ub = (v < 0 ? 0 : (size_t)v) + n_since_read;
    size_pending = false;
    if ((double)(size_ub + n) <= soft) { size_ub += n; return TFRA_OK; }
    truly_past_soft = (double)(v < 0 ? 0 : v) > soft;
  }
  if ((double)(size_ub + n) <= hard && !truly_past_soft) {
    if (!size_pending) {
      size_kernel<<<1, SIZE_SHARDS, 0, s>>>(view_of(cur), d_scalar + 1);
      HIP_TRY(hipMemcpyAsync(h_size, d_scalar + 1, sizeof(i64), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipEventRecord(size_event, s));
      size_pending = true;
      n_since_read = 0;
    }
    size_ub += n;
    n_since_read += n;
    return TFRA_OK;
  }
  size_t sz;
  int rc = read_size(s, &sz);
  if (rc) return rc;
  size_pending = false;
  if ((double)(sz + n) > soft && can_grow) {
    u64 max_nb = opts.max_capacity ? std::max<u64>(2, opts.max_capacity / SLOTS) : ~0ULL;
    u64 need = (u64)((double)(sz + n) / opts.max_load_factor / SLOTS) + 1;
    u64 tries[2] = {std::min(std::max(need, cur.nb * 2), max_nb), std::min(std::max(need, cur.nb + cur.nb / 4), max_nb)};
    rc = TFRA_ERR_OOM;
    // (a table in a mapped address range grows by doublings only: the 1.25x retry would ask for the same doubling again)
    for (int i = 0; i < (cur.vmm ? 1 : 2) && rc == TFRA_ERR_OOM; ++i) rc = grow(tries[i], s);
    if (rc == TFRA_ERR_OOM) {
      if ((double)(sz + n) > 0.98 * slots) return rc;  // cannot fit: report the allocation failure
      growth_blocked = true;                            // keep running denser instead
      g_last_error.clear();
    } else if (rc) {
      return rc;
    }
  }
  size_ub = sz + n;
  return TFRA_OK;
}

}  // namespace tfra

static int find_impl(Table* t, hipStream_t s, int field, size_t n, const int64_t* keys, void* values,
                     uint8_t* exists, const void* defaults, int full, const int64_t* d_n = nullptr) {
  if (n == 0) return TFRA_OK;
  if (!keys || !values || !defaults) return set_error(TFRA_ERR_INVALID, "find: null buffer");
  if (n >= (1ULL << 32)) return set_error(TFRA_ERR_INVALID, "find: more than 2^32-1 keys per call");
  if (field < 0 || field > t->opts.aux_fields) return set_error(TFRA_ERR_INVALID, "find: bad field");
  TableView v = t->view_of(t->cur);
  unsigned fo = field * t->field_bytes;
  int g = granule_of(t->field_bytes, values, defaults);
  if (fo) g = std::min(g, granule_of(fo, nullptr, nullptr));
  constexpr int U = 4;
  size_t waves = (n + 4 * U - 1) / (4 * U);
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  const i64* k = (const i64*)keys;
  unsigned char* o = (unsigned char*)values;
  const unsigned char* d = (const unsigned char*)defaults;
  const long long* dn = (const long long*)d_n;
  switch (g) {
    case 16:
      if (t->dense) find_kernel<16, U, true, true><<<grid, block, 0, s>>>(v, n, k, o, exists, d, full, fo, dn);
      else find_kernel<16, U><<<grid, block, 0, s>>>(v, n, k, o, exists, d, full, fo, dn);
      break;
    case 8: find_kernel<8, U><<<grid, block, 0, s>>>(v, n, k, o, exists, d, full, fo, dn); break;
    case 4: find_kernel<4, U><<<grid, block, 0, s>>>(v, n, k, o, exists, d, full, fo, dn); break;
    case 2: find_kernel<2, U><<<grid, block, 0, s>>>(v, n, k, o, exists, d, full, fo, dn); break;
    default: find_kernel<1, U><<<grid, block, 0, s>>>(v, n, k, o, exists, d, full, fo, dn); break;
  }
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

static int insert_impl(Table* t, hipStream_t s, int field, size_t n, const int64_t* keys, const void* values,
                       const uint64_t* scores, uint32_t flags) {
  if (n == 0) return TFRA_OK;
  if (!keys || !values) return set_error(TFRA_ERR_INVALID, "insert: null buffer");
  if (field < 0 || field > t->opts.aux_fields) return set_error(TFRA_ERR_INVALID, "insert: bad field");
  if (n >= (1ULL << 31)) return set_error(TFRA_ERR_INVALID, "insert: more than 2^31-1 keys per call");
  int rc = t->prepare_insert(n, s);
  if (rc) return rc;
  // TableWrapper::upsert epoch stepping (lookup_table_op_hkv.h:528-536)
  u64 epoch = t->global_epoch;
  unsigned fo = field * t->field_bytes;
  int g = granule_of(t->field_bytes, values, nullptr);
  if (fo) g = std::min(g, granule_of(fo, nullptr, nullptr));
  const i64* k = (const i64*)keys;
  const unsigned char* vals = (const unsigned char*)values;
  const u64* sc = (const u64*)scores;
  int strat = t->opts.strategy;
  constexpr int U = 4;
  size_t waves = (n + 4 * U - 1) / (4 * U);
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  // Hkv flavour at max_capacity: the table cannot grow, full home buckets evict by score (2 phases)
  const bool bounded = t->at_max_capacity() && field == 0;
  if (bounded && !(flags & TFRA_FLAG_UNIQUE_KEYS))
    return set_error(TFRA_ERR_UNSUPPORTED, "insert: a bounded (Hkv) table at max_capacity needs TFRA_FLAG_UNIQUE_KEYS "
                                           "(HKV's unique-keys contract) so that eviction is well defined");
  if ((flags & TFRA_FLAG_UNIQUE_KEYS) && field == 0) {
    // the single pass with bucket ownership whenever the batch is small for the table: no locks, no CAS
    bool taken = false;
    rc = own_upsert_unique(t, s, n, k, vals, sc, &taken);
    if (rc) return rc;
    if (taken) {
      if (strat == TFRA_EVICT_EPOCHLRU || strat == TFRA_EVICT_EPOCHLFU) {
        t->curr_step += 1;
        if (t->opts.step_per_epoch > 0 && t->curr_step > t->opts.step_per_epoch) { t->global_epoch += 1; t->curr_step = 1; }
      }
      return TFRA_OK;
    }
  }
  if (flags & TFRA_FLAG_UNIQUE_KEYS) {
    uint8_t* deferred = nullptr;
    if (bounded) {
      rc = t->ensure_scratch(n, s);
      if (rc) return rc;
      t->apply_P = 0;
      deferred = (uint8_t*)t->scratch;
    }
    TableView v = t->view_of(t->cur);
    const int bd = bounded ? (t->dense ? 2 : 1) : 0;
    switch (g) {
      case 16: insert_unique_kernel<16, U><<<grid, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, bd, deferred); break;
      case 8: insert_unique_kernel<8, U><<<grid, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, bd, deferred); break;
      case 4: insert_unique_kernel<4, U><<<grid, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, bd, deferred); break;
      case 2: insert_unique_kernel<2, U><<<grid, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, bd, deferred); break;
      default: insert_unique_kernel<1, U><<<grid, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, bd, deferred); break;
    }
    if (bounded) {
      dim3 grid2((unsigned)((n * 16 + 255) / 256));
      switch (g) {
        case 16: insert_evict_kernel<16><<<grid2, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, deferred); break;
        case 8: insert_evict_kernel<8><<<grid2, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, deferred); break;
        case 4: insert_evict_kernel<4><<<grid2, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, deferred); break;
        case 2: insert_evict_kernel<2><<<grid2, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, deferred); break;
        default: insert_evict_kernel<1><<<grid2, block, 0, s>>>(v, n, k, vals, sc, field, t->aux, strat, epoch, deferred); break;
      }
    }
  } else {
    rc = t->ensure_winner(s);
    if (rc) return rc;
    rc = t->ensure_scratch(n * sizeof(i64), s);
    if (rc) return rc;
    t->apply_P = 0;  // scratch head is overwritten below
    TableView v = t->view_of(t->cur);
    i64* slot_of = (i64*)t->scratch;
    insert_locate_kernel<U><<<grid, block, 0, s>>>(v, n, k, slot_of, field, t->aux);
    dim3 grid2((unsigned)((n * 16 + 255) / 256));
    switch (g) {
      case 16: insert_write_kernel<16><<<grid2, block, 0, s>>>(v, n, vals, sc, slot_of, field, strat, epoch); break;
      case 8: insert_write_kernel<8><<<grid2, block, 0, s>>>(v, n, vals, sc, slot_of, field, strat, epoch); break;
      case 4: insert_write_kernel<4><<<grid2, block, 0, s>>>(v, n, vals, sc, slot_of, field, strat, epoch); break;
      case 2: insert_write_kernel<2><<<grid2, block, 0, s>>>(v, n, vals, sc, slot_of, field, strat, epoch); break;
      default: insert_write_kernel<1><<<grid2, block, 0, s>>>(v, n, vals, sc, slot_of, field, strat, epoch); break;
    }
    rearm_winner_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(v, n, slot_of);
  }
  HIP_TRY(hipGetLastError());
  if (strat == TFRA_EVICT_EPOCHLRU || strat == TFRA_EVICT_EPOCHLFU) {
    t->curr_step += 1;
    if (t->opts.step_per_epoch > 0 && t->curr_step > t->opts.step_per_epoch) { t->global_epoch += 1; t->curr_step = 1; }
  }
  return TFRA_OK;
}

template <int DT>
static void launch_accum(int g, dim3 grid, hipStream_t s, TableView v, size_t n, const i64* k, const unsigned char* vod,
                         const uint8_t* ex, const u64* sc, unsigned dim, AuxInit ai, int strat, u64 epoch,
                         const u64* sorted_keys, const unsigned* sorted_idx, uint8_t* deferred, int bmode) {
  dim3 block(256);
  if (sorted_keys) {
    switch (g) {
      case 16: accum_segments_kernel<DT, 16><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx); break;
      case 8: accum_segments_kernel<DT, 8><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx); break;
      case 4: accum_segments_kernel<DT, 4><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx); break;
      case 2: accum_segments_kernel<DT, 2><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx); break;
      default: accum_segments_kernel<DT, 1><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx); break;
    }
    return;
  }
  switch (g) {
    case 16: accum_kernel<DT, 16><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, deferred, bmode); break;
    case 8: accum_kernel<DT, 8><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, deferred, bmode); break;
    case 4: accum_kernel<DT, 4><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, deferred, bmode); break;
    case 2: accum_kernel<DT, 2><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, deferred, bmode); break;
    default: accum_kernel<DT, 1><<<grid, block, 0, s>>>(v, n, k, vod, ex, sc, dim, ai, strat, epoch, deferred, bmode); break;
  }
}

static void launch_accum_dt(int dt, int g, dim3 grid, hipStream_t s, TableView v, size_t n, const i64* k,
                            const unsigned char* vod, const uint8_t* ex, const u64* sc, unsigned dim, AuxInit ai,
                            int strat, u64 epoch, const u64* sorted_keys, const unsigned* sorted_idx, uint8_t* deferred, int bmode) {
  switch (dt) {
    case TFRA_F32: launch_accum<TFRA_F32>(g, grid, s, v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx, deferred, bmode); break;
    case TFRA_F16: launch_accum<TFRA_F16>(g, grid, s, v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx, deferred, bmode); break;
    case TFRA_BF16: launch_accum<TFRA_BF16>(g, grid, s, v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx, deferred, bmode); break;
    case TFRA_I8: launch_accum<TFRA_I8>(g, grid, s, v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx, deferred, bmode); break;
    case TFRA_I32: launch_accum<TFRA_I32>(g, grid, s, v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx, deferred, bmode); break;
    case TFRA_I64: launch_accum<TFRA_I64>(g, grid, s, v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx, deferred, bmode); break;
    default: launch_accum<TFRA_F64>(g, grid, s, v, n, k, vod, ex, sc, dim, ai, strat, epoch, sorted_keys, sorted_idx, deferred, bmode); break;
  }
}

// ------------------------------------ C ABI --------------------------------------------------
extern "C" {

const char* tfra_last_error(void) { return g_last_error.c_str(); }
int tfra_abi_version(void) { return TFRA_ABI_VERSION; }

int tfra_table_create(const tfra_table_opts* o, const tfra_allocator* alloc, tfra_table_t** out) {
  if (!o || !out) return set_error(TFRA_ERR_INVALID, "null argument");
  if (o->struct_size != sizeof(tfra_table_opts)) return set_error(TFRA_ERR_INVALID, "tfra_table_opts size mismatch (ABI)");
  size_t es = dtype_size(o->value_dtype);
  if (!es) return set_error(TFRA_ERR_INVALID, "unsupported value_dtype");
  if (o->dim <= 0) return set_error(TFRA_ERR_INVALID, "dim must be positive");
  if (o->aux_fields < 0 || o->aux_fields > 4) return set_error(TFRA_ERR_INVALID, "aux_fields must be in [0,4]");
  if (o->strategy < -1 || o->strategy > 4) return set_error(TFRA_ERR_INVALID, "unknown eviction strategy");
  Table* t = new Table();
  t->opts = *o;
  if (alloc) t->alloc = *alloc;
  int dev = o->device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { delete t; return set_error(TFRA_ERR_HIP, "no HIP device"); }
  t->device = dev;
  if (hipSetDevice(dev) != hipSuccess) { delete t; return set_error(TFRA_ERR_HIP, "hipSetDevice failed"); }
  // hkv_hashtable_op_gpu.cu.cc:121-133: init 0 -> default, max < init -> max = init
  if (t->opts.init_capacity == 0) t->opts.init_capacity = t->opts.max_capacity ? (1ULL << 20) : 8192;
  if (t->opts.max_capacity && t->opts.max_capacity < t->opts.init_capacity) t->opts.max_capacity = t->opts.init_capacity;
  if (t->opts.max_load_factor <= 0.f || t->opts.max_load_factor > 1.f)
    t->opts.max_load_factor = t->opts.max_capacity ? 0.5f : 0.75f;
  t->field_bytes = (unsigned)(o->dim * es);
  t->row_stride = (unsigned)(((size_t)t->field_bytes * (1 + o->aux_fields) + 15) / 16 * 16);
  t->aux = make_aux_init(t->opts);
  hipStream_t s = nullptr;
  auto fail = [&](int rc) { tfra_table_destroy(reinterpret_cast<tfra_table_t*>(t)); return rc; };
  size_t ctr_bytes = (SIZE_SHARDS * SIZE_SHARD_STRIDE) * sizeof(u64);
  t->size_shards = (u64*)t->dalloc(ctr_bytes, s);
  t->reserved_present = (unsigned*)t->dalloc(64, s);  // [0,1] sentinel-key presence | +8 err_count | +10 dense flag | +12 two i64 scalars
  if (!t->size_shards || !t->reserved_present) return fail(set_error(TFRA_ERR_OOM, "counter allocation failed"));
  t->err_count = t->reserved_present + 8;
  t->d_dense = t->reserved_present + 10;
  t->d_scalar = (i64*)(t->reserved_present + 12);
  if (hipHostMalloc((void**)&t->h_scalar, 64) != hipSuccess) return fail(set_error(TFRA_ERR_OOM, "pinned scalar"));
  if (hipHostMalloc((void**)&t->own_stats_host, 64) != hipSuccess) { t->own_stats_host = nullptr; return fail(set_error(TFRA_ERR_OOM, "pinned sample words")); }
  t->own_stats_host[0] = t->own_stats_host[1] = 0;   // (allocated here, not at the first write-back: that one may run under stream capture)
  if (hipEventCreateWithFlags(&t->chain_event, hipEventDisableTiming) != hipSuccess) return fail(set_error(TFRA_ERR_HIP, "event"));
  if (hipEventCreateWithFlags(&t->size_event, hipEventDisableTiming) != hipSuccess) return fail(set_error(TFRA_ERR_HIP, "event"));
  t->h_size = t->h_scalar + 4;
  u64 nb = std::max<u64>(2, (u64)((double)t->opts.init_capacity / t->opts.max_load_factor / SLOTS) + 1);
  nb = t->lattice_nb(nb);   // bounded tables: slots <= max_capacity, and doublings end exactly there
  int rc = t->alloc_storage(nb, &t->cur, s);
  if (rc) return fail(rc);
  clear_kernel<<<2048, 256, 0, s>>>(t->view_of(t->cur), 1);
  if (hipStreamSynchronize(s) != hipSuccess) return fail(set_error(TFRA_ERR_HIP, "clear failed"));
  *out = reinterpret_cast<tfra_table_t*>(t);
  return TFRA_OK;
}

int tfra_table_destroy(tfra_table_t* tp) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t) return TFRA_OK;
  (void)hipSetDevice(t->device);
  (void)hipDeviceSynchronize();
  hipStream_t s = nullptr;
  destroy_own_plan(t);
  t->free_storage(t->cur, s);
  t->dfree(t->size_shards, s); t->dfree(t->reserved_present, s);
  t->dfree(t->winner, s); t->dfree(t->scratch, s); t->dfree(t->evict_flags, s); t->dfree(t->own_tags, s); t->dfree(t->own_ws, s);
  if (t->progress_host) (void)hipHostFree(t->progress_host);
  if (t->own_stats_host) (void)hipHostFree(t->own_stats_host);
  if (t->h_scalar) (void)hipHostFree(t->h_scalar);
  if (t->chain_event) (void)hipEventDestroy(t->chain_event);
  if (t->size_event) (void)hipEventDestroy(t->size_event);
  delete t;
  return TFRA_OK;
}

#define TABLE_ENTER()                                         \
  Table* t = reinterpret_cast<Table*>(tp);                    \
  if (!t) return set_error(TFRA_ERR_INVALID, "null table");   \
  hipStream_t s = (hipStream_t)stream;                        \
  std::lock_guard<std::mutex> lock(t->mu);                    \
  { int _rc = t->enter(s); if (_rc) return _rc; }

int tfra_table_find(tfra_table_t* tp, size_t n, const int64_t* keys, void* values, uint8_t* exists,
                    const void* defaults, int default_is_full, tfra_stream_t stream) {
  TABLE_ENTER();
  return find_impl(t, s, 0, n, keys, values, exists, defaults, default_is_full);
}

int tfra_table_find_n(tfra_table_t* tp, size_t n, const int64_t* d_n, const int64_t* keys, void* values, uint8_t* exists,
                      const void* defaults, int default_is_full, tfra_stream_t stream) {
  TABLE_ENTER();
  if (!d_n) return set_error(TFRA_ERR_INVALID, "find_n: null count");
  return find_impl(t, s, 0, n, keys, values, exists, defaults, default_is_full, d_n);
}

// Unique keys, their number on the device: the ownership pass or nothing (the locked kernels size their scratch by the host's n).
int tfra_table_insert_or_assign_n(tfra_table_t* tp, size_t n, const int64_t* d_n, const int64_t* keys, const void* values,
                                  const uint64_t* scores, tfra_stream_t stream) {
  TABLE_ENTER();
  if (n == 0) return TFRA_OK;
  if (!d_n || !keys || !values) return set_error(TFRA_ERR_INVALID, "insert_or_assign_n: null buffer");
  if (n >= (1ULL << 31)) return set_error(TFRA_ERR_INVALID, "insert: more than 2^31-1 keys per call");
  int rc = t->prepare_insert(n, s);   // (n is an upper bound of the keys: a growing table may grow a call early)
  if (rc) return rc;
  bool taken = false;
  rc = own_upsert_unique(t, s, n, (const i64*)keys, values, (const u64*)scores, &taken, nullptr, d_n);
  if (rc) return rc;
  if (!taken) return set_error(TFRA_ERR_UNSUPPORTED, "insert_or_assign_n: the single-pass write-back cannot take this call (owner tags off, "
                                                     "a bulk load, or no scratch while capturing): read the count and call tfra_table_insert_or_assign");
  const int strat = t->opts.strategy;
  if (strat == TFRA_EVICT_EPOCHLRU || strat == TFRA_EVICT_EPOCHLFU) {
    t->curr_step += 1;
    if (t->opts.step_per_epoch > 0 && t->curr_step > t->opts.step_per_epoch) { t->global_epoch += 1; t->curr_step = 1; }
  }
  return TFRA_OK;
}

int tfra_table_find_field(tfra_table_t* tp, int field, size_t n, const int64_t* keys, void* values,
                          uint8_t* exists, const void* defaults, int default_is_full, tfra_stream_t stream) {
  TABLE_ENTER();
  return find_impl(t, s, field, n, keys, values, exists, defaults, default_is_full);
}

int tfra_table_insert_or_assign(tfra_table_t* tp, size_t n, const int64_t* keys, const void* values,
                                const uint64_t* scores, uint32_t flags, tfra_stream_t stream) {
  TABLE_ENTER();
  return insert_impl(t, s, 0, n, keys, values, scores, flags);
}

int tfra_table_insert_field(tfra_table_t* tp, int field, size_t n, const int64_t* keys, const void* values,
                            uint32_t flags, tfra_stream_t stream) {
  TABLE_ENTER();
  return insert_impl(t, s, field, n, keys, values, nullptr, flags);
}

int tfra_table_accum_or_assign(tfra_table_t* tp, size_t n, const int64_t* keys, const void* vod,
                               const uint8_t* exists, const uint64_t* scores, uint32_t flags, tfra_stream_t stream) {
  TABLE_ENTER();
  if (n == 0) return TFRA_OK;
  if (!keys || !vod || !exists) return set_error(TFRA_ERR_INVALID, "accum: null buffer");
  if (n >= (1ULL << 31)) return set_error(TFRA_ERR_INVALID, "accum: more than 2^31-1 keys per call");
  int rc = t->prepare_insert(n, s);
  if (rc) return rc;
  int g = granule_of(t->field_bytes, vod, nullptr);
  TableView v = t->view_of(t->cur);
  const i64* k = (const i64*)keys;
  if (flags & TFRA_FLAG_UNIQUE_KEYS) {
    // unique keys (what TFRA hands the op: PY/dynamic_embedding_variable.py:1377-1378): the single pass with bucket ownership,
    // as tfra_table_insert_or_assign does, whenever the batch is small for the table; else the locked kernels
    bool taken = false;
    rc = own_upsert_unique(t, s, n, k, vod, (const u64*)scores, &taken, exists);
    if (rc) return rc;
    if (taken) return TFRA_OK;   // (TableWrapper::accum does not step the epoch: lookup_table_op_hkv.h:539-546)
    dim3 grid((unsigned)((n * 16 + 255) / 256));
    uint8_t* deferred;
    rc = t->bounded_flags(n, s, &deferred);
    if (rc) return rc;
    launch_accum_dt(t->opts.value_dtype, g, grid, s, v, n, k, (const unsigned char*)vod, exists, (const u64*)scores,
                    (unsigned)t->opts.dim, t->aux, t->opts.strategy, t->global_epoch, nullptr, nullptr, deferred,
                    deferred ? (t->dense ? 2 : 1) : 0);
    if (deferred) {  // phase 2: the absent keys that found no free slot replace a minimum-score entry
      const unsigned char* vals = (const unsigned char*)vod;
      const u64* sc = (const u64*)scores;
      const int strat = t->opts.strategy;
      const u64 epoch = t->global_epoch;
      dim3 block(256);
      switch (g) {
        case 16: insert_evict_kernel<16><<<grid, block, 0, s>>>(v, n, k, vals, sc, 0, t->aux, strat, epoch, deferred); break;
        case 8: insert_evict_kernel<8><<<grid, block, 0, s>>>(v, n, k, vals, sc, 0, t->aux, strat, epoch, deferred); break;
        case 4: insert_evict_kernel<4><<<grid, block, 0, s>>>(v, n, k, vals, sc, 0, t->aux, strat, epoch, deferred); break;
        case 2: insert_evict_kernel<2><<<grid, block, 0, s>>>(v, n, k, vals, sc, 0, t->aux, strat, epoch, deferred); break;
        default: insert_evict_kernel<1><<<grid, block, 0, s>>>(v, n, k, vals, sc, 0, t->aux, strat, epoch, deferred); break;
      }
    }
    HIP_TRY(hipGetLastError());
    return TFRA_OK;
  }
  {
    uint8_t* bounded_now;
    rc = t->bounded_flags(1, s, &bounded_now);
    if (rc) return rc;
    if (bounded_now)
      return set_error(TFRA_ERR_UNSUPPORTED, "accum: a bounded (Hkv) table at max_capacity needs TFRA_FLAG_UNIQUE_KEYS "
                                             "(HKV's unique-keys contract) so that eviction is well defined");
  }
  // Duplicate-safe mode: the reference applies the triples sequentially in index order (LaunchTensorsAccum on one
  // thread).  All on the device, no host copy and no synchronisation: a stable radix sort of (key, index) groups the
  // occurrences of a key with their indices ascending, and accum_segments_kernel walks each group in that order.
  {
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tmp_bytes, (const u64*)nullptr, (u64*)nullptr, (const unsigned*)nullptr,
                                      (unsigned*)nullptr, n, 0u, 64u, s));
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    rc = t->ensure_scratch(al(n * 8) + 2 * al(n * 4) + al(tmp_bytes), s);
    if (rc) return rc;
    t->apply_P = 0;  // scratch head is overwritten below
    unsigned char* w = (unsigned char*)t->scratch;
    u64* sorted_keys = (u64*)w; w += al(n * 8);
    unsigned* iota = (unsigned*)w; w += al(n * 4);
    unsigned* sorted_idx = (unsigned*)w; w += al(n * 4);
    void* tmp = w;
    iota_u32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(iota, n);
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, (const u64*)keys, sorted_keys, (const unsigned*)iota, sorted_idx, n, 0u, 64u, s));
    dim3 grid((unsigned)((n * 16 + 255) / 256));
    launch_accum_dt(t->opts.value_dtype, g, grid, s, v, n, k, (const unsigned char*)vod, exists, (const u64*)scores,
                    (unsigned)t->opts.dim, t->aux, t->opts.strategy, t->global_epoch, sorted_keys, sorted_idx, nullptr, 0);
  }
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_table_erase(tfra_table_t* tp, size_t n, const int64_t* keys, tfra_stream_t stream) {
  TABLE_ENTER();
  if (n == 0) return TFRA_OK;
  if (!keys) return set_error(TFRA_ERR_INVALID, "erase: null keys");
  erase_kernel<<<(unsigned)((n * 16 + 255) / 256), 256, 0, s>>>(t->view_of(t->cur), n, (const i64*)keys);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_table_clear(tfra_table_t* tp, tfra_stream_t stream) {
  TABLE_ENTER();
  clear_kernel<<<2048, 256, 0, s>>>(t->view_of(t->cur), 1);
  HIP_TRY(hipGetLastError());
  t->size_ub = 0;
  t->size_pending = false;
  t->n_since_read = 0;
  t->dense = false;
  return TFRA_OK;
}

int tfra_table_size(tfra_table_t* tp, size_t* out, tfra_stream_t stream) {
  TABLE_ENTER();
  if (!out) return set_error(TFRA_ERR_INVALID, "size: null out");
  int rc = t->read_size(s, out);
  if (rc) return rc;
  return t->check_errors(s);
}

int tfra_table_size_to_device(tfra_table_t* tp, int64_t* d_out, tfra_stream_t stream) {
  TABLE_ENTER();
  if (!d_out) return set_error(TFRA_ERR_INVALID, "size: null out");
  size_kernel<<<1, SIZE_SHARDS, 0, s>>>(t->view_of(t->cur), (i64*)d_out);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_table_check_errors(tfra_table_t* tp, tfra_stream_t stream) {
  TABLE_ENTER();
  return t->check_errors(s);
}

int tfra_table_slot_census(tfra_table_t* tp, uint64_t* out5, tfra_stream_t stream) {
  TABLE_ENTER();
  if (!out5) return set_error(TFRA_ERR_INVALID, "slot_census: null out");
  u64* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 5 * sizeof(u64)));
  HIP_TRY(hipMemsetAsync(d, 0, 5 * sizeof(u64), s));
  slot_census_kernel<<<2048, 256, 0, s>>>(t->view_of(t->cur), d);
  HIP_TRY(hipMemcpyAsync(out5, d, 5 * sizeof(u64), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipFree(d));
  return TFRA_OK;
}

int tfra_table_capacity(tfra_table_t* tp, size_t* out) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !out) return set_error(TFRA_ERR_INVALID, "capacity: null argument");
  std::lock_guard<std::mutex> lock(t->mu);
  *out = t->cur.nb * SLOTS + NUM_RESERVED;
  return TFRA_OK;
}

int tfra_table_growth_stats(tfra_table_t* tp, uint64_t* out4) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !out4) return set_error(TFRA_ERR_INVALID, "growth_stats: null argument");
  std::lock_guard<std::mutex> lock(t->mu);
  out4[0] = (uint64_t)t->n_rehash; out4[1] = (uint64_t)t->n_split; out4[2] = t->cur.vmm ? 1 : 0; out4[3] = (uint64_t)t->cur.mapped;
  return TFRA_OK;
}

int tfra_table_reserve(tfra_table_t* tp, size_t min_slots, tfra_stream_t stream) {
  TABLE_ENTER();
  u64 nb = (min_slots + SLOTS - 1) / SLOTS;
  if (t->opts.max_capacity) nb = std::min<u64>(nb, std::max<u64>(2, t->opts.max_capacity / SLOTS));
  return t->grow(nb, s);
}

int tfra_table_export_batch(tfra_table_t* tp, size_t n, size_t offset, size_t* d_counter, int64_t* keys,
                            void* values, uint64_t* scores, tfra_stream_t stream) {
  TABLE_ENTER();
  if (!d_counter || !keys) return set_error(TFRA_ERR_INVALID, "export: null buffer");
  TableView v = t->view_of(t->cur);
  u64 lo = offset, hi = offset + n, total = v.nb * SLOTS;
  if (n == 0 || lo >= total + NUM_RESERVED) return TFRA_OK;
  u64 fb = lo / SLOTS, lb = std::min<u64>(v.nb, (std::min<u64>(hi, total) + SLOTS - 1) / SLOTS);
  int g = granule_of(t->field_bytes, values, nullptr);
  if (lb > fb) {
    dim3 grid((unsigned)((lb - fb + 63) / 64)), block(256);
    u64* c = (u64*)d_counter;
    unsigned char* vo = (unsigned char*)values;
    switch (g) {
      case 16: export_kernel<16><<<grid, block, 0, s>>>(v, fb, lb, lo, hi, c, (i64*)keys, vo, (u64*)scores); break;
      case 8: export_kernel<8><<<grid, block, 0, s>>>(v, fb, lb, lo, hi, c, (i64*)keys, vo, (u64*)scores); break;
      case 4: export_kernel<4><<<grid, block, 0, s>>>(v, fb, lb, lo, hi, c, (i64*)keys, vo, (u64*)scores); break;
      case 2: export_kernel<2><<<grid, block, 0, s>>>(v, fb, lb, lo, hi, c, (i64*)keys, vo, (u64*)scores); break;
      default: export_kernel<1><<<grid, block, 0, s>>>(v, fb, lb, lo, hi, c, (i64*)keys, vo, (u64*)scores); break;
    }
  }
  if (hi > total)
    export_reserved_kernel<<<1, 64, 0, s>>>(v, lo, hi, (u64*)d_counter, (i64*)keys, (unsigned char*)values, (u64*)scores);
  HIP_TRY(hipGetLastError());
  return TFRA_OK;
}

int tfra_table_set_option(tfra_table_t* tp, int option, int64_t value) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t) return set_error(TFRA_ERR_INVALID, "null table");
  std::lock_guard<std::mutex> lock(t->mu);
  if (option == TFRA_OPTION_CAPTURE_SAFE) { t->capture_safe = value != 0; return TFRA_OK; }
  if (option == TFRA_OPTION_NO_OWNER_TAGS) { t->no_owner_tags = value != 0; return TFRA_OK; }
  if (option == TFRA_OPTION_KEY_BYTES_ON_DISK) {
    if (value != 4 && value != 8) return set_error(TFRA_ERR_INVALID, "TFRA_OPTION_KEY_BYTES_ON_DISK: 4 or 8");
    t->key_file_bytes = (int)value;
    return TFRA_OK;
  }
  return set_error(TFRA_ERR_INVALID, "unknown option");
}

int tfra_table_set_global_epoch(tfra_table_t* tp, uint64_t epoch) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t) return set_error(TFRA_ERR_INVALID, "null table");
  std::lock_guard<std::mutex> lock(t->mu);
  t->global_epoch = epoch;
  return TFRA_OK;
}

}  // extern "C"
