// Sparse optimizer write-back WITH duplicate ids: tfra_table_apply_sparse(ids[B], grads[B,D]).
// Three kernels, no global sort passes, no atomics on gradient data, deterministic.
//
// Reference: gradients of duplicate ids are summed, then ONE update per key
// (`_resource_apply_sparse_duplicate_indices` = unique + unsorted_segment_sum,
// PY/dynamic_embedding_optimizer.py:177-190), followed by (1+S) finds + dense apply + (1+S)
// upserts (:165-204).  A Zipf-1.2 batch of 131 072 ids has ~22 K unique keys and the hottest key
// repeats ~24 000 times, so the reduction must be parallel per key yet order-fixed:
//
//   A  tile_reduce   one block per TILE=512 ids: bitonic-sort (fmix64(id), idx) in LDS — equal ids
//      become adjacent with ascending idx, and because bucket = mulhi(hash, P) is monotone in the
//      hash the tile's unique keys come out grouped by bucket.  One descriptor (key, src) per
//      unique key per tile: ids occurring once in the tile point straight at their gradient row
//      (nothing is copied); runs of >= 2 are summed by 16-lane groups into a scratch row.
//   C  bucket_merge  one block per bucket: gathers the bucket's descriptors from all tiles (tile
//      order), bitonic-sorts (key, tile-order) in LDS, sums each key's partial rows in tile order
//      -> ONE (key, src) per unique key of the batch (again pass-through when there is one part).
//   apply_kernel<INDIRECT> (tfra_optim.hip) one 16-lane group per unique key: locate-or-insert
//      the row, read [p|m|v], apply, write back.
//
// Row reads are issued in batches of independent loads before the order-dependent adds, so the
// kernels are bound by memory-level parallelism, not by one latency per row.
// Summation tree per key = [ascending idx inside a (tile, position-chunk)] -> [chunks of the tile
// in order] -> [tiles in order (chunked the same way)]: fixed by the input => bit-reproducible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>

#include "../../include/tfra_mi355x.h"
#include "tfra_device.h"
#include "tfra_host.h"
#include "tfra_optim_device.h"

using namespace tfra;

namespace tfra {
// implemented in tfra_optim.hip
int launch_apply_indirect(Table* t, hipStream_t s, const tfra_opt_params* p, size_t max_n, const i64* keys,
                          const unsigned* src, const float* grads, const float* alt_rows, unsigned alt_base,
                          const float* default_row, const i64* d_n);
}  // namespace tfra

namespace {

constexpr int TILE = 512;    // ids per kernel-A block
constexpr int NTA = 512;     // kernel-A threads (32 groups of 16 lanes: 8 waves keep a CU busy)
constexpr int NT = 256;      // kernel-C threads (16 groups)
constexpr int CMAX = 1024;   // descriptors one kernel-C block can hold in LDS
constexpr int MAXCH = 4;     // D <= 256 (one float4 per lane per 64-column chunk)
constexpr unsigned SKIP = 0xffffffffu;
constexpr unsigned char F_HEAD = 1, F_SINGLE = 2;

template <int NCH> struct Batch { static constexpr int v = NCH == 1 ? 8 : (NCH == 2 ? 4 : 2); };

// in-LDS bitonic sort of n2 (power of two) 32-bit keys by NTH threads.
// Pair q of a stage touches elements i = 2j*(q/j) + q%j and i+j.  Thread t owns pairs t, t+NTH, ...
// so for j <= 64 every wave works inside its own 128-element windows: those stages need no block
// barrier (LDS operations of one wave execute in order) — only stages with distance >= 128 do.
// History: sorting (u64 hash, index) PAIRS with a __syncthreads per stage cost 14 us (512 keys) to
// 20 us (kernel C) per launch; grouping by an LDS hash first and sorting one packed u32 is ~5x less.
template <int NTH>
__device__ __forceinline__ void bitonic_sort_u32(unsigned* a, int n2) {
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int jm = j - 1;
      for (int q = threadIdx.x; q < (n2 >> 1); q += NTH) {
        int i = ((q & ~jm) << 1) | (q & jm);
        int l = i + j;
        unsigned x = a[i], y = a[l];
        unsigned lo = min(x, y), hi = max(x, y);
        bool up = ((i & k) == 0);
        a[i] = up ? lo : hi;
        a[l] = up ? hi : lo;
      }
      // block barrier iff this stage or the next one exchanges across waves (distance >= 128);
      // the stage after (k, 1) is (2k, k)
      if (j >= 128 || (j == 1 && (k >= 128 || k == n2))) __syncthreads();
      else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // same-wave LDS ordering only
    }
  }
}

// Group equal 64-bit keys of an LDS array: returns a slot id in [0, cap) that is the same for equal
// keys and different for different keys.  owner[] (cap entries, zeroed) holds 1 + the index of the
// element that claimed the slot; keys are compared through keys[owner-1].
__device__ __forceinline__ unsigned lds_group_slot(const i64* keys, unsigned* owner, unsigned cap, int me, u64 hash) {
  const i64 key = keys[me];
  unsigned slot = (unsigned)(hash >> 17) & (cap - 1);
  for (;;) {
    unsigned o = owner[slot];
    if (o == 0) {
      o = atomicCAS(&owner[slot], 0u, (unsigned)me + 1u);
      if (o == 0) return slot;
    }
    if (keys[o - 1] == key) return slot;
    slot = (slot + 1) & (cap - 1);
  }
}

// exclusive scan of one int per thread over the NTH-thread block
template <int NTH>
__device__ __forceinline__ int block_excl_scan(int v, int* sh /*[NTH/64]*/, int* total) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, incl = v;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) sh[w] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
  for (int k = 0; k < NTH / 64; ++k) { if (k < w) woff += sh[k]; tot += sh[k]; }
  __syncthreads();
  *total = tot;
  return woff + incl - v;
}

// ---------------------------------------------------------------------------------------------
// Ordered run sums over a sorted LDS sequence, shared by kernels A and C.
//   positions [0,n) carry flags (F_HEAD = first of its run, F_SINGLE = run of length 1 -> skipped)
//   group g (16 lanes) owns the contiguous chunk [g*span, (g+1)*span); rows are fetched BATCH at a
//   time (independent loads in flight) and added in position order.  A run crossing chunk borders
//   is finished by the group that holds its head ("owner"): later chunks leave their share in
//   s_left and the owner adds those in chunk order after one barrier.
//   row_of(p)  -> const float* of position p's row        out_of(p_head) -> float* for the run sum
template <int NCH, int NG, class RowOf, class OutOf>
__device__ __forceinline__ void ordered_run_sums(int n, int span, int dim, const unsigned char* s_flag,
                                                 float (*s_left)[64 * NCH], unsigned char* s_cont,
                                                 unsigned char* s_hashead, RowOf row_of, OutOf out_of) {
  constexpr int BATCH = Batch<NCH>::v;
  const int lane = threadIdx.x & 63, sub = lane & 15, g = threadIdx.x >> 4;
  const int gs = g * span, ge = min(gs + span, n);
  const bool has_any = gs < n;
  const bool cont_in = has_any && !(s_flag[gs] & F_HEAD);
  float4 acc[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  bool seen_head = false, owner_open = false, in_left = cont_in;
  int run_head = -1;  // position of the head of the run being accumulated (when owned)
  auto store_run = [&](int ph) {
    float* o = out_of(ph);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      int col = k * 64 + sub * 4;
      if (col < dim) *reinterpret_cast<float4*>(o + col) = acc[k];
    }
  };
  auto store_left = [&]() {
#pragma unroll
    for (int k = 0; k < NCH; ++k) *reinterpret_cast<float4*>(&s_left[g][k * 64 + sub * 4]) = acc[k];
  };
  if (has_any) {
    for (int p0 = gs; p0 < ge; p0 += BATCH) {
      float4 x[BATCH][NCH];
      // unconditional loads (skipped positions re-read the chunk's first row, an L2 hit) so that the
      // BATCH row fetches are all in flight before the first add; see find_kernel for why
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        int p = p0 + j;
        bool need = p < ge && !(s_flag[p] & F_SINGLE);
        const float* row = row_of(need ? p : gs);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          int col = k * 64 + sub * 4;
          x[j][k] = *reinterpret_cast<const float4*>(row + (col < dim ? col : 0));
        }
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        int p = p0 + j;
        if (p < ge) {
          unsigned char f = s_flag[p];
          if (f & F_HEAD) {
            // close the run accumulated so far
            if (in_left) { store_left(); in_left = false; }
            else if (run_head >= 0) store_run(run_head);
#pragma unroll
            for (int k = 0; k < NCH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            run_head = (f & F_SINGLE) ? -1 : p;
            if (p > gs) seen_head = true;
          }
          if (!(f & F_SINGLE)) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) { acc[k].x += x[j][k].x; acc[k].y += x[j][k].y; acc[k].z += x[j][k].z; acc[k].w += x[j][k].w; }
          }
        }
      }
    }
    bool spills_out = (ge == gs + span) && ge < n && !(s_flag[ge] & F_HEAD);
    if (in_left) store_left();                 // the whole chunk belongs to the spill-in run
    else if (run_head >= 0 && spills_out) owner_open = true;
    else if (run_head >= 0) store_run(run_head);
  }
  if (sub == 0) { s_cont[g] = cont_in; s_hashead[g] = seen_head; }
  __syncthreads();
  if (owner_open) {
    for (int g2 = g + 1; g2 < NG && s_cont[g2]; ++g2) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        float4 x = *reinterpret_cast<const float4*>(&s_left[g2][k * 64 + sub * 4]);
        acc[k].x += x.x; acc[k].y += x.y; acc[k].z += x.z; acc[k].w += x.w;
      }
      if (s_hashead[g2]) break;  // the run ended inside g2
    }
    store_run(run_head);
  }
}

// ---------------------------------------------------------------------------------------------
// kernel A.  Descriptor u of tile t lives at index t*TILE+u: part_keys[], part_src[] where
// src < rows_base -> gradient row `src` of the caller's buffer, else scratch row (src-rows_base).
// Sort key (31 bits) = bucket(11) | group slot(11) | position in tile(9): groups the tile by merge
// bucket, then by key, ascending input position inside a key.
template <int NCH>
__global__ __launch_bounds__(NTA) void tile_reduce_kernel(size_t n, const i64* __restrict__ ids,
                                                          const float* __restrict__ grads, int dim, unsigned P,
                                                          unsigned rows_base, i64* __restrict__ part_keys,
                                                          unsigned* __restrict__ part_src, float* __restrict__ scratch_rows,
                                                          unsigned short* __restrict__ tile_hist,
                                                          unsigned short* __restrict__ tile_start, int stop) {
  constexpr int NG = NTA / 16;
  constexpr unsigned GCAP = 2048;            // group table: 4x the tile => short probe chains
  __shared__ i64 s_key[TILE];                // ids of the tile, input order
  __shared__ unsigned s_sort[TILE];          // packed sort keys
  __shared__ unsigned s_owner[GCAP];
  __shared__ unsigned short s_u[TILE];       // unique rank (within the tile) of each head position
  __shared__ unsigned char s_flag[TILE + 1];
  __shared__ unsigned s_hist[2048];          // P <= 2048
  __shared__ float s_left[NG][64 * NCH];
  __shared__ unsigned char s_cont[NG], s_hashead[NG];
  __shared__ int s_scan[NTA / 64];
  const size_t tile = blockIdx.x, base = tile * TILE;
  const int nvalid = (int)min((size_t)TILE, n - base);
  static_assert(TILE == NTA && TILE == 512, "one id per thread; 9 position bits");
  const int p = threadIdx.x;
  const bool ok = p < nvalid;
  const i64 key = ok ? ids[base + p] : 0;
  s_key[p] = key;
  for (unsigned b = threadIdx.x; b < P; b += NTA) s_hist[b] = 0;
  for (unsigned b = threadIdx.x; b < GCAP; b += NTA) s_owner[b] = 0;
  __syncthreads();
  if (stop == 1) return;  // (tuning ablation, TFRA_DBG_STOP_A)
  const u64 h = fmix64((u64)key);
  unsigned packed = 0xffffffffu;
  if (ok) {
    unsigned slot = lds_group_slot(s_key, s_owner, GCAP, p, h);
    packed = ((unsigned)__umul64hi(h, (u64)P) << 20) | (slot << 9) | (unsigned)p;
  }
  s_sort[p] = packed;
  __syncthreads();
  bitonic_sort_u32<NTA>(s_sort, TILE);
  if (stop == 2) return;
  // heads / singles / unique ranks: one sorted position per thread
  const unsigned me = s_sort[p];
  const bool head = p < nvalid && (p == 0 || (s_sort[p - 1] >> 9) != (me >> 9));
  int ntile_unique;
  const int u = block_excl_scan<NTA>(head ? 1 : 0, s_scan, &ntile_unique);
  {
    bool single = head && (p + 1 >= nvalid || (s_sort[p + 1] >> 9) != (me >> 9));
    s_flag[p] = (head ? F_HEAD : 0) | (single ? F_SINGLE : 0);
    if (head) {
      atomicAdd(&s_hist[me >> 20], 1u);
      s_u[p] = (unsigned short)u;
      size_t d = base + (size_t)u;
      part_keys[d] = s_key[me & 511];
      part_src[d] = single ? (unsigned)(base + (me & 511)) : rows_base + (unsigned)d;
    }
  }
  if (threadIdx.x == 0) s_flag[TILE] = F_HEAD;
  __syncthreads();
  if (stop == 3) return;
  // per-bucket count and first unique rank (buckets ascend with the sort key)
  {
    int carry = 0;
    for (unsigned b0 = 0; b0 < P; b0 += NTA) {
      unsigned b = b0 + threadIdx.x;
      int v = b < P ? (int)s_hist[b] : 0, tot;
      int ex = block_excl_scan<NTA>(v, s_scan, &tot);
      if (b < P) {
        tile_hist[tile * P + b] = (unsigned short)v;
        tile_start[tile * P + b] = (unsigned short)(carry + ex);
      }
      carry += tot;
    }
  }
  if (stop == 4) return;
  ordered_run_sums<NCH, NG>(
      nvalid, TILE / NG, dim, s_flag, s_left, s_cont, s_hashead,
      [&](int q) { return grads + (base + (s_sort[q] & 511)) * (size_t)dim; },
      [&](int ph) { return scratch_rows + (base + (size_t)s_u[ph]) * (size_t)dim; });
}

// ---------------------------------------------------------------------------------------------
// kernel C.  Output: bucket b owns u_keys/u_src[off_b .. off_b+n_b) with n_b = its descriptor count
// and off_b = sum_t tile_start[t][b]; the first (#unique in bucket) entries are filled, the rest are
// SKIP.  Summed rows go to scratch row (sum_base - rows_base + index).  The last bucket publishes
// the total entry count.  Descriptors are gathered in tile order, grouped by key with an LDS hash
// and sorted on (group slot(11) | gather position(10)), so a key's parts stay in tile order.
// A bucket holding more than CMAX descriptors (several very hot keys hashing together) is
// processed in 2^k passes, pass q taking the keys with (hash & (2^k-1)) == q.
template <int NCH>
__global__ __launch_bounds__(NT) void bucket_merge_kernel(unsigned P, unsigned ntiles, int dim, unsigned rows_base,
                                                          unsigned sum_base, const float* __restrict__ grads,
                                                          const i64* __restrict__ part_keys,
                                                          const unsigned* __restrict__ part_src,
                                                          float* __restrict__ scratch_rows,
                                                          const unsigned short* __restrict__ tile_hist,
                                                          const unsigned short* __restrict__ tile_start,
                                                          i64* __restrict__ u_keys, unsigned* __restrict__ u_src,
                                                          i64* __restrict__ d_total, unsigned* overflow, int stop) {
  constexpr int NG = NT / 16;
  constexpr unsigned GCAP = 2 * CMAX;
  static_assert(CMAX == 1024, "10 position bits");
  __shared__ i64 e_key[CMAX];         // gathered keys, tile order
  __shared__ unsigned e_src[CMAX];    // their part_src
  __shared__ unsigned s_sort[CMAX];   // packed sort keys
  __shared__ unsigned s_owner[GCAP];
  __shared__ unsigned char s_flag[CMAX + 1];
  __shared__ unsigned short s_rank[CMAX];  // unique rank (within the pass) of each head position
  __shared__ float s_left[NG][64 * NCH];
  __shared__ unsigned char s_cont[NG], s_hashead[NG];
  __shared__ int s_scan[NT / 64];
  __shared__ long long s_off[NT / 64];
  const unsigned b = blockIdx.x;
  unsigned npass = 1;
  int n_b = 0, out_used = 0;
  long long off = 0;
  for (unsigned pass = 0; pass < npass; ++pass) {
    // gather this pass's descriptors from every tile, tile order; the first pass also finds the
    // bucket's totals (n_b descriptors, output offset off_b) and decides how many passes it needs
    int carry = 0, all = 0;
    long long off_acc = 0;
    bool too_many = false;
    for (unsigned t0 = 0; t0 < ntiles; t0 += NT) {
      unsigned t = t0 + threadIdx.x;
      int cnt = 0, st = 0, mine = 0, tot;
      if (t < ntiles) { cnt = tile_hist[(size_t)t * P + b]; st = tile_start[(size_t)t * P + b]; }
      if (npass == 1) {
        mine = cnt;
      } else {
        for (int j = 0; j < cnt; ++j)
          mine += ((unsigned)fmix64((u64)part_keys[t * TILE + st + j]) & (npass - 1)) == pass;
      }
      if (pass == 0) {  // off_b = sum over tiles of (descriptors of smaller buckets in that tile)
        int st_sum = st, cnt_sum = cnt;
        for (int o2 = 32; o2 > 0; o2 >>= 1) { st_sum += __shfl_xor(st_sum, o2); cnt_sum += __shfl_xor(cnt_sum, o2); }
        if ((threadIdx.x & 63) == 0) s_off[threadIdx.x >> 6] = ((long long)cnt_sum << 40) | (long long)st_sum;
      }
      int ex = block_excl_scan<NT>(mine, s_scan, &tot);  // (barriers inside also publish s_off)
      if (pass == 0) {
        long long both = s_off[0] + s_off[1] + s_off[2] + s_off[3];  // st sums < 2^40: no carry into cnt
        off_acc += both & ((1LL << 40) - 1);
        all += (int)(both >> 40);
      }
      if (carry + tot > CMAX) too_many = true;
      if (!too_many) {
        int w = carry + ex;
        for (int j = 0; j < cnt; ++j) {
          unsigned d = t * TILE + st + j;
          i64 key = part_keys[d];
          if (npass == 1 || ((unsigned)fmix64((u64)key) & (npass - 1)) == pass) {
            e_src[w] = part_src[d];
            e_key[w] = key;
            ++w;
          }
        }
      }
      carry += tot;
      __syncthreads();
    }
    if (pass == 0) {
      n_b = all;
      off = off_acc;
      if (b == P - 1 && threadIdx.x == 0) *d_total = off + n_b;
      if (n_b == 0) return;
      if (too_many && npass == 1) {  // restart with the bucket split by key hash
        while ((unsigned)n_b > (CMAX / 2) * npass && npass < 64) npass <<= 1;
        pass = (unsigned)-1;  // ++ -> 0
        continue;
      }
    }
    if (too_many) {  // one key alone exceeds CMAX parts: reported, not applied
      if (threadIdx.x == 0) atomicAdd(overflow, 1u);
      continue;
    }
    const int n = carry;
    if (n == 0) continue;
    if (stop == 1) continue;  // (tuning ablation, TFRA_DBG_STOP_C)
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    for (unsigned q = threadIdx.x; q < GCAP; q += NT) s_owner[q] = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < n2; q += NT) {
      unsigned packed = 0xffffffffu;
      if (q < n) packed = (lds_group_slot(e_key, s_owner, GCAP, q, fmix64((u64)e_key[q])) << 10) | (unsigned)q;
      s_sort[q] = packed;
    }
    __syncthreads();
    bitonic_sort_u32<NT>(s_sort, n2);
    if (stop == 2) continue;
    // flags + unique ranks; pass-through runs (exactly one part) need no row traffic
    int ccarry = 0;
    for (int pb = 0; pb < n; pb += NT) {
      int q = pb + threadIdx.x;
      unsigned me = q < n ? s_sort[q] : 0;
      bool hd = q < n && (q == 0 || (s_sort[q - 1] >> 10) != (me >> 10));
      bool single = hd && (q + 1 >= n || (s_sort[q + 1] >> 10) != (me >> 10));
      int tot;
      int ex = block_excl_scan<NT>(hd ? 1 : 0, s_scan, &tot);
      if (q < n) {
        s_flag[q] = (hd ? F_HEAD : 0) | (single ? F_SINGLE : 0);
        if (hd) {
          long long o = off + out_used + ccarry + ex;
          u_keys[o] = e_key[me & 1023];
          u_src[o] = single ? e_src[me & 1023] : sum_base + (unsigned)o;
          s_rank[q] = (unsigned short)(ccarry + ex);
        }
      }
      ccarry += tot;
    }
    if (threadIdx.x == 0) s_flag[n] = F_HEAD;
    __syncthreads();
    const long long obase = off + out_used;
    if (stop == 3) { out_used += ccarry; continue; }
    ordered_run_sums<NCH, NG>(
        n, (n + NG - 1) / NG, dim, s_flag, s_left, s_cont, s_hashead,
        [&](int q) {
          unsigned src = e_src[s_sort[q] & 1023];
          return src < rows_base ? grads + (size_t)src * dim : scratch_rows + (size_t)(src - rows_base) * dim;
        },
        [&](int ph) { return scratch_rows + (size_t)(sum_base - rows_base + (unsigned)(obase + s_rank[ph])) * dim; });
    out_used += ccarry;
    __syncthreads();
  }
  for (int i = out_used + threadIdx.x; i < n_b; i += NT) u_src[off + i] = SKIP;
}

}  // namespace

extern "C" int tfra_table_apply_sparse(tfra_table_t* tp, const tfra_opt_params* p, size_t n, const int64_t* ids,
                                       const float* grads, const float* param_default_row, tfra_stream_t stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !p) return set_error(TFRA_ERR_INVALID, "apply_sparse: null argument");
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(t->mu);
  int rc = t->enter(s);
  if (rc) return rc;
  if (n == 0) return TFRA_OK;
  if (!ids || !grads || !param_default_row) return set_error(TFRA_ERR_INVALID, "apply_sparse: null buffer");
  if (t->opts.value_dtype != TFRA_F32) return set_error(TFRA_ERR_UNSUPPORTED, "apply_sparse: value_dtype must be float32");
  if (p->kind < 0 || p->kind > TFRA_OPT_FTRL) return set_error(TFRA_ERR_INVALID, "apply_sparse: unknown kind");
  int need = p->kind == TFRA_OPT_SGD ? 0 : (p->kind == TFRA_OPT_ADAGRAD ? 1 : 2);
  if (t->opts.aux_fields < need) return set_error(TFRA_ERR_INVALID, "apply_sparse: table lacks optimizer slot fields");
  const int dim = t->opts.dim;
  if (dim % 4 != 0 || dim > 64 * MAXCH || (((uintptr_t)grads | (uintptr_t)param_default_row) & 15))
    return set_error(TFRA_ERR_UNSUPPORTED, "apply_sparse: needs dim % 4 == 0, dim <= 256 and 16-B aligned buffers "
                                           "(use tfra_unique + tfra_segment_sum + tfra_table_apply_optimizer otherwise)");
  if (n > (1ULL << 20))
    return set_error(TFRA_ERR_UNSUPPORTED, "apply_sparse: at most 2^20 ids per call (one key may hold ntiles <= CMAX "
                                           "partial rows); split the batch or use the unique + segment_sum path");
  rc = t->prepare_insert(n, s);
  if (rc) return rc;
  const size_t ntiles = (n + TILE - 1) / TILE, npad = ntiles * TILE;
  unsigned P = 64;
  while (P < 2048 && (size_t)P * 64 < n) P <<= 1;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  // scratch: part_keys | part_src | u_keys | u_src | tile_hist | tile_start | d_total | rows[2*npad]
  size_t bytes = 2 * al(npad * 8) + 2 * al(npad * 4) + 2 * al(ntiles * P * 2) + 256 + 2 * al(npad * (size_t)dim * 4);
  rc = t->ensure_scratch(bytes, s);
  if (rc) return rc;
  unsigned char* w = (unsigned char*)t->scratch;
  i64* part_keys = (i64*)w; w += al(npad * 8);
  i64* u_keys = (i64*)w; w += al(npad * 8);
  unsigned* part_src = (unsigned*)w; w += al(npad * 4);
  unsigned* u_src = (unsigned*)w; w += al(npad * 4);
  unsigned short* th = (unsigned short*)w; w += al(ntiles * P * 2);
  unsigned short* ts = (unsigned short*)w; w += al(ntiles * P * 2);
  i64* d_total = (i64*)w; w += 256;
  float* rows = (float*)w;
  const unsigned rows_base = (unsigned)npad, sum_base = (unsigned)(2 * npad);
  const int nch = (dim + 63) / 64;
  const i64* k = (const i64*)ids;
  dim3 ga((unsigned)ntiles), gc(P);
  static const int stop_a = getenv("TFRA_DBG_STOP_A") ? atoi(getenv("TFRA_DBG_STOP_A")) : 0;
  static const int stop_c = getenv("TFRA_DBG_STOP_C") ? atoi(getenv("TFRA_DBG_STOP_C")) : 0;
  switch (nch) {
    case 1: tile_reduce_kernel<1><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, part_keys, part_src, rows, th, ts, stop_a); break;
    case 2: tile_reduce_kernel<2><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, part_keys, part_src, rows, th, ts, stop_a); break;
    case 3: tile_reduce_kernel<3><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, part_keys, part_src, rows, th, ts, stop_a); break;
    default: tile_reduce_kernel<4><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, part_keys, part_src, rows, th, ts, stop_a); break;
  }
  switch (nch) {
    case 1: bucket_merge_kernel<1><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, part_keys, part_src, rows, th, ts, u_keys, u_src, d_total, t->err_count, stop_c); break;
    case 2: bucket_merge_kernel<2><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, part_keys, part_src, rows, th, ts, u_keys, u_src, d_total, t->err_count, stop_c); break;
    case 3: bucket_merge_kernel<3><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, part_keys, part_src, rows, th, ts, u_keys, u_src, d_total, t->err_count, stop_c); break;
    default: bucket_merge_kernel<4><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, part_keys, part_src, rows, th, ts, u_keys, u_src, d_total, t->err_count, stop_c); break;
  }
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "apply_sparse: launch failed");
  return launch_apply_indirect(t, s, p, npad, u_keys, u_src, grads, rows, rows_base, param_default_row, d_total);
}
