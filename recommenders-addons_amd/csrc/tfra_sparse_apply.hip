// Sparse optimizer write-back WITH duplicate ids: tfra_table_apply_sparse(ids[B], grads[B,D]).
// Three kernels, no global sort passes, no atomics on gradient data, results bit-reproducible.
//
// Reference: gradients of duplicate ids are summed, then ONE update per key
// (`_resource_apply_sparse_duplicate_indices` = unique + unsorted_segment_sum,
// PY/dynamic_embedding_optimizer.py:177-190), followed by (1+S) finds + dense apply + (1+S)
// upserts (:165-204).  A Zipf-1.2 batch of 131 072 ids has ~22 K unique keys and the hottest key
// repeats ~24 000 times, so the reduction must be parallel per key yet order-fixed:
//
//   A  tile_reduce   one block per TILE=512 ids (one id per thread): equal ids are grouped with an
//      LDS hash; a group's deterministic id is its FIRST POSITION in the tile, so the unique ranks
//      and the order of the multi-member groups come out of block scans, and the members of a group
//      are ranked by popcounts over a 512-bit position mask — no sort.  One descriptor
//      (key, src, tile<<9|rank) per unique key per tile is appended to the key's merge bucket
//      (bucket = mulhi(fmix64(key), P), one returning atomic per descriptor on a per-bucket cursor
//      that lives on its own 128-B line): ids occurring once in the tile point straight at their
//      gradient row (nothing is copied); runs of >= 2 are summed by 16-lane groups into a scratch
//      row, ascending input position.
//   C  bucket_merge  one block per bucket: loads its descriptors (one coalesced read), groups by key
//      (LDS hash; group id = smallest descriptor id), passes single-part keys through, orders the
//      few multi-part groups (register bitonic of <= 256 ids), ranks their members by tile with a
//      tile bitmask + popcounts and sums each key's parts in TILE ORDER -> ONE (key, src) per
//      unique key of the batch.
//   apply_kernel<INDIRECT> (tfra_optim.hip) one 16-lane group per unique key: locate-or-insert
//      the row, read [p|m|v], apply, write back; re-arms the cursors for the next call.
//   tfra_reduce_by_key = A + C + compact_gather_kernel (dense (key, sum) output in a deterministic
//      order) for callers that route the sums elsewhere (multi-GPU gradient alltoall).
//
// Row reads are issued in batches of independent loads before the order-dependent adds, so the
// kernels are bound by memory-level parallelism, not by one latency per row.
// Summation tree per key = [ascending idx inside a (tile, position-chunk)] -> [chunks of the tile
// in order] -> [tiles in order (chunked the same way)]: a function of the input alone.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <functional>
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
                          const float* default_row, const i64* d_n, unsigned* rearm, unsigned rearm_count,
                          unsigned rearm_stride);
}  // namespace tfra

namespace {

constexpr int TILE = 512;    // ids per kernel-A block = threads per block
constexpr int NTA = 512;     // kernel-A threads (32 groups of 16 lanes)
constexpr int NT = 256;      // kernel-C threads (16 groups)
constexpr int CMAX = 1024;   // descriptors one kernel-C pass holds in LDS = capacity of a bucket region
constexpr int MAXCH = 4;     // D <= 256 (one float4 per lane per 64-column chunk)
constexpr size_t MAX_IDS = (size_t)CMAX * TILE / 2;  // ids per call: a key present in every tile has n/TILE parts and
                                                     // must fit one merge pass together with the keys sharing the pass
constexpr unsigned SKIP = 0xffffffffu;
constexpr unsigned char F_HEAD = 1, F_SINGLE = 2;
constexpr unsigned CSTRIDE = 32;  // u32 words between bucket cursors: one 128-B line each (atomics on
                                  // neighbouring words of one line serialise: 18 us -> 3 us in kernel A)

template <int NCH> struct Batch { static constexpr int v = NCH == 1 ? 8 : (NCH == 2 ? 4 : 2); };

// ---------------------------------------------------------------------------------------------
// Bitonic sort of NTH*EPT 32-bit keys held in registers (element index i = thread*EPT + r).
// Distances < EPT stay inside a thread, < 64*EPT inside a wave (ds_bpermute shuffles, no barrier),
// only the last log2(NTH/64) distances of a round go through LDS with block barriers.
// (An LDS-resident version with a __syncthreads per stage cost 14-20 us per launch here.)
template <int NTH, int EPT, bool KV>
__device__ __forceinline__ void reg_bitonic_impl(unsigned (&x)[EPT], unsigned (&v)[EPT], unsigned* s_tmp, unsigned* s_tmpv,
                                                 int n2) {
  const int t = threadIdx.x;
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (EPT > 1 && j == 1) {  // constant register indices only (a runtime index would spill to scratch)
#pragma unroll
        for (int r = 0; r < EPT; r += 2) {
          bool up = (((t * EPT + r) & k) == 0);
          unsigned a = x[r], b = x[r + 1], av = v[r], bv = v[r + 1];
          bool sw = up ? (b < a) : (a < b);
          x[r] = sw ? b : a; x[r + 1] = sw ? a : b;
          if (KV) { v[r] = sw ? bv : av; v[r + 1] = sw ? av : bv; }
        }
      } else if (EPT > 2 && j == 2) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          bool up = (((t * EPT + r) & k) == 0);
          unsigned a = x[r], b = x[r + 2], av = v[r], bv = v[r + 2];
          bool sw = up ? (b < a) : (a < b);
          x[r] = sw ? b : a; x[r + 2] = sw ? a : b;
          if (KV) { v[r] = sw ? bv : av; v[r + 2] = sw ? av : bv; }
        }
      } else if (j < 64 * EPT) {
        const int dl = j / EPT;
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
          int i = t * EPT + r;
          unsigned y = (unsigned)__shfl_xor((int)x[r], dl);
          unsigned yv = KV ? (unsigned)__shfl_xor((int)v[r], dl) : 0u;
          bool keep_min = (((i & k) == 0) == ((i & j) == 0));
          bool take = keep_min ? (y < x[r]) : (y > x[r]);
          x[r] = take ? y : x[r];
          if (KV) v[r] = take ? yv : v[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < EPT; ++r) { s_tmp[t * EPT + r] = x[r]; if (KV) s_tmpv[t * EPT + r] = v[r]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
          int i = t * EPT + r;
          unsigned y = s_tmp[i ^ j];
          unsigned yv = KV ? s_tmpv[i ^ j] : 0u;
          bool keep_min = (((i & k) == 0) == ((i & j) == 0));
          bool take = keep_min ? (y < x[r]) : (y > x[r]);
          x[r] = take ? y : x[r];
          if (KV) v[r] = take ? yv : v[r];
        }
        __syncthreads();
      }
    }
  }
}

template <int NTH, int EPT>
__device__ __forceinline__ void reg_bitonic(unsigned (&x)[EPT], unsigned* s_tmp, int n2) {
  unsigned dummy[EPT] = {};
  reg_bitonic_impl<NTH, EPT, false>(x, dummy, s_tmp, nullptr, n2);
}

// key-value variant: keys must be unique (no tie-breaking on the payload)
template <int NTH, int EPT>
__device__ __forceinline__ void reg_bitonic_kv(unsigned (&x)[EPT], unsigned (&v)[EPT], unsigned* s_tmp, unsigned* s_tmpv,
                                               int n2) {
  reg_bitonic_impl<NTH, EPT, true>(x, v, s_tmp, s_tmpv, n2);
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
template <int NCH, int NG, int BATCH, class RowOf, class OutOf>
__device__ __forceinline__ void ordered_run_sums(int n, int span, int dim, const unsigned char* s_flag,
                                                 float (*s_left)[64 * NCH], unsigned char* s_cont,
                                                 unsigned char* s_hashead, RowOf row_of, OutOf out_of) {
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

// Descriptor store: bucket b owns entries [b*CMAX, b*CMAX+CMAX); cursor[b] counts every append,
// appends beyond CMAX go to the overflow list (rare: several very hot keys hashing together).
struct DescStore {
  i64* key;
  unsigned* src;
  unsigned* ord;       // tile << 9 | unique rank inside the tile: a deterministic id of the descriptor
  unsigned* cursor;    // [P] (this call's parity)
  i64* ovf_key;
  unsigned* ovf_src;
  unsigned* ovf_ord;
  unsigned* ovf_bucket;
  unsigned* ovf_count;
  unsigned ovf_cap;
};

// ---------------------------------------------------------------------------------------------
// kernel A.  src < rows_base -> gradient row `src` of the caller's buffer, else scratch row
// (src-rows_base).  No sort: a key's FIRST position in the tile is its deterministic group id, so
// unique ranks and the order of the multi-member groups come from block scans in position order
// (= tf.unique order); the members of a multi-member group are ranked by popcounts over a
// 512-bit position mask.
// PLAN: the id-only half (tfra_sparse_plan_build) — everything but the run sums; the run structure
// (member position | head flag | unique rank, one u32 per list entry) is stored for tile_sums_kernel.
template <int NCH, bool PLAN>
__global__ __launch_bounds__(NTA) void tile_reduce_kernel(size_t n, const i64* __restrict__ ids,
                                                          const float* __restrict__ grads, int dim, unsigned P,
                                                          unsigned rows_base, DescStore ds,
                                                          float* __restrict__ scratch_rows, unsigned* overflow, int stop,
                                                          unsigned* __restrict__ plan_entries,
                                                          unsigned* __restrict__ plan_len) {
  constexpr int NG = NTA / 16;
  constexpr unsigned GCAP = 2048;            // group table: 4x the tile => short probe chains
  constexpr int W = TILE / 32;               // words of a position bitmask
  constexpr int GM = GCAP / W;               // multi-member groups ranked per round (mask area = s_owner)
  __shared__ i64 s_key[TILE];                // ids of the tile, input order
  __shared__ unsigned s_owner[GCAP];         // group table; afterwards the position masks
  __shared__ unsigned s_rep[GCAP];           // smallest input position of the group; later (m << 16 | base)
  __shared__ unsigned s_cnt[GCAP];           // members per group
  __shared__ unsigned short s_list[TILE + 1];  // positions of the multi-member groups, run after run
  __shared__ unsigned short s_u[TILE];       // unique rank (within the tile) of the run starting at a list position
  __shared__ unsigned char s_flag[TILE + 1];
  __shared__ float s_left[NG][64 * NCH];
  __shared__ unsigned char s_cont[NG], s_hashead[NG];
  __shared__ int s_scan[NTA / 64];
  unsigned* const s_mask = s_owner;
  const size_t tile = blockIdx.x, base = tile * TILE;
  const int nvalid = (int)min((size_t)TILE, n - base);
  static_assert(TILE == NTA && TILE == 512, "one id per thread; 9 position bits");
  const int p = threadIdx.x;
  const bool ok = p < nvalid;
  const i64 key = ok ? ids[base + p] : 0;
  s_key[p] = key;
  for (unsigned b = threadIdx.x; b < GCAP; b += NTA) { s_owner[b] = 0; s_rep[b] = 0xffffffffu; s_cnt[b] = 0; }
  __syncthreads();
  if (stop == 1) return;  // (tuning ablation, TFRA_DBG_STOP_A)
  // Which hash slot a key lands in depends on the race between colliding keys, so the slot never
  // influences an order: a group is identified by its first input position.
  unsigned slot = 0;
  if (ok) {
    slot = lds_group_slot(s_key, s_owner, GCAP, p, fmix64((u64)key));
    atomicMin(&s_rep[slot], (unsigned)p);
    atomicAdd(&s_cnt[slot], 1u);
  }
  __syncthreads();
  const bool head = ok && s_rep[slot] == (unsigned)p;
  const unsigned cnt = ok ? s_cnt[slot] : 0;
  const bool single = head && cnt == 1, mhead = head && cnt > 1;
  int n_unique, n_multi, L;
  const int u = block_excl_scan<NTA>(head ? 1 : 0, s_scan, &n_unique);      // tf.unique rank
  const int m = block_excl_scan<NTA>(mhead ? 1 : 0, s_scan, &n_multi);      // multi groups in position order
  const int lbase = block_excl_scan<NTA>(mhead ? (int)cnt : 0, s_scan, &L);  // their runs in the member list
  if (stop == 2) return;
  // descriptor append: the returning atomic is issued now, its result is consumed after the ranking
  // rounds below so that its latency overlaps them
  unsigned desc_pos = 0, desc_bucket = 0;
  if (head) {
    desc_bucket = (unsigned)__umul64hi(fmix64((u64)key), (u64)P);
    desc_pos = atomicAdd(&ds.cursor[(size_t)desc_bucket * CSTRIDE], 1u);
  }
  s_flag[p] = 0;
  if (mhead) { s_rep[slot] = ((unsigned)m << 16) | (unsigned)lbase; s_u[lbase] = (unsigned short)u; }
  __syncthreads();  // s_rep now maps slot -> (m, base) for multi groups; s_owner is free
  for (int c0 = 0; c0 < n_multi; c0 += GM) {
    for (unsigned q = threadIdx.x; q < GCAP; q += NTA) s_mask[q] = 0;
    __syncthreads();
    const int mm = (ok && cnt > 1) ? (int)(s_rep[slot] >> 16) - c0 : -1;
    if (mm >= 0 && mm < GM) atomicOr(&s_mask[mm * W + (p >> 5)], 1u << (p & 31));
    __syncthreads();
    if (mm >= 0 && mm < GM) {
      int rank = __popc(s_mask[mm * W + (p >> 5)] & ((1u << (p & 31)) - 1u));
      for (int wds = 0; wds < (p >> 5); ++wds) rank += __popc(s_mask[mm * W + wds]);
      int pos = (int)(s_rep[slot] & 0xffffu) + rank;  // ascending input position inside the run
      s_list[pos] = (unsigned short)p;
      if (rank == 0) s_flag[pos] = F_HEAD;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) s_flag[L] = F_HEAD;
  __syncthreads();
  if (head) {
    const unsigned src = single ? (unsigned)(base + p) : rows_base + (unsigned)(base + u);
    const unsigned ordv = ((unsigned)tile << 9) | (unsigned)u;
    if (desc_pos < CMAX) {
      size_t d = (size_t)desc_bucket * CMAX + desc_pos;
      ds.key[d] = key; ds.src[d] = src; ds.ord[d] = ordv;
    } else {
      unsigned o = atomicAdd(ds.ovf_count, 1u);
      if (o < ds.ovf_cap) { ds.ovf_key[o] = key; ds.ovf_src[o] = src; ds.ovf_ord[o] = ordv; ds.ovf_bucket[o] = desc_bucket; }
      else atomicAdd(overflow, 1u);
    }
  }
  if (stop == 3) return;
  if (PLAN) {
    if (threadIdx.x == 0) plan_len[tile] = (unsigned)L;
    for (int q = threadIdx.x; q < L; q += NTA)
      plan_entries[base + q] = (unsigned)s_list[q] | ((unsigned)(s_flag[q] & F_HEAD) << 9) | ((unsigned)s_u[q] << 16);
    return;
  }
  if (L > 0) {
    ordered_run_sums<NCH, NG, 2 * Batch<NCH>::v>(  // one block per CU: registers for 16 rows in flight are free
        L, (L + NG - 1) / NG, dim, s_flag, s_left, s_cont, s_hashead,
        [&](int q) { return grads + (base + s_list[q]) * (size_t)dim; },
        [&](int ph) { return scratch_rows + (base + (size_t)s_u[ph]) * (size_t)dim; });
  }
}

// Last kernel of a plan build (behind the two plan kernels on the same stream): re-arms the bucket cursors and
// the overflow counter for the NEXT build of this plan (saves a memset launch per build), latches the error
// count for the gradient half, and publishes the build's generation to the host (pinned memory).
__global__ __launch_bounds__(256) void plan_finish_kernel(unsigned* cursors, unsigned P, unsigned* err_latched,
                                                          unsigned* built_host, unsigned gen) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i == P + 1) { *err_latched = cursors[(size_t)i * CSTRIDE]; cursors[(size_t)i * CSTRIDE] = 0; }  // error word
  else if (i <= P) cursors[(size_t)i * CSTRIDE] = 0;                                             // cursors + overflow count
  if (i == 0 && built_host) __hip_atomic_store(built_host, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// gradient half of kernel A for a planned batch: the run sums of one tile, following the stored run
// structure (same chunking as the fused kernel => the same summation tree, bit for bit).
template <int NCH>
__global__ __launch_bounds__(NTA) void tile_sums_kernel(const float* __restrict__ grads, int dim,
                                                        const unsigned* __restrict__ plan_entries,
                                                        const unsigned* __restrict__ plan_len,
                                                        float* __restrict__ scratch_rows, unsigned* progress,
                                                        unsigned progress_val) {
  constexpr int NG = NTA / 16;
  __shared__ unsigned short s_list[TILE + 1];
  __shared__ unsigned short s_u[TILE];
  // tfra_table_step_prefetch: host-visible progress counter (pinned memory) — this kernel running means the
  // lookup of step `progress_val` and every earlier step of the main stream are complete
  if (progress && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(progress, progress_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __shared__ unsigned char s_flag[TILE + 1];
  __shared__ float s_left[NG][64 * NCH];
  __shared__ unsigned char s_cont[NG], s_hashead[NG];
  const size_t tile = blockIdx.x, base = tile * TILE;
  const int L = (int)plan_len[tile];
  if (L == 0) return;
  for (int q = threadIdx.x; q < L; q += NTA) {
    const unsigned e = plan_entries[base + q];
    s_list[q] = (unsigned short)(e & 511u);
    s_flag[q] = (unsigned char)((e >> 9) & 1u);
    s_u[q] = (unsigned short)(e >> 16);
  }
  if (threadIdx.x == 0) s_flag[L] = F_HEAD;
  __syncthreads();
  ordered_run_sums<NCH, NG, 2 * Batch<NCH>::v>(
      L, (L + NG - 1) / NG, dim, s_flag, s_left, s_cont, s_hashead,
      [&](int q) { return grads + (base + s_list[q]) * (size_t)dim; },
      [&](int ph) { return scratch_rows + (base + (size_t)s_u[ph]) * (size_t)dim; });
}

// ---------------------------------------------------------------------------------------------
// kernel C.  Output: bucket b owns u_keys/u_src[off_b .. off_b+n_b), n_b = cursor[b], off_b =
// sum of the cursors of smaller buckets; the first (#unique in bucket) entries are filled, the rest
// are SKIP.  Summed rows go to scratch row (sum_base - rows_base + index).  The last bucket
// publishes the total entry count.
// A bucket with more than CMAX descriptors is processed in 2^k passes, pass q taking the keys with
// (hash & (2^k-1)) == q, reading its region plus its share of the overflow list.
// PLAN: the id-only half — (key, src) outputs as usual, no sums; per pass with multi-part keys the member
// list is stored as (src of the member, head flag | scratch row of the run's sum) for bucket_sums_kernel.
struct BucketPlan {
  unsigned* src;      // [d_total slots] src of list entry (off_b + position)
  unsigned* outf;     // [d_total slots] bit 31 = run head, low bits = scratch row receiving the run's sum
  uint2* pass;        // [P * MAXPASS] (start inside the bucket's slot range, L) of every pass with L > 0
  unsigned* npass;    // [P]
  unsigned* off;      // [P] first slot of the bucket
};
constexpr int MAXPASS = 64;

template <int NCH, bool PLAN>
__global__ __launch_bounds__(NT) void bucket_merge_kernel(unsigned P, unsigned ntiles, int dim, unsigned rows_base,
                                                          unsigned sum_base,
                                                          const float* __restrict__ grads, DescStore ds,
                                                          float* __restrict__ scratch_rows, i64* __restrict__ u_keys,
                                                          unsigned* __restrict__ u_src, i64* __restrict__ d_total,
                                                          unsigned* overflow, int stop, uint2* __restrict__ bucket_out,
                                                          BucketPlan bp) {
  constexpr int NG = NT / 16;
  constexpr unsigned GCAP = CMAX;   // group table as large as the pass (LDS budget: 4 blocks per CU)
  static_assert(CMAX == 1024, "10 entry bits");
  __shared__ i64 e_key[CMAX];
  __shared__ unsigned e_src[CMAX];
  __shared__ unsigned e_ord[CMAX];      // descriptor ids (tile << 9 | rank in tile)
  __shared__ unsigned s_sort[CMAX];     // group slot of each gathered entry
  __shared__ unsigned short s_list[CMAX + 1];  // entries of the multi-part groups, run after run, tile order
  __shared__ unsigned s_skey_m[256];    // (group id << 11 | slot) of the multi-part groups
  __shared__ unsigned s_owner[GCAP];
  __shared__ unsigned s_rep[GCAP];      // smallest descriptor id of the group (deterministic)
  __shared__ unsigned char s_flag[CMAX + 1];
  __shared__ unsigned short s_rank[CMAX];  // unique rank (within the pass) of each head position
  __shared__ float s_left[NG][64 * NCH];
  __shared__ unsigned char s_cont[NG], s_hashead[NG];
  __shared__ int s_scan[NT / 64];
  __shared__ long long s_red[NT / 64];
  const unsigned b = blockIdx.x;
  // output offset = sum of the cursors of the buckets before this one
  long long off = 0;
  {
    long long acc = 0;
    for (unsigned i = threadIdx.x; i < b; i += NT) acc += ds.cursor[(size_t)i * CSTRIDE];
    for (int o2 = 32; o2 > 0; o2 >>= 1) acc += __shfl_xor(acc, o2);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    off = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  }
  const int n_b = (int)ds.cursor[(size_t)b * CSTRIDE];
  const unsigned n_ovf = n_b > CMAX ? min(*ds.ovf_count, ds.ovf_cap) : 0u;
  __syncthreads();
  if (threadIdx.x == 0 && b == P - 1) *d_total = off + n_b;
  if (PLAN && threadIdx.x == 0) { bp.off[b] = (unsigned)off; bp.npass[b] = 0; }
  if (n_b == 0) {
    if (bucket_out && threadIdx.x == 0) bucket_out[b] = make_uint2((unsigned)off, 0u);
    return;
  }
  int plan_used = 0, plan_np = 0;  // PLAN: list slots and passes stored so far
  unsigned npass = 1;
  if (n_b > CMAX) while ((unsigned)n_b > (unsigned)(CMAX / 2) * npass && npass < 64) npass <<= 1;
  const int n_reg = min(n_b, CMAX);
  int out_used = 0;
  for (unsigned pass = 0; pass < npass; ++pass) {
    int n = 0;
    if (npass == 1) {  // the common case: one coalesced read of the bucket region
      for (int q = threadIdx.x; q < n_reg; q += NT) {
        size_t d = (size_t)b * CMAX + q;
        e_key[q] = ds.key[d]; e_src[q] = ds.src[d]; e_ord[q] = ds.ord[d];
      }
      n = n_reg;
      __syncthreads();
    } else {  // filtered gather from the region and from the overflow list
      int carry = 0;
      bool too_many = false;
      for (int q0 = 0; q0 < n_reg + (int)n_ovf; q0 += NT) {
        int q = q0 + threadIdx.x;
        i64 k = 0; unsigned sr = 0, od = 0; bool take = false;
        if (q < n_reg) {
          size_t d = (size_t)b * CMAX + q;
          k = ds.key[d]; sr = ds.src[d]; od = ds.ord[d]; take = true;
        } else if (q < n_reg + (int)n_ovf) {
          unsigned o = q - n_reg;
          if (ds.ovf_bucket[o] == b) { k = ds.ovf_key[o]; sr = ds.ovf_src[o]; od = ds.ovf_ord[o]; take = true; }
        }
        take = take && (((unsigned)(fmix64((u64)k) >> 40) & (npass - 1)) == pass);
        int tot;
        int ex = block_excl_scan<NT>(take ? 1 : 0, s_scan, &tot);
        if (carry + tot > CMAX) too_many = true;
        if (take && !too_many) { e_key[carry + ex] = k; e_src[carry + ex] = sr; e_ord[carry + ex] = od; }
        carry += tot;
      }
      __syncthreads();
      if (too_many) {  // one key alone exceeds CMAX parts: reported, not applied
        if (threadIdx.x == 0) atomicAdd(overflow, 1u);
        continue;
      }
      n = carry;
    }
    if (n == 0) continue;
    if (stop == 1) continue;  // (tuning ablation, TFRA_DBG_STOP_C)
    // ---- group by key: slot (arrival-order dependent, never used for ordering), deterministic
    //      group id = smallest descriptor id of the group, member count -------------------------
    unsigned* const s_cnt = reinterpret_cast<unsigned*>(&s_left[0][0]);  // [GCAP] until the run sums
    unsigned* const s_mask = s_owner;                                    // [GCAP] once grouping is done
    for (unsigned q = threadIdx.x; q < GCAP; q += NT) { s_owner[q] = 0; s_rep[q] = 0xffffffffu; s_cnt[q] = 0; }
    __syncthreads();
    for (int q = threadIdx.x; q < n; q += NT) {
      unsigned slot = lds_group_slot(e_key, s_owner, GCAP, q, fmix64((u64)e_key[q]));
      atomicMin(&s_rep[slot], e_ord[q]);
      atomicAdd(&s_cnt[slot], 1u);
      s_sort[q] = slot;
    }
    __syncthreads();
    // ---- the unique keys of this pass: singles first (gather order), then the multi-part groups
    //      in ascending group id.  Which output slot a key gets does not affect any sum. ---------
    const int W = (int)((ntiles + 31) >> 5);     // words of a tile bitmask
    const int GM = min(128, (int)GCAP / W);      // multi-part groups that fit the mask area
    int n_single = 0, n_multi = 0;
    for (int pb = 0; pb < n; pb += NT) {
      int q = pb + threadIdx.x;
      unsigned slot = q < n ? s_sort[q] : 0;
      bool hd = q < n && e_ord[q] == s_rep[slot];
      bool single = hd && s_cnt[slot] == 1;
      bool mhead = hd && !single;
      int ts, tm;
      int es = block_excl_scan<NT>(single ? 1 : 0, s_scan, &ts);
      int em = block_excl_scan<NT>(mhead ? 1 : 0, s_scan, &tm);
      if (single) {
        long long o = off + out_used + n_single + es;
        u_keys[o] = e_key[q];
        u_src[o] = e_src[q];
      }
      if (mhead && n_multi + em < NT) s_skey_m[n_multi + em] = (s_rep[slot] << 11) | slot;  // (id, slot) to be ordered
      n_single += ts;
      n_multi += tm;
    }
    __syncthreads();
    if (n_multi > NT) {  // more multi-part keys than one per thread: split the bucket further by key hash
      if (npass < 64) { npass <<= 1; pass = (unsigned)-1; out_used = 0; plan_used = 0; plan_np = 0; }
      else if (threadIdx.x == 0) atomicAdd(overflow, 1u);
      continue;
    }
    int L = 0;  // total members of multi-part groups
    if (n_multi > 0) {
      // order the (few) multi groups by id, give each its run [base, base+cnt) and mask row m
      int m2 = 2;
      while (m2 < n_multi) m2 <<= 1;
      unsigned x[1] = {threadIdx.x < n_multi ? s_skey_m[threadIdx.x] : 0xffffffffu};
      __syncthreads();
      reg_bitonic<NT, 1>(x, s_skey_m, m2);
      __syncthreads();
      const bool live = threadIdx.x < n_multi;
      const unsigned slot_m = x[0] & 2047u;
      int tot;
      int base = block_excl_scan<NT>(live ? (int)s_cnt[slot_m] : 0, s_scan, &tot);
      L = tot;
      if (live) {
        s_rep[slot_m] = ((unsigned)threadIdx.x << 16) | (unsigned)base;  // slot -> (m, base)
        long long o = off + out_used + n_single + threadIdx.x;
        u_keys[o] = e_key[s_owner[slot_m] - 1];
        u_src[o] = sum_base + (unsigned)o;
        s_rank[base] = (unsigned short)threadIdx.x;
      }
      for (int q = threadIdx.x; q <= L; q += NT) s_flag[q] = 0;
      __syncthreads();
      // rank of a member inside its run = number of members from smaller tiles: a tile bitmask per
      // group (a key has at most one part per tile) + popcounts; GM groups at a time
      for (int c0 = 0; c0 < n_multi; c0 += GM) {
        for (int q = threadIdx.x; q < GM * W; q += NT) s_mask[q] = 0;
        __syncthreads();
        for (int q = threadIdx.x; q < n; q += NT) {
          unsigned slot = s_sort[q];
          if (s_cnt[slot] >= 2) {
            int m = (int)(s_rep[slot] >> 16) - c0;
            unsigned tile = e_ord[q] >> 9;
            if (m >= 0 && m < GM) atomicOr(&s_mask[m * W + (tile >> 5)], 1u << (tile & 31));
          }
        }
        __syncthreads();
        for (int q = threadIdx.x; q < n; q += NT) {
          unsigned slot = s_sort[q];
          if (s_cnt[slot] >= 2) {
            unsigned mb = s_rep[slot];
            int m = (int)(mb >> 16) - c0;
            if (m >= 0 && m < GM) {
              unsigned tile = e_ord[q] >> 9;
              int rank = __popc(s_mask[m * W + (tile >> 5)] & ((1u << (tile & 31)) - 1u));
              for (unsigned wds = 0; wds < (tile >> 5); ++wds) rank += __popc(s_mask[m * W + wds]);
              int pos = (int)(mb & 0xffffu) + rank;   // tile order inside the run
              s_list[pos] = (unsigned short)q;
              if (rank == 0) s_flag[pos] = F_HEAD;
            }
          }
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) s_flag[L] = F_HEAD;
      __syncthreads();
    }
    const long long obase = off + out_used + n_single;
    const int ccarry = n_single + n_multi;
    if (stop == 2 || stop == 3) { out_used += ccarry; continue; }
    if (PLAN) {
      if (L > 0) {
        for (int q = threadIdx.x; q < L; q += NT) {
          bp.src[off + plan_used + q] = e_src[s_list[q]];
          bp.outf[off + plan_used + q] =
              (s_flag[q] & F_HEAD) ? (0x80000000u | (sum_base - rows_base + (unsigned)(obase + s_rank[q]))) : 0u;
        }
        if (threadIdx.x == 0) bp.pass[(size_t)b * MAXPASS + plan_np] = make_uint2((unsigned)plan_used, (unsigned)L);
        plan_used += L;
        plan_np += 1;
      }
      out_used += ccarry;
      __syncthreads();
      continue;
    }
    if (L > 0) {
      ordered_run_sums<NCH, NG, Batch<NCH>::v>(
          L, (L + NG - 1) / NG, dim, s_flag, s_left, s_cont, s_hashead,
          [&](int q) {
            unsigned src = e_src[s_list[q]];
            return src < rows_base ? grads + (size_t)src * dim : scratch_rows + (size_t)(src - rows_base) * dim;
          },
          [&](int ph) { return scratch_rows + (size_t)(sum_base - rows_base + (unsigned)(obase + s_rank[ph])) * dim; });
    }
    out_used += ccarry;
    __syncthreads();
  }
  for (int i = out_used + threadIdx.x; i < n_b; i += NT) u_src[off + i] = SKIP;
  if (bucket_out && threadIdx.x == 0) bucket_out[b] = make_uint2((unsigned)off, (unsigned)out_used);
  if (PLAN && threadIdx.x == 0) bp.npass[b] = (unsigned)plan_np;
}

// gradient half of kernel C for a planned batch: per pass the ordered sums of the multi-part keys'
// parts (same chunking as the fused kernel => the same summation tree).
template <int NCH>
__global__ __launch_bounds__(NT) void bucket_sums_kernel(int dim, unsigned rows_base, const float* __restrict__ grads,
                                                         float* __restrict__ scratch_rows, BucketPlan bp,
                                                         const unsigned* __restrict__ plan_err, unsigned* table_err) {
  constexpr int NG = NT / 16;
  __shared__ unsigned e_src[CMAX];
  __shared__ unsigned s_out[CMAX];
  __shared__ unsigned char s_flag[CMAX + 1];
  __shared__ float s_left[NG][64 * NCH];
  __shared__ unsigned char s_cont[NG], s_hashead[NG];
  const unsigned b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0 && *plan_err) atomicAdd(table_err, *plan_err);
  const unsigned np = bp.npass[b];
  if (np == 0) return;
  const size_t off = bp.off[b];
  for (unsigned ps = 0; ps < np; ++ps) {
    const uint2 pl = bp.pass[(size_t)b * MAXPASS + ps];
    const int L = (int)pl.y;
    for (int q = threadIdx.x; q < L; q += NT) {
      e_src[q] = bp.src[off + pl.x + q];
      const unsigned of = bp.outf[off + pl.x + q];
      s_out[q] = of & 0x7fffffffu;
      s_flag[q] = (of >> 31) ? F_HEAD : 0;
    }
    if (threadIdx.x == 0) s_flag[L] = F_HEAD;
    __syncthreads();
    ordered_run_sums<NCH, NG, Batch<NCH>::v>(
        L, (L + NG - 1) / NG, dim, s_flag, s_left, s_cont, s_hashead,
        [&](int q) {
          unsigned src = e_src[q];
          return src < rows_base ? grads + (size_t)src * dim : scratch_rows + (size_t)(src - rows_base) * dim;
        },
        [&](int ph) { return scratch_rows + (size_t)s_out[ph] * dim; });
    __syncthreads();
  }
}

// tfra_reduce_by_key epilogue: bucket b's unique keys (u_keys/u_src[off_b .. off_b+cnt_b)) go to the
// dense output at sum(cnt of the buckets before b); one 16-lane group per key copies the summed row.
__global__ __launch_bounds__(256) void compact_gather_kernel(unsigned P, int dim, unsigned rows_base,
                                                             const float* __restrict__ grads,
                                                             const float* __restrict__ scratch_rows,
                                                             const i64* __restrict__ u_keys,
                                                             const unsigned* __restrict__ u_src,
                                                             const uint2* __restrict__ bucket_out,
                                                             const unsigned* __restrict__ err, i64* __restrict__ keys_out,
                                                             float* __restrict__ rows_out, i64* __restrict__ d_count) {
  __shared__ unsigned s_red[4];
  const unsigned b = blockIdx.x;
  unsigned acc = 0;
  for (unsigned i = threadIdx.x; i < b; i += 256) acc += bucket_out[i].y;
  for (int o2 = 32; o2 > 0; o2 >>= 1) acc += __shfl_xor(acc, o2);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const unsigned before = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const uint2 me = bucket_out[b];
  if (b == P - 1 && threadIdx.x == 0) *d_count = *err ? (i64)-1 : (i64)(before + me.y);
  // Inside the bucket the merge kernel leaves single-part keys in descriptor-arrival order (atomic
  // cursors: differs from run to run).  Here the bucket's keys are written in ASCENDING KEY order, so
  // the output order is a function of the id set alone: rank = number of smaller keys in the bucket.
  constexpr unsigned LCAP = 2048;
  __shared__ i64 s_k[LCAP];
  const bool in_lds = me.y <= LCAP;
  if (in_lds) for (unsigned i = threadIdx.x; i < me.y; i += 256) s_k[i] = u_keys[me.x + i];
  __syncthreads();
  const int sub = threadIdx.x & 15;
  for (unsigned i = threadIdx.x >> 4; i < me.y; i += 16) {
    const i64 key = in_lds ? s_k[i] : u_keys[me.x + i];
    unsigned r = 0;
    for (unsigned j = sub; j < me.y; j += 16) r += (in_lds ? s_k[j] : u_keys[me.x + j]) < key;
    for (int o2 = 8; o2 > 0; o2 >>= 1) r += __shfl_xor(r, o2);
    const unsigned src = u_src[me.x + i];
    const float* row = src < rows_base ? grads + (size_t)src * dim : scratch_rows + (size_t)(src - rows_base) * dim;
    float* out = rows_out + (size_t)(before + r) * dim;
    for (int c = sub * 4; c < dim; c += 64) *reinterpret_cast<float4*>(out + c) = *reinterpret_cast<const float4*>(row + c);
    if (sub == 0) keys_out[before + r] = key;
  }
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
  if (n > MAX_IDS)
    return set_error(TFRA_ERR_UNSUPPORTED, "apply_sparse: at most 2^18 ids per call (one partial sum per key and 512-id "
                                           "tile; a merge pass holds 1024 partial sums, so a key in every tile must leave "
                                           "room for the keys sharing its pass); split the batch or use the unique + "
                                           "segment_sum path");
  rc = t->prepare_insert(n, s);
  if (rc) return rc;
  const size_t ntiles = (n + TILE - 1) / TILE, npad = ntiles * TILE;
  unsigned P = 64;
  while (P < 2048 && (size_t)P * 128 < n) P <<= 1;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  // scratch: cursors[P+1] | region key/src/ord [P*CMAX] | overflow key/src/ord/bucket [npad] |
  //          u_keys/u_src [npad] | d_total | rows [2*npad][dim]
  const size_t reg = (size_t)P * CMAX;
  size_t bytes = al((size_t)(P + 1) * CSTRIDE * 4) + al(reg * 8) + 2 * al(reg * 4) + al(npad * 8) + 3 * al(npad * 4) + al(npad * 8) +
                 al(npad * 4) + 256 + 2 * al(npad * (size_t)dim * 4);
  const bool fresh = t->scratch_bytes < bytes || t->apply_P != P;
  rc = t->ensure_scratch(bytes, s);
  if (rc) return rc;
  unsigned char* w = (unsigned char*)t->scratch;
  unsigned* cursors = (unsigned*)w; w += al((size_t)(P + 1) * CSTRIDE * 4);
  if (fresh) {  // cursors start at zero; afterwards the apply kernel of every call re-arms them
    if (hipMemsetAsync(cursors, 0, (size_t)(P + 1) * CSTRIDE * 4, s) != hipSuccess) return set_error(TFRA_ERR_HIP, "apply_sparse: memset");
    t->apply_P = P;
  }
  DescStore ds;
  ds.key = (i64*)w; w += al(reg * 8);
  ds.src = (unsigned*)w; w += al(reg * 4);
  ds.ord = (unsigned*)w; w += al(reg * 4);
  ds.ovf_key = (i64*)w; w += al(npad * 8);
  ds.ovf_src = (unsigned*)w; w += al(npad * 4);
  ds.ovf_ord = (unsigned*)w; w += al(npad * 4);
  ds.ovf_bucket = (unsigned*)w; w += al(npad * 4);
  i64* u_keys = (i64*)w; w += al(npad * 8);
  unsigned* u_src = (unsigned*)w; w += al(npad * 4);
  i64* d_total = (i64*)w; w += 256;
  float* rows = (float*)w;
  ds.cursor = cursors;
  ds.ovf_count = cursors + (size_t)P * CSTRIDE;
  ds.ovf_cap = (unsigned)npad;
  const unsigned rows_base = (unsigned)npad, sum_base = (unsigned)(2 * npad);
  const int nch = (dim + 63) / 64;
  const i64* k = (const i64*)ids;
  dim3 ga((unsigned)ntiles), gc(P);
#ifdef TFRA_WITH_TUNING  // phase ablation for scripts/ablate_apply.sh (build with TFRA_WITH_TUNING=1); never in the product build
  static const int stop_a = getenv("TFRA_DBG_STOP_A") ? atoi(getenv("TFRA_DBG_STOP_A")) : 0;
  static const int stop_c = getenv("TFRA_DBG_STOP_C") ? atoi(getenv("TFRA_DBG_STOP_C")) : 0;
#else
  constexpr int stop_a = 0, stop_c = 0;
#endif
  switch (nch) {
    case 1: tile_reduce_kernel<1, false><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, ds, rows, t->err_count, stop_a, nullptr, nullptr); break;
    case 2: tile_reduce_kernel<2, false><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, ds, rows, t->err_count, stop_a, nullptr, nullptr); break;
    case 3: tile_reduce_kernel<3, false><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, ds, rows, t->err_count, stop_a, nullptr, nullptr); break;
    default: tile_reduce_kernel<4, false><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, ds, rows, t->err_count, stop_a, nullptr, nullptr); break;
  }
  switch (nch) {
    case 1: bucket_merge_kernel<1, false><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, ds, rows, u_keys, u_src, d_total, t->err_count, stop_c, nullptr, BucketPlan{}); break;
    case 2: bucket_merge_kernel<2, false><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, ds, rows, u_keys, u_src, d_total, t->err_count, stop_c, nullptr, BucketPlan{}); break;
    case 3: bucket_merge_kernel<3, false><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, ds, rows, u_keys, u_src, d_total, t->err_count, stop_c, nullptr, BucketPlan{}); break;
    default: bucket_merge_kernel<4, false><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, ds, rows, u_keys, u_src, d_total, t->err_count, stop_c, nullptr, BucketPlan{}); break;
  }
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "apply_sparse: launch failed");
  // the apply kernel runs after every merge block has read the cursors: it zeroes them for the next
  // call (no host-side state, so the three launches can be captured into a HIP graph and replayed)
  return launch_apply_indirect(t, s, p, npad, u_keys, u_src, grads, rows, rows_base, param_default_row, d_total, cursors,
                               P + 1, CSTRIDE);
}

// unique + unsorted_segment_sum in one call = kernels A and C above + a compaction.
extern "C" int tfra_reduce_by_key(tfra_workspace_t* ws, size_t n, const int64_t* ids, int dim, const float* grads,
                                  int64_t* keys_out, float* rows_out, int64_t* d_count, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || !d_count) return set_error(TFRA_ERR_INVALID, "reduce_by_key: null argument");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) { if (hipSetDevice(ws->device) != hipSuccess) return set_error(TFRA_ERR_HIP, "reduce_by_key: hipSetDevice"); } }
  if (n == 0) {
    if (hipMemsetAsync(d_count, 0, sizeof(int64_t), s) != hipSuccess) return set_error(TFRA_ERR_HIP, "reduce_by_key: memset");
    return TFRA_OK;
  }
  if (!ids || !grads || !keys_out || !rows_out) return set_error(TFRA_ERR_INVALID, "reduce_by_key: null buffer");
  if (dim <= 0 || dim % 4 != 0 || dim > 64 * MAXCH || (((uintptr_t)grads | (uintptr_t)rows_out) & 15))
    return set_error(TFRA_ERR_UNSUPPORTED, "reduce_by_key: needs dim % 4 == 0, dim <= 256 and 16-B aligned buffers "
                                           "(use tfra_unique + tfra_segment_sum otherwise)");
  if (n > MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "reduce_by_key: at most 2^18 ids per call");
  const size_t ntiles = (n + TILE - 1) / TILE, npad = ntiles * TILE;
  unsigned P = 64;
  while (P < 2048 && (size_t)P * 128 < n) P <<= 1;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t reg = (size_t)P * CMAX;
  // cursors[P] | overflow count | error count (one 128-B line each) | bucket_out[P] | descriptor regions |
  // overflow list | u_keys/u_src | d_total | rows [2*npad][dim]
  const size_t head = al((size_t)(P + 2) * CSTRIDE * 4);
  size_t bytes = head + al((size_t)P * 8) + al(reg * 8) + 2 * al(reg * 4) + al(npad * 8) + 3 * al(npad * 4) + al(npad * 8) +
                 al(npad * 4) + 256 + 2 * al(npad * (size_t)dim * 4);
  int rc = ws->ensure(bytes, s);
  if (rc) return rc;
  unsigned char* w = (unsigned char*)ws->buf;
  unsigned* cursors = (unsigned*)w; w += head;
  if (hipMemsetAsync(cursors, 0, head, s) != hipSuccess) return set_error(TFRA_ERR_HIP, "reduce_by_key: memset");
  uint2* bucket_out = (uint2*)w; w += al((size_t)P * 8);
  DescStore ds;
  ds.key = (i64*)w; w += al(reg * 8);
  ds.src = (unsigned*)w; w += al(reg * 4);
  ds.ord = (unsigned*)w; w += al(reg * 4);
  ds.ovf_key = (i64*)w; w += al(npad * 8);
  ds.ovf_src = (unsigned*)w; w += al(npad * 4);
  ds.ovf_ord = (unsigned*)w; w += al(npad * 4);
  ds.ovf_bucket = (unsigned*)w; w += al(npad * 4);
  i64* u_keys = (i64*)w; w += al(npad * 8);
  unsigned* u_src = (unsigned*)w; w += al(npad * 4);
  i64* d_total = (i64*)w; w += 256;
  float* rows = (float*)w;
  ds.cursor = cursors;
  ds.ovf_count = cursors + (size_t)P * CSTRIDE;
  ds.ovf_cap = (unsigned)npad;
  unsigned* err = cursors + (size_t)(P + 1) * CSTRIDE;
  const unsigned rows_base = (unsigned)npad, sum_base = (unsigned)(2 * npad);
  const int nch = (dim + 63) / 64;
  const i64* k = (const i64*)ids;
  dim3 ga((unsigned)ntiles), gc(P);
  switch (nch) {
    case 1: tile_reduce_kernel<1, false><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, ds, rows, err, 0, nullptr, nullptr); break;
    case 2: tile_reduce_kernel<2, false><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, ds, rows, err, 0, nullptr, nullptr); break;
    case 3: tile_reduce_kernel<3, false><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, ds, rows, err, 0, nullptr, nullptr); break;
    default: tile_reduce_kernel<4, false><<<ga, NTA, 0, s>>>(n, k, grads, dim, P, rows_base, ds, rows, err, 0, nullptr, nullptr); break;
  }
  switch (nch) {
    case 1: bucket_merge_kernel<1, false><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, ds, rows, u_keys, u_src, d_total, err, 0, bucket_out, BucketPlan{}); break;
    case 2: bucket_merge_kernel<2, false><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, ds, rows, u_keys, u_src, d_total, err, 0, bucket_out, BucketPlan{}); break;
    case 3: bucket_merge_kernel<3, false><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, ds, rows, u_keys, u_src, d_total, err, 0, bucket_out, BucketPlan{}); break;
    default: bucket_merge_kernel<4, false><<<gc, NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, grads, ds, rows, u_keys, u_src, d_total, err, 0, bucket_out, BucketPlan{}); break;
  }
  compact_gather_kernel<<<gc, 256, 0, s>>>(P, dim, rows_base, grads, rows, u_keys, u_src, bucket_out, err, (i64*)keys_out, rows_out,
                                           (i64*)d_count);
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "reduce_by_key: launch failed");
  return TFRA_OK;
}

// ---------------------------------------------------------------------------------------------
// Planned write-back: the id-only half of tfra_table_apply_sparse (grouping, ranks, descriptors,
// (key, src) outputs, run structure) is built ahead — on any stream, typically concurrently with the
// lookup of the same ids or while the previous step is still running — and the gradient half
// (run sums of tiles, run sums of buckets, fused apply) follows it when the gradients exist.
struct tfra_sparse_plan {
  int device = 0;
  void* buf = nullptr;
  size_t bytes = 0;
  // filled by build
  size_t n = 0, npad = 0, ntiles = 0;
  unsigned P = 0;
  int dim = 0;
  unsigned* cursors = nullptr;
  unsigned* err = nullptr;
  unsigned* tile_entries = nullptr;
  unsigned* tile_len = nullptr;
  BucketPlan bp{};
  i64* u_keys = nullptr;
  unsigned* u_src = nullptr;
  i64* d_total = nullptr;
  float* rows = nullptr;
  // tfra_table_step_prefetch
  bool armed = false;              // cursors are zero (re-armed by plan_finish_kernel of the previous build)
  unsigned* err_latched = nullptr;
  unsigned* built_host = nullptr;  // pinned, device-visible: generation of the last build whose kernels have finished
  unsigned gen = 0;                // generation of the last enqueued build
  bool ev_recorded = false;        // the last build ran on a side stream (tfra_table_step_prefetch)
  unsigned last_used_step = 0;     // last step whose gradient half read this plan
};

extern "C" int tfra_sparse_plan_create(int device, tfra_sparse_plan_t** out) {
  if (!out) return set_error(TFRA_ERR_INVALID, "sparse_plan_create: null out");
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_create: no device");
  tfra_sparse_plan* pl = new tfra_sparse_plan();
  pl->device = device;
  *out = pl;
  return TFRA_OK;
}

extern "C" int tfra_sparse_plan_destroy(tfra_sparse_plan_t* pl) {
  if (!pl) return TFRA_OK;
  if (pl->buf) { (void)hipSetDevice(pl->device); (void)hipDeviceSynchronize(); (void)hipFree(pl->buf); }
  if (pl->built_host) (void)hipHostFree(pl->built_host);
  delete pl;
  return TFRA_OK;
}

extern "C" int tfra_sparse_plan_build(tfra_sparse_plan_t* pl, size_t n, const int64_t* ids, int dim, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pl) return set_error(TFRA_ERR_INVALID, "sparse_plan_build: null plan");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != pl->device) { if (hipSetDevice(pl->device) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: hipSetDevice"); } }
  pl->n = 0;
  if (n == 0) return TFRA_OK;
  if (!ids) return set_error(TFRA_ERR_INVALID, "sparse_plan_build: null ids");
  if (dim <= 0 || dim % 4 != 0 || dim > 64 * MAXCH)
    return set_error(TFRA_ERR_UNSUPPORTED, "sparse_plan_build: needs dim % 4 == 0 and dim <= 256");
  if (n > MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "sparse_plan_build: at most 2^18 ids per plan");
  const size_t ntiles = (n + TILE - 1) / TILE, npad = ntiles * TILE;
  unsigned P = 64;
  while (P < 2048 && (size_t)P * 128 < n) P <<= 1;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t reg = (size_t)P * CMAX;
  const size_t head = al((size_t)(P + 2) * CSTRIDE * 4);
  size_t bytes = head + al(reg * 8) + 2 * al(reg * 4) + al(npad * 8) + 3 * al(npad * 4)   // descriptors + overflow list
                 + al(npad * 8) + al(npad * 4) + 256                                       // u_keys, u_src, d_total
                 + al(npad * 4) + al(ntiles * 4)                                           // tile run structure
                 + 2 * al(npad * 4) + al((size_t)P * MAXPASS * 8) + 2 * al((size_t)P * 4)  // bucket run structure
                 + 2 * al(npad * (size_t)dim * 4);                                         // partial-sum rows
  if (pl->bytes < bytes) {
    if (pl->buf) {
      if (hipDeviceSynchronize() != hipSuccess || hipFree(pl->buf) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: free");
      pl->buf = nullptr; pl->bytes = 0;
    }
    hipError_t e = hipMalloc(&pl->buf, bytes);
    if (e != hipSuccess) { pl->buf = nullptr; return set_error(e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP, "sparse_plan_build: hipMalloc"); }
    pl->bytes = bytes;
    pl->armed = false;
  }
  unsigned char* w = (unsigned char*)pl->buf;
  pl->cursors = (unsigned*)w; w += head;
  const bool same_layout = pl->armed && pl->P == P;
  if (!same_layout && hipMemsetAsync(pl->cursors, 0, head, s) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: memset");
  DescStore ds;
  ds.key = (i64*)w; w += al(reg * 8);
  ds.src = (unsigned*)w; w += al(reg * 4);
  ds.ord = (unsigned*)w; w += al(reg * 4);
  ds.ovf_key = (i64*)w; w += al(npad * 8);
  ds.ovf_src = (unsigned*)w; w += al(npad * 4);
  ds.ovf_ord = (unsigned*)w; w += al(npad * 4);
  ds.ovf_bucket = (unsigned*)w; w += al(npad * 4);
  pl->u_keys = (i64*)w; w += al(npad * 8);
  pl->u_src = (unsigned*)w; w += al(npad * 4);
  pl->d_total = (i64*)w; w += 256;
  pl->tile_entries = (unsigned*)w; w += al(npad * 4);
  pl->tile_len = (unsigned*)w; w += al(ntiles * 4);
  pl->bp.src = (unsigned*)w; w += al(npad * 4);
  pl->bp.outf = (unsigned*)w; w += al(npad * 4);
  pl->bp.pass = (uint2*)w; w += al((size_t)P * MAXPASS * 8);
  pl->bp.npass = (unsigned*)w; w += al((size_t)P * 4);
  pl->bp.off = (unsigned*)w; w += al((size_t)P * 4);
  pl->rows = (float*)w;
  ds.cursor = pl->cursors;
  ds.ovf_count = pl->cursors + (size_t)P * CSTRIDE;
  ds.ovf_cap = (unsigned)npad;
  pl->err = pl->cursors + (size_t)(P + 1) * CSTRIDE;
  const unsigned rows_base = (unsigned)npad, sum_base = (unsigned)(2 * npad);
  tile_reduce_kernel<1, true><<<dim3((unsigned)ntiles), NTA, 0, s>>>(n, (const i64*)ids, nullptr, dim, P, rows_base, ds, nullptr,
                                                                     pl->err, 0, pl->tile_entries, pl->tile_len);
  bucket_merge_kernel<1, true><<<dim3(P), NT, 0, s>>>(P, (unsigned)ntiles, dim, rows_base, sum_base, nullptr, ds, nullptr,
                                                       pl->u_keys, pl->u_src, pl->d_total, pl->err, 0, nullptr, pl->bp);
  pl->gen += 1;
  pl->err_latched = pl->cursors + (size_t)(P + 1) * CSTRIDE + 1;   // same line as the error word, not re-armed
  plan_finish_kernel<<<dim3((P + 2 + 255) / 256), 256, 0, s>>>(pl->cursors, P, pl->err_latched, pl->built_host, pl->gen);
  pl->armed = true;
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: launch failed");
  pl->n = n; pl->npad = npad; pl->ntiles = ntiles; pl->P = P; pl->dim = dim;
  return TFRA_OK;
}

static int apply_planned_impl(tfra_table_t* tp, const tfra_opt_params* p, const tfra_sparse_plan_t* pl, const float* grads,
                              const float* param_default_row, tfra_stream_t stream, unsigned* progress, unsigned progress_val);

extern "C" int tfra_table_apply_planned(tfra_table_t* tp, const tfra_opt_params* p, const tfra_sparse_plan_t* pl,
                                        const float* grads, const float* param_default_row, tfra_stream_t stream) {
  return apply_planned_impl(tp, p, pl, grads, param_default_row, stream, nullptr, 0);
}

static int apply_planned_impl(tfra_table_t* tp, const tfra_opt_params* p, const tfra_sparse_plan_t* pl, const float* grads,
                              const float* param_default_row, tfra_stream_t stream, unsigned* progress, unsigned progress_val) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !p || !pl) return set_error(TFRA_ERR_INVALID, "apply_planned: null argument");
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(t->mu);
  int rc = t->enter(s);
  if (rc) return rc;
  if (pl->n == 0) return TFRA_OK;
  if (!grads || !param_default_row) return set_error(TFRA_ERR_INVALID, "apply_planned: null buffer");
  if (t->opts.value_dtype != TFRA_F32) return set_error(TFRA_ERR_UNSUPPORTED, "apply_planned: value_dtype must be float32");
  if (p->kind < 0 || p->kind > TFRA_OPT_FTRL) return set_error(TFRA_ERR_INVALID, "apply_planned: unknown kind");
  int need = p->kind == TFRA_OPT_SGD ? 0 : (p->kind == TFRA_OPT_ADAGRAD ? 1 : 2);
  if (t->opts.aux_fields < need) return set_error(TFRA_ERR_INVALID, "apply_planned: table lacks optimizer slot fields");
  if (t->opts.dim != pl->dim) return set_error(TFRA_ERR_INVALID, "apply_planned: the plan was built for another dim");
  if (t->opts.device != pl->device && t->opts.device >= 0) return set_error(TFRA_ERR_INVALID, "apply_planned: plan and table live on different devices");
  if ((((uintptr_t)grads | (uintptr_t)param_default_row) & 15))
    return set_error(TFRA_ERR_UNSUPPORTED, "apply_planned: gradient / default buffers must be 16-B aligned");
  rc = t->prepare_insert(pl->n, s);
  if (rc) return rc;
  const int dim = pl->dim;
  const unsigned rows_base = (unsigned)pl->npad;
  dim3 ga((unsigned)pl->ntiles), gc(pl->P);
  const int nch = (dim + 63) / 64;
  switch (nch) {
    case 1: tile_sums_kernel<1><<<ga, NTA, 0, s>>>(grads, dim, pl->tile_entries, pl->tile_len, pl->rows, progress, progress_val); break;
    case 2: tile_sums_kernel<2><<<ga, NTA, 0, s>>>(grads, dim, pl->tile_entries, pl->tile_len, pl->rows, progress, progress_val); break;
    case 3: tile_sums_kernel<3><<<ga, NTA, 0, s>>>(grads, dim, pl->tile_entries, pl->tile_len, pl->rows, progress, progress_val); break;
    default: tile_sums_kernel<4><<<ga, NTA, 0, s>>>(grads, dim, pl->tile_entries, pl->tile_len, pl->rows, progress, progress_val); break;
  }
  switch (nch) {
    case 1: bucket_sums_kernel<1><<<gc, NT, 0, s>>>(dim, rows_base, grads, pl->rows, pl->bp, pl->err_latched, t->err_count); break;
    case 2: bucket_sums_kernel<2><<<gc, NT, 0, s>>>(dim, rows_base, grads, pl->rows, pl->bp, pl->err_latched, t->err_count); break;
    case 3: bucket_sums_kernel<3><<<gc, NT, 0, s>>>(dim, rows_base, grads, pl->rows, pl->bp, pl->err_latched, t->err_count); break;
    default: bucket_sums_kernel<4><<<gc, NT, 0, s>>>(dim, rows_base, grads, pl->rows, pl->bp, pl->err_latched, t->err_count); break;
  }
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "apply_planned: launch failed");
  return launch_apply_indirect(t, s, p, pl->npad, pl->u_keys, pl->u_src, grads, pl->rows, rows_base, param_default_row,
                               pl->d_total, nullptr, 0, 0);
}

// One training step driven from C on two streams (no Python between the launches, no graph).
//   main : lookup(ids_cur) -> tile sums -> bucket sums -> fused update           (plan_cur)
//   side : build plan_next from ids_next, free-running
// Cross-queue events cost ~5 us (stream wait) / ~7 us (record) each between two kernels of the main stream,
// so the two streams are ordered through two host-visible counters in pinned memory instead, and the host
// only falls back to an event / a sync when a counter lags:
//   * table progress: written by the first block of the tile sums of step s  =>  every earlier step is done.
//     plan_next's buffers were last read by step plan_next->last_used_step; the build is enqueued once the
//     progress has passed it (with >= 3 plans in rotation that is always the case unless the host is far
//     ahead of the GPU, in which case it waits here instead of in a queue);
//   * plan built: written by a 1-thread kernel behind the build.  If it already shows plan_cur's generation
//     the gradient half is enqueued without any wait packet (the build's kernels have completed, so their
//     writes are in memory and the main-stream kernels start with a fresh cache view); otherwise — the host
//     got ahead of the side stream — the host waits for the side stream.
extern "C" int tfra_table_step_prefetch(tfra_table_t* tp, const tfra_opt_params* p, tfra_sparse_plan_t* plan_cur,
                                        const int64_t* ids_cur, void* rows_out, const void* find_default,
                                        const float* grads, const float* param_default_row,
                                        tfra_sparse_plan_t* plan_next, const int64_t* ids_next, size_t n_next,
                                        tfra_stream_t main_stream, tfra_stream_t side_stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !p || !plan_cur) return set_error(TFRA_ERR_INVALID, "step_prefetch: null argument");
  hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
  if (ms == ss && plan_next) return set_error(TFRA_ERR_INVALID, "step_prefetch: needs two different streams");
  if (plan_next == plan_cur) return set_error(TFRA_ERR_INVALID, "step_prefetch: plan_next must differ from plan_cur");
  std::lock_guard<std::mutex> step_lock(t->step_mu);   // one driver call at a time per table
  int rc = TFRA_OK;
  if (!t->progress_host) {
    if (hipHostMalloc((void**)&t->progress_host, 64, hipHostMallocDefault) != hipSuccess) { t->progress_host = nullptr; return set_error(TFRA_ERR_OOM, "step_prefetch: hipHostMalloc"); }
    *t->progress_host = 0;
  }
  const unsigned step = ++t->step_gen;
  if (plan_next) {
    if (!plan_next->built_host) {
      if (hipHostMalloc((void**)&plan_next->built_host, 64, hipHostMallocDefault) != hipSuccess) { plan_next->built_host = nullptr; return set_error(TFRA_ERR_OOM, "step_prefetch: hipHostMalloc"); }
      *plan_next->built_host = 0;
    }
    if (plan_next->last_used_step) {  // the gradient half that read plan_next's buffers must be over
      const unsigned need = plan_next->last_used_step + 1;
      volatile unsigned* prog = t->progress_host;
      bool ok = false;
      for (int it = 0; it < 200000 && !ok; ++it) ok = (int)(*prog - need) >= 0;   // ~ a few ms at most
      if (!ok && hipStreamSynchronize(ms) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_prefetch: sync");
    }
  }
  if (plan_cur->n && rows_out) {
    rc = tfra_table_find(tp, plan_cur->n, ids_cur, rows_out, nullptr, find_default, 0, main_stream);
    if (rc) return rc;
  }
  if (plan_next) {
    rc = tfra_sparse_plan_build(plan_next, n_next, ids_next, t->opts.dim, side_stream);
    if (rc) return rc;
    plan_next->ev_recorded = true;   // built on the side stream: the join below applies
  }
  if (plan_cur->ev_recorded) {  // built on the side stream by an earlier call
    const bool built = plan_cur->built_host && (int)(*(volatile unsigned*)plan_cur->built_host - plan_cur->gen) >= 0;
    if (!built && hipStreamSynchronize(ss) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_prefetch: join");   // rare
    plan_cur->ev_recorded = false;
  }
  plan_cur->last_used_step = step;
  if (plan_cur->n == 0) return TFRA_OK;   // no kernel publishes this step: a later slot check falls back to a sync
  return apply_planned_impl(tp, p, plan_cur, grads, param_default_row, main_stream, t->progress_host, step);
}
