// Block-level building blocks of the duplicate-id reductions (tfra_csr.hip: the CSR write-back plan and its
// gradient kernels).
#pragma once
#include <hip/hip_runtime.h>

#include "tfra_device.h"

namespace tfra {
namespace red {

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

}  // namespace red
}  // namespace tfra
