// Sparse optimizer write-back WITH duplicate ids, in two kernels (no sort passes, no atomics on
// gradient data, deterministic):   tfra_table_apply_sparse(ids[B], grads[B,D])
//
// Reference: gradients of duplicate ids are summed, then ONE update per key
// (`_resource_apply_sparse_duplicate_indices` = unique + unsorted_segment_sum,
// PY/dynamic_embedding_optimizer.py:177-190), followed by (1+S) finds + dense apply + (1+S)
// upserts (:165-204).  A Zipf-1.2 batch of 131 072 ids has ~22 K unique keys and the hottest key
// repeats ~24 000 times, so the reduction must be parallel per key yet order-fixed:
//
//   kernel A (one block per TILE=1024 ids): bitonic-sort (fmix64(id), idx) in LDS — equal ids
//     become adjacent, ascending idx, and because bucket = mulhi(hash, P) is monotone in the hash
//     the tile's unique keys come out grouped by bucket; 16-lane groups sum the gradient rows of
//     each run (fixed order) -> ONE partial row per unique key per tile, plus per-tile bucket
//     histogram/offsets.
//   kernel C (one block per bucket): gathers the bucket's partials from all tiles (tile order),
//     bitonic-sorts (key, src) in LDS, sums each key's partials in tile order, then does the
//     table work of tfra_optim.hip in place: locate-or-insert the row, apply, write back.
//
// Summation tree per key = [ascending idx inside (tile, 64-position group)] -> [groups of a tile
// in order] -> [tiles in order]: fixed by the input alone => bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <string>

#include "../../include/tfra_mi355x.h"
#include "tfra_device.h"
#include "tfra_host.h"
#include "tfra_optim_device.h"

using namespace tfra;

namespace {

constexpr int TILE = 1024;   // ids per kernel-A block
constexpr int NT = 256;      // threads per block (16 key groups)
constexpr int GRP_SPAN = TILE / 16;  // sorted positions walked by one 16-lane group
constexpr int CMAX = 2048;   // partials one kernel-C block can hold in LDS
constexpr int MAXCH = 4;     // D <= 256 (float4 per lane per 64-column chunk)

__device__ __forceinline__ bool less_hi(u64 ha, unsigned ia, u64 hb, unsigned ib) {
  return ha != hb ? ha < hb : ia < ib;
}

// in-LDS bitonic sort of n2 (power of two) composite (h, i) pairs by NT threads
template <class I>
__device__ __forceinline__ void bitonic_sort(u64* h, I* ix, int n2) {
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int q = threadIdx.x; q < (n2 >> 1); q += NT) {
        int i = ((q / j) * (j << 1)) + (q % j);
        int l = i + j;
        bool up = ((i & k) == 0);
        u64 hi_ = h[i], hl = h[l];
        I ii = ix[i], il = ix[l];
        bool sw = up ? less_hi(hl, (unsigned)il, hi_, (unsigned)ii) : less_hi(hi_, (unsigned)ii, hl, (unsigned)il);
        if (sw) { h[i] = hl; h[l] = hi_; ix[i] = il; ix[l] = ii; }
      }
      __syncthreads();
    }
  }
}

// exclusive scan over `vals` (one value per thread, NT threads); returns exclusive prefix, total in *total
__device__ __forceinline__ int block_excl_scan(int v, int* sh /*[NT/64+1]*/, int* total) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, incl = v;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) sh[w] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
  for (int k = 0; k < NT / 64; ++k) { if (k < w) woff += sh[k]; tot += sh[k]; }
  __syncthreads();
  *total = tot;
  return woff + incl - v;
}

// ---------------------------------------------------------------------------------------------
// kernel A
template <int NCH>
__global__ __launch_bounds__(NT) void tile_reduce_kernel(size_t n, const i64* __restrict__ ids,
                                                         const float* __restrict__ grads, int dim, unsigned P,
                                                         i64* __restrict__ part_keys, float* __restrict__ part_rows,
                                                         unsigned short* __restrict__ tile_hist,
                                                         unsigned short* __restrict__ tile_start) {
  __shared__ u64 s_h[TILE];
  __shared__ unsigned short s_ix[TILE];
  __shared__ unsigned short s_u[TILE];       // unique rank of the run each sorted position belongs to
  __shared__ unsigned s_hist[2048];          // P <= 2048
  __shared__ float s_left[16][64 * MAXCH];   // a group's sum of the run that spills in from the left
  __shared__ unsigned char s_cont[16], s_hashead[16];
  __shared__ int s_scan[NT / 64 + 1];
  const size_t tile = blockIdx.x, base = tile * TILE;
  const int nvalid = (int)min((size_t)TILE, n - base);
  for (int p = threadIdx.x; p < TILE; p += NT) {
    bool ok = p < nvalid;
    s_h[p] = ok ? fmix64((u64)ids[base + p]) : ~0ULL;
    s_ix[p] = ok ? (unsigned short)p : (unsigned short)0xffff;
  }
  for (unsigned b = threadIdx.x; b < P; b += NT) s_hist[b] = 0;
  __syncthreads();
  bitonic_sort<unsigned short>(s_h, s_ix, TILE);
  // heads (first position of every run of equal ids) and unique ranks; 4 consecutive positions/thread
  int p0 = threadIdx.x * 4, c = 0;
  bool head[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int p = p0 + k;
    head[k] = p < nvalid && (p == 0 || s_h[p] != s_h[p - 1]);
    c += head[k];
  }
  int ntile_unique;
  int u = block_excl_scan(c, s_scan, &ntile_unique);  // heads before this thread's positions
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int p = p0 + k;
    if (head[k]) {
      atomicAdd(&s_hist[(unsigned)__umul64hi(s_h[p], (u64)P)], 1u);
      s_u[p] = (unsigned short)u;  // unique rank of the run starting here
      ++u;
    } else {
      s_u[p] = (unsigned short)(u - 1);  // rank of the run this position continues
    }
  }
  __syncthreads();
  // per-bucket first unique rank = exclusive scan of the histogram (buckets ascend with the hash)
  {
    int carry = 0;
    for (unsigned b0 = 0; b0 < P; b0 += NT) {
      unsigned b = b0 + threadIdx.x;
      int v = b < P ? (int)s_hist[b] : 0, tot;
      int ex = block_excl_scan(v, s_scan, &tot);
      if (b < P) {
        tile_hist[tile * P + b] = (unsigned short)v;
        tile_start[tile * P + b] = (unsigned short)(carry + ex);
      }
      carry += tot;
    }
  }
  // ---- run sums: group g walks sorted positions [g*64, g*64+64) ------------------------------
  const int lane = threadIdx.x & 63, sub = lane & 15, g = threadIdx.x >> 4;
  const int gs = g * GRP_SPAN, ge = min(gs + GRP_SPAN, nvalid);
  float4 acc[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool has_any = gs < nvalid;
  const bool cont_in = has_any && gs > 0 && s_h[gs] == s_h[gs - 1];  // first run started in an earlier group
  bool seen_head = false;
  bool owner_open = false;   // this group owns a run that is still open at its right edge
  int open_u = -1;
  i64 open_key = 0;
  auto flush_global = [&](int uu, i64 key) {
    size_t r = base + (size_t)uu;
    float* o = part_rows + r * (size_t)dim;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      int col = k * 64 + sub * 4;
      if (col < dim) *reinterpret_cast<float4*>(o + col) = acc[k];
    }
    if (sub == 0) part_keys[r] = key;
  };
  if (has_any) {
    bool in_left = cont_in;  // currently accumulating the spill-in run
    int cur_u = s_u[gs];
    i64 cur_key = ids[base + s_ix[gs]];
    for (int p = gs; p < ge; ++p) {
      bool hd = (p > gs) && (s_h[p] != s_h[p - 1]);
      if (hd) {
        // close the previous run
        if (in_left) {
#pragma unroll
          for (int k = 0; k < NCH; ++k) *reinterpret_cast<float4*>(&s_left[g][k * 64 + sub * 4]) = acc[k];
          in_left = false;
        } else {
          flush_global(cur_u, cur_key);
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        cur_u = s_u[p];
        cur_key = ids[base + s_ix[p]];
        seen_head = true;
      }
      const float* row = grads + (base + s_ix[p]) * (size_t)dim;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        int col = k * 64 + sub * 4;
        if (col < dim) {
          float4 x = *reinterpret_cast<const float4*>(row + col);
          acc[k].x += x.x; acc[k].y += x.y; acc[k].z += x.z; acc[k].w += x.w;
        }
      }
    }
    // the run open at the right edge
    bool spills_out = (ge == gs + GRP_SPAN) && ge < nvalid && s_h[ge] == s_h[ge - 1];
    if (in_left) {
      // the whole range belongs to the spill-in run: contribute it as a left part
#pragma unroll
      for (int k = 0; k < NCH; ++k) *reinterpret_cast<float4*>(&s_left[g][k * 64 + sub * 4]) = acc[k];
    } else if (spills_out) {
      owner_open = true; open_u = cur_u; open_key = cur_key;
    } else {
      flush_global(cur_u, cur_key);
    }
  }
  if ((threadIdx.x & 15) == 0) { s_cont[g] = cont_in; s_hashead[g] = seen_head; }
  __syncthreads();
  if (owner_open) {
    for (int g2 = g + 1; g2 < 16 && s_cont[g2]; ++g2) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        float4 x = *reinterpret_cast<const float4*>(&s_left[g2][k * 64 + sub * 4]);
        acc[k].x += x.x; acc[k].y += x.y; acc[k].z += x.z; acc[k].w += x.w;
      }
      if (s_hashead[g2]) break;  // the run ended inside g2
    }
    flush_global(open_u, open_key);
  }
}

// ---------------------------------------------------------------------------------------------
// kernel C
template <int KIND, int NCH>
__global__ __launch_bounds__(NT) void bucket_apply_kernel(TableView v, OptP o, unsigned P, unsigned ntiles, int dim,
                                                          const i64* __restrict__ part_keys,
                                                          const float* __restrict__ part_rows,
                                                          const unsigned short* __restrict__ tile_hist,
                                                          const unsigned short* __restrict__ tile_start,
                                                          const float* __restrict__ defaults, float aux0, float aux1,
                                                          unsigned* overflow) {
  constexpr int S = NSlots<KIND>::v;
  __shared__ u64 e_key[CMAX];       // key biased to unsigned order (x ^ sign bit)
  __shared__ unsigned e_src[CMAX];  // partial index tile*TILE + rank  (ascending = tile order)
  __shared__ unsigned short seg_at[CMAX + 1];
  __shared__ int s_scan[NT / 64 + 1];
  __shared__ int s_n;
  const unsigned b = blockIdx.x;
  // gather this bucket's partial descriptors from every tile, tile order
  int carry = 0;
  bool too_many = false;
  for (unsigned t0 = 0; t0 < ntiles; t0 += NT) {
    unsigned t = t0 + threadIdx.x;
    int cnt = t < ntiles ? tile_hist[(size_t)t * P + b] : 0, tot;
    int ex = block_excl_scan(cnt, s_scan, &tot);
    if (carry + tot > CMAX) { too_many = true; break; }
    if (cnt) {
      unsigned st = tile_start[(size_t)t * P + b];
      for (int j = 0; j < cnt; ++j) {
        unsigned src = t * TILE + st + j;
        e_src[carry + ex + j] = src;
        e_key[carry + ex + j] = (u64)part_keys[src] ^ 0x8000000000000000ULL;
      }
    }
    carry += tot;
  }
  if (too_many) {  // extreme hash skew: reported to the host, nothing applied for this bucket
    if (threadIdx.x == 0) atomicAdd(overflow, 1u);
    return;
  }
  const int n = carry;
  if (n == 0) return;
  int n2 = 2;
  while (n2 < n) n2 <<= 1;
  for (int p = n + threadIdx.x; p < n2; p += NT) { e_key[p] = ~0ULL; e_src[p] = 0xffffffffu; }
  __syncthreads();
  bitonic_sort<unsigned>(e_key, e_src, n2);
  // run starts
  int nseg = 0;
  {
    int carry2 = 0;
    for (int p0 = 0; p0 < n; p0 += NT) {
      int p = p0 + threadIdx.x;
      int hd = p < n && (p == 0 || e_key[p] != e_key[p - 1]), tot;
      int ex = block_excl_scan(hd, s_scan, &tot);
      if (hd) seg_at[carry2 + ex] = (unsigned short)p;
      carry2 += tot;
    }
    nseg = carry2;
    if (threadIdx.x == 0) seg_at[nseg] = (unsigned short)n;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, g = threadIdx.x >> 4;
  int fresh = 0, failed = 0;
  // NB: every group executes the same number of loop trips (wave-level primitives inside)
  for (int s0 = 0; s0 < nseg; s0 += 16) {
    int sidx = s0 + g;
    bool act = sidx < nseg;
    if (act) {
      int pb = seg_at[sidx], pe = seg_at[sidx + 1];
      float4 gsum[NCH];
#pragma unroll
      for (int k = 0; k < NCH; ++k) gsum[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p = pb; p < pe; ++p) {
        const float* row = part_rows + (size_t)e_src[p] * dim;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          int col = k * 64 + sub * 4;
          if (col < dim) {
            float4 x = *reinterpret_cast<const float4*>(row + col);
            gsum[k].x += x.x; gsum[k].y += x.y; gsum[k].z += x.z; gsum[k].w += x.w;
          }
        }
      }
      i64 key = (i64)(e_key[pb] ^ 0x8000000000000000ULL);
      bool is_new;
      i64 rowi = locate_or_claim(v, key, sub, gshift, is_new);
      if (rowi < 0) {
        failed += (sub == 0);
      } else {
        fresh += (is_new && sub == 0);
        float* pr = reinterpret_cast<float*>(v.rows + (size_t)rowi * v.row_stride);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          int col = k * 64 + sub * 4;
          if (col < dim) {
            float4 p, s1 = make_float4(aux0, aux0, aux0, aux0), s2 = make_float4(aux1, aux1, aux1, aux1);
            if (is_new) {
              p = *reinterpret_cast<const float4*>(defaults + col);
            } else {
              p = *reinterpret_cast<const float4*>(pr + col);
              if (S >= 1) s1 = *reinterpret_cast<const float4*>(pr + dim + col);
              if (S >= 2) s2 = *reinterpret_cast<const float4*>(pr + 2 * dim + col);
            }
            apply_one<KIND>(o, gsum[k].x, p.x, s1.x, s2.x);
            apply_one<KIND>(o, gsum[k].y, p.y, s1.y, s2.y);
            apply_one<KIND>(o, gsum[k].z, p.z, s1.z, s2.z);
            apply_one<KIND>(o, gsum[k].w, p.w, s1.w, s2.w);
            *reinterpret_cast<float4*>(pr + col) = p;
            if (S >= 1) *reinterpret_cast<float4*>(pr + dim + col) = s1;
            if (S >= 2) *reinterpret_cast<float4*>(pr + 2 * dim + col) = s2;
          }
        }
        if (is_new && (int)v.n_fields - 1 > S) {
          for (int f = S + 1; f < (int)v.n_fields; ++f)
            for (int cidx = sub; cidx < dim; cidx += 16) pr[f * dim + cidx] = (f == 1 ? aux0 : aux1);
        }
        if (v.scores && sub == 0 && rowi < (i64)(v.nb * SLOTS))
          v.scores[((u64)rowi / SLOTS) * 16 + (u64)rowi % SLOTS] = wall_clock64();
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) { fresh += __shfl_xor(fresh, off); failed += __shfl_xor(failed, off); }
  if (lane == 0) {
    if (fresh) size_add(v, (u64)b * 4 + (threadIdx.x >> 6), fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

template <int KIND, int NCH>
void launch_c(dim3 grid, hipStream_t s, TableView v, OptP o, unsigned P, unsigned ntiles, int dim, const i64* pk,
              const float* pr, const unsigned short* th, const unsigned short* ts, const float* d, float a0, float a1,
              unsigned* ovf) {
  bucket_apply_kernel<KIND, NCH><<<grid, NT, 0, s>>>(v, o, P, ntiles, dim, pk, pr, th, ts, d, a0, a1, ovf);
}

template <int NCH>
void launch_c_kind(int kind, dim3 grid, hipStream_t s, TableView v, OptP o, unsigned P, unsigned ntiles, int dim,
                   const i64* pk, const float* pr, const unsigned short* th, const unsigned short* ts, const float* d,
                   float a0, float a1, unsigned* ovf) {
  switch (kind) {
    case TFRA_OPT_SGD: launch_c<TFRA_OPT_SGD, NCH>(grid, s, v, o, P, ntiles, dim, pk, pr, th, ts, d, a0, a1, ovf); break;
    case TFRA_OPT_ADAM: launch_c<TFRA_OPT_ADAM, NCH>(grid, s, v, o, P, ntiles, dim, pk, pr, th, ts, d, a0, a1, ovf); break;
    case TFRA_OPT_ADAGRAD: launch_c<TFRA_OPT_ADAGRAD, NCH>(grid, s, v, o, P, ntiles, dim, pk, pr, th, ts, d, a0, a1, ovf); break;
    default: launch_c<TFRA_OPT_FTRL, NCH>(grid, s, v, o, P, ntiles, dim, pk, pr, th, ts, d, a0, a1, ovf); break;
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
  if (n >= (1ULL << 31)) return set_error(TFRA_ERR_INVALID, "apply_sparse: too many ids");
  rc = t->prepare_insert(n, s);
  if (rc) return rc;
  const size_t ntiles = (n + TILE - 1) / TILE;
  unsigned P = 64;
  while (P < 2048 && (size_t)P * 256 < n) P <<= 1;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  size_t bytes = al(ntiles * TILE * sizeof(i64)) + al(ntiles * TILE * (size_t)dim * 4) + 2 * al(ntiles * P * 2) + 256;
  rc = t->ensure_scratch(bytes, s);
  if (rc) return rc;
  unsigned char* w = (unsigned char*)t->scratch;
  i64* part_keys = (i64*)w; w += al(ntiles * TILE * sizeof(i64));
  float* part_rows = (float*)w; w += al(ntiles * TILE * (size_t)dim * 4);
  unsigned short* th = (unsigned short*)w; w += al(ntiles * P * 2);
  unsigned short* ts = (unsigned short*)w; w += al(ntiles * P * 2);
  unsigned* ovf = (unsigned*)w;
  const int nch = (dim + 63) / 64;
  const i64* k = (const i64*)ids;
  switch (nch) {
    case 1: tile_reduce_kernel<1><<<(unsigned)ntiles, NT, 0, s>>>(n, k, grads, dim, P, part_keys, part_rows, th, ts); break;
    case 2: tile_reduce_kernel<2><<<(unsigned)ntiles, NT, 0, s>>>(n, k, grads, dim, P, part_keys, part_rows, th, ts); break;
    case 3: tile_reduce_kernel<3><<<(unsigned)ntiles, NT, 0, s>>>(n, k, grads, dim, P, part_keys, part_rows, th, ts); break;
    default: tile_reduce_kernel<4><<<(unsigned)ntiles, NT, 0, s>>>(n, k, grads, dim, P, part_keys, part_rows, th, ts); break;
  }
  TableView v = t->view_of(t->cur);
  OptP o{p->kind, p->lr, p->beta1, p->beta2, p->eps, p->l1, p->l2, p->lr_power};
  float a0 = t->opts.aux_init[0], a1 = t->opts.aux_init[1];
  dim3 grid(P);
  // the overflow word lives in the table's error counter: surfaced by tfra_table_size()
  switch (nch) {
    case 1: launch_c_kind<1>(p->kind, grid, s, v, o, P, (unsigned)ntiles, dim, part_keys, part_rows, th, ts, param_default_row, a0, a1, t->err_count); break;
    case 2: launch_c_kind<2>(p->kind, grid, s, v, o, P, (unsigned)ntiles, dim, part_keys, part_rows, th, ts, param_default_row, a0, a1, t->err_count); break;
    case 3: launch_c_kind<3>(p->kind, grid, s, v, o, P, (unsigned)ntiles, dim, part_keys, part_rows, th, ts, param_default_row, a0, a1, t->err_count); break;
    default: launch_c_kind<4>(p->kind, grid, s, v, o, P, (unsigned)ntiles, dim, part_keys, part_rows, th, ts, param_default_row, a0, a1, t->err_count); break;
  }
  (void)ovf;
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "apply_sparse: launch failed");
  return TFRA_OK;
}
