// Fused sparse-optimizer write-back on co-located rows [p | slot1 | slot2].
//
// The reference runs, per step and per embedding variable, (1+S) table finds, ONE stock
// TensorFlow dense apply kernel on the local [U,dim] buffers and (1+S) table upserts
// (PY/dynamic_embedding_optimizer.py:165-204; slots are separate hash tables, create_slots
// :870-958).  Here one kernel does it per unique key: locate-or-insert the row once, read
// p/slots, apply, write back = 8 + 7*Rb algorithmic bytes for Adam/FTRL instead of 4936 B.
// Update rules = TF's ResourceApply{GradientDescent,Adam,Adagrad[V2],Ftrl} (restated in
// oracle/optimizers.py), evaluated in fp32 in the same operation order
// (this file is compiled with -ffp-contract=off so no FMA contraction changes roundings).
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

// 16 lanes per key; lane `sub` owns elements sub*4..sub*4+3 (+64 per step): float4 everywhere
// when dim % 4 == 0 (VEC4), scalar otherwise.
// ST: storage type of the rows (tfra_dtype F32 / F16 / BF16); VEC4 only with float rows.
template <int KIND, bool VEC4, int ST = TFRA_F32>
__global__ __launch_bounds__(256) void apply_kernel(TableView v, OptP o, size_t n, const i64* __restrict__ keys,
                                                    const float* __restrict__ grads,
                                                    const float* __restrict__ defaults, int full, int dim,
                                                    float aux0, float aux1, const i64* __restrict__ d_n,
                                                    ScoreP sp, uint8_t* __restrict__ deferred) {
  constexpr int S = NSlots<KIND>::v;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const size_t g = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  int fresh = 0, failed = 0;
  if (o.d_lr) o.lr = *o.d_lr;
  if (d_n) { size_t dn = (size_t)*d_n; if (dn < n) n = dn; }
  if (g < n) {
    const i64 key = keys[g];
    bool is_new;
    i64 row;
    if (deferred) {  // bounded (Hkv) table at max_capacity: keys without a free slot go to phase 2
      u64 h;
      const u64 b0 = bucket0(key, v.nb, h);
      const i64 k0 = load_key_coherent(key_line(v, b0) + sub);
      row = locate_or_claim_from(v, key, h, b0, k0, sub, gshift, is_new, sp.bounded);
      if (sub == 0) deferred[g] = row == NEED_EVICT;
    } else {
      row = locate_or_claim(v, key, sub, gshift, is_new);
    }
    if (row < 0) {
      failed = (sub == 0 && row != NEED_EVICT);
    } else {
      fresh = (is_new && sub == 0);
      float* pr = reinterpret_cast<float*>(row_ptr(v, row));
      const float* gr = grads + g * (size_t)dim;
      const float* df = defaults + (full ? g * (size_t)dim : 0);
      if (VEC4) {
        for (int c = sub * 4; c < dim; c += 64) {
          // all loads unconditional and issued together (a brand-new row reads its own not yet
          // initialised bytes and discards them)
          float4 gg = *reinterpret_cast<const float4*>(gr + c);
          float4 p = *reinterpret_cast<const float4*>((is_new ? df : pr) + c);
          float4 s1 = *reinterpret_cast<const float4*>(pr + (S >= 1 ? dim : 0) + c);
          float4 s2 = *reinterpret_cast<const float4*>(pr + (S >= 2 ? 2 * dim : 0) + c);
          keep_live(gg, p, s1, s2);
          if (is_new || S < 1) s1 = make_float4(aux0, aux0, aux0, aux0);
          if (is_new || S < 2) s2 = make_float4(aux1, aux1, aux1, aux1);
          apply_one<KIND>(o, gg.x, p.x, s1.x, s2.x);
          apply_one<KIND>(o, gg.y, p.y, s1.y, s2.y);
          apply_one<KIND>(o, gg.z, p.z, s1.z, s2.z);
          apply_one<KIND>(o, gg.w, p.w, s1.w, s2.w);
          // write-through: the rows leave L2 during the kernel, not at the boundary to the next one
          store_wt16(pr + c, *reinterpret_cast<uint4*>(&p));
          if (S >= 1) store_wt16(pr + dim + c, *reinterpret_cast<uint4*>(&s1));
          if (S >= 2) store_wt16(pr + 2 * dim + c, *reinterpret_cast<uint4*>(&s2));
        }
      } else {
        typedef typename Stored<ST>::T V;
        V* sr = reinterpret_cast<V*>(row_ptr(v, row));
        for (int c = sub; c < dim; c += 16) {
          float p, s1 = aux0, s2 = aux1;
          if (is_new) {
            p = df[c];
          } else {
            p = load_stored<ST>(sr + c);
            if (S >= 1) s1 = load_stored<ST>(sr + dim + c);
            if (S >= 2) s2 = load_stored<ST>(sr + 2 * dim + c);
          }
          apply_one<KIND>(o, gr[c], p, s1, s2);
          sr[c] = to_stored<ST>(p);
          if (S >= 1) sr[dim + c] = to_stored<ST>(s1);
          if (S >= 2) sr[2 * dim + c] = to_stored<ST>(s2);
        }
        // aux fields the optimizer does not own (table created with more slots than it uses)
        if (is_new && (int)v.n_fields - 1 > S) {
          for (int f = S + 1; f < (int)v.n_fields; ++f)
            for (int c = sub; c < dim; c += 16) sr[f * dim + c] = to_stored<ST>(f == 1 ? aux0 : aux1);
        }
      }
      // aux fields the optimizer does not own (table created with more slots than it uses)
      if (VEC4 && is_new && (int)v.n_fields - 1 > S) {
        for (int f = S + 1; f < (int)v.n_fields; ++f)
          for (int c = sub; c < dim; c += 16) pr[f * dim + c] = (f == 1 ? aux0 : aux1);
      }
      update_score(v, row, is_new, sp.strategy, 1, sp.epoch, sub);  // one write-back = one upsert
    }
  } else if (deferred && g < n && sub == 0) {
    deferred[g] = 0;
  }
  for (int off = 32; off > 0; off >>= 1) { fresh += __shfl_xor(fresh, off); failed += __shfl_xor(failed, off); }
  if (lane == 0) {
    if (fresh) size_add(v, g >> 2, fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

// Phase 2 on a bounded (Hkv) table at max_capacity, after phase 1 has finished every hit and free-slot
// claim of the batch (so no row is being updated while it is evicted): each deferred key replaces the
// minimum-score entry of its two home buckets and starts from the default row / initial slot values —
// what find (miss -> default) + dense apply + upsert (evicting) give in the reference
// (PY/dynamic_embedding_optimizer.py:165-204 on an HkvHashTable, lookup_table_op_hkv.h:522-537).
template <int KIND, int ST = TFRA_F32>
__global__ __launch_bounds__(256) void apply_evict_kernel(TableView v, OptP o, size_t n, const i64* __restrict__ keys,
                                                          const float* __restrict__ grads,
                                                          const float* __restrict__ defaults, int full, int dim,
                                                          float aux0, float aux1, const i64* __restrict__ d_n,
                                                          ScoreP sp, const uint8_t* __restrict__ deferred) {
  constexpr int S = NSlots<KIND>::v;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const size_t g = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  int fresh = 0, failed = 0;
  if (o.d_lr) o.lr = *o.d_lr;
  if (d_n) { size_t dn = (size_t)*d_n; if (dn < n) n = dn; }
  if (g < n && deferred[g]) {
    const i64 key = keys[g];
    const bool lru_like = sp.strategy == TFRA_EVICT_LRU || sp.strategy == TFRA_EVICT_EPOCHLRU;
    const u64 in_score = sp.strategy == TFRA_EVICT_EPOCHLFU ? ((sp.epoch << 32) | 1) : 1;
    u64 word = 0;
    bool claimed_empty;
    i64 row = evict_and_lock(v, key, in_score, lru_like, sub, gshift, &word, claimed_empty);
    if (row >= 0) {
      typedef typename Stored<ST>::T V;
      V* pr = reinterpret_cast<V*>(row_ptr(v, row));
      const float* gr = grads + g * (size_t)dim;
      const float* df = defaults + (full ? g * (size_t)dim : 0);
      // write-through stores: the row is in memory before the key is published (publish_key)
      auto st = [](V* q, float x) { __hip_atomic_store(q, to_stored<ST>(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      for (int c = sub; c < dim; c += 16) {
        float p = df[c], s1 = aux0, s2 = aux1;
        apply_one<KIND>(o, gr[c], p, s1, s2);
        st(pr + c, p);
        if (S >= 1) st(pr + dim + c, s1);
        if (S >= 2) st(pr + 2 * dim + c, s2);
      }
      for (int f = S + 1; f < (int)v.n_fields; ++f)
        for (int c = sub; c < dim; c += 16) st(pr + f * dim + c, (f == 1 ? aux0 : aux1));
      if (sub == 0) store_wt8(score_word(v, word), 0);  // the slot starts a new life
      update_score<true>(v, row, true, sp.strategy, 1, sp.epoch, sub);
      publish_key(v, word, key, sub);
      fresh = (claimed_empty && sub == 0);
    } else if (row == -3) {
      failed = (sub == 0);
    }  // -1: not admitted (LFU-type score below every resident score): dropped, like HKV
  }
  for (int off = 32; off > 0; off >>= 1) { fresh += __shfl_xor(fresh, off); failed += __shfl_xor(failed, off); }
  if (lane == 0) {
    if (fresh) size_add(v, g >> 2, fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

template <int KIND>
void launch_apply(int dt, bool vec4, dim3 grid, hipStream_t s, TableView v, OptP o, size_t n, const i64* k, const float* g,
                  const float* d, int full, int dim, float a0, float a1, const i64* dn, ScoreP sp, uint8_t* deferred) {
  if (dt == TFRA_F16) {
    apply_kernel<KIND, false, TFRA_F16><<<grid, 256, 0, s>>>(v, o, n, k, g, d, full, dim, a0, a1, dn, sp, deferred);
    if (deferred) apply_evict_kernel<KIND, TFRA_F16><<<grid, 256, 0, s>>>(v, o, n, k, g, d, full, dim, a0, a1, dn, sp, deferred);
    return;
  }
  if (dt == TFRA_BF16) {
    apply_kernel<KIND, false, TFRA_BF16><<<grid, 256, 0, s>>>(v, o, n, k, g, d, full, dim, a0, a1, dn, sp, deferred);
    if (deferred) apply_evict_kernel<KIND, TFRA_BF16><<<grid, 256, 0, s>>>(v, o, n, k, g, d, full, dim, a0, a1, dn, sp, deferred);
    return;
  }
  if (vec4) apply_kernel<KIND, true><<<grid, 256, 0, s>>>(v, o, n, k, g, d, full, dim, a0, a1, dn, sp, deferred);
  else apply_kernel<KIND, false><<<grid, 256, 0, s>>>(v, o, n, k, g, d, full, dim, a0, a1, dn, sp, deferred);
  if (deferred) apply_evict_kernel<KIND><<<grid, 256, 0, s>>>(v, o, n, k, g, d, full, dim, a0, a1, dn, sp, deferred);
}

// one fused write-back counts as one upsert for the epoch strategies (lookup_table_op_hkv.h:528-536)
void step_epoch(Table* t) {
  if (t->epoch_hold) return;   // one logical write-back issued as several launches (apply_sparse_big): stepped once by the caller
  const int strat = t->opts.strategy;
  if (strat == TFRA_EVICT_EPOCHLRU || strat == TFRA_EVICT_EPOCHLFU) {
    t->curr_step += 1;
    if (t->opts.step_per_epoch > 0 && t->curr_step > t->opts.step_per_epoch) { t->global_epoch += 1; t->curr_step = 1; }
  }
}

}  // namespace

namespace tfra {
void step_epoch_public(Table* t) { step_epoch(t); }

}  // namespace tfra

extern "C" int tfra_table_apply_optimizer(tfra_table_t* tp, const tfra_opt_params* p, size_t n, const int64_t* keys,
                                          const float* grads, const void* param_defaults, int default_is_full,
                                          const int64_t* d_n, tfra_stream_t stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !p) return set_error(TFRA_ERR_INVALID, "apply_optimizer: null argument");
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(t->mu);
  int rc = t->enter(s);
  if (rc) return rc;
  if (n == 0) return TFRA_OK;
  if (!keys || !grads || !param_defaults) return set_error(TFRA_ERR_INVALID, "apply_optimizer: null buffer");
  const int dt = t->opts.value_dtype;
  if (dt != TFRA_F32 && dt != TFRA_F16 && dt != TFRA_BF16)
    return set_error(TFRA_ERR_UNSUPPORTED, "apply_optimizer: value_dtype must be float32, float16 or bfloat16 (gradients and defaults are float32)");
  int need = p->kind == TFRA_OPT_SGD ? 0 : (p->kind == TFRA_OPT_ADAGRAD ? 1 : 2);
  if (p->kind < 0 || p->kind > TFRA_OPT_FTRL) return set_error(TFRA_ERR_INVALID, "apply_optimizer: unknown kind");
  if (t->opts.aux_fields < need)
    return set_error(TFRA_ERR_INVALID, "apply_optimizer: table has " + std::to_string(t->opts.aux_fields) +
                                           " aux fields, optimizer needs " + std::to_string(need));
  rc = t->prepare_insert(n, s);
  if (rc) return rc;
  TableView v = t->view_of(t->cur);
  OptP o{p->kind, p->lr, p->beta1, p->beta2, p->eps, p->l1, p->l2, p->lr_power, p->d_lr};
  int dim = t->opts.dim;
  bool vec4 = dim % 4 == 0 && (((uintptr_t)grads | (uintptr_t)param_defaults) % 16 == 0);
  dim3 grid((unsigned)((n * 16 + 255) / 256));
  const i64* k = (const i64*)keys;
  const float* d = (const float*)param_defaults;
  float a0 = t->opts.aux_init[0], a1 = t->opts.aux_init[1];
  uint8_t* deferred;
  rc = t->bounded_flags(n, s, &deferred);
  if (rc) return rc;
  const ScoreP sp{t->opts.strategy, t->global_epoch, deferred ? (t->dense ? 2 : 1) : 0};
  switch (p->kind) {
    case TFRA_OPT_SGD: launch_apply<TFRA_OPT_SGD>(dt, vec4, grid, s, v, o, n, k, grads, d, default_is_full, dim, a0, a1, (const i64*)d_n, sp, deferred); break;
    case TFRA_OPT_ADAM: launch_apply<TFRA_OPT_ADAM>(dt, vec4, grid, s, v, o, n, k, grads, d, default_is_full, dim, a0, a1, (const i64*)d_n, sp, deferred); break;
    case TFRA_OPT_ADAGRAD: launch_apply<TFRA_OPT_ADAGRAD>(dt, vec4, grid, s, v, o, n, k, grads, d, default_is_full, dim, a0, a1, (const i64*)d_n, sp, deferred); break;
    default: launch_apply<TFRA_OPT_FTRL>(dt, vec4, grid, s, v, o, n, k, grads, d, default_is_full, dim, a0, a1, (const i64*)d_n, sp, deferred); break;
  }
  step_epoch(t);
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "apply_optimizer: launch failed");
  return TFRA_OK;
}
