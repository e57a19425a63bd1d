// Optimizer update rules shared by the fused write-back kernels (tfra_optim.hip,
// tfra_csr.hip).  TF's ResourceApply{GradientDescent,Adam,Adagrad[V2],Ftrl}, fp32, same
// operation order as oracle/optimizers.py; compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "../../include/tfra_mi355x.h"

namespace tfra {

struct OptP {
  int kind;
  float lr, beta1, beta2, eps, l1, l2, lr_power;
  const float* d_lr;  // optional device scalar overriding lr
};

__device__ __forceinline__ float sgnf(float z) { return (z > 0.f) ? 1.f : ((z < 0.f) ? -1.f : 0.f); }

template <int KIND>
__device__ __forceinline__ void apply_one(const OptP& o, float g, float& p, float& s1, float& s2) {
  if (KIND == TFRA_OPT_SGD) {
    p = p - o.lr * g;
  } else if (KIND == TFRA_OPT_ADAM) {
    // m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= lr_t*m/(sqrt(v)+eps)
    s1 = s1 + (g - s1) * (1.f - o.beta1);
    s2 = s2 + (g * g - s2) * (1.f - o.beta2);
    p = p - (s1 * o.lr) / (sqrtf(s2) + o.eps);
  } else if (KIND == TFRA_OPT_ADAGRAD) {
    s1 = s1 + g * g;
    if (o.eps < 0.f) p = p - o.lr * g / sqrtf(s1);
    else p = p - o.lr * g / (sqrtf(s1) + o.eps);
  } else {  // FTRL: s1 = accum, s2 = linear
    float a_new = s1 + g * g;
    float pa_new, pa;
    if (o.lr_power == -0.5f) { pa_new = sqrtf(a_new); pa = sqrtf(s1); }
    else { pa_new = powf(a_new, -o.lr_power); pa = powf(s1, -o.lr_power); }
    float sigma = (pa_new - pa) / o.lr;
    s2 = s2 + g - sigma * p;
    float q = pa_new / o.lr + 2.f * o.l2;
    p = (fabsf(s2) > o.l1) ? (sgnf(s2) * o.l1 - s2) / q : 0.f;
    s1 = a_new;
  }
}

// Rows of half / bfloat16 tables (GPU value types of the reference: hkv_hashtable_op_gpu.cu.cc:1133-1138) through the fused
// optimizers: the rule is evaluated in fp32 on the up-cast row and slots, the results are rounded to the storage type once,
// to nearest even.  ST: 0 float, 1 half, 2 bfloat16 (tfra_dtype).
template <int ST> struct Stored { typedef float T; };
template <> struct Stored<TFRA_F16> { typedef __half T; };
template <> struct Stored<TFRA_BF16> { typedef unsigned short T; };
template <int ST> __device__ __forceinline__ float load_stored(const typename Stored<ST>::T* p) { return *p; }
template <> __device__ __forceinline__ float load_stored<TFRA_F16>(const __half* p) { return __half2float(*p); }
template <> __device__ __forceinline__ float load_stored<TFRA_BF16>(const unsigned short* p) { return __uint_as_float((unsigned)*p << 16); }
template <int ST> __device__ __forceinline__ typename Stored<ST>::T to_stored(float x) { return x; }
template <> __device__ __forceinline__ __half to_stored<TFRA_F16>(float x) { return __float2half_rn(x); }
template <> __device__ __forceinline__ unsigned short to_stored<TFRA_BF16>(float x) {
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// score strategy of the table + its current epoch (update_score)
struct ScoreP { int strategy; u64 epoch; int bounded; };  // bounded: 0 / 1 / 2 (dense), see locate_or_claim_from

template <int KIND> struct NSlots { static constexpr int v = KIND == TFRA_OPT_SGD ? 0 : (KIND == TFRA_OPT_ADAGRAD ? 1 : 2); };


}  // namespace tfra
