// mke_gemm.h — internal interface of the hand-written f32 MFMA GEMMs (mke_gemm.hip) for the other translation units.
#pragma once
#include "mke_common.h"

namespace mke {

#define MKE_ACT_NONE 0
#define MKE_ACT_TANH 1
#define MKE_ACT_SIGMOID 2

// Fused epilogue of launch_gemm_f32_ex.  With v = alpha * (A B)[row][col] (+ bias[col]), in this order:
//   v = act(v);                                   act'(y) below is expressed in the activation's OUTPUT y
//   target:   e = v - target[row][col]; sumsq += e^2; v = target_scale * e * act'(v)      (loss tail: d loss / d pre-activation)
//   else:     sumsq += v^2  (when sumsq is given)
//   dact_y:   v *= dact_act'(dact_y[row][col])    (back-propagation through the previous layer's activation dact_act)
//   dot_with: dot += v * dot_with[row][col]
//   C[row][col] = v;  colsum[col] += v            (bias gradient)
// sumsq / dot are [MKE_LOSS_PARTIALS] double arrays that must be ZERO on entry; block b adds into slot b % MKE_LOSS_PARTIALS.
// colsum must be zero (or hold a running sum) on entry.  Everything except alpha needs a single K split.
struct GemmEpilogue {
  const float* alpha = nullptr;
  const float* bias = nullptr;
  int act = MKE_ACT_NONE;
  const float* dact_y = nullptr;
  int64_t ld_dact = 0;
  int dact_act = MKE_ACT_NONE;
  const float* target = nullptr;
  int64_t ld_target = 0;
  float target_scale = 0.f;
  double* sumsq = nullptr;
  const float* dot_with = nullptr;
  int64_t ld_dot = 0;
  double* dot = nullptr;
  float* colsum = nullptr;
};

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == MKE_ACT_TANH) return tanhf(v);
  if (act == MKE_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}
__device__ __forceinline__ float act_grad_from_output(float y, int act) {
  if (act == MKE_ACT_TANH) return 1.0f - y * y;
  if (act == MKE_ACT_SIGMOID) return y * (1.0f - y);
  return 1.0f;
}
__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int launch_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                    int M, int N, int K, int splits, int accumulate, hipStream_t st, double* tanh_sumsq_partials, int epi_plain);
bool launch_gemm_tallsplit_plus(const float* A0, int64_t a0_rs, int64_t a0_cs, const float* B0, int64_t b0_rs, float* C0, int64_t ldc0,
                                int M0, int N0, int K0, const float* A1, int64_t a1_rs, int64_t a1_cs, const float* B1, int64_t b1_rs,
                                int64_t b1_cs, float* C1, int64_t ldc1, int M1, int N1, int K1, hipStream_t st, int* rc);
int launch_gemm_f32_pair(const float* A0, int64_t a0_rs, int64_t a0_cs, const float* B0, int64_t b0_rs, int64_t b0_cs, float* C0,
                         int64_t ldc0, int M0, int N0, int K0, int splits0, int acc0, const float* A1, int64_t a1_rs, int64_t a1_cs,
                         const float* B1, int64_t b1_rs, int64_t b1_cs, float* C1, int64_t ldc1, int M1, int N1, int K1, int splits1,
                         int acc1, hipStream_t st);
// C (=|+=) epilogue(op(A) op(B)); A(i,k) = A[i*a_rs + k*a_cs], B(k,j) = B[k*b_rs + j*b_cs].  splits <= 0: chosen here.
int launch_gemm_f32_ex(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                       int M, int N, int K, int splits, int accumulate, hipStream_t st, const GemmEpilogue* epi);

}  // namespace mke
