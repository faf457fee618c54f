// mke_dense_opt.hip — Adam and Adadelta (TF1 `tf.train.AdamOptimizer` / `tf.train.AdadeltaOptimizer`, selectable through
// `args.optimizer`, code/MultiKE_model.py:15-25) for the embedding tables and the packed CNN / auto-encoder parameters.
//
// Unlike Adagrad and SGD these rules move a weight whose gradient is zero (Adam: m decays and keeps pushing; Adadelta:
// both accumulators decay), and TF applies them to the WHOLE variable because the gradient that reaches a table through
// `tf.nn.l2_normalize(table, 1)` is dense.  "Visit the touched rows only" is therefore not the same function: these
// kernels stream every row — grad, w and two slot rows in, 0, w and two slot rows out (8 row streams: ~0.5 GB per step at
// the DBP-WD shape, ~100 us) — which is what the reference's TF graph does with these optimizers, and why its default
// (and this build's tuned path) is Adagrad.
//
//   Adam     (slots m, v; step t >= 1):  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//                                        w -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)         (TF's "epsilon hat" form)
//   Adadelta (slots accum, accum_update): accum = rho accum + (1-rho) g^2 ; u = sqrt(accum_update+eps) rsqrt(accum+eps) g ;
//                                        w -= lr u ; accum_update = rho accum_update + (1-rho) u^2
#include "mke_common.h"

namespace mke {

struct OptHyper {
  int kind;
  float lr, b1, b2, eps, rho;
  float lr_t;  // Adam: lr * sqrt(1 - b2^t) / (1 - b1^t)
};

__device__ __forceinline__ float opt_step(const OptHyper& o, float w, float g, float& s1, float& s2) {
  if (o.kind == MKE_OPT_ADAM) {
    s1 = o.b1 * s1 + (1.0f - o.b1) * g;
    s2 = o.b2 * s2 + (1.0f - o.b2) * g * g;
    return w - o.lr_t * s1 / (sqrtf(s2) + o.eps);
  }
  s1 = o.rho * s1 + (1.0f - o.rho) * g * g;
  const float u = sqrtf(s2 + o.eps) * rsqrtf(s1 + o.eps) * g;
  s2 = o.rho * s2 + (1.0f - o.rho) * u * u;
  return w - o.lr * u;
}

struct DenseRowsParams {
  float* __restrict__ table;
  float* __restrict__ s1;
  float* __restrict__ s2;
  float* __restrict__ grad;
  int64_t n_rows;
  int stride, dim, normalize;
  OptHyper o;
};

// one 16-lane quarter-wave per row, every row of the table
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_rows_update_dense(const DenseRowsParams p) {
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  for (int64_t row = sub0; row < p.n_rows; row += nsub) {
    const int64_t off = row * (int64_t)p.stride + j;
    float g[FPL], w[FPL], a[FPL], b[FPL];
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      g[k] = p.grad[off + k * 16];
      w[k] = p.table[off + k * 16];
      a[k] = p.s1[off + k * 16];
      b[k] = p.s2[off + k * 16];
    }
    if (p.normalize) {
      float s = 0.f, dot = 0.f;
#pragma unroll
      for (int k = 0; k < FPL; ++k) {
        s = fmaf(w[k], w[k], s);
        dot = fmaf(w[k], g[k], dot);
      }
      s = sub16_sum(s);
      dot = sub16_sum(dot);
      const float inv = rsqrtf(fmaxf(s, MKE_L2_EPS));
      const float coef = (s > MKE_L2_EPS) ? dot * inv * inv : 0.f;
#pragma unroll
      for (int k = 0; k < FPL; ++k) g[k] = (g[k] - w[k] * coef) * inv;
    }
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      const bool col = k * 16 + j < p.dim;  // pad columns stay zero (eps would otherwise leak into them)
      const float nw = opt_step(p.o, w[k], g[k], a[k], b[k]);
      p.grad[off + k * 16] = 0.f;
      if (col) {
        p.table[off + k * 16] = nw;
        p.s1[off + k * 16] = a[k];
        p.s2[off + k * 16] = b[k];
      }
    }
  }
}

__global__ __launch_bounds__(MKE_BLOCK) void k_dense_update_opt(float* __restrict__ w, float* __restrict__ s1, float* __restrict__ s2,
                                                                float* __restrict__ g, int64_t n, const OptHyper o) {
  for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MKE_BLOCK) {
    const float gv = g[i];
    g[i] = 0.f;
    float a = s1[i], b = s2[i];
    w[i] = opt_step(o, w[i], gv, a, b);
    s1[i] = a;
    s2[i] = b;
  }
}

static int make_hyper(const mke_optimizer* opt, OptHyper& o, const char* who) {
  if (!opt) { set_error("%s: NULL optimizer", who); return MKE_E_NULL; }
  if (opt->kind != MKE_OPT_ADAM && opt->kind != MKE_OPT_ADADELTA) { set_error("%s: optimizer kind %d is not Adam / Adadelta (Adagrad and SGD use the touched-rows entry points)", who, opt->kind); return MKE_E_UNSUPPORTED; }
  o.kind = opt->kind; o.lr = opt->lr; o.b1 = opt->beta1; o.b2 = opt->beta2; o.eps = opt->epsilon; o.rho = opt->rho; o.lr_t = opt->lr;
  if (opt->kind == MKE_OPT_ADAM) {
    if (opt->step < 1) { set_error("%s: Adam needs step >= 1 (the number of this update)", who); return MKE_E_RANGE; }
    if (!(opt->beta1 >= 0.f && opt->beta1 < 1.f && opt->beta2 >= 0.f && opt->beta2 < 1.f)) { set_error("%s: betas must be in [0,1)", who); return MKE_E_RANGE; }
    const double t = (double)opt->step;
    o.lr_t = (float)((double)opt->lr * std::sqrt(1.0 - std::pow((double)opt->beta2, t)) / (1.0 - std::pow((double)opt->beta1, t)));
  }
  return MKE_OK;
}

}  // namespace mke

extern "C" int mke_rows_update_dense(float* table, float* slot1, float* slot2, float* grad, int64_t n_rows, int stride, int dim,
                                     int normalize, const mke_optimizer* opt, void* stream) {
  using namespace mke;
  DenseRowsParams p;
  int rc = make_hyper(opt, p.o, "mke_rows_update_dense");
  if (rc) return rc;
  if (n_rows < 0) { set_error("negative n_rows"); return MKE_E_SHAPE; }
  if (stride <= 0 || stride % 16 != 0 || dim <= 0 || dim > stride || stride > MKE_MAX_STRIDE) { set_error("bad stride/dim: stride=%d dim=%d", stride, dim); return MKE_E_SHAPE; }
  if (n_rows == 0) return MKE_OK;
  if (!table || !slot1 || !slot2 || !grad) { set_error("mke_rows_update_dense: NULL pointer"); return MKE_E_NULL; }
  p.table = table; p.s1 = slot1; p.s2 = slot2; p.grad = grad; p.n_rows = n_rows; p.stride = stride; p.dim = dim; p.normalize = normalize;
  int64_t blocks = (n_rows + (MKE_BLOCK / 16) - 1) / (MKE_BLOCK / 16);
  if (blocks > 8192) blocks = 8192;
  const int fpl = stride / 16;
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_rows_update_dense<FPL>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  });
  return check_launch("k_rows_update_dense");
}

extern "C" int mke_dense_update_opt(float* param, float* slot1, float* slot2, float* grad, int64_t n, const mke_optimizer* opt,
                                    void* stream) {
  using namespace mke;
  OptHyper o;
  int rc = make_hyper(opt, o, "mke_dense_update_opt");
  if (rc) return rc;
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!param || !slot1 || !slot2 || !grad) { set_error("mke_dense_update_opt: NULL pointer"); return MKE_E_NULL; }
  int64_t blocks = (n + MKE_BLOCK - 1) / MKE_BLOCK;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_dense_update_opt, dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, param, slot1, slot2, grad, n, o);
  return check_launch("k_dense_update_opt");
}
