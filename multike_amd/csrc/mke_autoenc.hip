// mke_autoenc.hip — training steps of the literal auto-encoder as one native call per epoch.
//
// What it computes is the graph of code/literal_encoder.py:41-91: encoder x W_i + b_i (optional sigmoid / tanh) down to the
// code, tf.nn.l2_normalize of the WHOLE code matrix (no axis, :65-66), mirrored decoder, loss = mean((decoded - x)^2),
// one Adagrad / SGD step over every weight and bias (:69).  How it is organised is new: every product is the hand-written
// f32 MFMA GEMM of mke_gemm.hip with the layer's elementwise work fused into its epilogue —
//   forward  : + bias, activation; the code layer also leaves the per-block partial sums of code^2; the first decoder
//              product takes the batch-wide 1/||code|| as a device scalar (alpha);  the last one turns its output straight
//              into d loss / d pre-activation = 2 (out - x) / size * act'(out) and leaves the loss partials;
//   backward : dH = dZ W^T with act'(H) multiplied in and the column sums (= the NEXT bias gradient) accumulated;
//              dW = H^T dZ split over K with atomic accumulation into the (zero-invariant) packed gradient buffer;
// no autograd, no library GEMM.  The backward through the batch-wide normalisation is two scalars and one small
// elementwise kernel.  Hand-derived gradients: oracle/literal_oracle.py holds the same derivation in NumPy.
#include "mke_gemm.h"

namespace mke {

// one block: out-of-band scalar work between two GEMMs.  Sums (and re-zeroes) a partials array.
//   mode 0: S = sum;  scal[0] = rsqrt(max(S, eps)) (or 1 when !normalize), scal[1] = S
//   mode 1: scal[2] = sum                                   (sum of dcn . code)
//   mode 2: loss_out[0] = sum / denom
__global__ __launch_bounds__(MKE_BLOCK) void k_ae_scalar(double* __restrict__ partials, float* __restrict__ scal, int mode,
                                                         double denom, double* __restrict__ loss_out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < MKE_LOSS_PARTIALS; i += MKE_BLOCK) {
    s += partials[i];
    partials[i] = 0.0;
  }
  __shared__ double sh[MKE_BLOCK / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < MKE_BLOCK / 64; ++w) t += sh[w];
    if (mode == 0) {
      scal[0] = (float)(1.0 / sqrt(t > (double)MKE_L2_EPS ? t : (double)MKE_L2_EPS));
      scal[1] = (float)t;
    } else if (mode == 1) {
      scal[2] = (float)t;
    } else {
      loss_out[0] = t / denom;
    }
  }
}

// Backward through out = code * inv, inv = rsqrt(max(S, eps)), S = sum code^2 over the whole batch, then through the
// code layer's activation:  dcode = inv * dcn - code * inv^3 * D  (D = sum dcn . code; the second term only when S > eps),
// dz = dcode * act'(code).  A block owns 16 rows (two 8-row strips x 128 columns); one atomic per column and strip for the bias gradient.
__global__ __launch_bounds__(MKE_BLOCK) void k_ae_norm_bwd(const float* __restrict__ dcn, const float* __restrict__ code,
                                                           float* __restrict__ dz, int64_t ld, int M, int d, int act,
                                                           int normalize, const float* __restrict__ scal,
                                                           float* __restrict__ colsum) {
  const float inv = normalize ? scal[0] : 1.0f;
  const float coef = (normalize && scal[1] > MKE_L2_EPS) ? inv * inv * inv * scal[2] : 0.f;
  // 16 rows per block; threads = (row half, column): two 8-row strips of up to 128 columns per pass
  const int r0 = blockIdx.x * 16 + (threadIdx.x >> 7) * 8, r1 = min(M, r0 + 8);
  for (int c = threadIdx.x & 127; c < d; c += 128) {
    float cs = 0.f;
    for (int r = r0; r < r1; ++r) {
      const float y = code[(int64_t)r * ld + c];
      const float v = (inv * dcn[(int64_t)r * ld + c] - coef * y) * act_grad_from_output(y, act);
      dz[(int64_t)r * ld + c] = v;
      cs += v;
    }
    if (r0 < r1) atomic_add_f32(colsum + c, cs);
  }
}

static inline int64_t pad4(int64_t x) { return (x + 3) & ~(int64_t)3; }

}  // namespace mke

extern "C" int64_t mke_ae_scratch_floats(const mke_ae_plan* pl, int64_t rows) {
  if (!pl || pl->n_layers < 1 || pl->n_layers > MKE_AE_MAX_LAYERS || rows < 0) return -1;
  int64_t per_row = 0;
  for (int i = 0; i <= pl->n_layers; ++i) per_row += mke::pad4(pl->dims[i]);
  // encoder activations H_1..H_n, decoder activations D_1..D_{n-1} and the output gradient (width dims[0]), two
  // ping-pong gradient buffers of the widest layer, the code-sized pair (dcn, dz): < 4 x the sum of the widths
  return rows * per_row * 4 + 64;
}

namespace mke {
// One batch, the phases selected by the bit mask (MKE_AE_ALL = the whole step).  m_global: rows of the GLOBAL batch (the loss is
// a mean over it; the ranks of a data-parallel run each hold M of them).
static int ae_batch(const mke_ae_plan* pl, const float* X, int M, int64_t ldx, int64_t m_global, int phases, double* loss_out,
                    hipStream_t st) {
  const int n = pl->n_layers;
  const int* d = pl->dims;
  const int act = pl->act;
  double* P_ssq = pl->partials;
  double* P_dot = pl->partials + MKE_LOSS_PARTIALS;
  double* P_loss = pl->partials + 2 * MKE_LOSS_PARTIALS;
  float* scal = pl->scalars;
  void* stream = (void*)st;
  int rc = MKE_OK;
#define AE_RUN(x) do { rc = (x); if (rc) return rc; } while (0)
  {
    // ---- scratch carve-up (leading dimensions padded to 4 floats = 16 bytes) ----
    float* sp = pl->scratch;
    auto take = [&](int width) { float* q = sp; sp += (int64_t)M * pad4(width); return q; };
    float* H[MKE_AE_MAX_LAYERS + 1];   // encoder activations, H[0] = X
    float* D[MKE_AE_MAX_LAYERS + 1];   // decoder activations, D[0] = code (normalisation folded into alpha), D[n] = dZ of the output
    H[0] = const_cast<float*>(X);
    for (int i = 1; i <= n; ++i) H[i] = take(d[i]);
    D[0] = H[n];
    for (int j = 1; j <= n; ++j) D[j] = take(d[n - j]);
    int wmax = 0;
    for (int i = 0; i <= n; ++i) wmax = d[i] > wmax ? d[i] : wmax;
    float* G[2] = {take(wmax), take(wmax)};
    float* dcn = take(d[n]);
    float* dzc = take(d[n]);
    auto ldH = [&](int i) { return i == 0 ? ldx : pad4(d[i]); };
    auto ldD = [&](int j) { return pad4(d[n - j]); };

    // ---- forward: encoder ----
    if (phases & MKE_AE_ENC)
    for (int i = 0; i < n; ++i) {
      GemmEpilogue e;
      e.bias = pl->params + pl->b_off[i];
      e.act = act;
      if (i == n - 1 && pl->normalize) e.sumsq = P_ssq;
      AE_RUN(launch_gemm_f32_ex(H[i], ldH(i), 1, pl->params + pl->w_off[i], pad4(d[i + 1]), 1, H[i + 1], ldH(i + 1), M, d[i + 1], d[i], 1, 0, st, &e));
    }
    if (phases & MKE_AE_DEC) {
    if (pl->normalize) hipLaunchKernelGGL(k_ae_scalar, dim3(1), dim3(MKE_BLOCK), 0, st, P_ssq, scal, 0, 1.0, (double*)nullptr);   // scal[0] = 1 / ||code||
    // ---- forward: decoder (layer j maps width d[n-j] -> d[n-j-1]) ----
    for (int j = 0; j < n; ++j) {
      GemmEpilogue e;
      e.bias = pl->params + pl->b_off[n + j];
      e.act = act;
      if (j == 0 && pl->normalize) e.alpha = scal;
      if (j == n - 1) {  // loss tail: D[n] = d loss / d pre-activation of the output layer; its column sums = the output bias gradient
        e.target = X; e.ld_target = ldx; e.target_scale = 2.0f / ((float)m_global * (float)d[0]);
        e.sumsq = P_loss;
        e.colsum = pl->grads + pl->b_off[n + j];
      }
      AE_RUN(launch_gemm_f32_ex(D[j], j == 0 ? ldH(n) : ldD(j), 1, pl->params + pl->w_off[n + j], pad4(d[n - j - 1]), 1, D[j + 1], ldD(j + 1), M,
                                d[n - j - 1], d[n - j], 1, 0, st, &e));
    }
    hipLaunchKernelGGL(k_ae_scalar, dim3(1), dim3(MKE_BLOCK), 0, st, P_loss, scal, 2, (double)m_global * (double)d[0], loss_out);

    // ---- backward: decoder.  dZ of layer j is `dz` (width d[n-j-1]); its input is D[j] (times inv for j = 0) ----
    const float* dz = D[n];
    int64_t ld_dz = ldD(n);
    int gsel = 0;
    for (int j = n - 1; j >= 0; --j) {
      const int din = d[n - j], dout = d[n - j - 1];
      const float* Win = pl->params + pl->w_off[n + j];
      {  // dW_j = D[j]^T dz  (x inv for the first decoder layer: its input was the normalised code)
        GemmEpilogue e;
        if (j == 0 && pl->normalize) e.alpha = scal;
        AE_RUN(launch_gemm_f32_ex(D[j], 1, j == 0 ? ldH(n) : ldD(j), dz, ld_dz, 1, pl->grads + pl->w_off[n + j], pad4(dout), din, dout, M, 0, 1, st, &e));
      }
      if (j > 0) {  // dZ_{j-1} = (dz W_j^T) * act'(D[j]);  column sums -> bias gradient of decoder layer j-1
        GemmEpilogue e;
        e.dact_act = act;
        e.dact_y = D[j]; e.ld_dact = ldD(j);
        e.colsum = pl->grads + pl->b_off[n + j - 1];
        float* out = G[gsel]; gsel ^= 1;
        AE_RUN(launch_gemm_f32_ex(dz, ld_dz, 1, Win, 1, pad4(dout), out, pad4(din), M, din, dout, 1, 0, st, &e));
        dz = out; ld_dz = pad4(din);
      } else {      // gradient w.r.t. the normalised code, and sum dcn . code for the normalisation's backward
        GemmEpilogue e;
        if (pl->normalize) { e.dot_with = H[n]; e.ld_dot = ldH(n); e.dot = P_dot; }
        AE_RUN(launch_gemm_f32_ex(dz, ld_dz, 1, Win, 1, pad4(dout), dcn, pad4(din), M, din, dout, 1, 0, st, &e));
      }
    }
    }   // MKE_AE_DEC
    if (phases & MKE_AE_BWD) {
    if (pl->normalize) hipLaunchKernelGGL(k_ae_scalar, dim3(1), dim3(MKE_BLOCK), 0, st, P_dot, scal, 1, 1.0, (double*)nullptr);
    if (M > 0)
      hipLaunchKernelGGL(k_ae_norm_bwd, dim3((M + 15) / 16), dim3(MKE_BLOCK), 0, st, dcn, H[n], dzc, pad4(d[n]), M, d[n], act, pl->normalize,
                         scal, pl->grads + pl->b_off[n - 1]);
    // ---- backward: encoder ----
    const float* dz = dzc;
    int64_t ld_dz = pad4(d[n]);
    int gsel = 0;
    for (int i = n - 1; i >= 0; --i) {
      const int din = d[i], dout = d[i + 1];
      AE_RUN(launch_gemm_f32_ex(H[i], 1, ldH(i), dz, ld_dz, 1, pl->grads + pl->w_off[i], pad4(dout), din, dout, M, 0, 1, st, nullptr));
      if (i > 0) {
        GemmEpilogue e;
        e.dact_act = act;
        e.dact_y = H[i]; e.ld_dact = ldH(i);
        e.colsum = pl->grads + pl->b_off[i - 1];
        float* out = G[gsel]; gsel ^= 1;
        AE_RUN(launch_gemm_f32_ex(dz, ld_dz, 1, pl->params + pl->w_off[i], 1, pad4(dout), out, pad4(din), M, din, dout, 1, 0, st, &e));
        dz = out; ld_dz = pad4(din);
      }
    }
    }   // MKE_AE_BWD
    if ((rc = check_launch("mke_ae_train_steps"))) return rc;
    if ((phases & MKE_AE_UPD) && pl->update) AE_RUN(mke_dense_update(pl->params, pl->optimizer == MKE_OPT_ADAGRAD ? pl->acc : nullptr, pl->grads, pl->n_params,
                                            pl->optimizer, pl->lr, stream));
  }
#undef AE_RUN
  return MKE_OK;
}

static int ae_check(const mke_ae_plan* pl, const char* who) {
  if (!pl) { set_error("%s: NULL plan", who); return MKE_E_NULL; }
  const int n = pl->n_layers;
  if (n < 1 || n > MKE_AE_MAX_LAYERS) { set_error("%s: n_layers must be in [1,%d]", who, MKE_AE_MAX_LAYERS); return MKE_E_SHAPE; }
  for (int i = 0; i <= n; ++i)
    if (pl->dims[i] < 1) { set_error("%s: bad layer width", who); return MKE_E_SHAPE; }
  if (pl->act != MKE_ACT_NONE && pl->act != MKE_ACT_TANH && pl->act != MKE_ACT_SIGMOID) { set_error("unknown activation %d", pl->act); return MKE_E_UNSUPPORTED; }
  if (pl->optimizer != MKE_OPT_ADAGRAD && pl->optimizer != MKE_OPT_SGD) { set_error("%s: Adagrad or SGD (update = 0 leaves the gradients to the caller)", who); return MKE_E_UNSUPPORTED; }
  if (!pl->params || !pl->grads || !pl->scratch || !pl->partials || !pl->scalars) { set_error("%s: NULL pointer", who); return MKE_E_NULL; }
  if (pl->update && pl->optimizer == MKE_OPT_ADAGRAD && !pl->acc) { set_error("Adagrad needs an accumulator"); return MKE_E_NULL; }
  return MKE_OK;
}
}  // namespace mke

extern "C" int mke_ae_train_steps(const mke_ae_plan* pl, const float* x, int64_t n_rows, int64_t ldx, int64_t batch_rows,
                                  double* loss_out, void* stream) {
  using namespace mke;
  int rc = ae_check(pl, "mke_ae_train_steps");
  if (rc) return rc;
  if (n_rows < 0 || batch_rows < 1 || ldx < pl->dims[0]) { set_error("mke_ae_train_steps: bad row counts / ldx"); return MKE_E_SHAPE; }
  if (n_rows == 0) return MKE_OK;
  if (!x || !loss_out) { set_error("mke_ae_train_steps: NULL pointer"); return MKE_E_NULL; }
  if (pl->scratch_floats < mke_ae_scratch_floats(pl, batch_rows < n_rows ? batch_rows : n_rows)) { set_error("mke_ae_train_steps: scratch too small"); return MKE_E_SHAPE; }
  int64_t batch_i = 0;
  for (int64_t r0 = 0; r0 < n_rows; r0 += batch_rows, ++batch_i) {
    const int M = (int)((n_rows - r0) < batch_rows ? (n_rows - r0) : batch_rows);
    if ((rc = ae_batch(pl, x + r0 * ldx, M, ldx, M, MKE_AE_ALL, loss_out + batch_i, (hipStream_t)stream))) return rc;
  }
  return MKE_OK;
}

// One batch cut at its batch-wide sums, for data-parallel training (every rank holds the replicated parameters and `rows` of the
// `global_rows` of the batch): see include/multike_hip.h.
extern "C" int mke_ae_step_phases(const mke_ae_plan* pl, const float* x, int64_t rows, int64_t ldx, int64_t global_rows, int phases,
                                  double* loss_out, void* stream) {
  using namespace mke;
  int rc = ae_check(pl, "mke_ae_step_phases");
  if (rc) return rc;
  if (phases <= 0 || phases > MKE_AE_ALL) { set_error("mke_ae_step_phases: bad phase mask"); return MKE_E_SHAPE; }
  if (rows < 0 || global_rows < rows || global_rows < 1 || rows > 0x7FFFFFFF || ldx < pl->dims[0]) { set_error("mke_ae_step_phases: bad row counts / ldx"); return MKE_E_SHAPE; }
  if ((rows > 0 && !x) || !loss_out) { set_error("mke_ae_step_phases: NULL pointer"); return MKE_E_NULL; }
  if (pl->scratch_floats < mke_ae_scratch_floats(pl, rows)) { set_error("mke_ae_step_phases: scratch too small"); return MKE_E_SHAPE; }
  return ae_batch(pl, x, (int)rows, ldx, global_rows, phases, loss_out, (hipStream_t)stream);
}

extern "C" int mke_ae_encode(const mke_ae_plan* pl, const float* x, int64_t n_rows, int64_t ldx, float* out, int64_t ld_out,
                             void* stream) {
  using namespace mke;
  if (!pl || !x || !out || !pl->params || !pl->scratch) { set_error("mke_ae_encode: NULL pointer"); return MKE_E_NULL; }
  const int n = pl->n_layers;
  if (n < 1 || n > MKE_AE_MAX_LAYERS || n_rows < 0 || ldx < pl->dims[0] || ld_out < pl->dims[n]) { set_error("mke_ae_encode: bad shape"); return MKE_E_SHAPE; }
  if (n_rows == 0) return MKE_OK;
  if (pl->scratch_floats < mke_ae_scratch_floats(pl, n_rows)) { set_error("mke_ae_encode: scratch too small"); return MKE_E_SHAPE; }
  const int* d = pl->dims;
  const int M = (int)n_rows;
  float* sp = pl->scratch;
  const float* in = x;
  int64_t ld_in = ldx;
  for (int i = 0; i < n; ++i) {
    float* o = i == n - 1 ? out : sp;
    const int64_t ld_o = i == n - 1 ? ld_out : pad4(d[i + 1]);
    if (i < n - 1) sp += (int64_t)M * ld_o;
    GemmEpilogue e;
    e.bias = pl->params + pl->b_off[i];
    e.act = pl->act;
    const int rc = launch_gemm_f32_ex(in, ld_in, 1, pl->params + pl->w_off[i], pad4(d[i + 1]), 1, o, ld_o, M, d[i + 1], d[i], 1, 0, (hipStream_t)stream, &e);
    if (rc) return rc;
    in = o; ld_in = ld_o;
  }
  return MKE_OK;
}

extern "C" int mke_dense_layer_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* b, int act, float* out,
                                   int64_t ld_out, int M, int N, int K, void* stream) {
  using namespace mke;
  if (M < 0 || N < 0 || K < 0 || ldx < K || ldw < N || ld_out < N) { set_error("mke_dense_layer_fwd: bad shape"); return MKE_E_SHAPE; }
  if (act != MKE_ACT_NONE && act != MKE_ACT_TANH && act != MKE_ACT_SIGMOID) { set_error("unknown activation %d", act); return MKE_E_UNSUPPORTED; }
  if (M == 0 || N == 0) return MKE_OK;
  if (!x || !w || !out) { set_error("mke_dense_layer_fwd: NULL pointer"); return MKE_E_NULL; }
  if (K == 0) { set_error("mke_dense_layer_fwd: K = 0"); return MKE_E_SHAPE; }
  GemmEpilogue e;
  e.bias = b;
  e.act = act;
  return launch_gemm_f32_ex(x, ldx, 1, w, ldw, 1, out, ld_out, M, N, K, 1, 0, (hipStream_t)stream, &e);
}
