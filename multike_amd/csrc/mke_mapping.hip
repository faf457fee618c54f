// mke_mapping.hip — the space-mapping step of the SSL driver (code/losses.py:53-63 `space_mapping_loss`, `orthogonal_loss`;
// graph code/MultiKE_model.py:241-261; loop :439-454) for gfx950.
//
// Per view k (name, relation, attribute) with mapping matrix M_k [d x d]:
//     P = V_k[idx] @ M_k ;  out = P / ||P||_F  (tf.nn.l2_normalize with NO axis: the whole [B, d] batch) ;
//     loss_k = sum (F - out)^2 + w ||M_k M_k^T - I||_F^2 + norm_w ||M_k||_F^2 ,   F = final (shared) embeddings of idx
// and one optimizer step over the three mapping matrices and the touched rows of the shared table.
//
// Shape of one step (everything enqueue-only): gather F; per view: gather V, MFMA GEMM with a sum-of-squares epilogue,
// tail 1 (difference, loss, gradient w.r.t. F accumulated over the views, g_out, partial sums of g_out.out), tail 2 (through
// the batch-wide normalisation), MFMA GEMM V^T dP (split-K) into the matrix gradient; then one block per matrix for the
// orthogonality / norm terms, a row scatter of the F gradient, and one launch for the row update + the dense update of the
// three matrices.  The batch-wide sums travel as per-block partials that the next kernel's blocks add up themselves.
#include "mke_common.h"

namespace mke {

int launch_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                    int M, int N, int K, int splits, int accumulate, hipStream_t st, double* tanh_sumsq_partials, int epi_plain);
int launch_rows_update_multi(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim, int optimizer,
                             float lr, hipStream_t st, const mke_count_job* count, const DenseJob* dense);

__device__ __forceinline__ double partials_total(const double* __restrict__ partials, double* s_w, double* s_tot) {
  double v = 0.0;
  for (int i = threadIdx.x; i < MKE_LOSS_PARTIALS; i += MKE_BLOCK) v += partials[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < MKE_BLOCK / 64; ++w) t += s_w[w];
    *s_tot = t;
  }
  __syncthreads();
  return *s_tot;
}

// out = P * inv ; diff = F - out ; loss = sum diff^2 ; GF (+)= 2 diff ; G = -2 diff (= dL/dout) ; dot = sum G * out
__global__ __launch_bounds__(MKE_BLOCK) void k_map_tail1(const float* __restrict__ P, const float* __restrict__ F, const double* __restrict__ ssq,
                                                         int64_t total, float* __restrict__ G, float* __restrict__ GF, int first_view,
                                                         double* __restrict__ lossp, double* __restrict__ dotp) {
  __shared__ double s_w[MKE_BLOCK / 64], s_tot;
  const double S = partials_total(ssq, s_w, &s_tot);
  const float inv = rsqrtf(fmaxf((float)S, MKE_L2_EPS));
  float loss = 0.f, dot = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * MKE_BLOCK) {
    const float out = P[i] * inv;
    const float diff = F[i] - out;
    loss = fmaf(diff, diff, loss);
    const float g = -2.0f * diff;
    G[i] = g;
    GF[i] = first_view ? 2.0f * diff : GF[i] + 2.0f * diff;
    dot = fmaf(g, out, dot);
  }
  const double lt = block_sum_double(loss);
  __syncthreads();
  const double dt = block_sum_double(dot);
  if (threadIdx.x == 0) {
    lossp[blockIdx.x] = lt;
    dotp[blockIdx.x] = dt;
    for (int k = blockIdx.x + gridDim.x; k < MKE_LOSS_PARTIALS; k += gridDim.x) { lossp[k] = 0.0; dotp[k] = 0.0; }   // slots no block owns
  }
}

// dP = inv * (G - out * T), out = P * inv, T = sum G * out  (inv * G when the batch norm sits on its epsilon floor); in place on G
__global__ __launch_bounds__(MKE_BLOCK) void k_map_tail2(const float* __restrict__ P, float* __restrict__ G, const double* __restrict__ ssq,
                                                         const double* __restrict__ dotp, int64_t total) {
  __shared__ double s_w[MKE_BLOCK / 64], s_tot;
  const double S = partials_total(ssq, s_w, &s_tot);
  __syncthreads();
  const double T = partials_total(dotp, s_w, &s_tot);
  const float inv = rsqrtf(fmaxf((float)S, MKE_L2_EPS));
  const float coef = (float)S > MKE_L2_EPS ? (float)T * inv * inv : 0.f;  // inv * out * T = P * inv^2 * T
  for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * MKE_BLOCK)
    G[i] = inv * G[i] - P[i] * coef;
}

// One block per mapping matrix: Q = M M^T - I in LDS; loss partial w * sum Q^2 + nw * sum M^2; gM += 4 w Q M + 2 nw M.
// d <= MKE_MAP_MAX_DIM so that Q and M fit in LDS together.
#define MKE_MAP_MAX_DIM 88   // 2 * d * (d + 1) floats <= 64 KB of dynamic LDS
__global__ __launch_bounds__(MKE_BLOCK) void k_map_ortho(const float* __restrict__ Ms, float* __restrict__ gMs, int d, float w, float nw,
                                                         double* __restrict__ lossp /* [MKE_LOSS_PARTIALS], zeroed beyond gridDim.x */) {
  extern __shared__ float sm[];
  float* M = sm;               // [d][d + 1]
  float* Q = sm + d * (d + 1); // [d][d + 1]
  const float* Mg = Ms + (int64_t)blockIdx.x * d * d;
  float* gM = gMs + (int64_t)blockIdx.x * d * d;
  const int ld = d + 1;
  float nrm = 0.f;
  for (int i = threadIdx.x; i < d * d; i += MKE_BLOCK) {
    const float v = Mg[i];
    M[(i / d) * ld + i % d] = v;
    nrm = fmaf(v, v, nrm);
  }
  __syncthreads();
  float orth = 0.f;
  for (int i = threadIdx.x; i < d * d; i += MKE_BLOCK) {
    const int r = i / d, c = i % d;
    float a = 0.f;
    for (int k = 0; k < d; ++k) a = fmaf(M[r * ld + k], M[c * ld + k], a);
    a -= (r == c) ? 1.0f : 0.f;
    Q[r * ld + c] = a;
    orth = fmaf(a, a, orth);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < d * d; i += MKE_BLOCK) {
    const int r = i / d, c = i % d;
    float a = 0.f;
    for (int k = 0; k < d; ++k) a = fmaf(Q[r * ld + k], M[k * ld + c], a);
    gM[i] += 4.0f * w * a + 2.0f * nw * M[r * ld + c];
  }
  const double t = block_sum_double(w * orth + nw * nrm);
  if (threadIdx.x == 0) {
    lossp[blockIdx.x] = t;
    for (int k = blockIdx.x + gridDim.x; k < MKE_LOSS_PARTIALS; k += gridDim.x) lossp[k] = 0.0;
  }
}

// grad[idx[i]][:] += rows[i][:] (dense [n][dim] rows into the strided gradient scratch), touched flags stored
__global__ __launch_bounds__(MKE_BLOCK) void k_scatter_dense_rows(const int32_t* __restrict__ idx, const float* __restrict__ rows, int64_t n,
                                                                  int dim, int stride, float* __restrict__ grad, int32_t* __restrict__ touched,
                                                                  int32_t tag) {
  const int64_t total = n * dim;
  for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * MKE_BLOCK) {
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    const int row = idx[r];
    atomic_add_f32(grad + (int64_t)row * stride + c, rows[i]);
    if (c == 0) touched[row] = tag;
  }
}

// phases (MKE_MAP_*): the single-GPU step is FWD | TAIL | BWD | UPD back to back; the row-sharded trainer
// (multike_amd/distributed_views.py) cuts it where the batch-wide sums live — between FWD and TAIL every rank's sum P_k^2,
// between TAIL and BWD every rank's sum G_k . out_k (it overwrites the partials with the all-reduced totals) — and all-reduces
// the data part of gM between BWD and UPD; the orthogonality / norm terms depend on the replicated matrices only and are added
// in UPD, after that reduction, identically on every rank.
static int mapping_step_impl(const mke_mapping_step_args* a, double* loss4, void* stream, int phases) {
  if (!a) { set_error("mke_mapping_step: NULL args"); return MKE_E_NULL; }
  const int d = a->dim;
  const int64_t n = a->n;
  if (n < 0 || d <= 0 || d > MKE_MAP_MAX_DIM || a->stride % 16 != 0 || d > a->stride || a->stride > MKE_MAX_STRIDE) { set_error("mke_mapping_step: bad n/dim/stride (dim <= %d)", MKE_MAP_MAX_DIM); return MKE_E_SHAPE; }
  if (a->n_views < 1 || a->n_views > MKE_MAPPING_MAX_VIEWS) { set_error("mke_mapping_step: n_views must be in [1,%d]", MKE_MAPPING_MAX_VIEWS); return MKE_E_SHAPE; }
  if (!a->ent_table || !a->M || !a->gM || !a->scratch || !a->partials || !loss4 || (n > 0 && !a->idx)) { set_error("mke_mapping_step: NULL pointer"); return MKE_E_NULL; }
  if (a->ent_grad && !a->ent_touched) { set_error("mke_mapping_step: NULL touched array"); return MKE_E_NULL; }
  if (!(phases & MKE_MAP_ALL)) { set_error("mke_mapping_step: empty phase mask"); return MKE_E_SHAPE; }
  for (int k = 0; k < a->n_views; ++k)
    if (!a->views[k].table) { set_error("mke_mapping_step: view %d has no table", k); return MKE_E_NULL; }
  hipStream_t st = (hipStream_t)stream;
  const int V = a->n_views;
  const int64_t total = n * d;
  float* F = a->scratch;          // F | GF | per view: V_k, P_k, G_k
  float* GF = F + total;
  auto Vr = [&](int k) { return GF + total + (int64_t)k * 3 * total; };
  auto Pm = [&](int k) { return Vr(k) + total; };
  auto Gm = [&](int k) { return Vr(k) + 2 * total; };
  auto ssq = [&](int k) { return a->partials + (int64_t)(2 * k) * MKE_LOSS_PARTIALS; };
  auto dot = [&](int k) { return a->partials + (int64_t)(2 * k + 1) * MKE_LOSS_PARTIALS; };
  int64_t eb = (total + MKE_BLOCK * 4 - 1) / (MKE_BLOCK * 4);
  eb = std::max<int64_t>(1, std::min<int64_t>(eb, MKE_LOSS_PARTIALS));
  int rc;
  auto zero = [&](double* p, int64_t blocks) -> int {
    hipError_t e = hipMemsetAsync(p, 0, sizeof(double) * blocks * MKE_LOSS_PARTIALS, st);
    if (e != hipSuccess) { set_error("mke_mapping_step: memset failed"); return (int)e; }
    return MKE_OK;
  };

  if (n == 0 && phases == MKE_MAP_ALL) {   // an empty step of the single-GPU loop: no loss, nothing moves
    return zero(loss4, MKE_MAPPING_MAX_VIEWS + 1);
  }
  if (phases & MKE_MAP_FWD) {
    if (n == 0) {
      for (int k = 0; k < V; ++k) if ((rc = zero(ssq(k), 1))) return rc;   // this part adds nothing to the batch-wide sums
    } else {
      if ((rc = mke_gather_rows(a->ent_table, a->ent_normalize, a->stride, d, a->idx, n, F, stream))) return rc;
      for (int k = 0; k < V; ++k) {
        if ((rc = mke_gather_rows(a->views[k].table, a->views[k].normalize, a->stride, d, a->idx, n, Vr(k), stream))) return rc;
        if ((rc = launch_gemm_f32(Vr(k), d, 1, a->M + (int64_t)k * d * d, d, 1, Pm(k), d, (int)n, d, d, 1, 0, st, ssq(k), 1))) return rc;  // P = V M, sum P^2
      }
    }
  }
  if (phases & MKE_MAP_TAIL) {
    if (n == 0) {
      for (int k = 0; k < V; ++k) {
        if ((rc = zero(dot(k), 1))) return rc;
        if ((rc = zero(loss4 + (int64_t)k * MKE_LOSS_PARTIALS, 1))) return rc;
      }
    } else {
      for (int k = 0; k < V; ++k) {
        // every partial slot is rewritten: a block clears the slots beyond the grid (each block re-adds all of them)
        hipLaunchKernelGGL(k_map_tail1, dim3((unsigned)eb), dim3(MKE_BLOCK), 0, st, Pm(k), F, ssq(k), total, Gm(k), GF, k == 0 ? 1 : 0,
                           loss4 + (int64_t)k * MKE_LOSS_PARTIALS, dot(k));
        if ((rc = check_launch("k_map_tail1"))) return rc;
      }
    }
    for (int k = V; k < MKE_MAPPING_MAX_VIEWS; ++k) if ((rc = zero(loss4 + (int64_t)k * MKE_LOSS_PARTIALS, 1))) return rc;
  }
  if ((phases & MKE_MAP_BWD) && n > 0) {
    for (int k = 0; k < V; ++k) {
      hipLaunchKernelGGL(k_map_tail2, dim3((unsigned)eb), dim3(MKE_BLOCK), 0, st, Pm(k), Gm(k), ssq(k), dot(k), total);
      if ((rc = check_launch("k_map_tail2"))) return rc;
      if ((rc = launch_gemm_f32(Vr(k), 1, d, Gm(k), d, 1, a->gM + (int64_t)k * d * d, d, d, d, (int)n, 16, 1, st, nullptr, 0))) return rc;  // gM += V^T dP
    }
  }
  if (phases & MKE_MAP_UPD) {
    hipLaunchKernelGGL(k_map_ortho, dim3(V), dim3(MKE_BLOCK), 2 * d * (d + 1) * sizeof(float), st, a->M, a->gM, d, a->orthogonal_weight, a->norm_w,
                       loss4 + (int64_t)MKE_MAPPING_MAX_VIEWS * MKE_LOSS_PARTIALS);
    if ((rc = check_launch("k_map_ortho"))) return rc;
    if (a->ent_grad && n > 0) {
      hipLaunchKernelGGL(k_scatter_dense_rows, dim3((unsigned)eb), dim3(MKE_BLOCK), 0, st, a->idx, GF, n, d, a->stride, a->ent_grad, a->ent_touched, a->tag);
      if ((rc = check_launch("k_scatter_dense_rows"))) return rc;
    }
    if (a->update) {
      if (a->optimizer != MKE_OPT_ADAGRAD && a->optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", a->optimizer); return MKE_E_UNSUPPORTED; }
      if (a->optimizer == MKE_OPT_ADAGRAD && (!a->accM || (a->ent_grad && !a->ent_acc))) { set_error("mke_mapping_step: Adagrad needs accumulators"); return MKE_E_NULL; }
      mke_update_table tab{a->ent_table, a->ent_acc, a->ent_grad, a->ent_touched, a->n_ent, a->ent_normalize, 1, nullptr};
      DenseJob dj{a->M, a->accM, a->gM, (int64_t)V * d * d, a->optimizer, a->lr, nullptr, 0, 0, 0};
      UpdateTouchedHint hint(a->n);
      if ((rc = launch_rows_update_multi(&tab, a->ent_grad ? 1 : 0, a->tag, a->stride, d, a->optimizer, a->lr, st, nullptr, &dj))) return rc;
    }
  }
  return MKE_OK;
}

}  // namespace mke

extern "C" int64_t mke_mapping_scratch_floats(int64_t n, int dim) {
  if (n < 0 || dim <= 0) return 0;
  return n * (int64_t)dim * (2 + 3 * MKE_MAPPING_MAX_VIEWS);  // F | GF | per view: V, P, G
}

extern "C" int mke_mapping_step(const mke_mapping_step_args* args, double* loss_partials, void* stream) {
  return mke::mapping_step_impl(args, loss_partials, stream, MKE_MAP_ALL);
}

extern "C" int mke_mapping_step_phases(const mke_mapping_step_args* args, double* loss_partials, int phases, void* stream) {
  return mke::mapping_step_impl(args, loss_partials, stream, phases);
}

extern "C" int mke_mapping_steps(const mke_mapping_step_args* args, const int64_t* step_off, int n_steps, double* loss_ring, int ring,
                                 void* stream) {
  using namespace mke;
  if (!args || !step_off || !loss_ring) { set_error("mke_mapping_steps: NULL pointer"); return MKE_E_NULL; }
  if (n_steps < 0 || ring < 1) { set_error("mke_mapping_steps: bad n_steps/ring"); return MKE_E_SHAPE; }
  if ((int64_t)args->tag + n_steps >= 0x7FFFFFFFLL) { set_error("tag overflow"); return MKE_E_RANGE; }
  for (int s = 0; s < n_steps; ++s) {
    const int64_t lo = step_off[s], hi = step_off[s + 1];
    if (lo < 0 || hi < lo) { set_error("mke_mapping_steps: step_off must be non-decreasing"); return MKE_E_SHAPE; }
    mke_mapping_step_args a = *args;
    a.idx = args->idx ? args->idx + lo : nullptr;
    a.n = hi - lo;
    a.tag = args->tag + s;
    const int rc = mapping_step_impl(&a, loss_ring + (int64_t)(s % ring) * (MKE_MAPPING_MAX_VIEWS + 1) * MKE_LOSS_PARTIALS, stream, MKE_MAP_ALL);
    if (rc) return rc;
  }
  return MKE_OK;
}
