// mke_rows.hip — the remaining row kernels (gfx950):
//   * loss ops over already-gathered dense rows (the losses.py surface itself, code/losses.py:4-69),
//   * the fused alignment term over table rows (code/MultiKE_model.py:229-236),
//   * normalised row gather (code/MultiKE_model.py:263-277, the `.eval()` read path).
// Same execution shape as the fused step: one 16-lane quarter-wave per row, lane j = columns j+16k.
#include "mke_common.h"

namespace mke {

template <int FPL>
__device__ __forceinline__ void load_dense(const float* __restrict__ base, int64_t row, int ld, int dim, int j,
                                           float (&v)[FPL]) {
  const float* p = base + row * (int64_t)ld + j;
#pragma unroll
  for (int k = 0; k < FPL; ++k) v[k] = (k * 16 + j < dim) ? p[k * 16] : 0.f;
}
template <int FPL>
__device__ __forceinline__ void store_dense(float* __restrict__ base, int64_t row, int ld, int dim, int j,
                                            const float (&v)[FPL], float sgn) {
  float* p = base + row * (int64_t)ld + j;
#pragma unroll
  for (int k = 0; k < FPL; ++k)
    if (k * 16 + j < dim) p[k * 16] = sgn * v[k];
}

// ---- gathered logistic loss: sum w * log(1+exp(sign*||h+r-t||^2)) --------------------------------
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_gathered_logistic(const float* __restrict__ hs, const float* __restrict__ rs,
                                                                 const float* __restrict__ ts, const float* __restrict__ ws,
                                                                 int64_t n, int dim, int ld, float sign,
                                                                 float* __restrict__ gh, float* __restrict__ gr,
                                                                 float* __restrict__ gt, double* __restrict__ lossp) {
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  float loss = 0.f;
  for (int64_t i = sub0; i < n; i += nsub) {
    float H[FPL], R[FPL], T[FPL];
    load_dense<FPL>(hs, i, ld, dim, j, H);
    load_dense<FPL>(rs, i, ld, dim, j, R);
    load_dense<FPL>(ts, i, ld, dim, j, T);
    float x = 0.f;
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      H[k] = (H[k] + R[k]) - T[k];
      x = fmaf(H[k], H[k], x);
    }
    x = sub16_sum(x);
    const float w = ws ? ws[i] : 1.0f;
    const float z = sign * x;
    loss += w * softplus_f(z);
    if (gh) {
      const float c = 2.0f * sign * w * sigmoid_f(z);
#pragma unroll
      for (int k = 0; k < FPL; ++k) H[k] *= c;
      store_dense<FPL>(gh, i, ld, dim, j, H, 1.0f);
      store_dense<FPL>(gr, i, ld, dim, j, H, 1.0f);
      store_dense<FPL>(gt, i, ld, dim, j, H, -1.0f);
    }
  }
  const double tot = block_sum_double(j == 0 ? loss : 0.f);
  if (threadIdx.x == 0) lossp[blockIdx.x] = tot;
}

// ---- gathered alignment loss: sum ||a-b||^2 ------------------------------------------------------
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_gathered_alignment(const float* __restrict__ a, const float* __restrict__ b,
                                                                  int64_t n, int dim, int ld, float* __restrict__ ga,
                                                                  float* __restrict__ gb, double* __restrict__ lossp) {
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  float loss = 0.f;
  for (int64_t i = sub0; i < n; i += nsub) {
    float A[FPL], B[FPL];
    load_dense<FPL>(a, i, ld, dim, j, A);
    load_dense<FPL>(b, i, ld, dim, j, B);
    float x = 0.f;
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      A[k] -= B[k];
      x = fmaf(A[k], A[k], x);
    }
    loss += sub16_sum(x);
    if (ga) {
#pragma unroll
      for (int k = 0; k < FPL; ++k) A[k] *= 2.0f;
      store_dense<FPL>(ga, i, ld, dim, j, A, 1.0f);
      store_dense<FPL>(gb, i, ld, dim, j, A, -1.0f);
    }
  }
  const double tot = block_sum_double(j == 0 ? loss : 0.f);
  if (threadIdx.x == 0) lossp[blockIdx.x] = tot;
}

// ---- fused alignment term over table rows --------------------------------------------------------
struct AlignParams {
  const float* __restrict__ ta;
  const float* __restrict__ tb;
  int a_norm, b_norm, stride, dim;
  const int32_t* __restrict__ ia;
  const int32_t* __restrict__ ib;
  int64_t n;
  float weight;
  float* __restrict__ ga;
  int32_t* __restrict__ toa;
  float* __restrict__ gb;
  int32_t* __restrict__ tob;
  int32_t tag;
  double* __restrict__ lossp;
};

template <int FPL>
__device__ __forceinline__ void align_block(const AlignParams& p) {
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  float loss = 0.f;
  for (int64_t i = sub0; i < p.n; i += nsub) {
    const int ra = p.ia[i], rb = p.ib[i];
    float A[FPL], B[FPL];
    load_row<FPL>(p.ta, ra, p.stride, j, A);
    load_row<FPL>(p.tb, rb, p.stride, j, B);
    l2_normalize_row<FPL>(A, p.a_norm);
    l2_normalize_row<FPL>(B, p.b_norm);
    float x = 0.f;
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      A[k] -= B[k];
      x = fmaf(A[k], A[k], x);
    }
    loss += sub16_sum(x);
    const float c = 2.0f * p.weight;
#pragma unroll
    for (int k = 0; k < FPL; ++k) A[k] *= c;
    if (p.ga) {
      atomic_add_row<FPL>(p.ga, ra, p.stride, p.dim, j, A, 1.0f);
      if (j == 0) p.toa[ra] = p.tag;
    }
    if (p.gb) {
      atomic_add_row<FPL>(p.gb, rb, p.stride, p.dim, j, A, -1.0f);
      if (j == 0) p.tob[rb] = p.tag;
    }
  }
  const double tot = block_sum_double(j == 0 ? loss : 0.f);
  if (threadIdx.x == 0) p.lossp[blockIdx.x] = tot * (double)p.weight;
}

template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_align(const AlignParams p) { align_block<FPL>(p); }

// all terms of one common-space step in one launch (blockIdx.y = term): one kernel floor instead of one per term
struct AlignBatch {
  AlignParams t[MKE_ALIGN_MAX_TERMS];
};
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_align_batch(const AlignBatch b) { align_block<FPL>(b.t[blockIdx.y]); }

// ---- normalised gather into a dense matrix -------------------------------------------------------
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_gather_rows(const float* __restrict__ table, int normalize, int stride,
                                                           int dim, const int32_t* __restrict__ idx, int64_t n,
                                                           float* __restrict__ out) {
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  for (int64_t i = sub0; i < n; i += nsub) {
    const int64_t row = idx ? (int64_t)idx[i] : i;
    float V[FPL];
    load_row<FPL>(table, row, stride, j, V);
    l2_normalize_row<FPL>(V, normalize);
    store_dense<FPL>(out, i, dim, dim, j, V, 1.0f);
  }
}

// Placement probe (read-only): row idx[i] of up to three arrays of the same shape at the same time — the access pattern of the
// relation step (table, accumulator, gradient scratch of one row together).  One float per visited row goes out so that the loads
// are not dead.  The tables' host side times it on candidate allocations of a large table's companion arrays and keeps the
// fastest combination (multike_amd/tables.py place_companions; EXPERIMENTS R5.10).
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_probe_rows(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ c, int stride,
                                                          const int32_t* __restrict__ idx, int64_t n, float* __restrict__ out) {
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  for (int64_t i = sub0; i < n; i += nsub) {
    const int64_t row = idx[i];
    float A[FPL], B[FPL], Cc[FPL];
    load_row<FPL>(a, row, stride, j, A);
    if (b) load_row<FPL>(b, row, stride, j, B);
    if (c) load_row<FPL>(c, row, stride, j, Cc);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < FPL; ++k) v += A[k] + (b ? B[k] : 0.f) + (c ? Cc[k] : 0.f);
    v = sub16_sum(v);
    if (j == 0) out[i] = v;
  }
}

// dense kernels bounds-check every column, so any supported FPL >= ceil(dim/16) is correct
static inline int dense_fpl(int dim) {
  static const int ok[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 16, 20};
  const int need = (dim + 15) / 16;
  for (int v : ok)
    if (v >= need) return v;
  return need;
}

static inline unsigned grid_for_rows(int64_t n) {
  int64_t b = (n + MKE_SUBS_PER_BLOCK - 1) / MKE_SUBS_PER_BLOCK;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (unsigned)b;
}

}  // namespace mke

extern "C" int mke_gathered_logistic_fwd_bwd(const float* hs, const float* rs, const float* ts, const float* ws,
                                             int64_t n, int dim, int ld, int sign, float* gh, float* gr, float* gt,
                                             double* loss_partials, void* stream) {
  using namespace mke;
  if (!loss_partials) { set_error("NULL loss_partials"); return MKE_E_NULL; }
  if (n < 0 || dim <= 0 || ld < dim || dim > MKE_MAX_STRIDE) { set_error("bad n/dim/ld: %lld %d %d", (long long)n, dim, ld); return MKE_E_SHAPE; }
  if (n > 0 && (!hs || !rs || !ts)) { set_error("NULL gathered rows"); return MKE_E_NULL; }
  if (sign != 1 && sign != -1) { set_error("sign must be +1 or -1"); return MKE_E_SHAPE; }
  const bool any = gh || gr || gt, all = gh && gr && gt;
  if (any && !all) { set_error("gh/gr/gt must be all given or all NULL"); return MKE_E_NULL; }
  const int fpl = dense_fpl(dim);
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_gathered_logistic<FPL>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, (hipStream_t)stream,
                       hs, rs, ts, ws, n, dim, ld, (float)sign, gh, gr, gt, loss_partials);
  });
  return check_launch("k_gathered_logistic");
}

extern "C" int mke_gathered_alignment_fwd_bwd(const float* a, const float* b, int64_t n, int dim, int ld, float* ga,
                                              float* gb, double* loss_partials, void* stream) {
  using namespace mke;
  if (!loss_partials) { set_error("NULL loss_partials"); return MKE_E_NULL; }
  if (n < 0 || dim <= 0 || ld < dim || dim > MKE_MAX_STRIDE) { set_error("bad n/dim/ld"); return MKE_E_SHAPE; }
  if (n > 0 && (!a || !b)) { set_error("NULL gathered rows"); return MKE_E_NULL; }
  if ((ga == nullptr) != (gb == nullptr)) { set_error("ga/gb must be both given or both NULL"); return MKE_E_NULL; }
  const int fpl = dense_fpl(dim);
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_gathered_alignment<FPL>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, (hipStream_t)stream, a,
                       b, n, dim, ld, ga, gb, loss_partials);
  });
  return check_launch("k_gathered_alignment");
}

extern "C" int mke_align_fwd_bwd(const float* table_a, int a_normalize, const float* table_b, int b_normalize,
                                 int stride, int dim, const int32_t* ia, const int32_t* ib, int64_t n, float weight,
                                 float* grad_a, int32_t* touched_a, float* grad_b, int32_t* touched_b, int32_t tag,
                                 double* loss_partials, void* stream) {
  using namespace mke;
  if (!table_a || !table_b || !loss_partials) { set_error("mke_align_fwd_bwd: NULL table/loss"); return MKE_E_NULL; }
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n > 0 && (!ia || !ib)) { set_error("NULL index stream"); return MKE_E_NULL; }
  if (stride <= 0 || stride % 16 != 0 || dim <= 0 || dim > stride || stride > MKE_MAX_STRIDE) { set_error("bad stride/dim"); return MKE_E_SHAPE; }
  if ((grad_a && !touched_a) || (grad_b && !touched_b)) { set_error("NULL touched array"); return MKE_E_NULL; }
  AlignParams p;
  p.ta = table_a; p.tb = table_b; p.a_norm = a_normalize; p.b_norm = b_normalize; p.stride = stride; p.dim = dim;
  p.ia = ia; p.ib = ib; p.n = n; p.weight = weight;
  p.ga = grad_a; p.toa = touched_a; p.gb = grad_b; p.tob = touched_b; p.tag = tag; p.lossp = loss_partials;
  const int fpl = stride / 16;
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_align<FPL>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  });
  return check_launch("k_align");
}

extern "C" int mke_align_steps(const mke_align_plan* pl, void* stream) {
  using namespace mke;
  if (!pl) { set_error("mke_align_steps: NULL plan"); return MKE_E_NULL; }
  if (pl->n_tables < 1 || pl->n_tables > MKE_ALIGN_MAX_TABLES || pl->n_terms < 1 || pl->n_terms > MKE_ALIGN_MAX_TERMS) { set_error("mke_align_steps: bad table / term count"); return MKE_E_SHAPE; }
  if (pl->n_steps < 0 || !pl->step_off || !pl->loss_partials) { set_error("mke_align_steps: NULL pointer or negative n_steps"); return MKE_E_NULL; }
  if ((int64_t)pl->tag_base + pl->n_steps >= 0x7FFFFFFFLL) { set_error("tag overflow"); return MKE_E_RANGE; }
  if (pl->stride <= 0 || pl->stride % 16 != 0 || pl->dim <= 0 || pl->dim > pl->stride || pl->stride > MKE_MAX_STRIDE) { set_error("mke_align_steps: bad stride/dim"); return MKE_E_SHAPE; }
  mke_update_table ut[MKE_ALIGN_MAX_TABLES];
  int nu = 0;
  for (int k = 0; k < pl->n_tables; ++k) {
    const mke_align_table& t = pl->tables[k];
    if (!t.table) { set_error("mke_align_steps: table %d is NULL", k); return MKE_E_NULL; }
    if (t.grad) {
      if (!t.touched) { set_error("mke_align_steps: table %d has no touched array", k); return MKE_E_NULL; }
      ut[nu++] = mke_update_table{t.table, t.acc, t.grad, t.touched, t.n_rows, t.normalize, 1, nullptr};
    }
  }
  for (int k = 0; k < pl->n_terms; ++k)
    if (pl->terms[k].a < 0 || pl->terms[k].a >= pl->n_tables || pl->terms[k].b < 0 || pl->terms[k].b >= pl->n_tables) { set_error("mke_align_steps: term %d names a table outside [0,%d)", k, pl->n_tables); return MKE_E_RANGE; }
  for (int s = 0; s < pl->n_steps; ++s) {
    const int64_t lo = pl->step_off[s], hi = pl->step_off[s + 1];
    if (lo < 0 || hi < lo) { set_error("mke_align_steps: step_off must be non-decreasing"); return MKE_E_SHAPE; }
    const int32_t tag = pl->tag_base + s;
    if (hi > lo && (!pl->ia || !pl->ib)) { set_error("mke_align_steps: NULL index stream"); return MKE_E_NULL; }
    AlignBatch ab;
    for (int k = 0; k < pl->n_terms; ++k) {
      const mke_align_table& a = pl->tables[pl->terms[k].a];
      const mke_align_table& b = pl->tables[pl->terms[k].b];
      AlignParams& q = ab.t[k];
      q.ta = a.table; q.tb = b.table; q.a_norm = a.normalize; q.b_norm = b.normalize; q.stride = pl->stride; q.dim = pl->dim;
      q.ia = pl->ia ? pl->ia + lo : nullptr; q.ib = pl->ib ? pl->ib + lo : nullptr; q.n = hi - lo; q.weight = pl->terms[k].weight;
      q.ga = a.grad; q.toa = a.touched; q.gb = b.grad; q.tob = b.touched; q.tag = tag;
      q.lossp = pl->loss_partials + ((int64_t)s * pl->n_terms + k) * MKE_LOSS_PARTIALS;
    }
    {
      const int fpl = pl->stride / 16;
      MKE_DISPATCH_FPL(fpl, {
        hipLaunchKernelGGL((k_align_batch<FPL>), dim3(MKE_LOSS_PARTIALS, pl->n_terms), dim3(MKE_BLOCK), 0, (hipStream_t)stream, ab);
      });
      const int rc = check_launch("k_align_batch");
      if (rc) return rc;
    }
    if (nu) {
      UpdateTouchedHint hint(pl->step_off[s + 1] - pl->step_off[s]);   // a term touches one row per batch entry and table
      const int rc = mke_rows_update_multi(ut, nu, tag, pl->stride, pl->dim, pl->optimizer, pl->lr, stream);
      if (rc) return rc;
    }
  }
  return MKE_OK;
}

extern "C" int mke_probe_rows(const float* a, const float* b, const float* c, int stride, const int32_t* idx, int64_t n, float* out,
                              void* stream) {
  using namespace mke;
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!a || !idx || !out) { set_error("mke_probe_rows: NULL pointer"); return MKE_E_NULL; }
  if (stride <= 0 || stride % 16 != 0 || stride > MKE_MAX_STRIDE) { set_error("bad stride"); return MKE_E_SHAPE; }
  const int fpl = stride / 16;
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_probe_rows<FPL>), dim3(grid_for_rows(n)), dim3(MKE_BLOCK), 0, (hipStream_t)stream, a, b, c, stride, idx, n, out);
  });
  return check_launch("k_probe_rows");
}

extern "C" int mke_gather_rows(const float* table, int normalize, int stride, int dim, const int32_t* idx, int64_t n,
                               float* out, void* stream) {
  using namespace mke;
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!table || !out) { set_error("mke_gather_rows: NULL pointer"); return MKE_E_NULL; }
  if (stride <= 0 || stride % 16 != 0 || dim <= 0 || dim > stride || stride > MKE_MAX_STRIDE) { set_error("bad stride/dim"); return MKE_E_SHAPE; }
  const int fpl = stride / 16;
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_gather_rows<FPL>), dim3(grid_for_rows(n)), dim3(MKE_BLOCK), 0, (hipStream_t)stream, table,
                       normalize, stride, dim, idx, n, out);
  });
  return check_launch("k_gather_rows");
}
