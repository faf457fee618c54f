/* CPU ORACLE — test infrastructure, not product code.  Included twice by mke_oracle.c with
 * REAL = float / double and SUF = f32 / f64.  See mke_oracle.c for the citations. */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* inverse norm of one row under tf.nn.l2_normalize semantics (code/base/initializers.py:26) */
static inline REAL FN(inv_norm)(const REAL* w, int dim, int on) {
  if (!on) return (REAL)1;
  REAL s = 0;
  for (int k = 0; k < dim; ++k) s += w[k] * w[k];
  if (s < (REAL)MKO_L2_EPS) s = (REAL)MKO_L2_EPS;
  return (REAL)1 / (REAL)sqrt((double)s);
}

/* one triple: loss term + gradient (normalised space) accumulated into the dense scratch rows */
static inline double FN(one_triple)(const REAL* H, const REAL* R, const REAL* T, REAL ih, REAL ir, REAL it, int dim,
                                    REAL sign, REAL w, REAL scale, REAL* gH, REAL* gR, REAL* gT, REAL* dbuf) {
  REAL x = 0;
  for (int k = 0; k < dim; ++k) {
    const REAL d = (H[k] * ih + R[k] * ir) - T[k] * it;
    dbuf[k] = d;
    x += d * d;
  }
  const double z = (double)(sign * x);
  const double sp = (z > 0 ? z : 0) + log1p(exp(-fabs(z))); /* log(1+exp(z)) — code/losses.py:9-10 */
  if (gH) {
    const REAL c = (REAL)2 * sign * w * scale * (REAL)(1.0 / (1.0 + exp(-z)));
    for (int k = 0; k < dim; ++k) {
      const REAL g = c * dbuf[k];
      gH[k] += g;
      gR[k] += g;
      gT[k] -= g;
    }
  }
  return (double)w * sp;
}

static inline void FN(row_update)(REAL* w, REAL* a, REAL* g, int dim, int normalize, REAL lr) {
  if (normalize) {
    REAL s = 0, dot = 0;
    for (int k = 0; k < dim; ++k) { s += w[k] * w[k]; dot += w[k] * g[k]; }
    const REAL sc = s < (REAL)MKO_L2_EPS ? (REAL)MKO_L2_EPS : s;
    const REAL inv = (REAL)1 / (REAL)sqrt((double)sc);
    const REAL coef = s > (REAL)MKO_L2_EPS ? dot * inv * inv : (REAL)0;
    for (int k = 0; k < dim; ++k) g[k] = (g[k] - w[k] * coef) * inv;
  }
  for (int k = 0; k < dim; ++k) { /* ApplyAdagrad — code/MultiKE_model.py:17 */
    a[k] += g[k] * g[k];
    w[k] -= lr * g[k] / (REAL)sqrt((double)a[k]);
    g[k] = 0;
  }
}

/* One relation-view train step (code/MultiKE_model.py:122-131,304-310; code/losses.py:4-12).
 * dense = 0: touched rows only (bit-identical to dense for untouched rows);
 * dense = 1: reference-faithful cost model — whole-table normalise, whole-table Jacobian + Adagrad.
 * gbuf_* are [n][dim] scratch, all-zero on entry; zero again on exit when do_update, else they hold the
 * gradient w.r.t. the normalised rows (for the scatter-kernel parity tests).  mark_* are [n] bytes, zero on entry/exit.
 * list_* are [n] int32 scratch.  norm_* are [n][dim] scratch (dense mode only, may be NULL otherwise).
 * Returns the loss. */
double FN(mko_relation_step)(REAL* ent, REAL* rel, REAL* acc_ent, REAL* acc_rel, int64_t n_ent, int64_t n_rel, int dim,
                             const int32_t* ph, const int32_t* pr, const int32_t* pt, const REAL* pw, int64_t n_pos,
                             const int32_t* nh, const int32_t* nr, const int32_t* nt, const REAL* nw, int64_t n_neg,
                             double scale, double lr, int ent_norm, int rel_norm, int dense, int do_update,
                             REAL* gbuf_ent, REAL* gbuf_rel, uint8_t* mark_ent, uint8_t* mark_rel, int32_t* list_ent,
                             int32_t* list_rel, REAL* norm_ent, REAL* norm_rel) {
  REAL dbuf[MKO_MAX_DIM];
  int64_t n_te = 0, n_tr = 0;
  double loss = 0.0;
  const REAL* E = ent;
  const REAL* Rt = rel;
  if (dense) { /* whole-table l2_normalize forward, as the TF graph does every step */
    for (int64_t i = 0; i < n_ent; ++i) {
      const REAL inv = FN(inv_norm)(ent + i * dim, dim, ent_norm);
      for (int k = 0; k < dim; ++k) norm_ent[i * dim + k] = ent[i * dim + k] * inv;
    }
    for (int64_t i = 0; i < n_rel; ++i) {
      const REAL inv = FN(inv_norm)(rel + i * dim, dim, rel_norm);
      for (int k = 0; k < dim; ++k) norm_rel[i * dim + k] = rel[i * dim + k] * inv;
    }
    E = norm_ent;
    Rt = norm_rel;
  }
  for (int pass = 0; pass < 2; ++pass) {
    const int32_t *hh = pass ? nh : ph, *rr = pass ? nr : pr, *tt = pass ? nt : pt;
    const REAL* ww = pass ? nw : pw;
    const int64_t n = pass ? n_neg : n_pos;
    const REAL sign = pass ? (REAL)-1 : (REAL)1;
    for (int64_t i = 0; i < n; ++i) {
      const int32_t h = hh[i], r = rr[i], t = tt[i];
      const REAL *H = E + (int64_t)h * dim, *R = Rt + (int64_t)r * dim, *T = E + (int64_t)t * dim;
      REAL ih = 1, ir = 1, it = 1;
      if (!dense) {
        ih = FN(inv_norm)(H, dim, ent_norm);
        ir = FN(inv_norm)(R, dim, rel_norm);
        it = FN(inv_norm)(T, dim, ent_norm);
      }
      loss += FN(one_triple)(H, R, T, ih, ir, it, dim, sign, ww ? ww[i] : (REAL)1, (REAL)scale,
                             gbuf_ent + (int64_t)h * dim, gbuf_rel + (int64_t)r * dim,
                             gbuf_ent + (int64_t)t * dim, dbuf);
      if (!mark_ent[h]) { mark_ent[h] = 1; list_ent[n_te++] = h; }
      if (!mark_ent[t]) { mark_ent[t] = 1; list_ent[n_te++] = t; }
      if (!mark_rel[r]) { mark_rel[r] = 1; list_rel[n_tr++] = r; }
    }
  }
  if (do_update) {
    if (dense) { /* TF applies the Jacobian and ApplyAdagrad to every row */
      for (int64_t i = 0; i < n_ent; ++i) FN(row_update)(ent + i * dim, acc_ent + i * dim, gbuf_ent + i * dim, dim, ent_norm, (REAL)lr);
      for (int64_t i = 0; i < n_rel; ++i) FN(row_update)(rel + i * dim, acc_rel + i * dim, gbuf_rel + i * dim, dim, rel_norm, (REAL)lr);
    } else {
      for (int64_t q = 0; q < n_te; ++q) { const int64_t i = list_ent[q]; FN(row_update)(ent + i * dim, acc_ent + i * dim, gbuf_ent + i * dim, dim, ent_norm, (REAL)lr); }
      for (int64_t q = 0; q < n_tr; ++q) { const int64_t i = list_rel[q]; FN(row_update)(rel + i * dim, acc_rel + i * dim, gbuf_rel + i * dim, dim, rel_norm, (REAL)lr); }
    }
  }
  for (int64_t q = 0; q < n_te; ++q) mark_ent[list_ent[q]] = 0;
  for (int64_t q = 0; q < n_tr; ++q) mark_rel[list_rel[q]] = 0;
  return loss * scale;
}

/* Multi-threaded variant for bench.py's `cpu_baseline` leg (OpenMP).  Same arithmetic as mko_relation_step, organised
 * without atomics: (1) parallel over triples: difference vector, loss, and the triple's gradient vector c*d into
 * gtrip[T][dim]; (2) parallel over ROW RANGES: each thread owns a contiguous range of entity (relation) ids, walks the
 * step's id streams in triple order and adds the vectors of the triples that reference its rows — so a row's gradient
 * is summed in triple order whatever the thread count (results do not depend on n_threads); (3) parallel over rows:
 * Jacobian + Adagrad.  One difference in bookkeeping from mko_relation_step: a touched row's inverse norm is computed
 * ONCE per step (inv_*), so the per-triple work of the touched-rows mode (dense = 0) and of the reference-faithful dense
 * mode (dense = 1: whole-table normalise, whole-table Jacobian + Adagrad, what the TF graph of
 * code/MultiKE_model.py:114-132 does every step) is identical and the dense mode is slower by exactly its whole-table
 * passes.  mark_* are [n] bytes (zero on entry / exit), inv_* are [n] scratch, gtrip is [n_pos + n_neg][dim] scratch. */
double FN(mko_relation_step_mt)(REAL* ent, REAL* rel, REAL* acc_ent, REAL* acc_rel, int64_t n_ent, int64_t n_rel, int dim,
                                const int32_t* ph, const int32_t* pr, const int32_t* pt, int64_t n_pos,
                                const int32_t* nh, const int32_t* nr, const int32_t* nt, int64_t n_neg, double lr,
                                int dense, REAL* gbuf_ent, REAL* gbuf_rel, uint8_t* mark_ent, uint8_t* mark_rel,
                                REAL* inv_ent, REAL* inv_rel, REAL* norm_ent, REAL* norm_rel, REAL* gtrip, int n_threads) {
  double loss = 0.0;
  const int64_t n_all = n_pos + n_neg;
  if (n_threads < 1) n_threads = 1;
  /* marks: benign races (every writer stores 1) */
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int64_t i = 0; i < n_all; ++i) {
    const int64_t k = i < n_pos ? i : i - n_pos;
    const int32_t h = i < n_pos ? ph[k] : nh[k], r = i < n_pos ? pr[k] : nr[k], t = i < n_pos ? pt[k] : nt[k];
    mark_ent[h] = 1; mark_ent[t] = 1; mark_rel[r] = 1;
  }
  if (dense) { /* whole-table l2_normalize forward */
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t i = 0; i < n_ent; ++i) {
      const REAL inv = FN(inv_norm)(ent + i * dim, dim, 1);
      inv_ent[i] = 1;
      for (int k = 0; k < dim; ++k) norm_ent[i * dim + k] = ent[i * dim + k] * inv;
    }
    for (int64_t i = 0; i < n_rel; ++i) {
      const REAL inv = FN(inv_norm)(rel + i * dim, dim, 1);
      inv_rel[i] = 1;
      for (int k = 0; k < dim; ++k) norm_rel[i * dim + k] = rel[i * dim + k] * inv;
    }
  } else {
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t i = 0; i < n_ent; ++i)
      if (mark_ent[i]) inv_ent[i] = FN(inv_norm)(ent + i * dim, dim, 1);
    for (int64_t i = 0; i < n_rel; ++i)
      if (mark_rel[i]) inv_rel[i] = FN(inv_norm)(rel + i * dim, dim, 1);
  }
  const REAL* E = dense ? norm_ent : ent;
  const REAL* Rt = dense ? norm_rel : rel;
#pragma omp parallel for schedule(static) reduction(+ : loss) num_threads(n_threads)
  for (int64_t i = 0; i < n_all; ++i) {
    const int64_t q = i < n_pos ? i : i - n_pos;
    const int32_t h = i < n_pos ? ph[q] : nh[q], r = i < n_pos ? pr[q] : nr[q], t = i < n_pos ? pt[q] : nt[q];
    const REAL sign = i < n_pos ? (REAL)1 : (REAL)-1;
    const REAL *H = E + (int64_t)h * dim, *R = Rt + (int64_t)r * dim, *T = E + (int64_t)t * dim;
    const REAL ih = inv_ent[h], ir = inv_rel[r], it = inv_ent[t];
    REAL* g = gtrip + i * dim;
    REAL x = 0;
    for (int k = 0; k < dim; ++k) {
      const REAL d = (H[k] * ih + R[k] * ir) - T[k] * it;
      g[k] = d;
      x += d * d;
    }
    const double z = (double)(sign * x);
    loss += (z > 0 ? z : 0) + log1p(exp(-fabs(z)));
    const REAL c = (REAL)2 * sign * (REAL)(1.0 / (1.0 + exp(-z)));
    for (int k = 0; k < dim; ++k) g[k] *= c;
  }
#pragma omp parallel num_threads(n_threads)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
    const int tid = 0, nth = 1;
#endif
    const int64_t elo = n_ent * tid / nth, ehi = n_ent * (tid + 1) / nth;
    const int64_t rlo = n_rel * tid / nth, rhi = n_rel * (tid + 1) / nth;
    for (int64_t i = 0; i < n_all; ++i) {
      const int64_t q = i < n_pos ? i : i - n_pos;
      const int32_t h = i < n_pos ? ph[q] : nh[q], r = i < n_pos ? pr[q] : nr[q], t = i < n_pos ? pt[q] : nt[q];
      const REAL* g = gtrip + i * dim;
      if (h >= elo && h < ehi) { REAL* o = gbuf_ent + (int64_t)h * dim; for (int k = 0; k < dim; ++k) o[k] += g[k]; }
      if (r >= rlo && r < rhi) { REAL* o = gbuf_rel + (int64_t)r * dim; for (int k = 0; k < dim; ++k) o[k] += g[k]; }
      if (t >= elo && t < ehi) { REAL* o = gbuf_ent + (int64_t)t * dim; for (int k = 0; k < dim; ++k) o[k] -= g[k]; }
    }
  }
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int64_t i = 0; i < n_ent; ++i) {
    if (dense || mark_ent[i]) FN(row_update)(ent + i * dim, acc_ent + i * dim, gbuf_ent + i * dim, dim, 1, (REAL)lr);
    mark_ent[i] = 0;
  }
  for (int64_t i = 0; i < n_rel; ++i) {
    if (dense || mark_rel[i]) FN(row_update)(rel + i * dim, acc_rel + i * dim, gbuf_rel + i * dim, dim, 1, (REAL)lr);
    mark_rel[i] = 0;
  }
  return loss;
}

#undef FN
#undef CAT
#undef CAT_
