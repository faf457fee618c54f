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

#undef FN
#undef CAT
#undef CAT_
