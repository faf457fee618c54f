// mke_score.hip — fused relation-view triple step for gfx950:
//   gather (h,r,t) rows -> normalise-on-read -> d = h + r - t -> ||d||^2 -> logistic loss
//   -> coefficient -> gradient rows (normalised space) scattered with fp32 atomics.
//
// What it computes is what the reference's relation-view graph computes
// (code/MultiKE_model.py:122-130 + code/losses.py:4-12 + code/base/initializers.py:26); how it is
// organised is new.  A 16-lane quarter-wave owns one "group" = a positive and (a slice of) its
// neg_per_pos negatives.  The positive's three rows are loaded and normalised once and stay in
// registers; a negative that differs from its positive in exactly one entity (what
// code/base/batch.py:86-116 produces) costs ONE more row gather and ONE row scatter; the gradients of
// the shared rows are pre-reduced in registers and flushed with one scatter per group.  Any other
// negative falls back to an independent (3 gathers, 3 scatters) triple — same arithmetic.
#include "mke_common.h"

namespace mke {

#ifndef MKE_SCORE_U
#define MKE_SCORE_U 2
#endif

struct ScoreParams {
  const float* ent;   // NOT __restrict__: ent_w below aliases it (in-place update of rows referenced once; such a row is never
                      // read again in the launch, but the compiler must not be told the two pointers cannot alias)
  const float* __restrict__ rel;
  int ent_norm, rel_norm;
  int stride, dim;
  const int32_t* __restrict__ ph;
  const int32_t* __restrict__ pr;
  const int32_t* __restrict__ pt;
  const float* __restrict__ pw;
  int64_t n_pos;
  const int32_t* __restrict__ nh;
  const int32_t* __restrict__ nr;
  const int32_t* __restrict__ nt;
  const float* __restrict__ nw;
  int64_t n_neg;
  int npp;     // negatives per positive (grouped) or 0
  int splits;  // quarter-waves sharing one group's negatives
  float scale;
  float* __restrict__ gent;
  float* __restrict__ grel;
  int grel_copies;
  int64_t grel_copy_elems;  // n_rel * stride
  int32_t* __restrict__ tent;
  int32_t* __restrict__ trel;
  int32_t tag;
  double* __restrict__ lossp;
  // exclusive-row fast path (nullable refcount = disabled)
  int32_t* __restrict__ refcount;
  float* ent_w;                 // writable alias of ent
  float* __restrict__ ent_acc;  // nullable (SGD)
  int optimizer;
  float lr;
  // deterministic mode (nullable): every gradient-row contribution is STORED into its own slot instead of added
  // atomically; mke_stage_reduce sums a row's slots in slot order afterwards
  float* __restrict__ stage_rows;   // [slots][stride]
  int64_t* __restrict__ stage_keys; // [slots]: (is_relation << 40) | row; untouched slots keep the caller's fill value
  // rider: the first count_blocks blocks of the grid count the NEXT step's entity references (they finish under the scoring
  // blocks; as riders of the update launch they were a 4 us tail of it)
  mke_count_job cj;
  int count_blocks;
  // hub rows (mke_hot_rows): the flush of a group's head / tail gradient goes to a private copy row when the row is listed
  const int32_t* __restrict__ hot_slot;
  int32_t n_hot, hot_copies;
  int32_t hot_row0;
};

#define MKE_STAGE_REL (1ll << 40)
// slot of contribution c (0 head row, 1 relation row, 2 tail row) of triple n of group g (n == npp: the group's flush)
__device__ __forceinline__ int64_t stage_slot(int64_t g, int npp, int n, int c) { return ((g * (npp + 1) + n) * 3 + c); }

// Address of column j of a row.  O32 (tables below 4 GB): byte offset in 32 bits with the compile-time stride, so that the
// access is `global_* v_off, s[base]` — the generic form costs a 64-bit multiply-add per row and pointer (v_mad_u64_u32 /
// v_lshl_add_u64: 34 such instructions per loop iteration of the training kernel, each worth four ordinary VALU slots)
template <int FPL, bool O32>
__device__ __forceinline__ float* row_at(const float* base, int row, int stride, int j) {
  if (O32) return (float*)((char*)const_cast<float*>(base) + (((uint32_t)row * (uint32_t)(FPL * 16) + (uint32_t)j) << 2));
  return const_cast<float*>(base) + (int64_t)row * stride + j;
}
template <int FPL>
__device__ __forceinline__ void load_at(const float* __restrict__ q, float (&v)[FPL]) {
#pragma unroll
  for (int k = 0; k < FPL; ++k) v[k] = q[k * 16];
}
template <int FPL>
__device__ __forceinline__ void atomic_add_at(float* __restrict__ q, int dim, int j, const float (&v)[FPL], float sgn) {
#pragma unroll
  for (int k = 0; k < FPL; ++k) {
    if (k * 16 + 16 <= dim || k * 16 + j < dim) atomic_add_f32(q + k * 16, sgn * v[k]);
  }
}

// one gradient-row contribution: atomic add + touched flag, or (deterministic mode) a plain store into its slot
template <int FPL, bool DET = false, bool O32 = false>
__device__ __forceinline__ void emit_row(const ScoreParams& p, bool is_rel, float* __restrict__ table_grad, int32_t* __restrict__ touched,
                                         int row, int64_t slot, int j, const float (&v)[FPL], float sgn, int grad_row = -1) {
  if (DET) {
    float* o = p.stage_rows + slot * p.stride + j;
#pragma unroll
    for (int k = 0; k < FPL; ++k) o[k * 16] = sgn * v[k];
    if (j == 0) p.stage_keys[slot] = (is_rel ? MKE_STAGE_REL : 0ll) | (int64_t)row;
  } else {
    // grad_row >= 0: a hub row's private copy takes the sum; the flag stays with the row itself
    atomic_add_at<FPL>(row_at<FPL, O32>(table_grad, grad_row >= 0 ? grad_row : row, p.stride, j), p.dim, j, v, sgn);
    if (j == 0) touched[row] = p.tag;
  }
}

// One triple scored on its own: 3 gathers, loss, 3 scatters.  sign=+1 positive, -1 negative.
// hub rows (mke_hot_rows): the scratch row that takes contribution `k` to entity row e (-1: the row itself)
template <bool DET>
__device__ __forceinline__ int hot_copy_row(const ScoreParams& p, int e, int64_t k) {
  if (DET || !p.hot_slot) return -1;
  const int s = p.hot_slot[e];
  return s >= 0 ? p.hot_row0 + (int)((uint32_t)k % (uint32_t)p.hot_copies) * p.n_hot + s : -1;   // k < 2^32 (the launcher checks the item count)
}

template <int FPL, bool DET = false>
__device__ __forceinline__ float independent_triple(const ScoreParams& p, float* __restrict__ grel, int j, int h, int r,
                                                    int t, float w, float sign, int64_t slot0) {
  float H[FPL], R[FPL], T[FPL];
  load_row<FPL>(p.ent, h, p.stride, j, H);
  load_row<FPL>(p.rel, r, p.stride, j, R);
  load_row<FPL>(p.ent, t, p.stride, j, T);
  l2_normalize_row<FPL>(H, p.ent_norm);
  l2_normalize_row<FPL>(R, p.rel_norm);
  l2_normalize_row<FPL>(T, p.ent_norm);
  float x = 0.f;
#pragma unroll
  for (int k = 0; k < FPL; ++k) {
    H[k] = (H[k] + R[k]) - T[k];  // d
    x = fmaf(H[k], H[k], x);
  }
  x = sub16_sum(x);
  const float z = sign * x;
  const float l = w * softplus_f(z);
  if (p.gent) {
    const float c = 2.0f * sign * w * p.scale * sigmoid_f(z);
#pragma unroll
    for (int k = 0; k < FPL; ++k) H[k] *= c;
    // positives-only steps (the cross-KG loops) score every triple here: a hub entity's row takes its private copy (slot0 / 3 = the
    // triple's index spreads the copies)
    emit_row<FPL, DET>(p, false, p.gent, p.tent, h, slot0, j, H, 1.0f, hot_copy_row<DET>(p, h, slot0 / 3));
    emit_row<FPL, DET>(p, true, grel, p.trel, r, slot0 + 1, j, H, 1.0f);
    emit_row<FPL, DET>(p, false, p.gent, p.tent, t, slot0 + 2, j, H, -1.0f, hot_copy_row<DET>(p, t, slot0 / 3));
  }
  return l;
}

// Grouped kernel: one 64-lane wavefront owns one (slice of a) group = a positive and its negatives.
// The four quarter-waves load the positive's rows redundantly (one TA broadcast), take the group's negatives
// round-robin, keep the shared rows' gradients in registers, reduce them across the quarters with two
// butterfly steps and flush them with ONE scatter of three rows (quarter 0 -> head row, 1 -> relation row,
// 2 -> tail row).  Lanes of one wave-instruction therefore never add to the same address (measured on
// MI355X: S quarter-waves of one wave adding to the same rows cost ~15-25 us per extra S at this shape).
// QPG (quarter-waves per group): 4 = the wavefront owns one group; 2 = each HALF of the wavefront owns a group of its own
// (short groups: with the reference's default 10 negatives a whole wavefront leaves a third of its quarter-wave slots
// idle in the last round and pays the positive's three row loads per 11 triples instead of per 22).
// DET: deterministic staging compiled in (a separate instantiation: its slot arithmetic costs the training kernel ten
// registers, 129 instead of 119 = three instead of four wavefronts per SIMD)
template <int FPL, int U, bool X, int QPG, bool DET = false, bool O32 = false, bool LID = false>  // X: exclusive-row fast path compiled in; O32: see row_at; LID: see LIDS
__global__ __launch_bounds__(MKE_BLOCK) void k_triple_score(const ScoreParams p) {
  constexpr bool LIDS = LID && X && !DET;   // ids and reference counts of a group fetched once, one negative per lane
  constexpr int LPG = 64 / (4 / QPG);            // lanes per group: the whole wavefront, or one half of it per group
  if ((int)blockIdx.x < p.count_blocks) {
    count_refs_range(p.cj, blockIdx.x, p.count_blocks);
    if (threadIdx.x == 0) p.lossp[blockIdx.x] = 0.0;
    return;
  }
  const int bid = blockIdx.x - p.count_blocks, nblk = gridDim.x - p.count_blocks;   // among the scoring blocks
  const int lane = threadIdx.x & 63;
  const int j = lane & 15;
  const int q4 = lane >> 4;
  constexpr int GPW = 4 / QPG;             // groups per wavefront
  const int sub = QPG == 4 ? 0 : (q4 >> 1);  // which group of the wavefront this lane works for
  const int q = QPG == 4 ? q4 : (q4 & 1);    // quarter-wave index inside the group
  const int64_t wave0 = ((int64_t)bid * MKE_BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)nblk * MKE_BLOCK) >> 6;
  const bool bwd = p.gent != nullptr;
  float loss = 0.f;  // identical on the 16 lanes of a quarter-wave; lane j==0 contributes

  const int npp = p.npp;
  if (npp > 0) {
    const int S = QPG == 2 ? 1 : p.splits;  // wavefronts sharing one group's negatives (two groups per wavefront: always 1 — the launcher's `half` needs splits == 1 — and a compile-time 1 folds every division by it below)
    const int64_t nitems = p.n_pos * S;
    const int64_t nwork = (nitems + GPW - 1) / GPW;
    for (int64_t wv = wave0; wv < nwork; wv += nwaves) {
      const int64_t wk_raw = wv * GPW + sub;
      const bool active = wk_raw < nitems;       // the second half of the last wavefront may have no group
      const int64_t wk = active ? wk_raw : 0;
      // group and slice of this work item: slices of one group are n_pos work items apart (different CUs).  Two groups per
      // wavefront (QPG == 2) means one slice per group: no division at all; otherwise UNSIGNED 32-bit ones (the launcher checks
      // n_pos * splits < 2^32) — the 64-bit `wk % n_pos`, `wk / n_pos`, `wk % copies` of a run-time divisor were ~400 of the
      // training kernel's ~2,400 vector instructions per wavefront, all of them ahead of its first load (EXPERIMENTS R5.28)
      int64_t g;
      int s;
      if constexpr (QPG == 2) { g = wk; s = 0; }
      else { s = (int)((uint32_t)wk / (uint32_t)p.n_pos); g = (int64_t)((uint32_t)wk - (uint32_t)s * (uint32_t)p.n_pos); }
      float* __restrict__ grel = bwd ? p.grel + (int64_t)((uint32_t)wk % (uint32_t)p.grel_copies) * p.grel_copy_elems : nullptr;
      const int ph = p.ph[g], pr = p.pr[g], pt = p.pt[g];
      // hub rows: looked up with the positive's rows, used at the flush (copy = group index modulo the number of copies)
      int hot_h = -1, hot_t = -1;
      if (!DET && p.hot_slot) {
        const int sh = p.hot_slot[ph], st = p.hot_slot[pt];
        const int cp = p.hot_row0 + (int)((uint32_t)g % (uint32_t)p.hot_copies) * p.n_hot;
        hot_h = sh >= 0 ? cp + sh : -1;
        hot_t = st >= 0 ? cp + st : -1;
      }
      // LIDS: the group's first block of negative ids is requested with the positive's ids, its reference counts with the
      // positive's rows (see below): two round trips fewer at the head of a wavefront's chain
      const int npp_ = p.npp;
      const int per_ = (npp_ + S - 1) / S;
      const int nlo_ = s * per_, nend_ = min(npp_, nlo_ + per_);
      const int lbase = lane & ~(LPG - 1);   // first lane of this lane's group
      int el = 0, fl = 0, rcl = 0, nblk = 0;
      float wl = 1.0f;
      bool slow_l = false;
      auto fetch_block = [&](int b0) {
        nblk = min(LPG, nend_ - b0);
        const bool has = active && lane - lbase < nblk;
        const int64_t idx = g * (int64_t)npp_ + (has ? b0 + (lane - lbase) : 0);   // an empty slice (b0 >= npp under a forced `score_splits`) reads the group's first id, never past the arrays
        const int nh = p.nh[idx], nr = p.nr[idx], nt = p.nt[idx];
        wl = p.nw ? p.nw[idx] : 1.0f;
        const bool dh = nh != ph, dt = nt != pt;
        const bool fastl = has && (nr == pr) && (dh != dt);
        el = dh ? nh : nt;
        fl = (fastl ? 1 : 0) | (dh ? 2 : 0);
        slow_l = has && !fastl;
        rcl = fastl ? p.refcount[el] : 0;
      };
      if constexpr (LIDS) fetch_block(nlo_);
      float H[FPL], R[FPL], T[FPL];
      load_at<FPL>(row_at<FPL, O32>(p.ent, ph, p.stride, j), H);
      load_at<FPL>(row_at<FPL, O32>(p.rel, pr, p.stride, j), R);
      load_at<FPL>(row_at<FPL, O32>(p.ent, pt, p.stride, j), T);
      l2_normalize_row<FPL>(H, p.ent_norm);
      l2_normalize_row<FPL>(R, p.rel_norm);
      l2_normalize_row<FPL>(T, p.ent_norm);
      float gH[FPL], gT[FPL];  // gT holds the NEGATED t-gradient (sum of c*d)
      float gP[FPL];           // the positive's own c*d (quarter 0 of slice 0, else 0): gR = gH + gT - gP at the flush
      float HR[FPL], RT[FPL];  // h + r and r - t of the positive: a negative's difference is one fma from them
#pragma unroll
      for (int k = 0; k < FPL; ++k) {
        gH[k] = gT[k] = gP[k] = 0.f;
        HR[k] = H[k] + R[k];
        RT[k] = R[k] - T[k];
      }

      if (s == 0 && q == 0 && active) {  // the positive itself
        const float w = p.pw ? p.pw[g] : 1.0f;
        float d[FPL];
        float x = 0.f;
#pragma unroll
        for (int k = 0; k < FPL; ++k) {
          d[k] = HR[k] - T[k];
          x = fmaf(d[k], d[k], x);
        }
        x = sub16_sum(x);
        loss += w * softplus_f(x);
        const float c = 2.0f * w * p.scale * sigmoid_f(x);
#pragma unroll
        for (int k = 0; k < FPL; ++k) {
          const float gd = c * d[k];
          gH[k] += gd; gT[k] += gd; gP[k] = gd;
        }
      }

      // this wave's slice of the group's negatives; quarter q takes n_lo+q, n_lo+q+4, ...
      const int per = (npp + S - 1) / S;
      const int n_lo = s * per;
      const int n_hi = active ? min(npp, n_lo + per) : n_lo;
      const int64_t nbase = g * (int64_t)npp;
      bool any_slow = false;
      // LIDS (the training instantiation): lane n of the wavefront fetches the ids AND the reference count of negative n of the
      // group once (two round trips for up to 64 negatives); the rounds below take them out of the lanes.  A round is then
      // one round trip instead of two (ids, then rows), and — the count being known before the gathers are issued — the
      // accumulator row is only fetched for rows that are finished in place (65 % of them): 10 % less traffic for a kernel
      // that runs at the copy rate beyond its fixed cost.
      const int n_end = min(npp, n_lo + per);
      for (int b0 = n_lo; b0 < (LIDS ? n_end : n_lo + 1); b0 += LPG) {
      if constexpr (LIDS) {
        if (b0 != n_lo) fetch_block(b0);
        any_slow |= __any(slow_l) != 0;
      }
      const int it_lo = LIDS ? 0 : n_lo + q, it_hi = LIDS ? nblk : n_hi, it_step = QPG * U;
      for (int n0 = it_lo; n0 < it_hi; n0 += it_step) {
        int e[U], cnt[U];
        bool fast[U], sideH[U];
        float w[U];
        float C[U][FPL];   // RAW corrupt rows (normalised on the fly: the raw values are needed for an in-place update)
        float A[X ? U : 1][X ? FPL : 1];
        // phase 1: ids
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if constexpr (LIDS) {
            const int slot = n0 + q + QPG * u;
            const bool live = slot < nblk;
            const int src = lbase + (live ? slot : 0);
            e[u] = __shfl(el, src, 64);
            const int f = __shfl(fl, src, 64);
            cnt[u] = __shfl(rcl, src, 64);
            w[u] = __shfl(wl, src, 64);
            fast[u] = live && (f & 1);
            sideH[u] = (f & 2) != 0;
          } else {
            const int n = n0 + QPG * u;
            const bool live = n < n_hi;
            const int64_t idx = nbase + (live ? n : n_lo);
            const int nh = p.nh[idx], nr = p.nr[idx], nt = p.nt[idx];
            w[u] = p.nw ? p.nw[idx] : 1.0f;
            const bool dh = nh != ph, dt = nt != pt;
            fast[u] = live && (nr == pr) && (dh != dt);
            sideH[u] = dh;
            e[u] = dh ? nh : nt;
            any_slow |= live && !fast[u];
          }
        }
        // phase 2: all corrupt-row gathers of the chunk in flight together
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if constexpr (!LIDS) cnt[u] = 0;
          if (fast[u]) {
            load_at<FPL>(row_at<FPL, O32>(p.ent, e[u], p.stride, j), C[u]);
            if constexpr (X) {
              if constexpr (!LIDS) cnt[u] = p.refcount[e[u]];
              if (p.ent_acc && (!LIDS || cnt[u] == 1)) load_at<FPL>(row_at<FPL, O32>(p.ent_acc, e[u], p.stride, j), A[u]);
            }
          }
        }
        // phase 3: score, loss, gradient
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (fast[u]) {
            float cinv = 1.0f;
            if (p.ent_norm) {
              float ss = 0.f;
#pragma unroll
              for (int k = 0; k < FPL; ++k) ss = fmaf(C[u][k], C[u][k], ss);
              cinv = rsqrtf(fmaxf(sub16_sum(ss), MKE_L2_EPS));
            }
            // corrupt head: d = c^ + (r - t);  corrupt tail: d = (h + r) - c^   ->   d = base + (+-cinv) * C
            float d[FPL];
            float y = 0.f;
            const float sc = sideH[u] ? cinv : -cinv;
#pragma unroll
            for (int k = 0; k < FPL; ++k) {
              d[k] = fmaf(sc, C[u][k], sideH[u] ? RT[k] : HR[k]);
              y = fmaf(d[k], d[k], y);
            }
            y = sub16_sum(y);
            // y >= 0: with t = exp(-y), softplus(-y) = log(1 + t) and sigmoid(-y) = t / (1 + t): one exp for both
            const float t_ = __expf(-y);
            const float s1 = 1.0f + t_;
            loss += w[u] * __logf(s1);
            if (bwd) {
              const float c = -2.0f * w[u] * p.scale * t_ * __builtin_amdgcn_rcpf(s1);
              const float cH = sideH[u] ? 0.f : c;   // a corrupted tail leaves the positive's head in the triple, and vice versa
              const float cT = sideH[u] ? c : 0.f;
#pragma unroll
              for (int k = 0; k < FPL; ++k) {
                gH[k] = fmaf(cH, d[k], gH[k]);
                gT[k] = fmaf(cT, d[k], gT[k]);
                d[k] *= c;
              }
              bool in_place = false;
              if constexpr (X) in_place = cnt[u] == 1;
              if (in_place) {
                // the only reference to row e in this step: Jacobian of the normalisation + optimizer, right here
                const float sg = sideH[u] ? 1.0f : -1.0f;
                float g[FPL];
                if (p.ent_norm) {
                  // g = (ghat - what (what . ghat)) / |w| with ghat = sg * d, what = C * cinv
                  float dot = 0.f;
#pragma unroll
                  for (int k = 0; k < FPL; ++k) dot = fmaf(C[u][k], d[k], dot);
                  dot = sub16_sum(dot) * (sg * cinv);
                  const float a1 = sg * cinv;
                  const float a2 = cinv < 0.99e6f ? -dot * cinv * cinv : 0.f;  // sum w^2 > eps  <=>  cinv < rsqrt(eps) = 1e6
#pragma unroll
                  for (int k = 0; k < FPL; ++k) g[k] = fmaf(a2, C[u][k], a1 * d[k]);
                } else {
#pragma unroll
                  for (int k = 0; k < FPL; ++k) g[k] = sg * d[k];
                }
                float* wp = row_at<FPL, O32>(p.ent_w, e[u], p.stride, j);
                if (p.optimizer == MKE_OPT_ADAGRAD) {
                  float* ap = row_at<FPL, O32>(p.ent_acc, e[u], p.stride, j);
#pragma unroll
                  for (int k = 0; k < FPL; ++k) {
                    const float a = fmaf(g[k], g[k], A[u][k]);
                    ap[k * 16] = a;
                    wp[k * 16] = C[u][k] - p.lr * g[k] * adagrad_scale(a);
                  }
                } else {
#pragma unroll
                  for (int k = 0; k < FPL; ++k) wp[k * 16] = C[u][k] - p.lr * g[k];
                }
                if (j == 0) p.refcount[e[u]] = 0;
              } else {
                emit_row<FPL, DET, O32>(p, false, p.gent, p.tent, e[u], stage_slot(g, npp, n0 + QPG * u, sideH[u] ? 0 : 2), j, d,
                              sideH[u] ? 1.0f : -1.0f);
              }
            }
          }
        }
      }
      }   // id blocks

      if (any_slow) {
        // rare: negatives that are not "the positive with exactly one entity replaced" (both sides or the
        // relation differ, or the negative equals the positive) are scored as independent triples
#pragma unroll 1
        for (int n = n_lo + q; n < n_hi; n += QPG) {
          const int64_t idx = nbase + n;
          const int nh = p.nh[idx], nr = p.nr[idx], nt = p.nt[idx];
          if (!((nr == pr) && ((nh != ph) != (nt != pt))))
            loss += independent_triple<FPL, DET>(p, grel, j, nh, nr, nt, p.nw ? p.nw[idx] : 1.0f, -1.0f, stage_slot(g, npp, n, 0));
        }
      }

      if (bwd) {
        // reduce the shared rows' gradients over the group's quarter-waves (all lanes converge here)
        // every negative's c*d went into exactly one of gH / gT and the positive's into both: the relation row's gradient
        // is their sum minus the positive's term once
        float gR[FPL];
#pragma unroll
        for (int k = 0; k < FPL; ++k) {
          gR[k] = (gH[k] + gT[k]) - gP[k];
          gH[k] += __shfl_xor(gH[k], 16, 64);
          gR[k] += __shfl_xor(gR[k], 16, 64);
          gT[k] += __shfl_xor(gT[k], 16, 64);
          if (QPG == 4) {
            gH[k] += __shfl_xor(gH[k], 32, 64);
            gR[k] += __shfl_xor(gR[k], 32, 64);
            gT[k] += __shfl_xor(gT[k], 32, 64);
          }
        }
        if (active && (s == 0 || n_lo < n_hi)) {
          if (QPG == 4) {
            // one scatter instruction stream covers the three rows: quarter 0 -> h, 1 -> r, 2 -> t
            float v[FPL];
#pragma unroll
            for (int k = 0; k < FPL; ++k) v[k] = q == 0 ? gH[k] : (q == 1 ? gR[k] : -gT[k]);
            float* __restrict__ base = q == 1 ? grel : p.gent;
            const int row = q == 0 ? ph : (q == 1 ? pr : pt);
            if (q < 3) emit_row<FPL, DET>(p, q == 1, base, q == 1 ? p.trel : p.tent, row, stage_slot(g, npp, npp, q), j, v, 1.0f,
                                          q == 0 ? hot_h : (q == 2 ? hot_t : -1));
          } else {
            // two quarter-waves per group: quarter 0 -> h, quarter 1 -> t, then quarter 0 -> r
            float v[FPL];
#pragma unroll
            for (int k = 0; k < FPL; ++k) v[k] = q == 0 ? gH[k] : -gT[k];
            const int row = q == 0 ? ph : pt;
            emit_row<FPL, DET>(p, false, p.gent, p.tent, row, stage_slot(g, npp, npp, q == 0 ? 0 : 2), j, v, 1.0f, q == 0 ? hot_h : hot_t);
            if (q == 0) emit_row<FPL, DET>(p, true, grel, p.trel, pr, stage_slot(g, npp, npp, 1), j, gR, 1.0f);
          }
        }
      }
    }
  } else {
    // no grouping: every positive / negative is an independent triple, one quarter-wave each
    const int64_t sub0 = ((int64_t)bid * MKE_BLOCK + threadIdx.x) >> 4;
    const int64_t nsub = ((int64_t)nblk * MKE_BLOCK) >> 4;
    for (int64_t i = sub0; i < p.n_pos + p.n_neg; i += nsub) {
      float* __restrict__ grel = bwd ? p.grel + (int64_t)((uint32_t)i % (uint32_t)p.grel_copies) * p.grel_copy_elems : nullptr;
      if (i < p.n_pos) {
        loss += independent_triple<FPL, DET>(p, grel, j, p.ph[i], p.pr[i], p.pt[i], p.pw ? p.pw[i] : 1.0f, 1.0f, i * 3);
      } else {
        const int64_t n = i - p.n_pos;
        loss += independent_triple<FPL, DET>(p, grel, j, p.nh[n], p.nr[n], p.nt[n], p.nw ? p.nw[n] : 1.0f, -1.0f, i * 3);
      }
    }
  }

  const double tot = block_sum_double(j == 0 ? loss : 0.f);
  if (threadIdx.x == 0) p.lossp[blockIdx.x] = tot * (double)p.scale;
}

__global__ __launch_bounds__(MKE_BLOCK) void k_count_refs(const int32_t* __restrict__ ph, const int32_t* __restrict__ pt,
                                                          int64_t n_pos, const int32_t* __restrict__ nh,
                                                          const int32_t* __restrict__ nt, int64_t n_neg, int npp,
                                                          int32_t* __restrict__ cnt) {
  const int64_t total = n_pos + n_neg;
  for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * MKE_BLOCK) {
    if (i < n_pos) {
      atomicAdd(&cnt[ph[i]], 1);
      atomicAdd(&cnt[pt[i]], 1);
    } else {
      const int64_t n = i - n_pos;
      const int64_t g = n / npp;
      const int a = nh[n], b = nt[n];
      if (a != ph[g]) atomicAdd(&cnt[a], 1);
      if (b != pt[g]) atomicAdd(&cnt[b], 1);
    }
  }
}

}  // namespace mke

static int score_impl(
    const float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel,
    int rel_normalize, int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t,
    const float* pos_w, int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t,
    const float* neg_w, int64_t n_neg, int neg_per_pos, float scale, float* grad_ent, float* grad_rel,
    int grad_rel_copies, int32_t* touched_ent, int32_t* touched_rel, int32_t tag, double* loss_partials, void* stream,
    int32_t* ref_count, float* ent_w, float* ent_acc, int optimizer, float lr, float* stage_rows = nullptr,
    int64_t* stage_keys = nullptr, int64_t stage_slots = 0, const mke_count_job* next_count = nullptr,
    const mke_hot_rows* hot = nullptr) {
  using namespace mke;
  if (!ent_table || !rel_table || !loss_partials) { set_error("mke_triple_score_fwd_bwd: NULL table/loss"); return MKE_E_NULL; }
  if (n_pos < 0 || n_neg < 0 || n_ent <= 0 || n_rel <= 0) { set_error("negative count"); return MKE_E_SHAPE; }
  if (n_pos > 0 && (!pos_h || !pos_r || !pos_t)) { set_error("NULL positive index stream"); return MKE_E_NULL; }
  if (n_neg > 0 && (!neg_h || !neg_r || !neg_t)) { set_error("NULL negative index stream"); return MKE_E_NULL; }
  if (stride <= 0 || stride % 16 != 0 || dim <= 0 || dim > stride || stride > MKE_MAX_STRIDE) {
    set_error("bad stride/dim: stride=%d dim=%d (stride %% 16 == 0, dim <= stride <= %d)", stride, dim, MKE_MAX_STRIDE);
    return MKE_E_SHAPE;
  }
  if (neg_per_pos < 0 || (neg_per_pos > 0 && n_neg != n_pos * (int64_t)neg_per_pos)) {
    set_error("grouped negatives: n_neg (%lld) != n_pos (%lld) * neg_per_pos (%d)", (long long)n_neg,
              (long long)n_pos, neg_per_pos);
    return MKE_E_SHAPE;
  }
  if ((grad_ent == nullptr) != (grad_rel == nullptr)) { set_error("grad_ent and grad_rel must both be given or both NULL"); return MKE_E_NULL; }
  if (grad_ent && (!touched_ent || !touched_rel)) { set_error("NULL touched array"); return MKE_E_NULL; }
  if (grad_ent && (grad_rel_copies < 1 || grad_rel_copies > 64)) { set_error("grad_rel_copies must be in [1,64]"); return MKE_E_SHAPE; }

  ScoreParams p;
  p.ent = ent_table; p.rel = rel_table; p.ent_norm = ent_normalize; p.rel_norm = rel_normalize;
  p.stride = stride; p.dim = dim;
  p.ph = pos_h; p.pr = pos_r; p.pt = pos_t; p.pw = pos_w; p.n_pos = n_pos;
  p.nh = neg_h; p.nr = neg_r; p.nt = neg_t; p.nw = neg_w; p.n_neg = n_neg;
  p.npp = neg_per_pos;
  const int64_t total_waves = (int64_t)MKE_LOSS_PARTIALS * (MKE_BLOCK / 64);
  int splits = 1;
  if (neg_per_pos > 0 && n_pos > 0) {
    int64_t s = total_waves / n_pos;  // spread small batches over the chip
    const int64_t smax = (neg_per_pos + 3) / 4;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    if (tune_score_splits() > 0) s = tune_score_splits();
    if (s > neg_per_pos) s = neg_per_pos;
    splits = (int)s;
  }
  if (stage_keys) {  // deterministic mode: one flush per group (its slot), every contribution has a slot of its own
    splits = 1;
    const int64_t need = (neg_per_pos > 0 ? n_pos * (neg_per_pos + 1) : n_pos + n_neg) * 3;
    if (!stage_rows || !grad_ent || need > stage_slots) { set_error("deterministic mode: %lld staging slots needed, %lld given", (long long)need, (long long)stage_slots); return MKE_E_SHAPE; }
  }
  p.stage_rows = stage_rows; p.stage_keys = stage_keys;
  p.hot_slot = nullptr; p.n_hot = 0; p.hot_copies = 1; p.hot_row0 = 0;
  if (hot && hot->slot && hot->n_hot > 0 && grad_ent && !stage_keys) {
    if (hot->copies < 1 || hot->row0 < n_ent || (hot->row0 + (int64_t)hot->copies * hot->n_hot) * stride >= (1ll << 31)) {
      set_error("hub rows: copies >= 1, row0 >= n_ent and the copy rows inside 2^31 floats of the scratch");
      return MKE_E_SHAPE;
    }
    p.hot_slot = hot->slot; p.n_hot = hot->n_hot; p.hot_copies = hot->copies; p.hot_row0 = (int32_t)hot->row0;
  }
  p.cj = mke_count_job{};
  p.count_blocks = 0;
  if (next_count && next_count->n_pos + next_count->n_neg > 0) {
    if (!next_count->ref_count || !next_count->pos_h || !next_count->pos_t || (next_count->n_neg > 0 && (!next_count->neg_h || !next_count->neg_t))) { set_error("count job: NULL pointer"); return MKE_E_NULL; }
    int64_t cb = (next_count->n_pos + next_count->n_neg + 4 * MKE_BLOCK - 1) / (4 * MKE_BLOCK);
    p.cj = *next_count;
    if (p.cj.neg_per_pos < 1) p.cj.neg_per_pos = 1;
    p.count_blocks = (int)std::min<int64_t>(cb, MKE_LOSS_PARTIALS / 4);
  }
  p.splits = splits;
  p.scale = scale;
  p.gent = grad_ent; p.grel = grad_rel; p.grel_copies = grad_rel_copies < 1 ? 1 : grad_rel_copies;
  p.grel_copy_elems = n_rel * (int64_t)stride;
  p.tent = touched_ent; p.trel = touched_rel; p.tag = tag;
  p.lossp = loss_partials;
  const bool excl = ref_count != nullptr && grad_ent != nullptr && neg_per_pos > 0;
  p.refcount = excl ? ref_count : nullptr; p.ent_w = ent_w; p.ent_acc = ent_acc; p.optimizer = optimizer; p.lr = lr;
  hipStream_t st = (hipStream_t)stream;
  const int fpl = stride / 16;
  // every row address of the launch fits 32 bits of byte offset (entity table = accumulator = gradient scratch in size)
  const int64_t grad_rows = p.hot_slot ? (int64_t)p.hot_row0 + (int64_t)p.hot_copies * p.n_hot : n_ent;
  const bool o32 = tune_score_o32() && grad_rows * (int64_t)stride < (1ll << 30) && n_rel * (int64_t)stride < (1ll << 30);
  MKE_DISPATCH_FPL(fpl, {
    // corrupt rows in flight per quarter-wave.  2, not 4, at FPL <= 5: with the accumulator rows of the exclusive-row path
    // U = 4 costs 144-153 registers = 3 waves per SIMD, U = 2 119 = 4 waves per SIMD, and the extra wave hides more
    // latency than the two extra gathers in flight did (41.9 -> 40.2 us at the C2 shape)
    constexpr int U = FPL <= 5 ? MKE_SCORE_U : (FPL <= 8 ? 2 : 1);
    // two groups per wavefront (a half-wave each): 5000 groups then need 2.4 wavefronts per SIMD instead of 4.9 and are all
    // resident at once (the kernel's 112+ registers allow four per SIMD).  Whole epochs, us per step, one / two groups per
    // wavefront (tools/half_groups_scan.py): dim 75: N 10 35.2 / 32.7, N 25 56.4 / 53.2, N 64 136 / 133; dim 128: N 25 89.7 / 83.6,
    // N 64 209 / 201; dim 200: N 25 143 / 140, N 40 208 / 210, N 64 312 / 326; dim 256, N 64 (C5): 238 / 277 us per launch.
    // Default: rows up to 128 floats always, wider rows up to 31 negatives (mke_set_option("score_half_groups", n): up to
    // n negatives at every width; 0: never).
    const int half_max = tune_score_half_max() >= 0 ? tune_score_half_max() : (FPL <= 8 ? 64 : 31);
    const bool half = neg_per_pos > 0 && neg_per_pos <= half_max && splits == 1;
    if (n_pos * (int64_t)splits > 0xFFFFFFFFll || n_pos + n_neg > 0xFFFFFFFFll) { set_error("more than 2^32 work items in one launch"); return MKE_E_RANGE; }
    if (stage_keys) {
      // deterministic mode: staging stores instead of atomics
      if (half) {
        if (excl) hipLaunchKernelGGL((k_triple_score<FPL, U, true, 2, true>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
        else hipLaunchKernelGGL((k_triple_score<FPL, U, false, 2, true>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
      } else {
        if (excl) hipLaunchKernelGGL((k_triple_score<FPL, U, true, 4, true>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
        else hipLaunchKernelGGL((k_triple_score<FPL, U, false, 4, true>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
      }
    } else if (excl && o32) {   // the training step on tables below 4 GB: 32-bit row offsets (row_at)
      if (half && tune_score_lane_ids()) hipLaunchKernelGGL((k_triple_score<FPL, U, true, 2, false, true, true>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
      else if (half) hipLaunchKernelGGL((k_triple_score<FPL, U, true, 2, false, true>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
      else if (tune_score_lane_ids()) hipLaunchKernelGGL((k_triple_score<FPL, U, true, 4, false, true, true>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
      else hipLaunchKernelGGL((k_triple_score<FPL, U, true, 4, false, true>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
    } else if (half) {
      if (excl) hipLaunchKernelGGL((k_triple_score<FPL, U, true, 2>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
      else hipLaunchKernelGGL((k_triple_score<FPL, U, false, 2>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
    } else {
      if (excl) hipLaunchKernelGGL((k_triple_score<FPL, U, true, 4>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
      else hipLaunchKernelGGL((k_triple_score<FPL, U, false, 4>), dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, st, p);
    }
  });
  return check_launch("k_triple_score");
}

extern "C" int mke_triple_score_fwd_bwd(
    const float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel,
    int rel_normalize, int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t,
    const float* pos_w, int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t,
    const float* neg_w, int64_t n_neg, int neg_per_pos, float scale, float* grad_ent, float* grad_rel,
    int grad_rel_copies, int32_t* touched_ent, int32_t* touched_rel, int32_t tag, double* loss_partials, void* stream) {
  return score_impl(ent_table, n_ent, ent_normalize, rel_table, n_rel, rel_normalize, stride, dim, pos_h, pos_r, pos_t, pos_w,
                    n_pos, neg_h, neg_r, neg_t, neg_w, n_neg, neg_per_pos, scale, grad_ent, grad_rel, grad_rel_copies,
                    touched_ent, touched_rel, tag, loss_partials, stream, nullptr, nullptr, nullptr, 0, 0.f);
}

extern "C" int mke_triple_score_fwd_bwd_xc(
    float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w,
    int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w, int64_t n_neg,
    int neg_per_pos, float scale, float* grad_ent, float* grad_rel, int grad_rel_copies, int32_t* touched_ent,
    int32_t* touched_rel, int32_t tag, int32_t* ref_count, float* ent_acc, int optimizer, float lr,
    const mke_count_job* next_count, double* loss_partials, void* stream) {
  using namespace mke;
  if (optimizer != MKE_OPT_ADAGRAD && optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", optimizer); return MKE_E_UNSUPPORTED; }
  if (ref_count && optimizer == MKE_OPT_ADAGRAD && !ent_acc) { set_error("exclusive-row path with Adagrad needs ent_acc"); return MKE_E_NULL; }
  if (ref_count && !grad_ent) { set_error("exclusive-row path needs the gradient scratch (it is a training step)"); return MKE_E_NULL; }
  return score_impl(ent_table, n_ent, ent_normalize, rel_table, n_rel, rel_normalize, stride, dim, pos_h, pos_r, pos_t, pos_w,
                    n_pos, neg_h, neg_r, neg_t, neg_w, n_neg, neg_per_pos, scale, grad_ent, grad_rel, grad_rel_copies,
                    touched_ent, touched_rel, tag, loss_partials, stream, ref_count, ent_table,
                    optimizer == MKE_OPT_ADAGRAD ? ent_acc : nullptr, optimizer, lr, nullptr, nullptr, 0, next_count);
}

extern "C" int mke_triple_score_fwd_bwd_xch(
    float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w,
    int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w, int64_t n_neg,
    int neg_per_pos, float scale, float* grad_ent, float* grad_rel, int grad_rel_copies, int32_t* touched_ent,
    int32_t* touched_rel, int32_t tag, int32_t* ref_count, float* ent_acc, int optimizer, float lr,
    const mke_count_job* next_count, const mke_hot_rows* hot, double* loss_partials, void* stream) {
  using namespace mke;
  if (optimizer != MKE_OPT_ADAGRAD && optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", optimizer); return MKE_E_UNSUPPORTED; }
  if (ref_count && optimizer == MKE_OPT_ADAGRAD && !ent_acc) { set_error("exclusive-row path with Adagrad needs ent_acc"); return MKE_E_NULL; }
  if (ref_count && !grad_ent) { set_error("exclusive-row path needs the gradient scratch (it is a training step)"); return MKE_E_NULL; }
  return score_impl(ent_table, n_ent, ent_normalize, rel_table, n_rel, rel_normalize, stride, dim, pos_h, pos_r, pos_t, pos_w,
                    n_pos, neg_h, neg_r, neg_t, neg_w, n_neg, neg_per_pos, scale, grad_ent, grad_rel, grad_rel_copies,
                    touched_ent, touched_rel, tag, loss_partials, stream, ref_count, ent_table,
                    optimizer == MKE_OPT_ADAGRAD ? ent_acc : nullptr, optimizer, lr, nullptr, nullptr, 0, next_count, hot);
}

extern "C" int mke_triple_score_fwd_bwd_t(
    float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w,
    int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w, int64_t n_neg,
    int neg_per_pos, float scale, float* grad_ent, float* grad_rel, int grad_rel_copies, int32_t* touched_ent,
    int32_t* touched_rel, int32_t tag, int32_t* ref_count, float* ent_acc, int optimizer, float lr,
    const mke_count_job* next_count, const mke_hot_rows* hot, const mke_tuning* tuning, double* loss_partials, void* stream) {
  mke::TuningScope scope(tuning);      // this call's knobs (NULL: the process defaults)
  return mke_triple_score_fwd_bwd_xch(ent_table, n_ent, ent_normalize, rel_table, n_rel, rel_normalize, stride, dim, pos_h, pos_r, pos_t,
                                      pos_w, n_pos, neg_h, neg_r, neg_t, neg_w, n_neg, neg_per_pos, scale, grad_ent, grad_rel,
                                      grad_rel_copies, touched_ent, touched_rel, tag, ref_count, ent_acc, optimizer, lr, next_count, hot,
                                      loss_partials, stream);
}

extern "C" int mke_triple_score_fwd_bwd_x(
    float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w,
    int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w, int64_t n_neg,
    int neg_per_pos, float scale, float* grad_ent, float* grad_rel, int grad_rel_copies, int32_t* touched_ent,
    int32_t* touched_rel, int32_t tag, int32_t* ref_count, float* ent_acc, int optimizer, float lr, double* loss_partials,
    void* stream) {
  using namespace mke;
  if (optimizer != MKE_OPT_ADAGRAD && optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", optimizer); return MKE_E_UNSUPPORTED; }
  if (ref_count && optimizer == MKE_OPT_ADAGRAD && !ent_acc) { set_error("exclusive-row path with Adagrad needs ent_acc"); return MKE_E_NULL; }
  if (ref_count && !grad_ent) { set_error("exclusive-row path needs the gradient scratch (it is a training step)"); return MKE_E_NULL; }
  return score_impl(ent_table, n_ent, ent_normalize, rel_table, n_rel, rel_normalize, stride, dim, pos_h, pos_r, pos_t, pos_w,
                    n_pos, neg_h, neg_r, neg_t, neg_w, n_neg, neg_per_pos, scale, grad_ent, grad_rel, grad_rel_copies,
                    touched_ent, touched_rel, tag, loss_partials, stream, ref_count, ent_table,
                    optimizer == MKE_OPT_ADAGRAD ? ent_acc : nullptr, optimizer, lr);
}

// ------------------------------------------------------------------------------------------------------------------------
// Deterministic mode: contributions were stored one per slot (emit_row); `order` = stable argsort of the slot keys, so the
// contributions of one row are adjacent and in slot order.  One quarter-wave per segment head sums its segment front to
// back and stores the row's gradient (the scratch is all-zero by invariant, so a store is an add).
namespace mke {
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_stage_reduce(const float* __restrict__ stage_rows, const int64_t* __restrict__ keys,
                                                            const int64_t* __restrict__ order, int64_t n, int stride,
                                                            float* __restrict__ gent, float* __restrict__ grel,
                                                            int32_t* __restrict__ tent, int32_t* __restrict__ trel, int32_t tag) {
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  for (int64_t i = sub0; i < n; i += nsub) {
    const int64_t key = keys[i];
    if (key >= (1ll << 41) || (i > 0 && keys[i - 1] == key)) continue;   // unused slot, or not the head of its segment
    double acc[FPL];   // a hub row's gradient is a cancelling sum of hundreds of fp32 terms: summed in double, rounded once
#pragma unroll
    for (int k = 0; k < FPL; ++k) acc[k] = 0.0;
    for (int64_t m = i; m < n && keys[m] == key; ++m) {
      const float* src = stage_rows + order[m] * stride + j;
#pragma unroll
      for (int k = 0; k < FPL; ++k) acc[k] += (double)src[k * 16];
    }
    const bool is_rel = (key & MKE_STAGE_REL) != 0;
    const int64_t row = key & (MKE_STAGE_REL - 1);
    float* o = (is_rel ? grel : gent) + row * stride + j;
#pragma unroll
    for (int k = 0; k < FPL; ++k) o[k * 16] = (float)acc[k];
    if (j == 0) (is_rel ? trel : tent)[row] = tag;
  }
}
}  // namespace mke

extern "C" int mke_triple_score_fwd_bwd_det(
    float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w,
    int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w, int64_t n_neg,
    int neg_per_pos, float scale, float* grad_ent, float* grad_rel, int32_t* touched_ent, int32_t* touched_rel, int32_t tag,
    int32_t* ref_count, float* ent_acc, int optimizer, float lr, float* stage_rows, int64_t* stage_keys, int64_t stage_slots,
    double* loss_partials, void* stream) {
  using namespace mke;
  if (optimizer != MKE_OPT_ADAGRAD && optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", optimizer); return MKE_E_UNSUPPORTED; }
  if (ref_count && optimizer == MKE_OPT_ADAGRAD && !ent_acc) { set_error("exclusive-row path with Adagrad needs ent_acc"); return MKE_E_NULL; }
  if (!stage_rows || !stage_keys) { set_error("mke_triple_score_fwd_bwd_det: NULL staging buffers"); return MKE_E_NULL; }
  return score_impl(ent_table, n_ent, ent_normalize, rel_table, n_rel, rel_normalize, stride, dim, pos_h, pos_r, pos_t, pos_w,
                    n_pos, neg_h, neg_r, neg_t, neg_w, n_neg, neg_per_pos, scale, grad_ent, grad_rel, 1, touched_ent, touched_rel, tag,
                    loss_partials, stream, ref_count, ent_table, optimizer == MKE_OPT_ADAGRAD ? ent_acc : nullptr, optimizer, lr,
                    stage_rows, stage_keys, stage_slots);
}

extern "C" int mke_stage_reduce(const float* stage_rows, const int64_t* sorted_keys, const int64_t* order, int64_t n_slots, int stride,
                                float* grad_ent, float* grad_rel, int32_t* touched_ent, int32_t* touched_rel, int32_t tag,
                                void* stream) {
  using namespace mke;
  if (n_slots < 0) { set_error("mke_stage_reduce: negative count"); return MKE_E_SHAPE; }
  if (n_slots == 0) return MKE_OK;
  if (!stage_rows || !sorted_keys || !order || !grad_ent || !grad_rel || !touched_ent || !touched_rel) { set_error("mke_stage_reduce: NULL pointer"); return MKE_E_NULL; }
  if (stride <= 0 || stride % 16 != 0 || stride > MKE_MAX_STRIDE) { set_error("mke_stage_reduce: bad stride %d", stride); return MKE_E_SHAPE; }
  int64_t blocks = (n_slots + MKE_SUBS_PER_BLOCK - 1) / MKE_SUBS_PER_BLOCK;
  if (blocks > 4096) blocks = 4096;
  const int fpl = stride / 16;
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_stage_reduce<FPL>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, stage_rows, sorted_keys,
                       order, n_slots, stride, grad_ent, grad_rel, touched_ent, touched_rel, tag);
  });
  return check_launch("k_stage_reduce");
}

extern "C" int mke_count_entity_refs(const int32_t* pos_h, const int32_t* pos_t, int64_t n_pos, const int32_t* neg_h,
                                     const int32_t* neg_t, int64_t n_neg, int neg_per_pos, int32_t* ref_count, void* stream) {
  using namespace mke;
  if (n_pos < 0 || n_neg < 0) { set_error("negative count"); return MKE_E_SHAPE; }
  if (n_pos + n_neg == 0) return MKE_OK;
  if (!ref_count || !pos_h || !pos_t || (n_neg > 0 && (!neg_h || !neg_t))) { set_error("mke_count_entity_refs: NULL pointer"); return MKE_E_NULL; }
  if (n_neg > 0 && (neg_per_pos < 1 || n_neg != n_pos * (int64_t)neg_per_pos)) { set_error("mke_count_entity_refs needs grouped negatives"); return MKE_E_SHAPE; }
  int64_t blocks = (n_pos + n_neg + MKE_BLOCK - 1) / MKE_BLOCK;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_count_refs, dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, pos_h, pos_t, n_pos, neg_h,
                     neg_t, n_neg, neg_per_pos < 1 ? 1 : neg_per_pos, ref_count);
  return check_launch("k_count_refs");
}
