// mke_oc_em.hip — entity-major second pass of the owner-computes (multi-GPU) relation-view step (gfx950).
//
// New design (the reference is single-device: one dense update per row per step, code/MultiKE_model.py:304-310).  Measured in
// round 5 (EXPERIMENTS R5.22-R5.26): at 8 ranks a rank's score launch was bound by its atomic row adds (C2: 36 of 49.5 us, C5:
// 266 of 270 us as empty kernels making the same visits), and k_oc_apply + the update launch then re-read the scratch those
// atomics filled.  Here the score launch (pass 1, mke_oc.hip with EM = true) stores ONE coefficient per (positive, owned
// negative) and this file does the rest:
//   * mke_oc_em_plan (per epoch, table-independent, on the plan's side stream): every reference of every global step to a row
//     this rank owns — its owned negatives, the positives' own terms it scores, the heads / tails whose gradient vector comes
//     back by the reduce-scatter, the relation rows' gradient vectors — listed in a fixed element order without atomics (the
//     negatives in code order, then five elements per position; count per wavefront range, prefix sum, fill — the owned
//     elements queued per wavefront and emitted 64 at a time, each with its (locator, coefficient index) pair), STABLY sorted
//     by (step, local row) (hipcub radix sort of 32-bit keys, payload = the element's position in the list: a row's references
//     keep the list's order), gathered into that order; then the touched rows of each step with their offsets, cut into WORK
//     ITEMS of at most 32 references (a long row's segments leave partial sums), each flagged when it reads a gradient vector;
//   * k_oc_em_pass2 (per global step): a quarter-wave per work item — raw row + accumulator read once, c^ formed once,
//     ghat = c^ sum coef + sum coef sg V + sum (+-) gv over the references in list order, Jacobian of the normalisation,
//     Adagrad / SGD, row and accumulator written (a relation row: this rank's partial gradient stored for the all-reduce);
//     k_oc_em_combine adds a long row's partials in segment order and finishes it.  No gradient scratch, no touched flags, no
//     reference counts, no atomics; the summation order per row is the list's: bit-reproducible run to run.  em_mode lets the
//     loop run the items that read no gradient vector while the step's reduce-scatter is on the wire (mke_oc_loop.hip).
#include "mke_common.h"

#include <hipcub/hipcub.hpp>

namespace mke {

// kinds of a reference inside its positive: 0 .. 63 = negative n
#define EM_KIND_OWN 64      // the positive's own term, scored on this rank
#define EM_KIND_GV_H 65     // head row <- + gv[HR slot]
#define EM_KIND_GV_T 66     // tail row <- - gv[RT slot]
#define EM_KIND_REL_H 67    // relation row (local row n_local + r) <- + gv[HR slot]   (HR = h^ + r^)
#define EM_KIND_REL_T 68    // relation row <- + gv[RT slot]                             (RT = r^ - t^)
// locator: bit 31 = gradient vector (gv) instead of a base vector; bits 30:28 chunk; 27:24 owner rank (base vectors) / bit 24 =
// "always +" (gradient vectors into a relation row); bit 23 RT half; 22:0 slot
#define EM_LOC_GV 0x80000000u
#define EM_LOC_PLUS (1u << 24)

extern int g_oc_em_keys64;   // option "oc_em_keys64" (mke_api.hip)

struct EmPlanParams {
  mke_oc_em_plan_args a;
  uint32_t ep;          // elements per positive: neg_per_pos + 5
  uint64_t ep_magic;    // ceil(2^40 / ep)
  int64_t n_codes;      // n_all * neg_per_pos: elements [0, n_codes) are the negatives in code order, [n_codes, n_codes + 5 n_all) the
                        // five other elements of every position
  uint64_t n_magic;     // ceil(2^40 / neg_per_pos)
  int64_t total;        // n_codes + 5 n_all
  int g_shift;          // log2(n_ranks) when it is a power of two, else -1
  int64_t rows_tot;     // n_local + n_rel: the relation rows follow the shard's rows
  int64_t per;          // elements per wavefront range (a multiple of 64)
  int n_waves;          // wavefront ranges
  int32_t* wave_cnt;    // [n_waves + 1] owned elements per range
  int32_t* wave_off;    // [n_waves + 1] exclusive prefix (wave_off[n_waves] = all)
};

__device__ __forceinline__ int em_step_of(const int64_t* __restrict__ step_lo, int n_steps, int64_t p) {
  int lo = 0, hi = n_steps;      // step_lo[lo] <= p < step_lo[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (step_lo[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// element t = (epoch position p, element n of the position): n < N negative n; then the own term, the head / tail rows' gradient
// vectors, the relation row's two gradient vectors.  Owned by this rank -> (step * rows_tot + local row, (position in step, kind)).
// The elements are enumerated position-major, kinds ascending: a STABLE sort by (step, row) of the owned elements in this order
// leaves every row's references sorted by (positive, kind) — the summation order of the second pass.
// The loads of an element are unconditional (clamped indices) and independent of each other, so that the EM_U elements a lane
// handles per round are in flight together: with one dependent load per round the walks ran at the latency of a round trip per
// 64 elements and wavefront (2.8 + 4.5 ms per epoch at the C5 shape with 8 ranks).
struct EmElem { int ent; int kind; int64_t p; int code, sh, st, ph, pt; };
// Element t of the epoch: t < n_codes — negative t % N of position t / N, ONE load (its code; the position and the positive's slots
// are looked up only for the eighth this rank owns); then five elements per position — own term, the head / tail rows' gradient
// vectors, the relation row's two — from the position's slots and ids.  (Round 6 first enumerated position-major, N + 5 elements
// per position with the five loads for every element: 62 + 148 us per epoch share at C2 with 8 ranks for the two walks.)  A
// STABLE sort by (step, row) of the owned elements in THIS order leaves every row's references in a fixed order — its negatives by
// (positive, n), then the other kinds by (positive, kind) — the summation order of the second pass.
__device__ __forceinline__ EmElem em_elem_of(const EmPlanParams& pp, int64_t t) {
  const mke_oc_em_plan_args& a = pp.a;
  EmElem e;
  e.sh = e.st = e.ph = e.pt = 0;
  e.p = -1;
  const bool in = t < pp.total;
  const int64_t tc = in ? t : 0;
  // the code load is unconditional (clamped) so that the EM_U elements of a lane are in flight together
  e.code = pp.n_codes > 0 ? a.codes[tc < pp.n_codes ? tc : pp.n_codes - 1] : 0;
  if (tc < pp.n_codes) {
    e.kind = -1;                                   // a negative: its n = t - p N when the position is looked up (em_locate)
    e.ent = (e.code & 0x3FFFFFFF) >> 1;
  } else {
    const int64_t q = tc - pp.n_codes;
    const int64_t p = q / 5;
    const int kk = (int)(q - p * 5);
    e.p = p;
    e.sh = a.slot_h[p]; e.st = a.slot_t[p]; e.ph = a.pos_h[p]; e.pt = a.pos_t[p];
    e.kind = kk == 0 ? EM_KIND_OWN : (kk == 1 ? EM_KIND_GV_H : (kk == 2 ? EM_KIND_GV_T : (kk == 3 ? EM_KIND_REL_H : EM_KIND_REL_T)));
    const int own_ent = e.sh >= 0 ? e.pt : e.ph;       // own term: the owner of t when HR travels, else the owner of h
    const int head_ent = e.sh >= 0 ? e.ph : -1;        // the owner of the head receives sum dL/dHR: head row and relation row
    const int tail_ent = e.st >= 0 ? e.pt : -1;        // the owner of the tail receives sum dL/dRT: tail row and relation row
    e.ent = kk == 0 ? own_ent : ((kk == 1 || kk == 3) ? head_ent : tail_ent);
  }
  if (!in) e.ent = -1;
  return e;
}
// an OWNED negative's position, index in its group, and the positive's slots / ids
__device__ __forceinline__ void em_locate(const EmPlanParams& pp, int64_t t, EmElem& e) {
  if (e.kind >= 0) return;
  const mke_oc_em_plan_args& a = pp.a;
  const int N = a.neg_per_pos;
  const int64_t p = pp.n_codes <= 0xFFFFFFFFll ? (int64_t)(((uint64_t)(uint32_t)t * pp.n_magic) >> 40) : t / N;
  e.p = p;
  e.kind = (int)(t - p * N);
  e.sh = a.slot_h[p]; e.st = a.slot_t[p]; e.ph = a.pos_h[p]; e.pt = a.pos_t[p];
}
__device__ __forceinline__ bool em_owned(const EmPlanParams& pp, const EmElem& e) {
  if (e.ent < 0) return false;
  const int G = pp.a.n_ranks;
  const int owner = pp.g_shift >= 0 ? (e.ent & (G - 1)) : (int)((uint32_t)e.ent % (uint32_t)G);     // uniform choice
  return owner == pp.a.rank;
}
#define EM_U 4

// A wavefront per contiguous range of elements, in two launches with a prefix sum between them: count the owned ones; append
// (key, descriptor) at the range's offset + ballot rank — no cursor atomic (a returning atomic on one address costs ~12 ns:
// one per 64 elements was 5.2 ms of a 6.2 ms plan at the C2 shape with 8 ranks), and the list comes out in element order.
__global__ __launch_bounds__(MKE_BLOCK) void k_em_count(const EmPlanParams pp) {
  const int64_t total = pp.total;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 6;
  if (wave >= pp.n_waves) return;
  const int64_t t0 = wave * pp.per, t1 = t0 + pp.per < total ? t0 + pp.per : total;
  int cnt = 0;
  for (int64_t t = t0 + lane; t - lane < t1; t += 64 * EM_U) {
    EmElem e[EM_U];
#pragma unroll
    for (int u = 0; u < EM_U; ++u) e[u] = em_elem_of(pp, t + 64 * u < t1 ? t + 64 * u : total);
#pragma unroll
    for (int u = 0; u < EM_U; ++u) cnt += em_owned(pp, e[u]) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (lane == 0) pp.wave_cnt[wave] = cnt;
  if (wave == 0 && lane == 0) pp.wave_cnt[pp.n_waves] = 0;
}

// (locator, coefficient index) of an owned element — everything it needs is in the element's registers already
__device__ __forceinline__ uint2 em_ref_of(const EmPlanParams& pp, const EmElem& e, int s, int64_t i) {
  const mke_oc_em_plan_args& a = pp.a;
  const int N = a.neg_per_pos, G = a.n_ranks;
  const int64_t lo = a.step_lo[s], size = a.step_lo[s + 1] - lo;
  const int64_t part = (size + a.chunks - 1) / a.chunks;
  const uint32_t chunk = (uint32_t)(i / (part > 0 ? part : 1));
  bool rt;
  uint32_t loc, cidx = 0;
  if (e.kind >= EM_KIND_GV_H) {
    rt = e.kind == EM_KIND_GV_T || e.kind == EM_KIND_REL_T;
    loc = EM_LOC_GV | (e.kind >= EM_KIND_REL_H ? EM_LOC_PLUS : 0u) | (uint32_t)(rt ? e.st : e.sh);
  } else {
    if (e.kind == EM_KIND_OWN) {
      rt = e.sh < 0;                                        // HR travels: d = HR - t^, else d = h^ + RT
      cidx = (uint32_t)(i * (N + 1) + N);
    } else {
      rt = (e.code & 1) != 0;                               // corrupted head: d = c^ + RT
      cidx = (uint32_t)(i * (N + 1) + e.kind);
    }
    const uint32_t ent = (uint32_t)(rt ? e.pt : e.ph);
    const uint32_t owner = pp.g_shift >= 0 ? (ent & (uint32_t)(G - 1)) : ent % (uint32_t)G;
    loc = (owner << 24) | (uint32_t)(rt ? e.st : e.sh);
  }
  loc |= (chunk << 28) | (rt ? (1u << 23) : 0u);
  return make_uint2(loc, cidx);
}

// The owned elements of a round are an eighth of its lanes at 8 ranks, and what an owned element needs (its position, the
// positive's slots and ids, its step, key and reference) is ~100 instructions and five dependent loads: done under `if (mine)` the whole
// wavefront pays for them in nearly every round.  The rounds therefore only QUEUE the owned element indices (ballot-compacted, in
// order, in the wavefront's own LDS strip), and whenever 64 are waiting all 64 lanes take one each — the same output order.
__device__ __forceinline__ void em_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename KEY>
__device__ __forceinline__ void em_emit(const EmPlanParams& pp, int64_t t, int64_t k, int& s_hint, KEY* __restrict__ keys, uint32_t* __restrict__ vals,
                                        uint2* __restrict__ refs_unsorted) {
  const mke_oc_em_plan_args& a = pp.a;
  const int G = a.n_ranks;
  EmElem e = em_elem_of(pp, t);
  em_locate(pp, t, e);
  int s = s_hint;                                           // a step at or before the element's, or anything after the seam
  if (a.step_lo[s] > e.p) s = em_step_of(a.step_lo, a.n_steps, e.p);
  while (s + 1 < a.n_steps && a.step_lo[s + 1] <= e.p) ++s;
  s_hint = s;
  const uint64_t row = e.kind >= EM_KIND_REL_H ? (uint64_t)a.n_local + (uint32_t)a.pos_r[e.p]
                                               : (uint64_t)(pp.g_shift >= 0 ? (uint32_t)e.ent >> pp.g_shift : (uint32_t)e.ent / (uint32_t)G);
  const uint64_t srow = (uint64_t)s * (uint64_t)pp.rows_tot + row;
  if (k < a.capacity) { keys[k] = (KEY)srow; vals[k] = (uint32_t)k; refs_unsorted[k] = em_ref_of(pp, e, s, e.p - a.step_lo[s]); }
}

template <typename KEY>
__global__ __launch_bounds__(MKE_BLOCK) void k_em_fill(const EmPlanParams pp, KEY* __restrict__ keys, uint32_t* __restrict__ vals, uint2* __restrict__ refs_unsorted) {
  const mke_oc_em_plan_args& a = pp.a;
  __shared__ int64_t s_q[MKE_BLOCK / 64][128];              // per wavefront: owned element indices waiting for their turn
  const int64_t total = pp.total;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 6;
  if (wave >= pp.n_waves) return;
  if (wave == 0 && lane == 0) a.n_refs[0] = pp.wave_off[pp.n_waves];
  const int64_t t0 = wave * pp.per, t1 = t0 + pp.per < total ? t0 + pp.per : total;
  if (t0 >= total) return;                                  // wave-uniform
  int64_t base = pp.wave_off[wave];
  int qn = 0;                                               // wave-uniform: elements waiting
  int s_hint = 0;                                           // per lane: the step of the last element this lane emitted
  int64_t* q = s_q[wv];
  for (int64_t t = t0 + lane; t - lane < t1; t += 64 * EM_U) {     // wave-uniform trip count (ballots inside)
    EmElem e[EM_U];
#pragma unroll
    for (int u = 0; u < EM_U; ++u) e[u] = em_elem_of(pp, t + 64 * u < t1 ? t + 64 * u : total);
#pragma unroll
    for (int u = 0; u < EM_U; ++u) {
      const bool mine = em_owned(pp, e[u]);
      const uint64_t m = __ballot(mine);
      if (mine) q[qn + __popcll(m & ((1ull << lane) - 1ull))] = t + 64 * u;
      qn += __popcll(m);
      if (qn >= 64) {                                       // wave-uniform
        em_wave_lds_sync();
        const int64_t tq = q[lane];
        const int64_t spill = lane + 64 < qn ? q[lane + 64] : 0;
        em_emit<KEY>(pp, tq, base + lane, s_hint, keys, vals, refs_unsorted);
        em_wave_lds_sync();
        if (lane + 64 < qn) q[lane] = spill;              // at most 63 stay behind
        base += 64;
        qn -= 64;
        em_wave_lds_sync();
      }
    }
  }
  if (qn > 0) {
    em_wave_lds_sync();
    if (lane < qn) em_emit<KEY>(pp, q[lane], base + lane, s_hint, keys, vals, refs_unsorted);
  }
}

// sorted reference k <- the (locator, coefficient index) its element left at its unsorted position, and the "a new (step, row)
// starts here" flag; entry `capacity` (and every sentinel) is the end marker
template <typename KEY>
__global__ __launch_bounds__(MKE_BLOCK) void k_em_resolve(const EmPlanParams pp, const KEY* __restrict__ sorted, const uint32_t* __restrict__ src,
                                                          const uint2* __restrict__ refs_unsorted) {
  const mke_oc_em_plan_args& a = pp.a;
  const int64_t k = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x;
  if (k > a.capacity) return;
  const KEY SENT = (KEY)~(KEY)0;
  const KEY srow = k < a.capacity ? sorted[k] : SENT;
  const KEY prev = k > 0 ? sorted[k - 1] : SENT;
  if (srow == SENT) {
    a.flags[k] = (k == 0 || prev != SENT) ? 1 : 0;        // the first sentinel closes the last row's list
    return;
  }
  a.flags[k] = (k == 0 || prev != srow) ? 1 : 0;
  reinterpret_cast<uint2*>(a.refs)[k] = refs_unsorted[src[k]];
}

// flagged k: the scan[k]-th touched (step, row) starts at reference k; step boundaries fall out of the same walk
template <typename KEY>
__global__ __launch_bounds__(MKE_BLOCK) void k_em_rows(const EmPlanParams pp, const KEY* __restrict__ sorted) {
  const mke_oc_em_plan_args& a = pp.a;
  const int64_t k = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x;
  if (k > a.capacity || !a.flags[k]) return;
  const KEY SENT = (KEY)~(KEY)0;
  const KEY srow = k < a.capacity ? sorted[k] : SENT;
  const int u = a.scan[k];
  a.off[u] = (int32_t)k;
  int s_cur = a.n_steps;
  if (srow != SENT) {
    s_cur = (int)((uint64_t)srow / (uint64_t)pp.rows_tot);
    a.rows[u] = (int32_t)((uint64_t)srow - (uint64_t)s_cur * (uint64_t)pp.rows_tot);
  }
  const int s_prev = k > 0 ? (int)((uint64_t)sorted[k - 1] / (uint64_t)pp.rows_tot) : -1;
  for (int s = s_prev + 1; s <= s_cur; ++s) a.step_row0[s] = u;
}

// ---- work items of the second pass: a row's list in segments of at most EM_SEG references ------------------------------------
// A row with a long list (a hub entity of a skewed KG, a frequent relation: thousands of references per step) walked by ONE
// quarter-wave is a chain of thousands of round trips — the whole launch waits for it (measured at Zipf(1.0): the second pass 20 ->
// hundreds of us).  Its list is cut into segments; each segment is a work item of its own that leaves a PARTIAL sum (the gradient
// vector and the coefficient sum: neither needs the row), and a combine launch adds a long row's partials in segment order and
// finishes the row — still one fixed summation order.  Rows whose list fits one segment (all but a handful) are finished by
// their single item as before.
#define EM_SEG 32
struct EmItemParams {
  const int32_t* rows_u; const int32_t* off_u; const int64_t* step_row0; int n_steps; int64_t cap;
  int32_t* nseg; int32_t* lflag; int32_t* lns;                 // per touched row: segments, long?, segments if long
  const int32_t* itemoff; const int32_t* lidx; const int32_t* part0;   // their exclusive scans
  int32_t* item_row; int32_t* item_off; int32_t* item_part; int32_t* long_row; int32_t* long_part0;
  int64_t* step_item0; int64_t* step_long0; int64_t* step_part0;
};
__global__ __launch_bounds__(MKE_BLOCK) void k_em_nseg(const EmItemParams p) {
  const int64_t u = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x;
  if (u > p.cap) return;
  const int64_t U = p.step_row0[p.n_steps];
  int ns = 0;
  if (u < U) ns = (p.off_u[u + 1] - p.off_u[u] + EM_SEG - 1) / EM_SEG;
  p.nseg[u] = ns;
  p.lflag[u] = ns > 1;
  p.lns[u] = ns > 1 ? ns : 0;
}
__global__ __launch_bounds__(MKE_BLOCK) void k_em_items(const EmItemParams p) {
  const int64_t u = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x;
  const int64_t U = p.step_row0[p.n_steps];
  if (u <= p.n_steps) {     // the steps' first item / long row / partial
    const int64_t r0 = p.step_row0[u];
    p.step_item0[u] = p.itemoff[r0]; p.step_long0[u] = p.lidx[r0]; p.step_part0[u] = p.part0[r0];
  }
  if (u > U || u > p.cap) return;
  if (u == U) {             // end markers
    p.item_off[p.itemoff[U]] = p.off_u[U];
    p.long_part0[p.lidx[U]] = p.part0[U];
    return;
  }
  const int ns = p.nseg[u], w0 = p.itemoff[u], lo = p.off_u[u];
  const bool lng = ns > 1;
  for (int k = 0; k < ns; ++k) {
    p.item_row[w0 + k] = p.rows_u[u] | (lng ? (int32_t)0x80000000 : 0);
    p.item_off[w0 + k] = lo + k * EM_SEG;
    p.item_part[w0 + k] = lng ? p.part0[u] + k : -1;
  }
  if (lng) { p.long_row[p.lidx[u]] = p.rows_u[u]; p.long_part0[p.lidx[u]] = p.part0[u]; }
}

// bit 30 of item_row: the item's references include a gradient vector (gv) — it can only run after the step's reduce-scatter; the
// others (at 8 ranks: ~95 % of the rows, the corrupt entities of negatives) need only the coefficients and the all-gathered vectors
// and may run WHILE the reduce-scatter is on the wire (mke_oc_step.em_mode)
#define EM_ITEM_SEG 0x80000000u
#define EM_ITEM_GV 0x40000000u
#define EM_ITEM_ROW 0x3FFFFFFFu
__global__ __launch_bounds__(MKE_BLOCK) void k_em_item_flags(const EmItemParams p, const uint32_t* __restrict__ refs) {
  const int64_t w = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x;
  const int64_t U = p.step_row0[p.n_steps];
  const int64_t W = p.itemoff[U];
  if (w >= W) return;
  const int lo = p.item_off[w], hi = p.item_off[w + 1];
  bool gv = ((uint32_t)p.item_row[w] & EM_ITEM_SEG) != 0;      // a long row's segments wait too: their combine follows the flagged items
  for (int k = lo; k < hi; ++k) gv = gv || (refs[2 * (int64_t)k] & EM_LOC_GV) != 0;
  if (gv) p.item_row[w] |= (int32_t)EM_ITEM_GV;
}

static inline int bits_for(uint64_t v) {   // smallest b with v < 2^b
  int b = 0;
  while (b < 64 && (v >> b)) ++b;
  return b;
}

template <typename KEY>
static int em_plan_sorted(const EmPlanParams& pp, int key_bits, hipStream_t st) {
  const mke_oc_em_plan_args& a = pp.a;
  KEY* keys = reinterpret_cast<KEY*>(a.keys);
  KEY* keys_alt = reinterpret_cast<KEY*>(a.keys_alt);
  uint32_t* vals = reinterpret_cast<uint32_t*>(a.flags);           // the unsorted positions are dead after the sort
  uint2* refs_unsorted = reinterpret_cast<uint2*>(a.scratch8);     // the references at their unsorted positions (gathered after the sort)
  hipError_t e;
  if ((e = hipMemsetAsync(keys, 0xFF, (size_t)(a.capacity + 1) * sizeof(KEY), st)) != hipSuccess) { set_error("mke_oc_em_plan: %s", hipGetErrorString(e)); return (int)e; }
  const int64_t total = a.n_all * (int64_t)pp.ep;
  const dim3 wgrid((unsigned)((pp.n_waves * 64 + MKE_BLOCK - 1) / MKE_BLOCK));
  size_t tb = (size_t)a.temp_bytes;
  if (total > 0) {
    hipLaunchKernelGGL(k_em_count, wgrid, dim3(MKE_BLOCK), 0, st, pp);
    int rc = check_launch("k_em_count");
    if (rc) return rc;
    if ((e = hipcub::DeviceScan::ExclusiveSum(a.temp, tb, pp.wave_cnt, pp.wave_off, pp.n_waves + 1, st)) != hipSuccess) { set_error("mke_oc_em_plan: scan: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL((k_em_fill<KEY>), wgrid, dim3(MKE_BLOCK), 0, st, pp, keys, vals, refs_unsorted);
    if ((rc = check_launch("k_em_fill"))) return rc;
  } else if ((e = hipMemsetAsync(a.n_refs, 0, sizeof(int64_t), st)) != hipSuccess) { set_error("mke_oc_em_plan: %s", hipGetErrorString(e)); return (int)e; }
  const int n = (int)(a.capacity + 1);
  tb = (size_t)a.temp_bytes;
  if ((e = hipcub::DeviceRadixSort::SortPairs(a.temp, tb, keys, keys_alt, vals, a.vals_alt, n, 0, key_bits, st)) != hipSuccess) { set_error("mke_oc_em_plan: radix sort: %s", hipGetErrorString(e)); return (int)e; }
  const dim3 grid((unsigned)((a.capacity + 1 + MKE_BLOCK - 1) / MKE_BLOCK));
  hipLaunchKernelGGL((k_em_resolve<KEY>), grid, dim3(MKE_BLOCK), 0, st, pp, (const KEY*)keys_alt, (const uint32_t*)a.vals_alt, (const uint2*)refs_unsorted);
  int rc = check_launch("k_em_resolve");
  if (rc) return rc;
  tb = (size_t)a.temp_bytes;
  if ((e = hipcub::DeviceScan::ExclusiveSum(a.temp, tb, a.flags, a.scan, n, st)) != hipSuccess) { set_error("mke_oc_em_plan: scan: %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL((k_em_rows<KEY>), grid, dim3(MKE_BLOCK), 0, st, pp, (const KEY*)keys_alt);
  if ((rc = check_launch("k_em_rows"))) return rc;
  // work items (segments of at most EM_SEG references), the long rows and their partial slots; the scratch is the sort's, dead now
  EmItemParams ip;
  ip.rows_u = a.rows; ip.off_u = a.off; ip.step_row0 = a.step_row0; ip.n_steps = a.n_steps; ip.cap = a.capacity;
  int32_t* k0 = reinterpret_cast<int32_t*>(a.keys);
  int32_t* k1 = reinterpret_cast<int32_t*>(a.keys_alt);
  ip.nseg = a.flags; ip.lflag = k0; ip.lns = k1;
  int32_t* itemoff = a.scan; int32_t* lidx = k0 + (a.capacity + 1); int32_t* part0 = k1 + (a.capacity + 1);
  ip.itemoff = itemoff; ip.lidx = lidx; ip.part0 = part0;
  ip.item_row = a.item_row; ip.item_off = a.item_off; ip.item_part = a.item_part; ip.long_row = a.long_row; ip.long_part0 = a.long_part0;
  ip.step_item0 = a.step_item0; ip.step_long0 = a.step_long0; ip.step_part0 = a.step_part0;
  // the per-row arrays are at most n_steps x (n_local + n_rel) + 1 long (the scans over the whole reference capacity were 24 us each)
  int64_t nu64 = (int64_t)a.n_steps * pp.rows_tot + 1;
  if (nu64 > a.capacity + 1) nu64 = a.capacity + 1;
  const int nu = (int)nu64;
  ip.cap = nu64 - 1;
  const dim3 ugrid((unsigned)((nu64 + MKE_BLOCK - 1) / MKE_BLOCK));
  hipLaunchKernelGGL(k_em_nseg, ugrid, dim3(MKE_BLOCK), 0, st, ip);
  if ((rc = check_launch("k_em_nseg"))) return rc;
  int32_t* const ins[3] = {ip.nseg, ip.lflag, ip.lns};
  int32_t* const outs[3] = {itemoff, lidx, part0};
  for (int k = 0; k < 3; ++k) {
    tb = (size_t)a.temp_bytes;
    if ((e = hipcub::DeviceScan::ExclusiveSum(a.temp, tb, ins[k], outs[k], nu, st)) != hipSuccess) { set_error("mke_oc_em_plan: scan: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(k_em_items, ugrid, dim3(MKE_BLOCK), 0, st, ip);
  if ((rc = check_launch("k_em_items"))) return rc;
  hipLaunchKernelGGL(k_em_item_flags, grid, dim3(MKE_BLOCK), 0, st, ip, (const uint32_t*)a.refs);     // at most one item per reference
  return check_launch("k_em_item_flags");
}

// ---- pass 2 -------------------------------------------------------------------------------------------------------------
// ghat = c^ csum + g;  Jacobian of x * rsqrt(max(sum x^2, eps)):  g_raw = (ghat - c^ (c^ . ghat)) * inv;  optimizer; stores
template <int FPL>
__device__ __forceinline__ void em_finish_row(const mke_oc_step& s, float* wp, float* ap, float (&w)[FPL], float (&acc)[FPL], float (&g)[FPL], float csum) {
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < FPL; ++k) ss = fmaf(w[k], w[k], ss);
  ss = sub16_sum(ss);
  const float inv = rsqrtf(fmaxf(ss, MKE_L2_EPS));
  const float ci = csum * inv;
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < FPL; ++k) {
    g[k] = fmaf(ci, w[k], g[k]);
    dot = fmaf(w[k], g[k], dot);
  }
  dot = sub16_sum(dot);
  const float coef = (ss > MKE_L2_EPS) ? dot * inv * inv : 0.f;
#pragma unroll
  for (int k = 0; k < FPL; ++k) g[k] = (g[k] - w[k] * coef) * inv;
  if (s.optimizer == MKE_OPT_ADAGRAD) {
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      acc[k] = fmaf(g[k], g[k], acc[k]);
      w[k] -= s.lr * g[k] * adagrad_scale(acc[k]);
    }
#pragma unroll
    for (int k = 0; k < FPL; ++k) ap[k * 16] = acc[k];
  } else {
#pragma unroll
    for (int k = 0; k < FPL; ++k) w[k] -= s.lr * g[k];
  }
#pragma unroll
  for (int k = 0; k < FPL; ++k) wp[k * 16] = w[k];
}

// a quarter-wave per work item (a touched owned row, or one segment of a long row's list)
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_oc_em_pass2(const mke_oc_step s) {
  constexpr int STRIDE = FPL * 16;
  constexpr int U = FPL <= 8 ? 4 : 2;            // vectors in flight per quarter-wave
  const int lane = threadIdx.x & 63, j = lane & 15, qb = lane & 48;
  const int64_t u = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  bool act = u < s.em_n_rows;
  const uint32_t rowf = act ? (uint32_t)s.em_rows[u] : 0u;
  const bool seg = (rowf & EM_ITEM_SEG) != 0;    // bit 31: a segment of a long row — leaves a partial sum, does not touch the row
  // em_mode 1: only the items WITHOUT gradient-vector references (they may run while the reduce-scatter is on the wire), 2: only
  // those with (after it); 0: all
  if ((s.em_mode == 1 && (rowf & EM_ITEM_GV)) || (s.em_mode == 2 && !(rowf & EM_ITEM_GV))) act = false;
  const int row = (int)(rowf & EM_ITEM_ROW);
  const int lo = act ? s.em_off[u] : 0, hi = act ? s.em_off[u + 1] : 0;
  // rows [n_local, n_local + n_rel) are the (replicated) relation table's: this rank's PARTIAL gradient — the sum of the gradient
  // vectors of its owned slots — is stored (not added: one writer per row and step) for the all-reduce and the relation update
  const bool is_rel = row >= s.n_local;
  float* wp = s.ent + (int64_t)row * STRIDE + j;
  const bool adagrad = s.optimizer == MKE_OPT_ADAGRAD;
  float* ap = adagrad ? s.ent_acc + (int64_t)row * STRIDE + j : nullptr;
  float w[FPL], acc[FPL], g[FPL];
#pragma unroll
  for (int k = 0; k < FPL; ++k) { w[k] = 0.f; acc[k] = 0.f; g[k] = 0.f; }
  if (act && !is_rel && !seg) {
#pragma unroll
    for (int k = 0; k < FPL; ++k) w[k] = wp[k * 16];
    if (adagrad) {
#pragma unroll
      for (int k = 0; k < FPL; ++k) acc[k] = ap[k * 16];
    }
  }
  float csum = 0.f;
  const int64_t C = s.capacity;
  for (int base = lo; __ballot(base < hi); base += 16) {     // wave-uniform trip count: the quarters walk their lists together
    // the quarter's next 16 references, one per lane; their coefficients gathered by the lanes that hold them
    uint32_t myloc = 0, mycidx = 0;
    const bool has = base + j < hi;
    if (has) {
      const uint2 r = *reinterpret_cast<const uint2*>(s.em_refs + 2 * (int64_t)(base + j));
      myloc = r.x; mycidx = r.y;
    }
    float mycoef = 0.f;
    if (has && !(myloc & EM_LOC_GV)) mycoef = s.em_coef[mycidx];
    const int n = min(16, hi - base);                        // <= 0 for a quarter that is done
    for (int t0 = 0; __ballot(t0 < n); t0 += U) {
      float V[U][FPL];
      float wgt[U];
      uint32_t loc[U];
#pragma unroll
      for (int x = 0; x < U; ++x) {
        loc[x] = (uint32_t)__shfl((int)myloc, qb + min(t0 + x, 15), 64);
        const bool live = t0 + x < n;
        if (live) {
          const uint32_t l = loc[x];
          const uint32_t chunk = (l >> 28) & 7u;
          const bool gvk = (l & EM_LOC_GV) != 0;
          const float* b0 = gvk ? s.em_gv[0] : s.em_v[0];
          if (chunk == 1) b0 = gvk ? s.em_gv[1] : s.em_v[1];
          if (chunk == 2) b0 = gvk ? s.em_gv[2] : s.em_v[2];
          if (chunk == 3) b0 = gvk ? s.em_gv[3] : s.em_v[3];
          const int64_t slot = (int64_t)(l & 0x7FFFFFu) + ((l & (1u << 23)) ? C : 0);
          const float* vp = b0 + (gvk ? 0 : (int64_t)((l >> 24) & 15u) * s.em_block_floats) + slot * STRIDE + j;
#pragma unroll
          for (int k = 0; k < FPL; ++k) V[x][k] = vp[k * 16];
        } else {
#pragma unroll
          for (int k = 0; k < FPL; ++k) V[x][k] = 0.f;
        }
      }
#pragma unroll
      for (int x = 0; x < U; ++x) {
        const float c = __shfl(mycoef, qb + min(t0 + x, 15), 64);
        const bool live = t0 + x < n;
        const bool rt = (loc[x] & (1u << 23)) != 0;
        const bool gvk = (loc[x] & EM_LOC_GV) != 0;
        // base vector: + coef sg V (sg = +1 with RT: d = c^ + RT; -1 with HR: d = HR - c^) and coef to the c^ term;
        // gradient vector: + gv for a head (HR = h^ + r^), - gv for a tail (RT = r^ - t^), + for a relation row
        wgt[x] = !live ? 0.f : (gvk ? ((rt && !(loc[x] & EM_LOC_PLUS)) ? -1.0f : 1.0f) : (rt ? c : -c));
        csum += (live && !gvk) ? c : 0.f;
      }
#pragma unroll
      for (int x = 0; x < U; ++x) {
#pragma unroll
        for (int k = 0; k < FPL; ++k) g[k] = fmaf(wgt[x], V[x][k], g[k]);
      }
    }
  }
  if (!act) return;
  if (seg) {               // a long row's segment: the partial gradient vector and coefficient sum (EM_PART floats per slot)
    float* pp = s.em_partials + ((int64_t)s.em_part[u] - s.em_part0) * (STRIDE + 16) + j;
#pragma unroll
    for (int k = 0; k < FPL; ++k) pp[k * 16] = g[k];
    pp[STRIDE] = csum;
    return;
  }
  if (is_rel) {
    float* gp = s.rel_grad + (int64_t)(row - s.n_local) * STRIDE + j;
#pragma unroll
    for (int k = 0; k < FPL; ++k) gp[k * 16] = g[k];
    return;
  }
  em_finish_row<FPL>(s, wp, ap, w, acc, g, csum);
}

// a quarter-wave per LONG row: its segments' partials added in segment order, then the row is finished as above
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_oc_em_combine(const mke_oc_step s) {
  constexpr int STRIDE = FPL * 16;
  constexpr int PS = STRIDE + 16;
  const int j = threadIdx.x & 15;
  const int64_t l = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  if (l >= s.em_n_long) return;
  const int row = s.em_long_rows[l];
  const int p0 = (int)(s.em_long_part0[l] - s.em_part0), p1 = (int)(s.em_long_part0[l + 1] - s.em_part0);
  const bool is_rel = row >= s.n_local;
  float* wp = s.ent + (int64_t)row * STRIDE + j;
  const bool adagrad = s.optimizer == MKE_OPT_ADAGRAD;
  float* ap = adagrad ? s.ent_acc + (int64_t)row * STRIDE + j : nullptr;
  float w[FPL], acc[FPL], g[FPL];
#pragma unroll
  for (int k = 0; k < FPL; ++k) { w[k] = 0.f; acc[k] = 0.f; g[k] = 0.f; }
  if (!is_rel) {
#pragma unroll
    for (int k = 0; k < FPL; ++k) w[k] = wp[k * 16];
    if (adagrad) {
#pragma unroll
      for (int k = 0; k < FPL; ++k) acc[k] = ap[k * 16];
    }
  }
  float csum = 0.f;
  for (int p = p0; p < p1; p += 2) {              // two partials in flight
    const bool two = p + 1 < p1;
    const float* a0 = s.em_partials + (int64_t)p * PS + j;
    const float* a1 = s.em_partials + (int64_t)(two ? p + 1 : p) * PS + j;
    float x0[FPL], x1[FPL];
#pragma unroll
    for (int k = 0; k < FPL; ++k) { x0[k] = a0[k * 16]; x1[k] = a1[k * 16]; }
    const float c0 = a0[STRIDE], c1 = a1[STRIDE];
#pragma unroll
    for (int k = 0; k < FPL; ++k) { g[k] += x0[k]; if (two) g[k] += x1[k]; }
    csum += c0;
    if (two) csum += c1;
  }
  if (is_rel) {
    float* gp = s.rel_grad + (int64_t)(row - s.n_local) * STRIDE + j;
#pragma unroll
    for (int k = 0; k < FPL; ++k) gp[k * 16] = g[k];
    return;
  }
  em_finish_row<FPL>(s, wp, ap, w, acc, g, csum);
}

}  // namespace mke

extern "C" int64_t mke_oc_em_plan_temp_bytes(int64_t capacity) {
  if (capacity < 0 || capacity >= 0x7FFFFFFFll) return -1;
  size_t a = 0, b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)(capacity + 1), 0, 64, (hipStream_t)0);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, (int)(capacity + 1), (hipStream_t)0);
  return (int64_t)(a > b ? a : b) + 256;
}

extern "C" int mke_oc_em_plan(const mke_oc_em_plan_args* args, void* stream) {
  using namespace mke;
  if (!args) { set_error("mke_oc_em_plan: NULL args"); return MKE_E_NULL; }
  const mke_oc_em_plan_args& a = *args;
  if (a.n_steps < 0 || a.n_all < 0 || a.neg_per_pos < 0 || a.neg_per_pos > 64 || a.chunks < 1 || a.chunks > MKE_OC_EM_MAX_CHUNKS ||
      a.n_ranks < 1 || a.n_ranks > MKE_OC_MAX_RANKS || a.rank < 0 || a.rank >= a.n_ranks || a.n_local < 1 || a.n_rel < 1 || a.max_step < 0) {
    set_error("mke_oc_em_plan: bad n_steps / n_all / neg_per_pos / chunks (1..%d) / ranks / n_local", MKE_OC_EM_MAX_CHUNKS);
    return MKE_E_SHAPE;
  }
  if (a.capacity < 1 || a.capacity >= 0x7FFFFFFFll) { set_error("mke_oc_em_plan: capacity must be in [1, 2^31)"); return MKE_E_RANGE; }
  if (!a.keys || !a.keys_alt || !a.vals_alt || !a.scratch8 || !a.wave_scratch || !a.refs || !a.rows || !a.off || !a.flags || !a.scan || !a.step_row0 || !a.n_refs || !a.temp ||
      !a.item_row || !a.item_off || !a.item_part || !a.long_row || !a.long_part0 || !a.step_item0 || !a.step_long0 || !a.step_part0) { set_error("mke_oc_em_plan: NULL output / scratch"); return MKE_E_NULL; }
  if (a.n_all > 0 && (!a.pos_h || !a.pos_r || !a.pos_t || !a.slot_h || !a.slot_t || !a.step_lo || (a.neg_per_pos > 0 && !a.codes))) { set_error("mke_oc_em_plan: NULL input"); return MKE_E_NULL; }
  if (a.temp_bytes < mke_oc_em_plan_temp_bytes(a.capacity)) { set_error("mke_oc_em_plan: temp storage below mke_oc_em_plan_temp_bytes"); return MKE_E_SHAPE; }
  if (a.max_step >= (1ll << 25) || a.max_step * (a.neg_per_pos + 1) > 0x7FFFFFFFll) { set_error("mke_oc_em_plan: a global step holds at most 2^25 positives"); return MKE_E_RANGE; }
  EmPlanParams pp;
  pp.a = a;
  pp.ep = (uint32_t)a.neg_per_pos + 5u;
  pp.ep_magic = ((1ull << 40) + pp.ep - 1) / pp.ep;
  pp.n_codes = a.n_all * (int64_t)a.neg_per_pos;
  pp.n_magic = a.neg_per_pos > 0 ? ((1ull << 40) + a.neg_per_pos - 1) / a.neg_per_pos : 0;
  pp.total = pp.n_codes + 5 * a.n_all;
  pp.g_shift = (a.n_ranks & (a.n_ranks - 1)) == 0 ? __builtin_ctz((unsigned)a.n_ranks) : -1;
  pp.rows_tot = a.n_local + a.n_rel;
  const int64_t total = a.n_all * (int64_t)pp.ep;
  int64_t nw = (total + 511) / 512;                       // >= 8 rounds of 64 elements per wavefront
  if (nw > MKE_OC_EM_WAVES) nw = MKE_OC_EM_WAVES;
  if (nw < 1) nw = 1;
  pp.n_waves = (int)nw;
  pp.per = ((total + nw - 1) / nw + 63) & ~63ll;
  pp.wave_cnt = a.wave_scratch;
  pp.wave_off = a.wave_scratch + (MKE_OC_EM_WAVES + 1);
  // + 1: an all-ones (step, row) never occurs, so the all-ones sentinel of the unused tail sorts last
  const int key_bits = bits_for((uint64_t)a.n_steps * (uint64_t)pp.rows_tot + 1ull);
  if (key_bits <= 32 && !g_oc_em_keys64) return em_plan_sorted<uint32_t>(pp, key_bits, (hipStream_t)stream);
  return em_plan_sorted<uint64_t>(pp, key_bits, (hipStream_t)stream);
}

extern "C" int mke_oc_pass2(const mke_oc_step* s, void* stream) {
  using namespace mke;
  if (!s) { set_error("mke_oc_pass2: NULL step"); return MKE_E_NULL; }
  if (!s->em_coef) { set_error("mke_oc_pass2: the step is not entity-major (em_coef == NULL)"); return MKE_E_UNSUPPORTED; }
  if (s->stride <= 0 || s->stride % 16 != 0 || s->dim <= 0 || s->dim > s->stride || s->stride > MKE_MAX_STRIDE) { set_error("mke_oc_pass2: bad stride/dim"); return MKE_E_SHAPE; }
  if (s->em_n_rows < 0 || s->em_chunks < 1 || s->em_chunks > MKE_OC_EM_MAX_CHUNKS || s->n_peers) { set_error("mke_oc_pass2: bad em_n_rows / em_chunks, or peer-direct"); return MKE_E_SHAPE; }
  if (s->em_n_rows == 0) return MKE_OK;
  if (!s->em_refs || !s->em_rows || !s->em_off || !s->ent || !s->rel_grad) { set_error("mke_oc_pass2: NULL reference lists / table"); return MKE_E_NULL; }
  for (int c = 0; c < s->em_chunks; ++c)
    if (!s->em_v[c] || !s->em_gv[c]) { set_error("mke_oc_pass2: NULL vector block of chunk %d", c); return MKE_E_NULL; }
  if (s->optimizer != MKE_OPT_ADAGRAD && s->optimizer != MKE_OPT_SGD) { set_error("mke_oc_pass2: Adagrad or SGD"); return MKE_E_UNSUPPORTED; }
  if (s->optimizer == MKE_OPT_ADAGRAD && !s->ent_acc) { set_error("mke_oc_pass2: Adagrad needs ent_acc"); return MKE_E_NULL; }
  const int fpl = s->stride / 16;
  const dim3 grid((unsigned)((s->em_n_rows + MKE_SUBS_PER_BLOCK - 1) / MKE_SUBS_PER_BLOCK));
  if (s->em_n_long < 0 || (s->em_n_long > 0 && (!s->em_part || !s->em_long_rows || !s->em_long_part0 || !s->em_partials))) { set_error("mke_oc_pass2: long rows without their lists / partial buffer"); return MKE_E_NULL; }
  MKE_DISPATCH_FPL(fpl, { hipLaunchKernelGGL((k_oc_em_pass2<FPL>), grid, dim3(MKE_BLOCK), 0, (hipStream_t)stream, *s); });
  int rc = check_launch("k_oc_em_pass2");
  if (rc || s->em_n_long == 0 || s->em_mode == 1) return rc;     // the long rows are combined after the LAST pass-2 launch of the step
  const dim3 lgrid((unsigned)((s->em_n_long + MKE_SUBS_PER_BLOCK - 1) / MKE_SUBS_PER_BLOCK));
  MKE_DISPATCH_FPL(fpl, { hipLaunchKernelGGL((k_oc_em_combine<FPL>), lgrid, dim3(MKE_BLOCK), 0, (hipStream_t)stream, *s); });
  return check_launch("k_oc_em_combine");
}
