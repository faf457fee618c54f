// mke_oc.hip — "owner computes" kernels of the entity-row sharded (multi-GPU) relation-view step (gfx950).
//
// New design (the reference has no multi-device code).  Entity rows are sharded id % G.  Moving rows to
// the triples costs one row in and one gradient row out per (corrupted) negative; here the NEGATIVES move to the rows:
// a negative of positive (h, r, t) differs from it in one entity c, and its score needs only c's row and one of two
// vectors of the positive,
//       corrupted head:  d = c^ + (r^ - t^) = c^ + RT_p          corrupted tail:  d = (h^ + r^) - c^ = HR_p - c^ ,
// so per global step every rank
//   (0) per EPOCH (table-independent): every rank packs the negatives of its own positives as (corrupt entity, side) codes
//                    (k_oc_pack_codes); one all-gather per epoch gives every rank the codes of all negatives;
//   (1) k_oc_bases : builds HR_p for the positives whose head it owns and RT_p for those whose tail it owns (relation
//                    table replicated: no row leaves its owner) — but ONLY the vector a positive's negatives need: the
//                    reference's sampler tosses one coin per ROUND (code/base/batch.py:97-105), so the negatives of a
//                    positive almost always corrupt the same side and ONE of the two vectors is enough (both only for the
//                    few positives whose re-draw rounds fell on the other side).  The group's need is carried in the top
//                    two bits of its first code (MKE_OC_NEED_HR / MKE_OC_NEED_RT), the slot of an unneeded vector is -1;
//   ... all-gather of the blocks: every rank now holds the needed vector(s) of ALL G x P positives ...
//   (2) k_oc_count : reference counts of its own rows over the whole global step (exclusive-row fast path; needs only the
//                    codes, so it runs while the all-gather is on the wire);
//   (3) k_oc_score : one wavefront per positive of the GLOBAL step scores the negatives whose corrupt entity THIS rank
//                    owns — the corrupt row is local: updated in place when referenced once, else scattered into the
//                    local gradient scratch — and writes its partial dL/dHR_p, dL/dRT_p into the slot the vector came
//                    from; the positive's own term d = HR_p - t^ (or h^ + RT_p when only RT_p travels) is one more such
//                    term, scored by the owner of t (of h) against its local row;
//   ... reduce-scatter of the gradient vectors: the owner of h_p receives sum dL/dHR_p, the owner of t_p sum dL/dRT_p ...
//   (4) k_oc_apply : adds them to the head / tail rows' gradient (local) and to the replicated relation gradient;
//   ... all-reduce of the relation gradient; mke_rows_update_multi on the shard + the relation table (every row is
//       updated once per step from the sum of all its contributions: dense-Adagrad-equivalent).
// Per rank and step that is ~1 vector out and ~1 gradient vector back per positive instead of N rows + N gradient rows:
// (G-1)/G * 2 * ~1 * P * stride * 4 bytes against (G-1)/G * 2 * P * (N + 2) * stride * 4 — 27x less at N = 25, and no
// row-set construction, id exchange or remap.  Same arithmetic as the fused single-GPU kernel (mke_score.hip).
#include "mke_common.h"

namespace mke {


struct OcParams {
  mke_oc_step s;
  float* send;          // k_oc_bases: this rank's block
  const float* v_all;   // [G][block_floats]
  int64_t block_floats; // 2 C stride
  float* g_all;         // [G][2 C stride]
  const float* gv;      // [2 C stride]
  double* lossp;
  int count_blocks;     // k_oc_bases: the first count_blocks blocks of the launch count the step's references instead
};

// codes of the negatives of home rank g's positives of this part: [n_mine_g][neg_per_pos]
__device__ __forceinline__ const int32_t* oc_codes(const OcParams& p, int g) { return p.s.codes + p.s.code_off[g]; }

// id / G and id % G for the (non-negative) entity ids: a shift and a mask when the world size is a power of two (2, 4, 8 GPUs),
// else an UNSIGNED 32-bit division.  The plain `int / int` of a run-time divisor is ~30 vector instructions on this part and the
// 64-bit `i / per` ~150; SQ counters of k_oc_score_q as rank 0 of 8 showed the launch bound by instruction issue (1,660 vector
// instructions per wavefront x 10 wavefronts per SIMD: EXPERIMENTS R5.26), a quarter of them these divisions.
struct OcDiv { uint32_t g; int shift; };
__device__ __forceinline__ OcDiv oc_divisor(int g) { OcDiv d; d.g = (uint32_t)g; d.shift = (g & (g - 1)) == 0 ? __builtin_ctz((unsigned)g) : -1; return d; }
__device__ __forceinline__ int oc_div(const OcDiv& d, int x) { return d.shift >= 0 ? (int)((uint32_t)x >> d.shift) : (int)((uint32_t)x / d.g); }
__device__ __forceinline__ int oc_mod(const OcDiv& d, int x) { return d.shift >= 0 ? (int)((uint32_t)x & (d.g - 1u)) : (int)((uint32_t)x % d.g); }
// the score kernels take the power-of-two case as a template parameter (with the run-time test the compiler computes BOTH forms and
// selects: nothing saved)
template <bool P2> struct OcDivP { uint32_t g; int shift; };
template <bool P2> __device__ __forceinline__ OcDivP<P2> oc_divisor_p(int g) { OcDivP<P2> d; d.g = (uint32_t)g; d.shift = P2 ? __builtin_ctz((unsigned)g) : 0; return d; }
template <bool P2> __device__ __forceinline__ int oc_div(const OcDivP<P2>& d, int x) {
  if constexpr (P2) return (int)((uint32_t)x >> d.shift); else return (int)((uint32_t)x / d.g);
}
template <bool P2> __device__ __forceinline__ int oc_mod(const OcDivP<P2>& d, int x) {
  if constexpr (P2) return (int)((uint32_t)x & (d.g - 1u)); else return (int)((uint32_t)x % d.g);
}
// home rank of global-step position i (i < n_pos < 2^31, checked on the host): 32-bit
__device__ __forceinline__ int oc_home(const mke_oc_step& s, int64_t i) { return (int)((uint32_t)i / (uint32_t)s.per); }

// Hub rows of the shard (mke_oc_step.hot; the fused kernel's mke_hot_rows on this rank's rows): an entity that is head or tail of
// many positives of EVERY global step receives that many same-address atomic row adds from k_oc_apply and from the positives'
// own terms (measured as rank 0 of 8 on Zipf(1.0) triples: apply 7.4 -> 60 us, score 50 -> 76, the counting 13 -> 39).  Their
// contributions go to one of `copies` private rows behind the shard's own rows in the gradient scratch (the update launch
// adds the copies when it visits the row), they are not reference-counted (a hub row is never finished in place).
__device__ __forceinline__ bool oc_is_hot(const mke_oc_step& s, int row) { return s.hot.slot && s.hot.slot[row] >= 0; }
__device__ __forceinline__ int64_t oc_grad_row(const mke_oc_step& s, int row, int64_t k) {
  if (!s.hot.slot) return row;
  const int hs = s.hot.slot[row];
  return hs >= 0 ? s.hot.row0 + (k % s.hot.copies) * s.hot.n_hot + hs : (int64_t)row;
}

// (corrupt entity << 1) | corrupted-head, one per negative; a negative equal to its positive counts as a corrupted tail.
// Bits 30 / 31 of a group's FIRST code carry its need flags (entity ids stay below 2^29).
#define OC_CODE_MASK 0x3FFFFFFF
__device__ __forceinline__ int oc_code(int32_t c) { return c & OC_CODE_MASK; }
// GS lanes per positive (16 / 32 / 64 >= neg_per_pos), lane = slot: the reads of the sampler's output are coalesced and the group's
// need flags — MKE_OC_NEED_RT when any negative corrupts the head, MKE_OC_NEED_HR when any corrupts the tail or none corrupts
// the head (the positive's own term needs one of the two) — are two ballots; slot 0 stores them with its code.  (As two kernels,
// pack then a thread per positive marking its group, the marking alone took 569 us per epoch share at the C5 shape.)
template <int GS>
__global__ __launch_bounds__(MKE_BLOCK) void k_oc_pack_codes(const int32_t* __restrict__ pos_h, const int32_t* __restrict__ neg_h,
                                                             const int32_t* __restrict__ neg_t, int64_t n_pos, int neg_per_pos,
                                                             int32_t* __restrict__ codes) {
  const int lane = threadIdx.x & 63, gl = lane & (GS - 1), gbase = lane & ~(GS - 1);
  const int64_t g0 = (((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 6) * (64 / GS) + lane / GS;
  const int64_t gstep = (((int64_t)gridDim.x * MKE_BLOCK) >> 6) * (64 / GS);
  const int64_t iters = (n_pos + gstep - 1) / gstep;                 // wave-uniform trip count (ballots inside)
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t i = g0 + it * gstep;
    const bool has = i < n_pos && gl < neg_per_pos;
    uint32_t code = 0;
    if (has) {
      const int64_t e = i * neg_per_pos + gl;
      const int nh = neg_h[e], nt = neg_t[e];
      code = nh != pos_h[i] ? (((uint32_t)nh << 1) | 1u) : ((uint32_t)nt << 1);
    }
    const uint64_t gm = GS == 64 ? ~0ull : (((1ull << (GS & 63)) - 1ull) << gbase);
    const bool any_h = (__ballot(has && (code & 1u)) & gm) != 0, any_t = (__ballot(has && !(code & 1u)) & gm) != 0;
    if (has) {
      if (gl == 0) code |= (any_h ? MKE_OC_NEED_RT : 0u) | ((any_t || !any_h) ? MKE_OC_NEED_HR : 0u);
      codes[i * neg_per_pos + gl] = (int32_t)code;
    }
  }
}

// reference counts of the rows this rank owns over the whole global step: the negatives' codes of every home rank + the
// heads / tails of the step's positives that it owns (each is referenced once more: by mke_oc_apply or by the positive's own
// term in mke_oc_score).  Block `block` of `n_blocks` (its own launch, or rider blocks of k_oc_bases: the counts need
// only the epoch's codes, nothing of this step's)
__device__ __forceinline__ void oc_count_range(const OcParams& p, int block, int n_blocks) {
  const mke_oc_step& s = p.s;
  const int64_t n_codes = s.n_pos * s.neg_per_pos;
  const int64_t total = n_codes + 2 * s.n_pos;
  const OcDiv dv = oc_divisor(s.n_ranks);
  for (int64_t e = (int64_t)block * MKE_BLOCK + threadIdx.x; e < total; e += (int64_t)n_blocks * MKE_BLOCK) {
    if (e < n_codes) {
      const int64_t i = n_codes <= 0xFFFFFFFFll ? (int64_t)((uint32_t)e / (uint32_t)s.neg_per_pos) : e / s.neg_per_pos;   // uniform choice
      const int g = oc_home(s, i);
      const int c = oc_code(oc_codes(p, g)[(i - (int64_t)g * s.per) * s.neg_per_pos + (e - i * s.neg_per_pos)]) >> 1;
      if (oc_mod(dv, c) == s.rank) atomicAdd(&s.ref_count[oc_div(dv, c)], 1);
    } else {
      const int64_t k = e - n_codes;
      const int ent = k < s.n_pos ? s.pos_h[k] : s.pos_t[k - s.n_pos];
      if (oc_mod(dv, ent) == s.rank && !oc_is_hot(s, oc_div(dv, ent))) atomicAdd(&s.ref_count[oc_div(dv, ent)], 1);
    }
  }
}

// quarter-wave per owned slot (HR slots, then RT slots)
template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_oc_bases(const OcParams p) {
  if ((int)blockIdx.x < p.count_blocks) {  // block-uniform
    oc_count_range(p, blockIdx.x, p.count_blocks);
    return;
  }
  const mke_oc_step& s = p.s;
  const int j = threadIdx.x & 15;
  const int64_t sub = (((int64_t)blockIdx.x - p.count_blocks) * MKE_BLOCK + threadIdx.x) >> 4;
  if (sub >= s.n_own_h + s.n_own_t) return;
  const bool is_h = sub < s.n_own_h;
  const int64_t k = is_h ? sub : sub - s.n_own_h;
  const int32_t pos = is_h ? s.own_h[k] : s.own_t[k];
  const int e = is_h ? s.pos_h[pos] : s.pos_t[pos];
  float E[FPL], R[FPL];
  load_row<FPL>(s.ent, oc_div(oc_divisor(s.n_ranks), e), s.stride, j, E);
  load_row<FPL>(s.rel, s.pos_r[pos], s.stride, j, R);
  l2_normalize_row<FPL>(E, true);
  l2_normalize_row<FPL>(R, true);
  float* o = p.send + ((is_h ? 0 : s.capacity) + k) * (int64_t)s.stride + j;
#pragma unroll
  for (int q = 0; q < FPL; ++q) o[q * 16] = is_h ? E[q] + R[q] : R[q] - E[q];
}

__global__ __launch_bounds__(MKE_BLOCK) void k_oc_count(const OcParams p) { oc_count_range(p, blockIdx.x, gridDim.x); }

// The positive's own term, scored by the owner of the entity the travelling vector lacks (a quarter-wave): with HR_p on the
// wire the owner of t forms d = HR_p - t^, with only RT_p the owner of h forms d = h^ + RT_p; loss and coefficient of a
// POSITIVE (code/losses.py:4-12), c d added to the vector's gradient, -+ c d scattered to the local row (never in place:
// the row is also referenced by nothing else only by accident, and the update launch visits it anyway).
template <int FPL, bool EM = false>
__device__ __forceinline__ float oc_positive_term(const mke_oc_step& s, int STRIDE, int j, bool use_hr, int ent_local, float pw, int64_t i,
                                                  const float (&HR)[FPL], const float (&RT)[FPL], float (&gHR)[FPL], float (&gRT)[FPL]) {
  float d[FPL];
  load_row<FPL>(s.ent, ent_local, STRIDE, j, d);
  l2_normalize_row<FPL>(d, true);
  float x = 0.f;
  const float sg = use_hr ? -1.0f : 1.0f;
#pragma unroll
  for (int k = 0; k < FPL; ++k) {
    const float v = use_hr ? HR[k] : RT[k];
    d[k] = fmaf(sg, d[k], v);
    x = fmaf(d[k], d[k], x);
  }
  x = sub16_sum(x);
  const float c = 2.0f * s.scale * pw * sigmoid_f(x);
  const float ch = use_hr ? c : 0.0f, ct = use_hr ? 0.0f : c;
#pragma unroll
  for (int k = 0; k < FPL; ++k) {
    gHR[k] = fmaf(ch, d[k], gHR[k]);
    gRT[k] = fmaf(ct, d[k], gRT[k]);
  }
  if constexpr (EM) {   // entity-major: the row's owner-side pass re-forms sg c d = c (e^ + sg V) from this one scalar
    if (j == 0) s.em_coef[(s.em_pos0 + i) * (s.neg_per_pos + 1) + s.neg_per_pos] = c;
  } else {
    // the row's own gradient is sg * c * d: the scale rides on the sign argument (no second scaled copy of d)
    atomic_add_row<FPL>(s.ent_grad, oc_grad_row(s, ent_local, i), STRIDE, s.dim, j, d, sg * c);
    if (j == 0) s.ent_touched[ent_local] = s.tag;
  }
  return pw * softplus_f(x);
}

// One wavefront per positive of the global step.  Lane l holds the code of negative l (neg_per_pos <= 64); the negatives
// this rank owns are dealt round-robin to the four quarter-waves (the (4 round + q)-th set bit of the ballot), U of them
// in flight per quarter.
template <int FPL, int U, bool P2, bool EM = false>
__global__ __launch_bounds__(MKE_BLOCK) void k_oc_score(const OcParams p) {
  constexpr int STRIDE = FPL * 16;   // == s.stride (the dispatch picks FPL from it): row offsets by shift-add, not a 64-bit multiply
  const mke_oc_step& s = p.s;
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int64_t wave0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * MKE_BLOCK) >> 6;
  const int G = s.n_ranks, N = s.neg_per_pos;
  const OcDivP<P2> dv = oc_divisor_p<P2>(G);
  const int64_t C = s.capacity;
  float loss = 0.f;
  for (int64_t i = wave0; i < s.n_pos; i += nwaves) {
    const int ph = s.pos_h[i], pt = s.pos_t[i];
    const int home = oc_home(s, i);
    const int sh = s.slot_h[i], st = s.slot_t[i];   // -1: that vector does not travel (no negative of this positive needs it)
    // the vectors' home: this rank's all-gathered copy, or (peer-direct) the owner's own send block over xGMI
    const float* vh = (s.n_peers ? s.peer_v[oc_mod(dv, ph)] : p.v_all + (int64_t)oc_mod(dv, ph) * p.block_floats) + (int64_t)max(sh, 0) * STRIDE;
    const float* vt = (s.n_peers ? s.peer_v[oc_mod(dv, pt)] : p.v_all + (int64_t)oc_mod(dv, pt) * p.block_floats) + (C + max(st, 0)) * STRIDE;
    // codes first (one negative per lane), then the owned rows' reference counts together with the positive's vector(s): a
    // round below is one round trip, and the accumulator row is gathered only for rows that are finished in place
    int code = 0;
    if (lane < N) code = oc_code(oc_codes(p, home)[(i - (int64_t)home * s.per) * N + lane]);
    const bool mine = lane < N && oc_mod(dv, code >> 1) == s.rank;
    int rcl = 0;
    if constexpr (!EM) rcl = (mine && s.ref_count) ? (oc_is_hot(s, oc_div(dv, code >> 1)) ? 2 : s.ref_count[oc_div(dv, code >> 1)]) : 0;   // a hub row is never finished in place
    float HR[FPL], RT[FPL], gHR[FPL], gRT[FPL];
#pragma unroll
    for (int k = 0; k < FPL; ++k) HR[k] = RT[k] = gHR[k] = gRT[k] = 0.f;
    if (sh >= 0) load_row<FPL>(vh, 0, STRIDE, j, HR);
    if (st >= 0) load_row<FPL>(vt, 0, STRIDE, j, RT);
    const uint64_t mask = __ballot(mine);
    const int total = __popcll(mask);
    float* const coefp = EM ? s.em_coef + (s.em_pos0 + i) * (N + 1) : nullptr;   // this positive's coefficients (entity-major)

    // the positive itself: with HR on the wire the owner of t scores it, else the owner of h (wave-uniform test)
    if (oc_mod(dv, sh >= 0 ? pt : ph) == s.rank && q == 0) {
      const float pw = s.pos_w ? s.pos_w[i] : 1.0f;   // weighted positives: code/losses.py:44-50
      loss += oc_positive_term<FPL, EM>(s, STRIDE, j, sh >= 0, oc_div(dv, sh >= 0 ? pt : ph), pw, i, HR, RT, gHR, gRT);
    }

    // quarter q takes the q-th, (q+4)-th, ... set bit of the ballot: a running copy of the mask with the bits already
    // dealt removed (4 bit-clears per negative instead of a scan from bit 0)
    uint64_t rest = mask;
    for (int k = 0; k < q; ++k) rest &= rest - 1;
    for (int base = 0; base < total; base += 4 * U) {
      int e[U], cnt[U], nidx[U];
      bool live[U], sideH[U];
      float Cr[U][FPL], A[U][FPL];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        live[u] = rest != 0;
        const int src = live[u] ? __builtin_ctzll(rest) : 0;
        const int cd = __shfl(code, src, 64);
        cnt[u] = EM ? 0 : __shfl(rcl, src, 64);
        nidx[u] = src;
        sideH[u] = cd & 1;
        e[u] = oc_div(dv, cd >> 1);
        rest &= rest - 1; rest &= rest - 1; rest &= rest - 1; rest &= rest - 1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (live[u]) {
          load_row<FPL>(s.ent, e[u], STRIDE, j, Cr[u]);
          if constexpr (!EM) { if (s.ref_count && s.ent_acc && cnt[u] == 1) load_row<FPL>(s.ent_acc, e[u], STRIDE, j, A[u]); }
        } else {
#pragma unroll
          for (int k = 0; k < FPL; ++k) Cr[u][k] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < FPL; ++k) ss = fmaf(Cr[u][k], Cr[u][k], ss);
        const float cinv = rsqrtf(fmaxf(sub16_sum(ss), MKE_L2_EPS));
        float d[FPL];
        float y = 0.f;
        const float sc = sideH[u] ? cinv : -cinv;
#pragma unroll
        for (int k = 0; k < FPL; ++k) {
          d[k] = fmaf(sc, Cr[u][k], sideH[u] ? RT[k] : HR[k]);
          y = fmaf(d[k], d[k], y);
        }
        y = sub16_sum(y);
        const float t_ = __expf(-y);
        const float s1 = 1.0f + t_;
        loss += __logf(s1);
        const float c = -2.0f * s.scale * t_ * __builtin_amdgcn_rcpf(s1);
        const float cHR = sideH[u] ? 0.f : c, cRT = sideH[u] ? c : 0.f;
        if constexpr (EM) {   // entity-major: one scalar out per (positive, negative); the row's owner-side pass does the rest
#pragma unroll
          for (int k = 0; k < FPL; ++k) {
            gHR[k] = fmaf(cHR, d[k], gHR[k]);
            gRT[k] = fmaf(cRT, d[k], gRT[k]);
          }
          if (j == 0) coefp[nidx[u]] = c;
          continue;
        }
#pragma unroll
        for (int k = 0; k < FPL; ++k) {
          gHR[k] = fmaf(cHR, d[k], gHR[k]);
          gRT[k] = fmaf(cRT, d[k], gRT[k]);
          d[k] *= c;
        }
        const float sg = sideH[u] ? 1.0f : -1.0f;
        if (s.ref_count && cnt[u] == 1) {  // the only reference to this row in the whole global step: finish it here
          float dot = 0.f;
#pragma unroll
          for (int k = 0; k < FPL; ++k) dot = fmaf(Cr[u][k], d[k], dot);
          dot = sub16_sum(dot) * (sg * cinv);
          const float a1 = sg * cinv;
          const float a2 = cinv < 0.99e6f ? -dot * cinv * cinv : 0.f;
          float g[FPL];
#pragma unroll
          for (int k = 0; k < FPL; ++k) g[k] = fmaf(a2, Cr[u][k], a1 * d[k]);
          float* wp = s.ent + (int64_t)e[u] * STRIDE + j;
          if (s.optimizer == MKE_OPT_ADAGRAD) {
            float* ap = s.ent_acc + (int64_t)e[u] * STRIDE + j;
#pragma unroll
            for (int k = 0; k < FPL; ++k) {
              const float a = fmaf(g[k], g[k], A[u][k]);
              ap[k * 16] = a;
              wp[k * 16] = Cr[u][k] - s.lr * g[k] * adagrad_scale(a);
            }
          } else {
#pragma unroll
            for (int k = 0; k < FPL; ++k) wp[k * 16] = Cr[u][k] - s.lr * g[k];
          }
          if (j == 0) s.ref_count[e[u]] = 0;
        } else {
          atomic_add_row<FPL>(s.ent_grad, e[u], STRIDE, s.dim, j, d, sg);
          if (j == 0) s.ent_touched[e[u]] = s.tag;
        }
      }
    }

    // partial gradient vectors of this positive (zero when this rank owns none of its negatives): quarter 0 -> HR slot,
    // quarter 1 -> RT slot.  Every slot of g_all is written by exactly one wavefront per step: no atomics, no clearing.
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      gHR[k] += __shfl_xor(gHR[k], 16, 64); gHR[k] += __shfl_xor(gHR[k], 32, 64);
      gRT[k] += __shfl_xor(gRT[k], 16, 64); gRT[k] += __shfl_xor(gRT[k], 32, 64);
    }
    if (q < 2 && (q == 0 ? sh : st) >= 0) {
      const int64_t gb = 2 * C * (int64_t)STRIDE;
      const int own = oc_mod(dv, q == 0 ? ph : pt);
      float* o = (s.n_peers ? s.peer_g[own] : p.g_all + (int64_t)own * gb) +
                 (q == 0 ? (int64_t)sh : C + st) * STRIDE + j;
#pragma unroll
      for (int k = 0; k < FPL; ++k) o[k * 16] = q == 0 ? gHR[k] : gRT[k];
    }
  }
  const double tot = block_sum_double(j == 0 ? loss : 0.f);
  if (threadIdx.x == 0) p.lossp[blockIdx.x] = tot * (double)s.scale;
}

// The same step with a QUARTER-wave per positive (four positives per wavefront), for the multi-rank shapes: at G ranks a rank owns
// ~N / G of a positive's negatives (3 of 25 at G = 8), so k_oc_score's wavefront-per-positive spends its instructions on the per-
// positive overhead (ids, codes, two vector loads, two butterfly reductions, two slot writes) with most quarter-wave slots of its
// single round idle — measured at the C2 shape as rank 0 of 8: 56.7 us for 40,000 positives against 39.0 us for 5,000 positives
// with all 25 negatives each (profiles/r04_oc_rank_compute.md).  Here a quarter owns its positive outright: its 16 lanes fetch the
// codes 16 at a time, the owned ones are visited one after the other (the four quarters of a wavefront iterate together until the
// busiest is done), the partial gradient vectors need no cross-quarter reduction.  Same arithmetic, same slots, same in-place /
// scatter rule per corrupt row as k_oc_score.
template <int FPL, bool P2, bool EM = false>
__global__ __launch_bounds__(MKE_BLOCK) void k_oc_score_q(const OcParams p) {
  constexpr int STRIDE = FPL * 16;
  const mke_oc_step& s = p.s;
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;   // quarter-wave index
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  const int G = s.n_ranks, N = s.neg_per_pos;
  const OcDivP<P2> dv = oc_divisor_p<P2>(G);
  const int64_t C = s.capacity;
  float loss = 0.f;
  const int64_t iters = (s.n_pos + nsub - 1) / nsub;          // wave-uniform trip count (ballots inside)
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t i_raw = sub0 + it * nsub;
    const bool act = i_raw < s.n_pos;
    const int64_t i = act ? i_raw : 0;
    const int ph = s.pos_h[i], pt = s.pos_t[i];
    const int home = oc_home(s, i);
    const int sh = act ? s.slot_h[i] : -1, st = act ? s.slot_t[i] : -1;   // -1: that vector does not travel
    const float* vh = (s.n_peers ? s.peer_v[oc_mod(dv, ph)] : p.v_all + (int64_t)oc_mod(dv, ph) * p.block_floats) + (int64_t)max(sh, 0) * STRIDE;
    const float* vt = (s.n_peers ? s.peer_v[oc_mod(dv, pt)] : p.v_all + (int64_t)oc_mod(dv, pt) * p.block_floats) + (C + max(st, 0)) * STRIDE;
    float HR[FPL], RT[FPL], gHR[FPL], gRT[FPL];
#pragma unroll
    for (int k = 0; k < FPL; ++k) HR[k] = RT[k] = gHR[k] = gRT[k] = 0.f;
    if (sh >= 0) load_row<FPL>(vh, 0, STRIDE, j, HR);
    if (st >= 0) load_row<FPL>(vt, 0, STRIDE, j, RT);
    // the positive itself: with HR on the wire the owner of t scores it, else the owner of h
    if (act && oc_mod(dv, sh >= 0 ? pt : ph) == s.rank) {
      const float pw = s.pos_w ? s.pos_w[i] : 1.0f;
      loss += oc_positive_term<FPL, EM>(s, STRIDE, j, sh >= 0, oc_div(dv, sh >= 0 ? pt : ph), pw, i, HR, RT, gHR, gRT);
    }
    const int32_t* cp = oc_codes(p, home) + (i - (int64_t)home * s.per) * N;
    float* const coefp = EM ? s.em_coef + (s.em_pos0 + i) * (N + 1) : nullptr;   // this positive's coefficients (entity-major)
    for (int c0 = 0; c0 < N; c0 += 16) {                      // the group's codes, 16 per quarter at a time
      int code = 0;
      const bool has = act && c0 + j < N;
      if (has) code = oc_code(cp[c0 + j]);
      const bool mine = has && oc_mod(dv, code >> 1) == s.rank;
      int rcl = 0;
      if constexpr (!EM) rcl = (mine && s.ref_count) ? (oc_is_hot(s, oc_div(dv, code >> 1)) ? 2 : s.ref_count[oc_div(dv, code >> 1)]) : 0;
      const uint64_t mall = __ballot(mine);
      unsigned rest = (unsigned)(mall >> (16 * q)) & 0xFFFFu;  // this quarter's owned negatives of the chunk
      while (__ballot(rest != 0)) {                            // the four quarters visit their next owned negative together
        const bool live = rest != 0;
        const int bit = live ? __builtin_ctz(rest) : 0;
        const int src = 16 * q + bit;
        const int cd = __shfl(code, src, 64);
        const int cnt = EM ? 0 : __shfl(rcl, src, 64);
        rest &= rest - 1;
        if (!live) continue;
        const bool sideH = cd & 1;
        const int e = oc_div(dv, cd >> 1);
        float Cr[FPL], A[FPL];
        load_row<FPL>(s.ent, e, STRIDE, j, Cr);
        const bool in_place = !EM && s.ref_count && cnt == 1;
        if (in_place && s.ent_acc) load_row<FPL>(s.ent_acc, e, STRIDE, j, A);
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < FPL; ++k) ss = fmaf(Cr[k], Cr[k], ss);
        const float cinv = rsqrtf(fmaxf(sub16_sum(ss), MKE_L2_EPS));
        float d[FPL];
        float y = 0.f;
        const float sc = sideH ? cinv : -cinv;
#pragma unroll
        for (int k = 0; k < FPL; ++k) {
          d[k] = fmaf(sc, Cr[k], sideH ? RT[k] : HR[k]);
          y = fmaf(d[k], d[k], y);
        }
        y = sub16_sum(y);
        const float t_ = __expf(-y);
        const float s1 = 1.0f + t_;
        loss += __logf(s1);
        const float c = -2.0f * s.scale * t_ * __builtin_amdgcn_rcpf(s1);
        const float cHR = sideH ? 0.f : c, cRT = sideH ? c : 0.f;
        if constexpr (EM) {   // entity-major: one scalar out per (positive, negative)
#pragma unroll
          for (int k = 0; k < FPL; ++k) {
            gHR[k] = fmaf(cHR, d[k], gHR[k]);
            gRT[k] = fmaf(cRT, d[k], gRT[k]);
          }
          if (j == 0) coefp[c0 + bit] = c;
          continue;
        }
#pragma unroll
        for (int k = 0; k < FPL; ++k) {
          gHR[k] = fmaf(cHR, d[k], gHR[k]);
          gRT[k] = fmaf(cRT, d[k], gRT[k]);
          d[k] *= c;
        }
        const float sg = sideH ? 1.0f : -1.0f;
        if (in_place) {  // the only reference to this row in the whole global step: finish it here
          float dot = 0.f;
#pragma unroll
          for (int k = 0; k < FPL; ++k) dot = fmaf(Cr[k], d[k], dot);
          dot = sub16_sum(dot) * (sg * cinv);
          const float a1 = sg * cinv;
          const float a2 = cinv < 0.99e6f ? -dot * cinv * cinv : 0.f;
          float g[FPL];
#pragma unroll
          for (int k = 0; k < FPL; ++k) g[k] = fmaf(a2, Cr[k], a1 * d[k]);
          float* wp = s.ent + (int64_t)e * STRIDE + j;
          if (s.optimizer == MKE_OPT_ADAGRAD) {
            float* ap = s.ent_acc + (int64_t)e * STRIDE + j;
#pragma unroll
            for (int k = 0; k < FPL; ++k) {
              const float a = fmaf(g[k], g[k], A[k]);
              ap[k * 16] = a;
              wp[k * 16] = Cr[k] - s.lr * g[k] * adagrad_scale(a);
            }
          } else {
#pragma unroll
            for (int k = 0; k < FPL; ++k) wp[k * 16] = Cr[k] - s.lr * g[k];
          }
          if (j == 0) s.ref_count[e] = 0;
        } else {
          atomic_add_row<FPL>(s.ent_grad, e, STRIDE, s.dim, j, d, sg);
          if (j == 0) s.ent_touched[e] = s.tag;
        }
      }
    }
    {   // this positive's partial gradient vector(s): every live slot of g_all is written by exactly one quarter-wave per step
      const int64_t gb = 2 * C * (int64_t)STRIDE;
      float* oh = (s.n_peers ? s.peer_g[oc_mod(dv, ph)] : p.g_all + (int64_t)oc_mod(dv, ph) * gb) + (int64_t)max(sh, 0) * STRIDE + j;
      float* ot = (s.n_peers ? s.peer_g[oc_mod(dv, pt)] : p.g_all + (int64_t)oc_mod(dv, pt) * gb) + (C + max(st, 0)) * STRIDE + j;
      if (sh >= 0) {
#pragma unroll
        for (int k = 0; k < FPL; ++k) oh[k * 16] = gHR[k];
      }
      if (st >= 0) {
#pragma unroll
        for (int k = 0; k < FPL; ++k) ot[k * 16] = gRT[k];
      }
    }
  }
  const double tot = block_sum_double(j == 0 ? loss : 0.f);
  if (threadIdx.x == 0) p.lossp[blockIdx.x] = tot * (double)s.scale;
}

// quarter-wave per owned slot: the summed gradient vector goes to the head (+) / tail (-) row's gradient and to the
// relation row's
template <int FPL, bool ENT = true>   // ENT false (entity-major step): the relation rows only — the head / tail rows take gv in mke_oc_pass2
__global__ __launch_bounds__(MKE_BLOCK) void k_oc_apply(const OcParams p) {
  const mke_oc_step& s = p.s;
  const int j = threadIdx.x & 15;
  const int64_t sub = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  if (sub >= s.n_own_h + s.n_own_t) return;
  const bool is_h = sub < s.n_own_h;
  const int64_t k = is_h ? sub : sub - s.n_own_h;
  const int32_t pos = is_h ? s.own_h[k] : s.own_t[k];
  const int row = ENT ? oc_div(oc_divisor(s.n_ranks), is_h ? s.pos_h[pos] : s.pos_t[pos]) : 0;
  const int r = s.pos_r[pos];
  float v[FPL];
  load_row<FPL>(p.gv, (is_h ? 0 : s.capacity) + k, s.stride, j, v);
  if (s.n_peers) {  // peer-direct: gv is this rank's inbox [n_ranks][2 C][stride]; sum the writers' slices in rank order
    const int64_t gb = 2 * s.capacity * (int64_t)s.stride;
    for (int r = 1; r < s.n_ranks; ++r) {
      float w[FPL];
      load_row<FPL>(p.gv + r * gb, (is_h ? 0 : s.capacity) + k, s.stride, j, w);
#pragma unroll
      for (int c = 0; c < FPL; ++c) v[c] += w[c];
    }
  }
  if constexpr (ENT) atomic_add_row<FPL>(s.ent_grad, oc_grad_row(s, row, sub), s.stride, s.dim, j, v, is_h ? 1.0f : -1.0f);
  float* grel = s.rel_grad + (sub % s.rel_grad_copies) * (s.n_rel * (int64_t)s.stride);
  atomic_add_row<FPL>(grel, r, s.stride, s.dim, j, v, 1.0f);
  if (j == 0) {
    if constexpr (ENT) s.ent_touched[row] = s.tag;
    s.rel_touched[r] = s.tag;
  }
}

// ---- per-epoch plan: the slot of every positive's HR / RT vector in its owner's block -------------------------------------
// Part k of the epoch = epoch positions [part_lo[k], part_lo[k + 1]).  Inside a part, the positives whose head (tail) is
// owned by rank g = id % G AND whose negatives need HR (RT) — the flag bits of the group's first code, codes laid out by
// epoch position; a positive without negatives needs HR — get slots 0, 1, 2, ... of g's block in epoch order (a stable
// counting sort by owner with G buckets), every other positive slot -1; rank `rank`'s own positives are also listed, as positions inside the part, in slot order, at
// own_*[part_lo[k] + slot] (a part's list starts at the part's own offset: no prefix over parts is needed), and the
// per-(part, owner) counts go out for the host (block capacity, list lengths).  One block of 1024 threads per (part, h | t):
// a chunk of 1024 positives is ranked by a ballot per distinct owner in each wavefront, the 16 wavefronts' counts meet in LDS.
// Replaces two argsorts + bincount + cumsum + scatter of the whole epoch in torch (2.3 ms per epoch at the C2 shape,
// 12 us per step of 184; at 8 ranks an epoch is 23 global steps and the same plan cost 70 us per step).
#define OC_PLAN_THREADS 1024
__global__ __launch_bounds__(OC_PLAN_THREADS) void k_oc_plan(const int32_t* __restrict__ pos_h, const int32_t* __restrict__ pos_t,
                                                             const int32_t* __restrict__ codes, int neg_per_pos,
                                                             const int64_t* __restrict__ part_lo, int n_parts, int G, int rank,
                                                             int32_t* __restrict__ slot_h, int32_t* __restrict__ slot_t,
                                                             int32_t* __restrict__ own_h, int32_t* __restrict__ own_t,
                                                             int32_t* __restrict__ counts) {
  constexpr int NWV = OC_PLAN_THREADS / 64;
  __shared__ int s_cnt[MKE_OC_MAX_RANKS];            // owners' running counts inside this part
  __shared__ int s_wcnt[NWV][MKE_OC_MAX_RANKS];      // this chunk: per wavefront and owner
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int x = blockIdx.y;                          // 0: heads, 1: tails
  const int32_t* __restrict__ ids = x ? pos_t : pos_h;
  int32_t* __restrict__ slot = x ? slot_t : slot_h;
  int32_t* __restrict__ own = x ? own_t : own_h;
  for (int k = blockIdx.x; k < n_parts; k += gridDim.x) {
    const int64_t lo = part_lo[k], hi = part_lo[k + 1];
    if (tid < MKE_OC_MAX_RANKS) s_cnt[tid] = 0;
    for (int i = tid; i < NWV * MKE_OC_MAX_RANKS; i += OC_PLAN_THREADS) (&s_wcnt[0][0])[i] = 0;
    __syncthreads();
    // 4,096 positions per round of the block: wavefront wv takes 256 CONSECUTIVE ones as four ballots of 64, its running
    // per-owner counts in the lanes (lane o = owner o), so the block synchronises three times per 4,096 positions instead of per
    // 1,024 (the kernel was bound by those: 139 us per epoch share at the C2 shape with 8 ranks, the largest term of the plan)
    constexpr int RND = 4;
    for (int64_t base = lo; base < hi; base += (int64_t)OC_PLAN_THREADS * RND) {
      int rk[RND], ow[RND];
      int run = 0;                                   // lane o: this wavefront's positives of owner o so far in this round of the block
      uint32_t need_[RND];
      int id_[RND];
#pragma unroll
      for (int r = 0; r < RND; ++r) {                // the four rounds' loads first: in flight together (the first code of a group is a
        const int64_t i = base + (int64_t)wv * (64 * RND) + r * 64 + lane;   // strided access: one cache line per lane)
        need_[r] = i < hi ? (neg_per_pos ? (uint32_t)codes[i * neg_per_pos] : MKE_OC_NEED_HR) : 0u;
        id_[r] = i < hi ? ids[i] : 0;
      }
#pragma unroll
      for (int r = 0; r < RND; ++r) {
        const bool valid = (need_[r] & (x ? MKE_OC_NEED_RT : MKE_OC_NEED_HR)) != 0;
        const int o = valid ? oc_mod(oc_divisor(G), id_[r]) : -1;
        ow[r] = o;
        rk[r] = 0;
        uint64_t todo = __ballot(valid);
        while (todo) {                               // wave-uniform: one round per distinct owner present in the wavefront
          const int o0 = __shfl(o, __builtin_ctzll(todo), 64);
          const uint64_t m = __ballot(valid && o == o0);
          const int before = __shfl(run, o0, 64);
          if (valid && o == o0) rk[r] = before + __popcll(m & ((1ull << lane) - 1ull));
          if (lane == o0) run += __popcll(m);
          todo &= ~m;
        }
      }
      if (lane < MKE_OC_MAX_RANKS) s_wcnt[wv][lane] = run;
      __syncthreads();
      int pre = 0;                                   // lane o: positives of owner o in the wavefronts before this one
      if (lane < MKE_OC_MAX_RANKS) {
        pre = s_cnt[lane];
        for (int w = 0; w < wv; ++w) pre += s_wcnt[w][lane];
      }
#pragma unroll
      for (int r = 0; r < RND; ++r) {
        const int64_t i = base + (int64_t)wv * (64 * RND) + r * 64 + lane;
        const int o = ow[r];
        const int first = __shfl(pre, o < 0 ? 0 : o, 64);
        if (o >= 0) {
          const int sl = first + rk[r];
          slot[i] = sl;
          if (o == rank) own[lo + sl] = (int32_t)(i - lo);
        } else if (i < hi) {
          slot[i] = -1;
        }
      }
      __syncthreads();
      if (tid < G) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) c += s_wcnt[w][tid];
        s_cnt[tid] += c;
      }
      __syncthreads();
    }
    if (tid < G) counts[((int64_t)x * n_parts + k) * G + tid] = s_cnt[tid];
    __syncthreads();
  }
}

static int oc_check(const mke_oc_step* s, const char* who) {
  if (!s) { set_error("%s: NULL step", who); return MKE_E_NULL; }
  if (s->n_ranks < 1 || s->n_ranks > MKE_OC_MAX_RANKS || s->rank < 0 || s->rank >= s->n_ranks) { set_error("%s: bad rank / n_ranks", who); return MKE_E_SHAPE; }
  if (s->stride <= 0 || s->stride % 16 != 0 || s->dim <= 0 || s->dim > s->stride || s->stride > MKE_MAX_STRIDE) { set_error("%s: bad stride/dim", who); return MKE_E_SHAPE; }
  if (s->n_pos > 0x7FFFFFFFll || s->per > 0x7FFFFFFFll) { set_error("%s: n_pos / per beyond 2^31", who); return MKE_E_RANGE; }
  if (s->n_pos < 0 || s->per < 1 || s->per * s->n_ranks < s->n_pos || s->neg_per_pos < 0 || s->neg_per_pos > 64) { set_error("%s: bad n_pos / per / neg_per_pos (<= 64)", who); return MKE_E_SHAPE; }
  if (s->n_own_h < 0 || s->n_own_t < 0 || s->n_own_h > s->capacity || s->n_own_t > s->capacity) { set_error("%s: owned vectors exceed the capacity", who); return MKE_E_SHAPE; }
  if (s->rel_grad_copies < 1 || s->rel_grad_copies > 64) { set_error("%s: rel_grad_copies must be in [1,64]", who); return MKE_E_SHAPE; }
  if (s->n_pos > 0 && (!s->pos_h || !s->pos_r || !s->pos_t || !s->slot_h || !s->slot_t)) { set_error("%s: NULL positive / slot stream", who); return MKE_E_NULL; }
  if (s->n_pos * s->neg_per_pos > 0 && !s->codes) { set_error("%s: NULL negative codes", who); return MKE_E_NULL; }
  if (s->n_peers != 0 && s->n_peers != s->n_ranks) { set_error("%s: n_peers must be 0 or n_ranks", who); return MKE_E_SHAPE; }
  for (int g = 0; g < s->n_peers; ++g)
    if (!s->peer_v[g] || !s->peer_g[g]) { set_error("%s: NULL peer block %d", who, g); return MKE_E_NULL; }
  if ((s->n_own_h > 0 && !s->own_h) || (s->n_own_t > 0 && !s->own_t)) { set_error("%s: NULL owned-slot list", who); return MKE_E_NULL; }
  if (!s->ent || !s->rel) { set_error("%s: NULL table", who); return MKE_E_NULL; }
  if (s->optimizer != MKE_OPT_ADAGRAD && s->optimizer != MKE_OPT_SGD) { set_error("%s: Adagrad or SGD", who); return MKE_E_UNSUPPORTED; }
  if (s->hot.slot && (s->hot.n_hot < 1 || s->hot.copies < 1 || s->hot.copies > 64 || s->hot.row0 < s->n_local)) { set_error("%s: bad hub-row declaration", who); return MKE_E_SHAPE; }
  if (s->em_coef) {   // entity-major second pass
    if (s->n_peers) { set_error("%s: the entity-major pass does not run peer-direct", who); return MKE_E_UNSUPPORTED; }
    if (s->em_pos0 < 0 || s->em_n_rows < 0 || s->em_chunks < 1 || s->em_chunks > MKE_OC_EM_MAX_CHUNKS) { set_error("%s: bad em_pos0 / em_n_rows / em_chunks (1..%d)", who, MKE_OC_EM_MAX_CHUNKS); return MKE_E_SHAPE; }
    if ((s->em_pos0 + s->n_pos) * (s->neg_per_pos + 1) > 0x7FFFFFFFll) { set_error("%s: the step's coefficients exceed 2^31", who); return MKE_E_RANGE; }
    if (s->capacity >= (1 << 23)) { set_error("%s: entity-major locators hold slots below 2^23", who); return MKE_E_RANGE; }
  }
  return MKE_OK;
}

static inline unsigned oc_blocks(int64_t n, int per_block, int64_t cap) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

}  // namespace mke

extern "C" int64_t mke_oc_block_floats(int64_t capacity, int stride) {
  if (capacity < 0 || stride < 0) return -1;
  return 2 * capacity * stride;
}

extern "C" int mke_oc_pack_codes(const int32_t* pos_h, const int32_t* neg_h, const int32_t* neg_t, int64_t n_pos, int neg_per_pos,
                                 int32_t* codes, void* stream) {
  using namespace mke;
  if (n_pos < 0 || neg_per_pos < 0) { set_error("mke_oc_pack_codes: negative count"); return MKE_E_SHAPE; }
  if (n_pos * neg_per_pos == 0) return MKE_OK;
  if (!pos_h || !neg_h || !neg_t || !codes) { set_error("mke_oc_pack_codes: NULL pointer"); return MKE_E_NULL; }
  if (neg_per_pos > 64) { set_error("mke_oc_pack_codes: neg_per_pos <= 64"); return MKE_E_SHAPE; }
  const int gs = neg_per_pos <= 16 ? 16 : (neg_per_pos <= 32 ? 32 : 64);
  const dim3 grid(oc_blocks(n_pos, (MKE_BLOCK / 64) * (64 / gs), 16384)), blk(MKE_BLOCK);
  if (gs == 16) hipLaunchKernelGGL((k_oc_pack_codes<16>), grid, blk, 0, (hipStream_t)stream, pos_h, neg_h, neg_t, n_pos, neg_per_pos, codes);
  else if (gs == 32) hipLaunchKernelGGL((k_oc_pack_codes<32>), grid, blk, 0, (hipStream_t)stream, pos_h, neg_h, neg_t, n_pos, neg_per_pos, codes);
  else hipLaunchKernelGGL((k_oc_pack_codes<64>), grid, blk, 0, (hipStream_t)stream, pos_h, neg_h, neg_t, n_pos, neg_per_pos, codes);
  return check_launch("k_oc_pack_codes");
}

extern "C" int mke_oc_plan(const int32_t* pos_h, const int32_t* pos_t, const int32_t* codes, int neg_per_pos, const int64_t* part_lo,
                           int n_parts, int n_ranks, int rank, int32_t* slot_h, int32_t* slot_t, int32_t* own_h, int32_t* own_t,
                           int32_t* counts, void* stream) {
  using namespace mke;
  if (n_parts < 0 || n_ranks < 1 || n_ranks > MKE_OC_MAX_RANKS || rank < 0 || rank >= n_ranks) { set_error("mke_oc_plan: bad n_parts / n_ranks / rank"); return MKE_E_SHAPE; }
  if (n_parts == 0) return MKE_OK;
  if (!pos_h || !pos_t || !part_lo || !slot_h || !slot_t || !own_h || !own_t || !counts) { set_error("mke_oc_plan: NULL pointer"); return MKE_E_NULL; }
  if (neg_per_pos < 0 || (neg_per_pos > 0 && !codes)) { set_error("mke_oc_plan: neg_per_pos > 0 needs the epoch's codes"); return MKE_E_NULL; }
  hipLaunchKernelGGL(k_oc_plan, dim3((unsigned)(n_parts < 32768 ? n_parts : 32768), 2), dim3(OC_PLAN_THREADS), 0, (hipStream_t)stream,
                     pos_h, pos_t, codes, neg_per_pos, part_lo, n_parts, n_ranks, rank, slot_h, slot_t, own_h, own_t, counts);
  return check_launch("k_oc_plan");
}

namespace mke {
// bases, with the step's reference counting on rider blocks of the same launch when with_count
static int oc_bases_impl(const mke_oc_step* s, float* send_block, bool with_count, void* stream) {
  int rc = oc_check(s, "mke_oc_bases");
  if (rc) return rc;
  if (!send_block) { set_error("mke_oc_bases: NULL send block"); return MKE_E_NULL; }
  OcParams p{};
  p.s = *s; p.send = send_block;
  const int64_t subs = s->n_own_h + s->n_own_t;
  const int64_t n_count = (with_count && s->ref_count) ? s->n_pos * (s->neg_per_pos + 2) : 0;
  if (subs == 0) return n_count ? mke_oc_count(s, stream) : MKE_OK;
  p.count_blocks = n_count ? (int)oc_blocks(n_count, MKE_BLOCK, 1024) : 0;
  const int fpl = s->stride / 16;
  const unsigned blocks = (unsigned)((subs + MKE_SUBS_PER_BLOCK - 1) / MKE_SUBS_PER_BLOCK) + p.count_blocks;
  MKE_DISPATCH_FPL(fpl, { hipLaunchKernelGGL((k_oc_bases<FPL>), dim3(blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, p); });
  return check_launch("k_oc_bases");
}
}  // namespace mke

extern "C" int mke_oc_bases(const mke_oc_step* s, float* send_block, void* stream) {
  return mke::oc_bases_impl(s, send_block, false, stream);
}

extern "C" int mke_oc_count(const mke_oc_step* s, void* stream) {
  using namespace mke;
  int rc = oc_check(s, "mke_oc_count");
  if (rc) return rc;
  if (!s->ref_count) return MKE_OK;
  const int64_t total = s->n_pos * (s->neg_per_pos + 2);
  if (total == 0) return MKE_OK;
  OcParams p{};
  p.s = *s;
  hipLaunchKernelGGL(k_oc_count, dim3(oc_blocks(total, MKE_BLOCK, 2048)), dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  return check_launch("k_oc_count");
}

extern "C" int mke_oc_score(const mke_oc_step* s, const float* v_all, int64_t block_floats, float* g_all, double* loss_partials,
                            void* stream) {
  mke::TuningScope scope((s && s->tuning) ? s->tuning : nullptr);   // the step's knobs for the duration of this call
  using namespace mke;
  int rc = oc_check(s, "mke_oc_score");
  if (rc) return rc;
  const bool em = s->em_coef != nullptr;
  if ((!s->n_peers && (!v_all || !g_all)) || !loss_partials || !s->rel_grad || !s->rel_touched || (!em && (!s->ent_grad || !s->ent_touched))) { set_error("mke_oc_score: NULL pointer"); return MKE_E_NULL; }
  if (!em && s->ref_count && s->optimizer == MKE_OPT_ADAGRAD && !s->ent_acc) { set_error("mke_oc_score: the exclusive-row path with Adagrad needs ent_acc"); return MKE_E_NULL; }
  OcParams p{};
  p.s = *s; p.v_all = v_all; p.block_floats = block_floats; p.g_all = g_all; p.lossp = loss_partials;
  const int fpl = s->stride / 16;
  // a quarter-wave per positive when a rank owns only a few of a positive's negatives (N / G <= 8 at G >= 4, rows up to 128 floats:
  // wider rows leave two wavefronts per SIMD at 194 registers) — k_oc_score_q;
  // option "oc_score_quarter": -1 = by shape (default), 0 = never, 1 = always
  const bool pow2 = (s->n_ranks & (s->n_ranks - 1)) == 0;     // id / G, id % G as shift / mask (2, 4, 8 ranks)
  const int oq = tune_oc_score_quarter();
  const bool quarter = oq < 0 ? (s->n_ranks >= 4 && s->neg_per_pos <= 8 * s->n_ranks && s->stride <= 128) : oq != 0;
  const dim3 grid(MKE_LOSS_PARTIALS), blk(MKE_BLOCK);
  hipStream_t st = (hipStream_t)stream;
  if (quarter) {
    MKE_DISPATCH_FPL(fpl, {
      if (em) { if (pow2) hipLaunchKernelGGL((k_oc_score_q<FPL, true, true>), grid, blk, 0, st, p); else hipLaunchKernelGGL((k_oc_score_q<FPL, false, true>), grid, blk, 0, st, p); }
      else { if (pow2) hipLaunchKernelGGL((k_oc_score_q<FPL, true>), grid, blk, 0, st, p); else hipLaunchKernelGGL((k_oc_score_q<FPL, false>), grid, blk, 0, st, p); }
    });
    return check_launch("k_oc_score_q");
  }
  MKE_DISPATCH_FPL(fpl, {
    constexpr int U = FPL <= 5 ? 2 : 1;
    if (em) { if (pow2) hipLaunchKernelGGL((k_oc_score<FPL, U, true, true>), grid, blk, 0, st, p); else hipLaunchKernelGGL((k_oc_score<FPL, U, false, true>), grid, blk, 0, st, p); }
    else { if (pow2) hipLaunchKernelGGL((k_oc_score<FPL, U, true>), grid, blk, 0, st, p); else hipLaunchKernelGGL((k_oc_score<FPL, U, false>), grid, blk, 0, st, p); }
  });
  return check_launch("k_oc_score");
}

extern "C" int mke_oc_apply(const mke_oc_step* s, const float* gv, void* stream) {
  using namespace mke;
  int rc = oc_check(s, "mke_oc_apply");
  if (rc) return rc;
  const int64_t subs = s->n_own_h + s->n_own_t;
  if (subs == 0) return MKE_OK;
  const bool em = s->em_coef != nullptr;
  if (!gv || !s->rel_grad || !s->rel_touched || (!em && (!s->ent_grad || !s->ent_touched))) { set_error("mke_oc_apply: NULL pointer"); return MKE_E_NULL; }
  OcParams p{};
  p.s = *s; p.gv = gv;
  const int fpl = s->stride / 16;
  const dim3 grid((unsigned)((subs + MKE_SUBS_PER_BLOCK - 1) / MKE_SUBS_PER_BLOCK));
  MKE_DISPATCH_FPL(fpl, {
    if (em) hipLaunchKernelGGL((k_oc_apply<FPL, false>), grid, dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_oc_apply<FPL>), grid, dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  });
  return check_launch("k_oc_apply");
}

namespace mke {
int launch_rows_update_multi(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim, int optimizer,
                             float lr, hipStream_t st, const mke_count_job* count, const struct DenseJob* dense);
}

// Several phases of one part's step in ONE call (host overhead: the step is 5 launches of 5-50 us each): bit 0 bases, 1 count,
// 2 score, 3 apply, 4 the row update (relation table: every row; shard: touched rows).  What lies between two collectives
// goes into one call: single rank: 31;  G > 1: 1 | all-gather | 6 | reduce-scatter | 8 | all-reduce | 16.
extern "C" int mke_oc_run(const mke_oc_step* s, int phases, float* send_block, const float* v_all, int64_t block_floats, float* g_all,
                          const float* gv, double* loss_partials, void* stream) {
  mke::TuningScope scope((s && s->tuning) ? s->tuning : nullptr);   // the step's knobs for the duration of this call
  using namespace mke;
  int rc = MKE_OK;
  if (s && s->em_coef) phases &= ~MKE_OC_COUNT;                              // entity-major: nothing is reference-counted
  const bool both = (phases & MKE_OC_BASES) && (phases & MKE_OC_COUNT);   // one launch: the counting rides with the bases
  if ((phases & MKE_OC_BASES) && (rc = oc_bases_impl(s, send_block, both, stream))) return rc;
  if ((phases & MKE_OC_COUNT) && !both && (rc = mke_oc_count(s, stream))) return rc;
  if ((phases & MKE_OC_SCORE) && (rc = mke_oc_score(s, v_all, block_floats, g_all, loss_partials, stream))) return rc;
  if ((phases & MKE_OC_APPLY) && (rc = mke_oc_apply(s, gv, stream))) return rc;
  if ((phases & MKE_OC_PASS2) && (rc = mke_oc_pass2(s, stream))) return rc;
  if (phases & MKE_OC_UPDATE) {
    if ((rc = oc_check(s, "mke_oc_run"))) return rc;
    if (s->optimizer == MKE_OPT_ADAGRAD && ((!s->em_coef && !s->ent_acc) || !s->rel_acc)) { set_error("mke_oc_run: Adagrad needs ent_acc and rel_acc"); return MKE_E_NULL; }
    mke_update_table ut[2] = {};
    ut[0].table = const_cast<float*>(s->rel); ut[0].acc = s->rel_acc; ut[0].grad = s->rel_grad; ut[0].touched = nullptr;
    ut[0].n_rows = s->n_rel; ut[0].normalize = 1; ut[0].grad_copies = s->rel_grad_copies;
    ut[1].table = s->ent; ut[1].acc = s->ent_acc; ut[1].grad = s->ent_grad; ut[1].touched = s->ent_touched;
    ut[1].n_rows = s->n_local; ut[1].normalize = 1; ut[1].grad_copies = 1; ut[1].ref_count = s->ref_count; ut[1].hot = s->hot;
    // entity-major: the shard's rows were finished by mke_oc_pass2 — the relation table alone
    if ((rc = launch_rows_update_multi(ut, s->em_coef ? 1 : 2, s->tag, s->stride, s->dim, s->optimizer, s->lr, (hipStream_t)stream, nullptr, nullptr))) return rc;
  }
  return MKE_OK;
}
