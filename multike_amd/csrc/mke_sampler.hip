// mke_sampler.hip — on-device negative sampler + known-triple hash set (gfx950).
//
// Distribution = code/base/batch.py:86-116 generate_neg_triples_fast: per positive, up
// to max_try rounds; one fair coin per round chooses the corrupted side; `need` distinct candidates are
// drawn without replacement from that side's candidate list; known triples are dropped except in the
// last round; stop at neg_per_pos.  The reference draws from CPython's Mersenne Twister through
// Manager-queue worker processes; here the stream is Philox4x32-10 keyed by (seed, stream id) and
// indexed by (positive, round, slot, attempt), so any positive can be sampled by any wavefront in any
// order and the CPU oracle (oracle/sampler_oracle.py) reproduces the device output bit for bit.
//
// Shape: one group of 16 (neg_per_pos <= 15: four positives per wavefront), 32 (<= 32: two) or 64 lanes per positive, lane q of
// the group = slot q of the round (neg_per_pos <= 64).
#include "mke_common.h"

namespace mke {

struct SampleParams {
  const int32_t* __restrict__ ph;
  const int32_t* __restrict__ pr;
  const int32_t* __restrict__ pt;
  const uint8_t* __restrict__ pos_kg;
  int64_t n_pos, pos_offset;
  const int32_t* __restrict__ pos_index;  // nullable: epoch position of positive i (else pos_offset + i)
  int npp, max_try;
  mke_kg_side side[2];
  uint32_t seed_lo, seed_hi, sid;
  int32_t* __restrict__ nh;
  int32_t* __restrict__ nr;
  int32_t* __restrict__ nt;
};

// Next accepted bounded draw in [0,n) for (positive gi, round, slot): attempts are consumed in order;
// attempt a uses word (a&3) of Philox block (a>>2).  Lemire's multiply-shift with exact rejection.
__host__ __device__ __forceinline__ uint32_t draw_next(uint32_t gi, uint32_t round, uint32_t slot, uint32_t sid,
                                                       uint32_t k0, uint32_t k1, uint32_t n, uint32_t& attempt) {
  for (;;) {
    const uint32_t blk = attempt >> 2, word = attempt & 3u;
    const Philox4 ph = philox4x32_10(gi, round | (blk << 8), slot, sid, k0, k1);
    const uint32_t x = ph.v[word];
    ++attempt;
    const uint64_t m = (uint64_t)x * (uint64_t)n;
    const uint32_t l = (uint32_t)m;
    if (l < n) {
      const uint32_t thresh = (0u - n) % n;
      if (l < thresh) continue;
    }
    return (uint32_t)(m >> 32);
  }
}

__device__ __forceinline__ bool set_contains(const uint64_t* __restrict__ keys, uint64_t cap, uint64_t key) {
  uint64_t slot = mix64(key) & (cap - 1);
  for (;;) {
    const uint64_t k = keys[slot];
    if (k == key) return true;
    if (k == MKE_EMPTY_KEY) return false;
    slot = (slot + 1) & (cap - 1);
  }
}

// GS lanes per positive (GS = 32 when neg_per_pos <= 32: two positives per wavefront, else 64).  Everything that is
// per positive (round count, collected, coin, candidate list) lives in the group's lanes; wave-wide primitives
// (__shfl, __ballot) are used with group-relative indices / masks.
//
// FAST (round 4; same stream, same output bit for bit — the kernel is bound by VALU issue, 32-bit integer multiplies being
// quarter rate: two Philox4x32-10 evaluations = 80 of them per round):
//  * the round's coin block (counter word 2 = 0xFFFFFFFF) is evaluated by the group's LAST lane in the same Philox
//    evaluation in which lanes < need evaluate their first draw block (word 2 = slot) — one evaluation per round instead
//    of two; needs an idle lane (neg_per_pos < GS), else the coin keeps its own evaluation;
//  * "is any first draw a duplicate of an earlier slot's" is answered by inserting the draws into a 2 GS-slot open-addressing
//    table of the group in LDS (compare-and-swap, ~1.2 probes) instead of neg_per_pos rounds of shuffles; the exact sequential
//    fix-up below runs only when the table saw an equal value (0.3 % of the groups at 25 draws from 100K).
template <int GS, bool FAST>
__global__ __launch_bounds__(MKE_BLOCK) void k_neg_sample(const SampleParams p) {
  constexpr int PPW = 64 / GS;
  constexpr int TS = 2 * GS;                                // slots of a group's duplicate table
  __shared__ uint32_t s_dup[FAST ? (MKE_BLOCK / GS) * TS : 1];
  const int lane = threadIdx.x & 63;
  const int gl = lane & (GS - 1);        // slot inside the group
  const int gbase = lane & ~(GS - 1);    // first lane of the group
  const int64_t wave = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 6;
  const int64_t i = wave * PPW + (lane / GS);
  const bool live = i < p.n_pos;
  const int64_t ii = live ? i : 0;
  const int h = p.ph[ii], r = p.pr[ii], t = p.pt[ii];
  const int kg = p.pos_kg ? (p.pos_kg[ii] != 0) : 0;
  const mke_kg_side& sd = p.side[kg];
  const uint32_t sid = p.sid + (uint32_t)kg;
  const uint64_t* __restrict__ keys = sd.known_keys;
  const uint32_t gi = p.pos_index ? (uint32_t)p.pos_index[ii] : (uint32_t)(ii + p.pos_offset);
  const int N = p.npp;
  const uint64_t gmask_all = GS == 64 ? ~0ull : (((1ull << (GS & 63)) - 1ull) << gbase);
  int collected = live ? 0 : N;
  for (int round = 0; round < p.max_try; ++round) {
    if (!__ballot(collected < N)) break;  // every positive of the wave is done
    const int need = N - collected;       // 0 for finished groups
    const bool active = gl < need;
    const bool merged = FAST && N < GS;   // the group's last lane is never a draw lane: it evaluates the coin block
    bool corrupt_head;
    uint32_t first_word = 0;
    if (merged) {
      const Philox4 ph = philox4x32_10(gi, (uint32_t)round, gl == GS - 1 ? 0xFFFFFFFFu : (uint32_t)gl, sid, p.seed_lo, p.seed_hi);
      first_word = ph.v[0];               // draw lanes: attempt 0 = word 0 of block 0 of (positive, round, slot)
      corrupt_head = ((uint32_t)__shfl((int)ph.v[0], gbase + GS - 1, 64) >> 31) != 0;
    } else {
      const Philox4 cph = philox4x32_10(gi, (uint32_t)round, 0xFFFFFFFFu, sid, p.seed_lo, p.seed_hi);
      corrupt_head = (cph.v[0] >> 31) != 0;
    }
    const int x = corrupt_head ? h : t;
    const bool use_tbl = sd.cand_table != nullptr && (sd.cand_valid == nullptr || sd.cand_valid[x] != 0);
    const uint32_t n = use_tbl ? (uint32_t)sd.cand_k : (uint32_t)sd.n_ent;
    uint32_t attempt = 0;
    uint32_t pos = 0xFFFFFFFFu;
    if (active) {
      bool have = false;
      if (merged) {                       // draw_next's first iteration on the word already at hand
        attempt = 1;
        const uint64_t m = (uint64_t)first_word * (uint64_t)n;
        const uint32_t l = (uint32_t)m;
        have = l >= n || l >= (0u - n) % n;
        pos = (uint32_t)(m >> 32);
      }
      if (!have) pos = draw_next(gi, (uint32_t)round, (uint32_t)gl, sid, p.seed_lo, p.seed_hi, n, attempt);
    }
    // duplicate detection among first draws (q runs to the largest `need` of the wave)
    int need_max = max(need, __shfl_xor(need, 32, 64));
    if (GS == 16) need_max = max(need_max, __shfl_xor(need_max, 16, 64));
    bool dup = false;
    if (FAST) {
      uint32_t* tbl = s_dup + (threadIdx.x / GS) * TS;
      tbl[gl] = 0xFFFFFFFFu;
      tbl[gl + GS] = 0xFFFFFFFFu;
      if (active) {
        uint32_t sl = (pos * 0x9E3779B1u) >> (32 - (GS == 16 ? 5 : GS == 32 ? 6 : 7));
        for (;;) {
          const uint32_t old = atomicCAS(&tbl[sl], 0xFFFFFFFFu, pos);
          if (old == 0xFFFFFFFFu) break;
          if (old == pos) { dup = true; break; }
          sl = (sl + 1) & (TS - 1);
        }
      }
    } else {
      for (int q = 0; q < need_max; ++q) {
        const uint32_t v = (uint32_t)__shfl((int)pos, gbase + q, 64);
        dup |= active && q < need && gl > q && pos == v;
      }
    }
    if (__ballot(dup)) {
      // sequential without-replacement semantics: slot q must differ from the final draws of slots < q
      for (int q = 1; q < need_max; ++q) {
        for (;;) {
          const uint32_t v = (uint32_t)__shfl((int)pos, gbase + q, 64);
          const bool hit = q < need && gl < q && pos == v;
          const uint64_t hb = __ballot(hit);
          if (!hb) break;
          if ((hb & gmask_all) && gl == q) pos = draw_next(gi, (uint32_t)round, (uint32_t)gl, sid, p.seed_lo, p.seed_hi, n, attempt);
        }
      }
    }
    int ent = 0;
    if (active) {
      ent = use_tbl ? sd.cand_table[(int64_t)x * sd.cand_k + pos]
                    : (sd.ent_list ? sd.ent_list[pos] : sd.ent_lo + (int32_t)pos);
    }
    const int nh = corrupt_head ? ent : h;
    const int nt = corrupt_head ? t : ent;
    bool keep = active;
    if (active && round < p.max_try - 1 && keys != nullptr) {
      keep = !set_contains(keys, sd.known_capacity, triple_key((uint32_t)nh, (uint32_t)r, (uint32_t)nt));
    }
    const uint64_t mask = (__ballot(keep) & gmask_all) >> gbase;  // this group's kept slots
    if (keep) {
      const int rank = __popcll(mask & ((1ull << gl) - 1ull));
      const int64_t o = i * (int64_t)N + collected + rank;
      p.nh[o] = nh; p.nr[o] = r; p.nt[o] = nt;
    }
    collected += __popcll(mask);
  }
}

__global__ __launch_bounds__(MKE_BLOCK) void k_tripleset_build(const int32_t* __restrict__ h, const int32_t* __restrict__ r,
                                                               const int32_t* __restrict__ t, int64_t n,
                                                               uint64_t* __restrict__ keys, uint64_t cap) {
  const int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = triple_key((uint32_t)h[i], (uint32_t)r[i], (uint32_t)t[i]);
  uint64_t slot = mix64(key) & (cap - 1);
  for (;;) {
    const unsigned long long prev =
        atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)MKE_EMPTY_KEY, (unsigned long long)key);
    if (prev == MKE_EMPTY_KEY || prev == key) return;
    slot = (slot + 1) & (cap - 1);
  }
}

__global__ __launch_bounds__(MKE_BLOCK) void k_tripleset_query(const int32_t* __restrict__ h, const int32_t* __restrict__ r,
                                                               const int32_t* __restrict__ t, int64_t n,
                                                               const uint64_t* __restrict__ keys, uint64_t cap,
                                                               uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x;
  if (i >= n) return;
  out[i] = set_contains(keys, cap, triple_key((uint32_t)h[i], (uint32_t)r[i], (uint32_t)t[i])) ? 1 : 0;
}

}  // namespace mke

namespace mke {
int validate_side(const mke_kg_side& sd, int neg_per_pos) {
  // random.sample raises ValueError when the population is smaller than the sample (batch.py:98,101)
  if (sd.n_ent < neg_per_pos) { set_error("candidate population (%d) smaller than neg_per_pos (%d)", sd.n_ent, neg_per_pos); return MKE_E_SHAPE; }
  if (sd.cand_table && sd.cand_k < neg_per_pos) { set_error("neighbour list (%d) shorter than neg_per_pos (%d)", sd.cand_k, neg_per_pos); return MKE_E_SHAPE; }
  if (sd.known_keys && (sd.known_capacity == 0 || (sd.known_capacity & (sd.known_capacity - 1)) != 0)) { set_error("known_capacity must be a power of two"); return MKE_E_SHAPE; }
  return MKE_OK;
}

int launch_neg_sample(const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, int64_t n_pos, int64_t pos_offset,
                      const int32_t* pos_index,
                      const uint8_t* pos_kg, const mke_kg_side* sides, int neg_per_pos, int max_try, uint32_t seed_lo,
                      uint32_t seed_hi, uint32_t stream_id, int32_t* neg_h, int32_t* neg_r, int32_t* neg_t, hipStream_t st) {
  SampleParams p;
  p.ph = pos_h; p.pr = pos_r; p.pt = pos_t; p.pos_kg = pos_kg; p.n_pos = n_pos; p.pos_offset = pos_offset; p.pos_index = pos_index;
  p.npp = neg_per_pos; p.max_try = max_try;
  p.side[0] = sides[0];
  p.side[1] = pos_kg ? sides[1] : sides[0];
  p.seed_lo = seed_lo; p.seed_hi = seed_hi; p.sid = stream_id;
  p.nh = neg_h; p.nr = neg_r; p.nt = neg_t;
  // lanes per positive: 16 (four positives per wavefront; fast form only, which needs an idle lane: neg_per_pos <= 15), 32, 64
  const int gs = (tune_sampler_fast() && neg_per_pos <= 15) ? 16 : (neg_per_pos <= 32 ? 32 : 64);
  const int64_t pos_per_block = (MKE_BLOCK / 64) * (64 / gs);
  const int64_t blocks = (n_pos + pos_per_block - 1) / pos_per_block;
  const dim3 grid((unsigned)blocks), blk(MKE_BLOCK);
  if (tune_sampler_fast()) {
    if (gs == 16) hipLaunchKernelGGL((k_neg_sample<16, true>), grid, blk, 0, st, p);
    else if (gs == 32) hipLaunchKernelGGL((k_neg_sample<32, true>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((k_neg_sample<64, true>), grid, blk, 0, st, p);
  } else {
    if (gs == 32) hipLaunchKernelGGL((k_neg_sample<32, false>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((k_neg_sample<64, false>), grid, blk, 0, st, p);
  }
  return check_launch("k_neg_sample");
}

// ---------------------------------------------------------------------------------------------------------------
// random.sample(list, batch) for every step of an epoch in one launch (code/MultiKE_model.py:358,380,402,425,446: the
// cross-KG inference and common-space loops draw `batch` distinct list positions per step, independently per step).
// out[s * batch + i] = pi_s(i), pi_s a keyed pseudo-random permutation of [0, n): a 6-round Feistel network over the
// smallest even-width power-of-two domain >= n, cycle-walked back into [0, n) (Black & Rogaway 2002).  Distinct
// inputs give distinct outputs by construction, so no dedupe pass and no n-sized shuffle per step is needed.
// Round keys: Philox4x32-10 of (step, block 0/1, 0x5A4D504C, stream_id) under (seed_lo, seed_hi).
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t feistel_round_fn(uint32_t x, uint32_t key) {
  x = x * 0x9E3779B1u + key;
  x ^= x >> 15; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(MKE_BLOCK) void k_sample_distinct(uint32_t n, int batch, int n_steps, uint32_t seed_lo, uint32_t seed_hi,
                                                               uint32_t stream_id, int half_bits, int32_t* __restrict__ out) {
  const int step = blockIdx.y;
  const Philox4 ka = philox4x32_10((uint32_t)step, 0u, 0x5A4D504Cu, stream_id, seed_lo, seed_hi);
  const Philox4 kb = philox4x32_10((uint32_t)step, 1u, 0x5A4D504Cu, stream_id, seed_lo, seed_hi);
  const uint32_t keys[6] = {ka.v[0], ka.v[1], ka.v[2], ka.v[3], kb.v[0], kb.v[1]};
  const uint32_t mask = (1u << half_bits) - 1u;
  for (int i = blockIdx.x * MKE_BLOCK + threadIdx.x; i < batch; i += gridDim.x * MKE_BLOCK) {
    uint32_t x = (uint32_t)i;
    do {
      uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const uint32_t t = l ^ (feistel_round_fn(r, keys[k]) & mask);
        l = r;
        r = t;
      }
      x = (l << half_bits) | r;
    } while (x >= n);
    out[(int64_t)step * batch + i] = (int32_t)x;
  }
}

}  // namespace mke

static int neg_sample_checked(const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, int64_t n_pos, int64_t pos_offset,
                              const int32_t* pos_index, const uint8_t* pos_kg, const mke_kg_side* sides, int neg_per_pos, int max_try,
                              uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id, int32_t* neg_h, int32_t* neg_r, int32_t* neg_t,
                              void* stream) {
  using namespace mke;
  if (n_pos < 0) { set_error("negative n_pos"); return MKE_E_SHAPE; }
  if (n_pos == 0 || neg_per_pos == 0) return MKE_OK;
  if (!pos_h || !pos_r || !pos_t || !neg_h || !neg_r || !neg_t || !sides) { set_error("mke_neg_sample: NULL pointer"); return MKE_E_NULL; }
  if (neg_per_pos < 0 || neg_per_pos > 64) { set_error("neg_per_pos must be in [0,64], got %d", neg_per_pos); return MKE_E_UNSUPPORTED; }
  if (max_try < 1 || max_try > 255) { set_error("max_try must be in [1,255]"); return MKE_E_SHAPE; }
  for (int k = 0; k < (pos_kg ? 2 : 1); ++k) {
    const int rc = validate_side(sides[k], neg_per_pos);
    if (rc) return rc;
  }
  return launch_neg_sample(pos_h, pos_r, pos_t, n_pos, pos_offset, pos_index, pos_kg, sides, neg_per_pos, max_try, seed_lo, seed_hi,
                           stream_id, neg_h, neg_r, neg_t, (hipStream_t)stream);
}

extern "C" int mke_neg_sample(const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, int64_t n_pos,
                              int64_t pos_offset, const uint8_t* pos_kg, const mke_kg_side* sides, int neg_per_pos,
                              int max_try, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id, int32_t* neg_h,
                              int32_t* neg_r, int32_t* neg_t, void* stream) {
  return neg_sample_checked(pos_h, pos_r, pos_t, n_pos, pos_offset, nullptr, pos_kg, sides, neg_per_pos, max_try, seed_lo, seed_hi,
                            stream_id, neg_h, neg_r, neg_t, stream);
}

extern "C" int mke_neg_sample_at(const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, int64_t n_pos,
                                 const int32_t* pos_index, const uint8_t* pos_kg, const mke_kg_side* sides, int neg_per_pos,
                                 int max_try, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id, int32_t* neg_h,
                                 int32_t* neg_r, int32_t* neg_t, void* stream) {
  if (n_pos > 0 && !pos_index) { mke::set_error("mke_neg_sample_at: NULL pos_index"); return MKE_E_NULL; }
  return neg_sample_checked(pos_h, pos_r, pos_t, n_pos, 0, pos_index, pos_kg, sides, neg_per_pos, max_try, seed_lo, seed_hi,
                            stream_id, neg_h, neg_r, neg_t, stream);
}

extern "C" int mke_tripleset_build(const int32_t* h, const int32_t* r, const int32_t* t, int64_t n, uint64_t* keys,
                                   uint64_t capacity, void* stream) {
  using namespace mke;
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!h || !r || !t || !keys) { set_error("mke_tripleset_build: NULL pointer"); return MKE_E_NULL; }
  if (capacity == 0 || (capacity & (capacity - 1)) != 0 || capacity < (uint64_t)n + 1) {
    set_error("capacity must be a power of two > n");
    return MKE_E_SHAPE;
  }
  const int64_t blocks = (n + MKE_BLOCK - 1) / MKE_BLOCK;
  hipLaunchKernelGGL(k_tripleset_build, dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, h, r, t, n,
                     keys, capacity);
  return check_launch("k_tripleset_build");
}

extern "C" int mke_tripleset_query(const int32_t* h, const int32_t* r, const int32_t* t, int64_t n,
                                   const uint64_t* keys, uint64_t capacity, uint8_t* out, void* stream) {
  using namespace mke;
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!h || !r || !t || !keys || !out) { set_error("mke_tripleset_query: NULL pointer"); return MKE_E_NULL; }
  if (capacity == 0 || (capacity & (capacity - 1)) != 0) { set_error("capacity must be a power of two"); return MKE_E_SHAPE; }
  const int64_t blocks = (n + MKE_BLOCK - 1) / MKE_BLOCK;
  hipLaunchKernelGGL(k_tripleset_query, dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, h, r, t, n,
                     keys, capacity, out);
  return check_launch("k_tripleset_query");
}

extern "C" int mke_sample_distinct(int64_t n, int batch, int n_steps, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id,
                                   int32_t* out, void* stream) {
  using namespace mke;
  if (n < 0 || batch < 0 || n_steps < 0 || n > 0x7FFFFFFFLL || (int64_t)batch > n) { set_error("mke_sample_distinct: need 0 <= batch <= n < 2^31 (n=%lld batch=%d)", (long long)n, batch); return MKE_E_SHAPE; }
  if (batch == 0 || n_steps == 0) return MKE_OK;
  if (!out) { set_error("mke_sample_distinct: NULL output"); return MKE_E_NULL; }
  if (n_steps > 65535) { set_error("mke_sample_distinct: more than 65535 steps"); return MKE_E_SHAPE; }
  int half = 1;
  while ((1ll << (2 * half)) < n) ++half;   // domain 4^half >= n, < 4n
  int bx = (batch + MKE_BLOCK - 1) / MKE_BLOCK;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_sample_distinct, dim3(bx, n_steps), dim3(MKE_BLOCK), 0, (hipStream_t)stream, (uint32_t)n, batch, n_steps, seed_lo,
                     seed_hi, stream_id, half, out);
  return check_launch("k_sample_distinct");
}
