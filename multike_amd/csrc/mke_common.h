// mke_common.h — device helpers shared by the gfx950 kernels of libmultike_hip.so.
//
// Execution shape used by every row kernel: one 16-lane quarter-wavefront ("sub16") owns one embedding
// row; lane j of the quarter holds columns {j, j+16, j+32, ...} (FPL = stride/16 floats per lane).
// A 64-wide wavefront therefore works on 4 rows at a time, each row access is a run of 64-byte
// contiguous segments, reductions over a row are four DPP adds that never leave the 16-lane DPP row,
// and a gradient row is scattered with FPL `global_atomic_add_f32`, each covering 64 contiguous bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/multike_hip.h"

#define MKE_BLOCK 256                 // threads per workgroup: 4 wavefronts, 16 quarter-waves
#define MKE_SUBS_PER_BLOCK (MKE_BLOCK / 16)
#define MKE_L2_EPS 1e-12f             // tf.nn.l2_normalize epsilon (code/base/initializers.py:26)

namespace mke {

// host-side error plumbing (mke_api.hip)
void set_error(const char* fmt, ...);
int check_launch(const char* what);

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 16 lanes of a DPP row; every lane receives the total.
__device__ __forceinline__ float sub16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror: lane i <-> 7-i inside each 8
  v += dpp_mov<0x140>(v);  // row_mirror:      lane i <-> 15-i
  return v;
}

template <int FPL>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int64_t row, int stride, int j,
                                         float (&v)[FPL]) {
  const float* p = base + row * (int64_t)stride + j;
#pragma unroll
  for (int k = 0; k < FPL; ++k) v[k] = p[k * 16];
}

// x * rsqrt(max(sum x^2, eps)) — tf.nn.l2_normalize(x, 1).  Returns the inverse norm factor used.
template <int FPL>
__device__ __forceinline__ float l2_normalize_row(float (&v)[FPL], bool on) {
  if (!on) return 1.0f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < FPL; ++k) s = fmaf(v[k], v[k], s);
  s = sub16_sum(s);
  const float inv = rsqrtf(fmaxf(s, MKE_L2_EPS));
#pragma unroll
  for (int k = 0; k < FPL; ++k) v[k] *= inv;
  return inv;
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  // relaxed, agent scope, no return value -> global_atomic_add_f32 (built with -munsafe-fp-atomics)
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Atomically add a row held in sub16 layout into grad[row]; columns >= dim are never written.
template <int FPL>
__device__ __forceinline__ void atomic_add_row(float* __restrict__ grad, int64_t row, int stride, int dim,
                                               int j, const float (&v)[FPL], float sgn) {
  float* p = grad + row * (int64_t)stride + j;
#pragma unroll
  for (int k = 0; k < FPL; ++k) {
    if (k * 16 + 16 <= dim || k * 16 + j < dim) atomic_add_f32(p + k * 16, sgn * v[k]);
  }
}

// log(1 + exp(x)) and sigmoid, written to stay finite for any x, on the hardware transcendentals (v_exp_f32 / v_log_f32 /
// v_rcp_f32, ~1 ulp each: absolute error of a term <= ~2e-7, far inside the 1e-4 loss tolerance).  The library forms
// (log1pf, expf, an IEEE division) are 10-30 VALU instructions each, and the fused step is VALU-issue bound: rocprofv3 PMC
// on k_triple_score showed 3500 VALU instructions per wavefront, ~30 us of pure issue time at 5 waves per SIMD.
__device__ __forceinline__ float softplus_f(float x) {
  return fmaxf(x, 0.f) + __logf(1.0f + __expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh on v_exp_f32 / v_rcp_f32: (1 - e) / (1 + e), e = exp(-2 |x|) — 7 instructions against ~25 for the library's tanhf
// (the attribute CNN evaluates 16 of them per lane and triple, twice: the backward recomputes the forward).  |error| <= ~2e-7
// absolute (the subtraction 1 - e loses relative accuracy only where tanh itself is below 1e-3).
__device__ __forceinline__ float tanh_f(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}
// g / sqrt(a) of the Adagrad rule on v_rsq_f32 (an IEEE sqrt + division is ~20 instructions per element)
__device__ __forceinline__ float adagrad_scale(float a) { return __builtin_amdgcn_rsqf(a); }

// Reference counting of a step's entity references (mke_count_entity_refs semantics) by `n_blocks` rider blocks of some
// other kernel's grid; `block` = index among them.
__device__ __forceinline__ void count_refs_range(const mke_count_job& c, int64_t block, int64_t n_blocks) {
  const int64_t total = c.n_pos + c.n_neg;
  for (int64_t i = block * MKE_BLOCK + threadIdx.x; i < total; i += n_blocks * MKE_BLOCK) {
    if (i < c.n_pos) {
      atomicAdd(&c.ref_count[c.pos_h[i]], 1);
      atomicAdd(&c.ref_count[c.pos_t[i]], 1);
    } else {
      const int64_t n = i - c.n_pos;
      const int64_t g = n / c.neg_per_pos;
      const int a = c.neg_h[n], b = c.neg_t[n];
      if (a != c.pos_h[g]) atomicAdd(&c.ref_count[a], 1);
      if (b != c.pos_t[g]) atomicAdd(&c.ref_count[b], 1);
    }
  }
}

// Upper bound of the rows the NEXT k_rows_update_multi launch of this host thread will find touched (0 = unknown).  A step that
// touches at most a 16th of a large table (an attribute step's 5,000 heads of 200K rows, a positives-only step) is walked in
// 64-row chunks — a quarter of the wavefronts, each still finding about one row — instead of 16-row chunks (mke_update.hip).
extern thread_local int64_t g_update_touched_hint;
struct UpdateTouchedHint {
  explicit UpdateTouchedHint(int64_t rows) { g_update_touched_hint = rows; }
  ~UpdateTouchedHint() { g_update_touched_hint = 0; }
};

// Per-call tuning (mke_tuning): the entry point that carries one installs it for the duration of the call (thread-local, so two
// host threads with different plans do not see each other); the launchers read a knob through tune(): the call's value, else the
// process default set by mke_set_option.
extern thread_local const mke_tuning* tl_tuning;
struct TuningScope {
  const mke_tuning* prev;
  explicit TuningScope(const mke_tuning* t) : prev(tl_tuning) { if (t) tl_tuning = t; }
  ~TuningScope() { tl_tuning = prev; }
};
extern int g_score_splits, g_score_half_max, g_score_o32, g_score_lane_ids, g_count_in_score, g_update_chunk, g_oc_score_quarter,
    g_attr_fused_bwd, g_sampler_fast;
#define MKE_TUNE(field, dflt) ((mke::tl_tuning && mke::tl_tuning->field != MKE_TUNE_DEFAULT) ? mke::tl_tuning->field : (dflt))
inline int tune_score_splits() { const int v = MKE_TUNE(score_splits, g_score_splits); return v < 0 ? 0 : v; }
inline int tune_score_half_max() { const int v = MKE_TUNE(score_half_groups, g_score_half_max); return v < 0 ? -1 : (v > 64 ? 64 : v); }
inline int tune_score_o32() { return MKE_TUNE(score_offsets32, g_score_o32) != 0; }
inline int tune_score_lane_ids() { return MKE_TUNE(score_lane_ids, g_score_lane_ids) != 0; }
inline int tune_count_in_score() { return MKE_TUNE(count_in_score, g_count_in_score) != 0; }
inline int tune_update_chunk() { const int v = MKE_TUNE(update_chunk, g_update_chunk); return v == 16 ? 16 : (v == 64 ? 64 : 0); }
inline int tune_oc_score_quarter() { const int v = MKE_TUNE(oc_score_quarter, g_oc_score_quarter); return v < 0 ? -1 : (v != 0); }
inline int tune_attr_fused_bwd() { return MKE_TUNE(attr_fused_bwd, g_attr_fused_bwd) != 0; }
inline int tune_sampler_fast() { return MKE_TUNE(sampler_fast, g_sampler_fast) != 0; }

// Block-wide sum of one float per thread, accumulated in double; thread 0 gets the result.
__device__ __forceinline__ double block_sum_double(float v) {
  __shared__ double s_part[MKE_BLOCK / 64];
  double d = (double)v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_part[wave] = d;
  __syncthreads();
  double tot = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < MKE_BLOCK / 64; ++w) tot += s_part[w];
  }
  return tot;
}

// Dense Adagrad / SGD over n contiguous floats (the CNN's packed parameters), gradient consumed (zeroed).  The first
// ws_n gradients may additionally live in ws_copies privatised copies (row stride ws_stride) left by k_attr_conv's
// backward; they are added up and zeroed here.  Runs as its own kernel or as rider blocks of k_rows_update_multi.
struct DenseJob {
  float* w;
  float* acc;
  float* g;
  int64_t n;
  int optimizer;
  float lr;
  float* ws;
  int ws_n, ws_stride, ws_copies;
};
__device__ __forceinline__ void dense_update_range(const DenseJob& j, int64_t block, int64_t n_blocks) {
  for (int64_t i = block * MKE_BLOCK + threadIdx.x; i < j.n; i += n_blocks * MKE_BLOCK) {
    float gv = j.g[i];
    if (j.ws && i < j.ws_n) {  // all loads first (independent, in flight together), then the zeroing stores
#pragma unroll 8
      for (int c = 0; c < j.ws_copies; ++c) gv += j.ws[(size_t)c * j.ws_stride + i];
      for (int c = 0; c < j.ws_copies; ++c) j.ws[(size_t)c * j.ws_stride + i] = 0.f;
    }
    j.g[i] = 0.f;
    if (j.optimizer == MKE_OPT_ADAGRAD) {
      const float a = fmaf(gv, gv, j.acc[i]);
      j.acc[i] = a;
      j.w[i] -= j.lr * gv * adagrad_scale(a);
    } else {
      j.w[i] -= j.lr * gv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) — counter-based RNG of the negative sampler.
// ---------------------------------------------------------------------------------------------
struct Philox4 {
  uint32_t v[4];
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// known-triple key: h<<38 | t<<12 | r
__host__ __device__ __forceinline__ uint64_t triple_key(uint32_t h, uint32_t r, uint32_t t) {
  return ((uint64_t)h << 38) | ((uint64_t)t << 12) | (uint64_t)r;
}
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
#define MKE_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

}  // namespace mke

// Dispatch a kernel template on FPL = stride/16 (1..20).
#define MKE_DISPATCH_FPL(fpl, ...)                                                   \
  switch (fpl) {                                                                     \
    case 1: { constexpr int FPL = 1; __VA_ARGS__; } break;                           \
    case 2: { constexpr int FPL = 2; __VA_ARGS__; } break;                           \
    case 3: { constexpr int FPL = 3; __VA_ARGS__; } break;                           \
    case 4: { constexpr int FPL = 4; __VA_ARGS__; } break;                           \
    case 5: { constexpr int FPL = 5; __VA_ARGS__; } break;                           \
    case 6: { constexpr int FPL = 6; __VA_ARGS__; } break;                           \
    case 7: { constexpr int FPL = 7; __VA_ARGS__; } break;                           \
    case 8: { constexpr int FPL = 8; __VA_ARGS__; } break;                           \
    case 10: { constexpr int FPL = 10; __VA_ARGS__; } break;                         \
    case 12: { constexpr int FPL = 12; __VA_ARGS__; } break;                         \
    case 13: { constexpr int FPL = 13; __VA_ARGS__; } break;                         \
    case 16: { constexpr int FPL = 16; __VA_ARGS__; } break;                         \
    case 20: { constexpr int FPL = 20; __VA_ARGS__; } break;                         \
    default:                                                                         \
      mke::set_error("unsupported stride %d (stride/16 must be one of 1-8,10,12,13,16,20)", (fpl) * 16); \
      return MKE_E_UNSUPPORTED;                                                      \
  }
