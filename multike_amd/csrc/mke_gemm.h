// mke_gemm.h — internal interface of the hand-written f32 MFMA GEMMs (mke_gemm.hip) for the other translation units.
#pragma once
#include "mke_common.h"

namespace mke {

#define MKE_ACT_NONE 0
#define MKE_ACT_TANH 1
#define MKE_ACT_SIGMOID 2

// Fused epilogue of launch_gemm_f32_ex.  With v = alpha * (A B)[row][col] (+ bias[col]), in this order:
//   v = act(v);                                   act'(y) below is expressed in the activation's OUTPUT y
//   target:   e = v - target[row][col]; sumsq += e^2; v = target_scale * e * act'(v)      (loss tail: d loss / d pre-activation)
//   else:     sumsq += v^2  (when sumsq is given)
//   dact_y:   v *= dact_act'(dact_y[row][col])    (back-propagation through the previous layer's activation dact_act)
//   dot_with: dot += v * dot_with[row][col]
//   C[row][col] = v;  colsum[col] += v            (bias gradient)
// sumsq / dot are [MKE_LOSS_PARTIALS] double arrays that must be ZERO on entry; block b adds into slot b % MKE_LOSS_PARTIALS.
// colsum must be zero (or hold a running sum) on entry.  Everything except alpha needs a single K split.
struct GemmEpilogue {
  const float* alpha = nullptr;
  const float* bias = nullptr;
  int act = MKE_ACT_NONE;
  const float* dact_y = nullptr;
  int64_t ld_dact = 0;
  int dact_act = MKE_ACT_NONE;
  const float* target = nullptr;
  int64_t ld_target = 0;
  float target_scale = 0.f;
  double* sumsq = nullptr;
  const float* dot_with = nullptr;
  int64_t ld_dot = 0;
  double* dot = nullptr;
  float* colsum = nullptr;
};

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == MKE_ACT_TANH) return tanhf(v);
  if (act == MKE_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}
__device__ __forceinline__ float act_grad_from_output(float y, int act) {
  if (act == MKE_ACT_TANH) return 1.0f - y * y;
  if (act == MKE_ACT_SIGMOID) return y * (1.0f - y);
  return 1.0f;
}
__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct GemmParams {
  const float* __restrict__ A;
  const float* __restrict__ B;
  float* __restrict__ C;
  int M, N, K;
  int64_t a_rs, a_cs, b_rs, b_cs;  // element strides: A(i,k) = A[i*a_rs + k*a_cs], B(k,j) = B[k*b_rs + j*b_cs]
  int64_t ldc;
  int k_per_split;
  int atomic;  // != 0: C += (atomicAdd), else C = (only with a single split)
  // epilogue (single split only): C = tanh(acc) and partials[block] = sum of C^2 over the block's tile; the blocks also
  // zero partials[block + k * n_blocks] up to MKE_LOSS_PARTIALS so that a reader can add all of them up
  double* partials;
  int epi_plain;  // with partials: store acc itself (no tanh) and the per-block sums of its squares
  int gx, gy, gz;  // this problem's grid (k_gemm_f32_batch decodes its linear block index with it)
  int ext;         // != 0: the GemmEpilogue below replaces the partials / epi_plain epilogue
  GemmEpilogue e;
};

// (moved here from mke_gemm.hip in round 4: the attribute step's convolution-backward launch carries the weight-gradient
// product on rider blocks — mke_attr_cnn.hip — and needs the block function in its own translation unit)
// Tall-and-skinny products with the fused epilogue (the attribute step's [5000 x 301] x [301 x 75] and the mapping step's
// [5000 x 75] x [75 x 75]): 64 x 64 tiles give 158 blocks of one wave per SIMD, each walking the whole K through LDS with a
// barrier pair per slab — 17 us for 0.23 GFLOP.  Here a block owns 16 rows x all N (<= 96) columns and its four wavefronts
// split K: every operand a wave needs (<= 20 k-steps of 4: one A scalar and up to six B scalars each) is requested up
// front, straight from global memory (B is L2-resident), multiplied with v_mfma_f32_16x16x4_f32, and the four partial
// products meet once in LDS.  313 blocks for M = 5000, no barrier inside the K loop: 10.5 us for the attribute step's
// product (forcing all 120 loads of a wave ahead of its first MFMA with a scheduling barrier: 12.8 us; guarded instead of
// clamped loads or run-time tile counts: 44-61 us of register shuffling).
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NCT = column tiles of 16 (N <= 16 NCT), KS = k-steps of 4 per wavefront (K <= 16 KS): compile-time, so that the operand
// arrays stay in registers and the MFMA sequence is straight-line code.  Loads are unconditional from clamped addresses
// (a guarded load makes the compiler wait for each one before its select): a k-step past K gets a = 0, so whatever b
// holds there multiplies to nothing; columns past N and rows past M are computed from valid memory and never stored.
// EPI: tanh (unless epi_plain) + per-block sums of squares — the forward products, K <= 16 KS.  !EPI: a K SPLIT of a product
// with a short M (the attribute step's weight gradient [flat, 1]^T dz: 301 x 75 over K = 5000): block (bx, by) owns rows
// [16 bx, 16 bx + 16) and k in [by k_per_split, (by + 1) k_per_split), k_per_split <= 16 KS, and adds its 16 x N partial
// product to C atomically — 304 blocks of ONE round of loads each and 0.39 M atomics, where 64 x 64 tiles through LDS were
// 320 blocks of five dependent K slabs and 1.3 M atomics (a 64-wide column tile holds 11 of N = 75 columns).
// A(i, k) = A[i a_rs + k a_cs] (offsets fit 32 bits: the launcher checks), B n-contiguous.
template <int NCT, int KS, bool EPI>
__device__ __forceinline__ void gemm_tall_block(const GemmParams& p, int bx, int by, float (*s_part)[NCT][4][64]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = bx * 16;
  const int k_lo = EPI ? 0 : by * p.k_per_split;
  const int k_hi = EPI ? p.K : min(p.K, k_lo + p.k_per_split);
  const int k0 = k_lo + 4 * KS * wv + kq;  // this lane's first k
  const int row = min(m0 + r16, p.M - 1);
  const float* ap = p.A + (int64_t)row * p.a_rs;
  const int a_cs = (int)p.a_cs;
  int bcol[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) bcol[c] = min(16 * c + r16, p.N - 1);
  float a[KS], b[KS][NCT];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int kc = min(k0 + 4 * i, p.K - 1);
    a[i] = EPI ? ap[kc] : ap[kc * a_cs];
    const float* bp = p.B + (int64_t)kc * p.b_rs;
#pragma unroll
    for (int c = 0; c < NCT; ++c) b[i][c] = bp[bcol[c]];
  }
  f32x4 acc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const float ai = (k0 + 4 * i < k_hi) ? a[i] : 0.f;
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, b[i][c], acc[c], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_part[wv][c][r][lane] = acc[c][r];
  __syncthreads();
  // wave w finishes column tiles w, w + 4: C/D map of the 16x16 forms: col = lane & 15, row = 4 * (lane >> 4) + reg
  float ssq = 0.f;
  for (int c = wv; c < NCT; c += MKE_BLOCK / 64) {
    const int col = 16 * c + r16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = s_part[0][c][r][lane] + s_part[1][c][r][lane] + s_part[2][c][r][lane] + s_part[3][c][r][lane];
      const int orow = m0 + 4 * kq + r;
      if (col < p.N && orow < p.M) {
        if constexpr (EPI) {
          if (!p.epi_plain) v = tanhf(v);
          p.C[(int64_t)orow * p.ldc + col] = v;
          ssq = fmaf(v, v, ssq);
        } else {
          atomic_add_f32(p.C + (int64_t)orow * p.ldc + col, v);
        }
      }
    }
  }
  if constexpr (EPI) {
    const double tot = block_sum_double(ssq);
    if (tid == 0) {
      const int nb = gridDim.x, bi = blockIdx.x;
      p.partials[bi] = tot;
      for (int k = bi + nb; k < MKE_LOSS_PARTIALS; k += nb) p.partials[k] = 0.0;
    }
  }
}


// dz = inv (g - z coef) (1 - z^2): the backward of the attribute step's batch-wide l2_normalize + tanh, applied where an operand is
// loaded (inv = rsqrt(max(S, eps)), coef = S > eps ? T inv^2 : 0; S = sum z^2, T = sum g . z over the whole batch)
__device__ __forceinline__ float dz_of(float g, float z, float inv, float coef) { return inv * (g - z * coef) * (1.0f - z * z); }

// gemm_tall_block's K-split form (atomic accumulation into C) whose B operand is dz, computed on the way into the MFMAs from g (= p.B)
// and z (same layout): KS k-steps per wavefront in NQ rounds of KS / NQ — the next round's g and z fragments are in flight under the
// current round's MFMAs; all of them at once would be 2 KS NCT live registers.
template <int NCT, int KS, int NQ>
__device__ __forceinline__ void gemm_tall_block_dz(const GemmParams& p, int bx, int by, float (*s_part)[NCT][4][64],
                                                   const float* __restrict__ zmat, float inv, float coef) {
  static_assert(KS % NQ == 0, "rounds of equal length");
  constexpr int KH = KS / NQ;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = bx * 16;
  const int k_lo = by * p.k_per_split;
  const int k_hi = min(p.K, k_lo + p.k_per_split);
  const int k0 = k_lo + 4 * KS * wv + kq;  // this lane's first k
  const int row = min(m0 + r16, p.M - 1);
  const float* ap = p.A + (int64_t)row * p.a_rs;
  const int a_cs = (int)p.a_cs;
  const int64_t z_off = zmat - p.B;
  int bcol[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) bcol[c] = min(16 * c + r16, p.N - 1);
  float a[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) a[i] = ap[min(k0 + 4 * i, p.K - 1) * a_cs];
  f32x4 acc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float g[2][KH][NCT], z[2][KH][NCT];
  auto load_round = [&](int h, int buf) {
#pragma unroll
    for (int i = 0; i < KH; ++i) {
      const float* bp = p.B + (int64_t)min(k0 + 4 * (h * KH + i), p.K - 1) * p.b_rs;
#pragma unroll
      for (int c = 0; c < NCT; ++c) { g[buf][i][c] = bp[bcol[c]]; z[buf][i][c] = bp[z_off + bcol[c]]; }
    }
  };
  load_round(0, 0);
#pragma unroll
  for (int h = 0; h < NQ; ++h) {
    const int buf = h & 1;
    if (h + 1 < NQ) load_round(h + 1, buf ^ 1);
#pragma unroll
    for (int i = 0; i < KH; ++i) {
      const float ai = (k0 + 4 * (h * KH + i) < k_hi) ? a[h * KH + i] : 0.f;
#pragma unroll
      for (int c = 0; c < NCT; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, dz_of(g[buf][i][c], z[buf][i][c], inv, coef), acc[c], 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_part[wv][c][r][lane] = acc[c][r];
  __syncthreads();
  for (int c = wv; c < NCT; c += MKE_BLOCK / 64) {
    const int col = 16 * c + r16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = s_part[0][c][r][lane] + s_part[1][c][r][lane] + s_part[2][c][r][lane] + s_part[3][c][r][lane];
      const int orow = m0 + 4 * kq + r;
      if (col < p.N && orow < p.M) atomic_add_f32(p.C + (int64_t)orow * p.ldc + col, v);
    }
  }
}

int launch_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                    int M, int N, int K, int splits, int accumulate, hipStream_t st, double* tanh_sumsq_partials, int epi_plain);
bool launch_gemm_tallsplit_plus(const float* A0, int64_t a0_rs, int64_t a0_cs, const float* B0, int64_t b0_rs, float* C0, int64_t ldc0,
                                int M0, int N0, int K0, const float* A1, int64_t a1_rs, int64_t a1_cs, const float* B1, int64_t b1_rs,
                                int64_t b1_cs, float* C1, int64_t ldc1, int M1, int N1, int K1, hipStream_t st, int* rc);
int launch_gemm_f32_pair(const float* A0, int64_t a0_rs, int64_t a0_cs, const float* B0, int64_t b0_rs, int64_t b0_cs, float* C0,
                         int64_t ldc0, int M0, int N0, int K0, int splits0, int acc0, const float* A1, int64_t a1_rs, int64_t a1_cs,
                         const float* B1, int64_t b1_rs, int64_t b1_cs, float* C1, int64_t ldc1, int M1, int N1, int K1, int splits1,
                         int acc1, hipStream_t st);
// C (=|+=) epilogue(op(A) op(B)); A(i,k) = A[i*a_rs + k*a_cs], B(k,j) = B[k*b_rs + j*b_cs].  splits <= 0: chosen here.
int launch_gemm_f32_ex(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                       int M, int N, int K, int splits, int accumulate, hipStream_t st, const GemmEpilogue* epi);

}  // namespace mke
