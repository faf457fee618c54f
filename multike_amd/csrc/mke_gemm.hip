// mke_gemm.hip — small dense f32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: plain f32 fma chains), with arbitrary operand strides (so A^T / B^T need no copies) and optional split-K with atomic accumulation.
// It exists so that a whole attribute-view step (conv stack -> dense layer -> loss tail -> backward) can be enqueued by
// ONE native call: the dense layer's three products are [n,4d]x[4d,d], [4d,n]x[n,d] and [n,d]x[d,4d] with n = 5000,
// d = 75 — a few hundred MFLOP each, far below where a library call's launch + dispatch overhead is amortised.
//
// Block = 256 threads = 4 wavefronts, block tile 64 x 64, each wavefront one 32 x 32 MFMA accumulator; K is consumed
// in slabs of 32 staged through LDS (zero-filled at the edges), the next slab prefetched into registers while the
// current one is multiplied.  Measured on MI355X for the attribute step's products (n = 5000, d = 75): 15.9 / 13.0 /
// 9.1 us (was 44 / 31 / 14 us with 16-wide slabs, no prefetch and per-element 64-bit address arithmetic); slab widths
// 16..128 are within 15 % of each other, i.e. the rest is occupancy (158..395 blocks on 256 CUs), not the slab size.
#include "mke_gemm.h"

namespace mke {

typedef float f32x16 __attribute__((ext_vector_type(16)));


// Extended epilogue (mke_gemm.h) for one wavefront's 32 x 32 accumulator whose top-left element is (row0, col0) of C.
// C/D map of the 32x32 MFMA forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ void gemm_epilogue_ext(const GemmEpilogue& e, const f32x16& acc, float* __restrict__ C, int64_t ldc,
                                                  int M, int N, int row0, int col0, int lane, int atomic, int block_linear) {
  const int half = lane >> 5, col = col0 + (lane & 31);
  const float alpha = e.alpha ? *e.alpha : 1.0f;
  float ssq = 0.f, dot = 0.f, csum = 0.f;
  if (col < N) {
    const float bias = e.bias ? e.bias[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = row0 + 4 * half + (reg & 3) + 8 * (reg >> 2);
      if (row < M) {
        float v = fmaf(alpha, acc[reg], bias);
        if (atomic) { atomic_add_f32(C + (int64_t)row * ldc + col, v); continue; }
        v = act_fwd(v, e.act);
        if (e.target) {
          const float d = v - e.target[(int64_t)row * e.ld_target + col];
          ssq = fmaf(d, d, ssq);
          v = e.target_scale * d * act_grad_from_output(v, e.act);
        } else if (e.sumsq) {
          ssq = fmaf(v, v, ssq);
        }
        if (e.dact_y) v *= act_grad_from_output(e.dact_y[(int64_t)row * e.ld_dact + col], e.dact_act);
        if (e.dot_with) dot = fmaf(v, e.dot_with[(int64_t)row * e.ld_dot + col], dot);
        C[(int64_t)row * ldc + col] = v;
        csum += v;
      }
    }
  }
  if (atomic) return;
  if (e.colsum) {
    csum += __shfl_xor(csum, 32, 64);
    if (half == 0 && col < N) atomic_add_f32(e.colsum + col, csum);
  }
  if (e.sumsq) {  // block-uniform
    const double tot = block_sum_double(ssq);
    if (threadIdx.x == 0) atomic_add_f64(e.sumsq + (block_linear % MKE_LOSS_PARTIALS), tot);
  }
  if (e.dot) {
    __syncthreads();
    const double tot = block_sum_double(dot);
    if (threadIdx.x == 0) atomic_add_f64(e.dot + (block_linear % MKE_LOSS_PARTIALS), tot);
  }
}

#define GT 64
#define GK 32

// Block tile 64 x 64 x GK.  Thread -> slab-element map (element e of GE per operand), chosen so that the global reads
// of a wavefront are contiguous whichever way the operand is laid out:
//   A k-contiguous (a_cs == 1):  (m, k) = (tid / GK + e * (256 / GK), tid % GK)     else  (tid % 64, tid / 64 + e * 4)
//   B n-contiguous (b_cs == 1):  (k, n) = (tid / 64 + e * 4,          tid % 64)     else  (tid % GK, tid / GK + e * (256 / GK))
// Addresses are one 64-bit base per operand plus a constant step per element and per slab: 64-bit multiplies are
// quarter rate on CDNA and, at GE of them per slab, were what the first version of this kernel spent its time on.
__device__ __forceinline__ void gemm_block(const GemmParams& p, int bx, int by, int bz) {
  constexpr int GE = GT * GK / MKE_BLOCK;
  // both tiles k-contiguous with a 16-byte aligned row stride: a lane's operands for the 16 MFMAs of a slab are 16
  // consecutive floats = 4 ds_read_b128 per operand, all issued before the first MFMA (the first version read one
  // dword per operand per MFMA and, at one wave per SIMD, exposed the LDS latency 160 times per block)
  __shared__ __attribute__((aligned(16))) float As[GT][GK + 4];
  __shared__ __attribute__((aligned(16))) float Bt[GT][GK + 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = by * GT, n0 = bx * GT;
  const int k_lo = bz * p.k_per_split;
  const int k_hi = min(p.K, k_lo + p.k_per_split);
  const bool akf = p.a_cs == 1, bnf = p.b_cs == 1;
  const int a_m = akf ? tid / GK : tid % GT, a_k = akf ? tid % GK : tid / GT;
  const int a_dm = akf ? MKE_BLOCK / GK : 0, a_dk = akf ? 0 : MKE_BLOCK / GT;
  const int b_k = bnf ? tid / GT : tid % GK, b_n = bnf ? tid % GT : tid / GK;
  const int b_dk = bnf ? MKE_BLOCK / GT : 0, b_dn = bnf ? 0 : MKE_BLOCK / GK;
  const float* pa = p.A + (int64_t)(m0 + a_m) * p.a_rs + (int64_t)(k_lo + a_k) * p.a_cs;
  const float* pb = p.B + (int64_t)(k_lo + b_k) * p.b_rs + (int64_t)(n0 + b_n) * p.b_cs;
  const int64_t a_step = a_dm * p.a_rs + a_dk * p.a_cs, b_step = b_dk * p.b_rs + b_dn * p.b_cs;
  const int64_t a_slab = GK * p.a_cs, b_slab = GK * p.b_rs;
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float ra[GE], rb[GE];
  auto fetch = [&](int k0) {  // one slab into registers, zero beyond the matrix / split edges; all loads independent
    const float* qa = pa;
    const float* qb = pb;
#pragma unroll
    for (int e = 0; e < GE; ++e) {
      ra[e] = (m0 + a_m + e * a_dm < p.M && k0 + a_k + e * a_dk < k_hi) ? *qa : 0.f;
      rb[e] = (k0 + b_k + e * b_dk < k_hi && n0 + b_n + e * b_dn < p.N) ? *qb : 0.f;
      qa += a_step;
      qb += b_step;
    }
    pa += a_slab;
    pb += b_slab;
  };
  fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += GK) {
#pragma unroll
    for (int e = 0; e < GE; ++e) {
      As[a_m + e * a_dm][a_k + e * a_dk] = ra[e];
      Bt[b_n + e * b_dn][b_k + e * b_dk] = rb[e];
    }
    __syncthreads();
    if (k0 + GK < k_hi) fetch(k0 + GK);  // the next slab is in flight during the MFMAs below
    // lane (l31, half) feeds k = half * 16 + j at MFMA j: any assignment that covers the slab's 32 k exactly once is a
    // valid K order; columns past the matrix / split edge are zero-filled, so short tail slabs need no special case
    float4 av[GK / 8], bv[GK / 8];
#pragma unroll
    for (int q = 0; q < GK / 8; ++q) {
      av[q] = *reinterpret_cast<const float4*>(&As[wm * 32 + l31][half * (GK / 2) + q * 4]);
      bv[q] = *reinterpret_cast<const float4*>(&Bt[wn * 32 + l31][half * (GK / 2) + q * 4]);
    }
#pragma unroll
    for (int q = 0; q < GK / 8; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (p.ext) {
    gemm_epilogue_ext(p.e, acc, p.C, p.ldc, p.M, p.N, m0 + wm * 32, n0 + wn * 32, lane, p.atomic, (bz * p.gy + by) * p.gx + bx);
    return;
  }
  // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int col = n0 + wn * 32 + l31;
  float ssq = 0.f;
  if (col < p.N) {
    float* c = p.C + (int64_t)(m0 + wm * 32 + 4 * half) * p.ldc + col;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int dr = (reg & 3) + 8 * (reg >> 2);
      if (m0 + wm * 32 + 4 * half + dr < p.M) {
        if (p.partials) {
          const float v = p.epi_plain ? acc[reg] : tanhf(acc[reg]);
          c[dr * p.ldc] = v;
          ssq = fmaf(v, v, ssq);
        } else if (p.atomic) atomic_add_f32(c + dr * p.ldc, acc[reg]);
        else c[dr * p.ldc] = acc[reg];
      }
    }
  }
  if (p.partials) {  // block-uniform
    const double tot = block_sum_double(ssq);
    if (tid == 0) {
      const int nb = p.gx * p.gy, b = by * p.gx + bx;
      p.partials[b] = tot;
      for (int k = b + nb; k < MKE_LOSS_PARTIALS; k += nb) p.partials[k] = 0.0;
    }
  }
}

template <int NCT, int KS>
__global__ __launch_bounds__(MKE_BLOCK) void k_gemm_tall(const GemmParams p) {
  __shared__ float s_part[MKE_BLOCK / 64][NCT][4][64];
  gemm_tall_block<NCT, KS, true>(p, blockIdx.x, 0, s_part);
}

__global__ __launch_bounds__(MKE_BLOCK) void k_gemm_f32(const GemmParams p) { gemm_block(p, blockIdx.x, blockIdx.y, blockIdx.z); }

// several independent products in one launch (one kernel floor instead of one per product, and their block counts add
// up to fill the chip): block b belongs to the problem whose [first, first + gx*gy*gz) range contains it
#define GEMM_BATCH_MAX 4
struct GemmBatch {
  GemmParams g[GEMM_BATCH_MAX];
  int first[GEMM_BATCH_MAX + 1];
  int n;
};
__global__ __launch_bounds__(MKE_BLOCK) void k_gemm_f32_batch(const GemmBatch b) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < GEMM_BATCH_MAX; ++k)
    if (k < b.n && (int)blockIdx.x >= b.first[k]) i = k;
  const GemmParams& p = b.g[i];
  int r = blockIdx.x - b.first[i];
  const int bx = r % p.gx;
  r /= p.gx;
  gemm_block(p, bx, r % p.gy, r / p.gy);
}

static bool gemm_setup(GemmParams& p, const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C,
                       int64_t ldc, int M, int N, int K, int splits, int accumulate, double* partials, int epi_plain = 0) {
  if (M <= 0 || N <= 0 || K <= 0) return false;
  if (splits < 1 || partials) splits = 1;
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.a_rs = a_rs; p.a_cs = a_cs; p.b_rs = b_rs; p.b_cs = b_cs; p.ldc = ldc;
  int kps = (K + splits - 1) / splits;
  kps = (kps + 3) / 4 * 4;
  p.k_per_split = kps;
  const int nz = (K + kps - 1) / kps;
  p.atomic = (accumulate || nz > 1) ? 1 : 0;
  p.partials = partials;
  p.epi_plain = epi_plain;
  p.ext = 0;
  p.e = GemmEpilogue{};
  p.gx = (N + GT - 1) / GT; p.gy = (M + GT - 1) / GT; p.gz = nz;
  return true;
}

int launch_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                    int M, int N, int K, int splits, int accumulate, hipStream_t st, double* tanh_sumsq_partials, int epi_plain) {
  GemmParams p;
  if (!gemm_setup(p, A, a_rs, a_cs, B, b_rs, b_cs, C, ldc, M, N, K, splits, accumulate, tanh_sumsq_partials, epi_plain)) return MKE_OK;
  // tall and skinny with the fused epilogue: K split over the four wavefronts of a 16-row block (see k_gemm_tall)
  if (tanh_sumsq_partials && a_cs == 1 && b_cs == 1 && N <= 80 && K <= 320 && M >= 1024 && (M + 15) / 16 <= MKE_LOSS_PARTIALS) {
    const dim3 grid((M + 15) / 16);
    if (K <= 80) hipLaunchKernelGGL((k_gemm_tall<5, 5>), grid, dim3(MKE_BLOCK), 0, st, p);
    else if (K <= 160) hipLaunchKernelGGL((k_gemm_tall<5, 10>), grid, dim3(MKE_BLOCK), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_tall<5, 20>), grid, dim3(MKE_BLOCK), 0, st, p);
    return check_launch("k_gemm_tall");
  }
  if (tanh_sumsq_partials && p.gx * p.gy > MKE_LOSS_PARTIALS) { set_error("gemm epilogue: more than %d blocks", MKE_LOSS_PARTIALS); return MKE_E_SHAPE; }
  hipLaunchKernelGGL(k_gemm_f32, dim3(p.gx, p.gy, p.gz), dim3(MKE_BLOCK), 0, st, p);
  return check_launch("k_gemm_f32");
}

// the attribute step's two gradient products in one launch: blocks [0, t.gx t.gz) the K-split tall product t (dW), the
// rest the 64 x 64-tile product g (dflat)
struct GemmTallPlus {
  GemmParams t, g;
  int tall_blocks;
};
// three blocks per CU: at the compiler's own 158 + 20 registers (two) 512 of the launch's 699 blocks were resident — the two
// products (~11 us each alone) took 16.2 us together, 14.9 us now
__global__ __launch_bounds__(MKE_BLOCK, 3) void k_gemm_tallsplit_plus(const GemmTallPlus b) {
  if ((int)blockIdx.x < b.tall_blocks) {   // block-uniform
    __shared__ float s_part[MKE_BLOCK / 64][5][4][64];
    gemm_tall_block<5, 20, false>(b.t, blockIdx.x % b.t.gx, blockIdx.x / b.t.gx, s_part);
    return;
  }
  int r = blockIdx.x - b.tall_blocks;
  const int bx = r % b.g.gx;
  r /= b.g.gx;
  gemm_block(b.g, bx, r % b.g.gy, r / b.g.gy);
}

// C0 += A0^T-style tall product, split over K (M0 x N0, N0 <= 80, atomically into a zeroed / running C0) and C1 = A1 B1 in one
// launch.  Returns false when the first product's shape does not fit (the caller uses launch_gemm_f32_pair).
bool launch_gemm_tallsplit_plus(const float* A0, int64_t a0_rs, int64_t a0_cs, const float* B0, int64_t b0_rs, float* C0, int64_t ldc0,
                                int M0, int N0, int K0, const float* A1, int64_t a1_rs, int64_t a1_cs, const float* B1, int64_t b1_rs,
                                int64_t b1_cs, float* C1, int64_t ldc1, int M1, int N1, int K1, hipStream_t st, int* rc) {
  if (N0 > 80 || N0 < 1 || M0 < 1 || K0 < 1 || (int64_t)K0 * a0_cs >= (1LL << 31) || a0_cs < 1) return false;
  GemmTallPlus b;
  GemmParams& t = b.t;
  t = GemmParams{};
  t.A = A0; t.B = B0; t.C = C0; t.M = M0; t.N = N0; t.K = K0; t.a_rs = a0_rs; t.a_cs = a0_cs; t.b_rs = b0_rs; t.b_cs = 1; t.ldc = ldc0;
  t.k_per_split = 320;   // 16 x KS(20) k per block: 4 wavefronts x 20 k-steps of 4
  t.gx = (M0 + 15) / 16; t.gy = 1; t.gz = (K0 + t.k_per_split - 1) / t.k_per_split; t.atomic = 1;
  b.tall_blocks = t.gx * t.gz;
  int g_blocks = 0;
  if (gemm_setup(b.g, A1, a1_rs, a1_cs, B1, b1_rs, b1_cs, C1, ldc1, M1, N1, K1, 1, 0, nullptr)) g_blocks = b.g.gx * b.g.gy * b.g.gz;
  else b.g = GemmParams{};
  hipLaunchKernelGGL(k_gemm_tallsplit_plus, dim3(b.tall_blocks + g_blocks), dim3(MKE_BLOCK), 0, st, b);
  *rc = check_launch("k_gemm_tallsplit_plus");
  return true;
}

// two independent products in one launch (the attribute step's dW and dflat)
int launch_gemm_f32_pair(const float* A0, int64_t a0_rs, int64_t a0_cs, const float* B0, int64_t b0_rs, int64_t b0_cs, float* C0,
                         int64_t ldc0, int M0, int N0, int K0, int splits0, int acc0, const float* A1, int64_t a1_rs, int64_t a1_cs,
                         const float* B1, int64_t b1_rs, int64_t b1_cs, float* C1, int64_t ldc1, int M1, int N1, int K1, int splits1,
                         int acc1, hipStream_t st) {
  GemmBatch b;
  b.n = 0;
  b.first[0] = 0;
  if (gemm_setup(b.g[b.n], A0, a0_rs, a0_cs, B0, b0_rs, b0_cs, C0, ldc0, M0, N0, K0, splits0, acc0, nullptr)) {
    b.first[b.n + 1] = b.first[b.n] + b.g[b.n].gx * b.g[b.n].gy * b.g[b.n].gz;
    ++b.n;
  }
  if (gemm_setup(b.g[b.n], A1, a1_rs, a1_cs, B1, b1_rs, b1_cs, C1, ldc1, M1, N1, K1, splits1, acc1, nullptr)) {
    b.first[b.n + 1] = b.first[b.n] + b.g[b.n].gx * b.g[b.n].gy * b.g[b.n].gz;
    ++b.n;
  }
  if (b.n == 0) return MKE_OK;
  hipLaunchKernelGGL(k_gemm_f32_batch, dim3(b.first[b.n]), dim3(MKE_BLOCK), 0, st, b);
  return check_launch("k_gemm_f32_batch");
}

// ---------------------------------------------------------------------------------------------------------------------
// Large dense products (the literal auto-encoder: M = 5000, K / N up to 1500; code/literal_encoder.py:63-91).
// Same 64 x 64 block tile / one 32 x 32 accumulator per wavefront as gemm_block — at the f32 MFMA rate (64 cycles per
// v_mfma_f32_32x32x2_f32 per SIMD) a 32 x 32 wave tile needs only 8 bytes of LDS per lane per MFMA, and 64 x 64 blocks
// quantise well on 256 CUs (5000 x 1024: 1264 blocks = 4.94 per CU) — but every global access is a 16-byte load along the
// operand's contiguous dimension (gemm_block's dword loads: 16 per thread and slab = as many vector-memory cycles as the
// slab's MFMAs take; here 4), the LDS image keeps the operand's own orientation (k-contiguous: fragments by ds_read_b128;
// m / n-contiguous: by conflict-free ds_read_b32), and the block -> tile map gives every XCD a contiguous band of row
// tiles so that an A row tile is fetched through ONE L2 instead of eight.
//   A_KC: A[m * lda + k]   (!A_KC: A[k * lda + m])      B_KC: B[n * ldb + k]   (!B_KC: B[k * ldb + n])
struct GemmVParams {
  const float* __restrict__ A;
  const float* __restrict__ B;
  float* __restrict__ C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int k_per_split, atomic;
  int gx, gy, gz;
  GemmEpilogue e;
};

#define GV_SK 36  // row stride (floats) of a k-contiguous 64 x 32 image
#define GV_SM 68  // row stride of an m/n-contiguous 32 x 64 image

template <bool KC>
__device__ __forceinline__ void gv_fetch(const float* __restrict__ base, int64_t ld, int dim_mn, int mn0, int k0, int K, int tid,
                                         float4 (&r)[2]) {
  // Every access is one aligned 16-byte load along the contiguous dimension; the caller guarantees that the contiguous
  // extent rounded up to 4 fits in ld, so a float4 that straddles the logical edge still reads the row's own padding.
  // Rows / columns past the M / N edge are clamped to valid memory and never stored.  NOTHING is computed on the loaded
  // values here: they are only consumed by gv_stage one iteration later, so the wait for them can sit behind the slab's
  // MFMAs (a zero-fill select right after the load pinned the wait in front of them: 175 -> 150 us at 5000 x 1024 x 1500).
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    int mn, k;
    if (KC) { mn = mn0 + (tid >> 3) + 32 * e; k = k0 + 4 * (tid & 7); }
    else    { k = k0 + (tid >> 4) + 16 * e;   mn = mn0 + 4 * (tid & 15); }
    const int kpad = (K + 3) & ~3, mpad = (dim_mn + 3) & ~3;
    const int kc = min(k, (KC ? kpad - 4 : K - 1)), mc = min(mn, (KC ? dim_mn - 1 : mpad - 4));
    r[e] = *reinterpret_cast<const float4*>(KC ? base + (int64_t)mc * ld + kc : base + (int64_t)kc * ld + mc);
  }
}
// the two addresses a thread loads from in the slab starting at k0 (rows / columns clamped once: they do not depend on k)
template <bool KC>
__device__ __forceinline__ void gv_ptrs(const float* __restrict__ base, int64_t ld, int dim_mn, int mn0, int k0, int tid,
                                        const float* (&q)[2]) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (KC) q[e] = base + (int64_t)min(mn0 + (tid >> 3) + 32 * e, dim_mn - 1) * ld + (k0 + 4 * (tid & 7));
    else    q[e] = base + (int64_t)(k0 + (tid >> 4) + 16 * e) * ld + min(mn0 + 4 * (tid & 15), ((dim_mn + 3) & ~3) - 4);
  }
}
// interior slabs (wholly inside [0, K)): plain loads from the running pointers, which then advance by one slab — no
// 64-bit multiply, no clamp in the loop (quarter-rate 64-bit address arithmetic per load cost 8 us of 167 here)
template <bool KC>
__device__ __forceinline__ void gv_fetch_fast(const float* (&q)[2], int64_t slab_step, float4 (&r)[2]) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    r[e] = *reinterpret_cast<const float4*>(q[e]);
    q[e] += slab_step;
  }
}

// registers -> LDS image; on the slab that crosses the K edge (k_hi) what must contribute nothing is zeroed first: whole
// float4s past k_hi, and — k-contiguous operands with K % 4 != 0 — the components past it
template <bool KC>
__device__ __forceinline__ void gv_stage(float* __restrict__ s, int tid, float4 (&r)[2], int k0, int k_hi) {
  if (k0 + 32 > k_hi) {  // block-uniform
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = KC ? k0 + 4 * (tid & 7) : k0 + (tid >> 4) + 16 * e;
      if (KC) {
        r[e].x = k < k_hi ? r[e].x : 0.f; r[e].y = k + 1 < k_hi ? r[e].y : 0.f;
        r[e].z = k + 2 < k_hi ? r[e].z : 0.f; r[e].w = k + 3 < k_hi ? r[e].w : 0.f;
      } else if (k >= k_hi) {
        r[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (KC) *reinterpret_cast<float4*>(s + ((tid >> 3) + 32 * e) * GV_SK + 4 * (tid & 7)) = r[e];
    else    *reinterpret_cast<float4*>(s + ((tid >> 4) + 16 * e) * GV_SM + 4 * (tid & 15)) = r[e];
  }
}
// the 16 operand values of one lane for the slab's 16 MFMAs: k = 16 * half + j
template <bool KC>
__device__ __forceinline__ void gv_frag(const float* __restrict__ s, int w32, int l31, int half, float (&f)[16]) {
  if (KC) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(s + (w32 + l31) * GV_SK + half * 16 + 4 * q);
      f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = s[(half * 16 + j) * GV_SM + w32 + l31];
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(MKE_BLOCK) void k_gemm_vec(const GemmVParams p) {
  // ONE LDS image per operand and two barriers per slab: double-buffering the images (one barrier per slab) measured
  // SLOWER (5000 x 1024 x 1500: 191 vs 174 us) — 37 KB of LDS and 73 registers leave 4 waves per SIMD instead of 8, and
  // it is the number of co-resident blocks that keeps the matrix pipe fed across the barriers.  Raising the wave priority
  // around the 16-MFMA burst (s_setprio 2 ... 0) also measured slower (190 vs 175 us).  PMC: the matrix pipe is busy 74 %
  // of the kernel's cycles at an effective clock of 1.84 GHz under this load (profiles/r02_ae_step.md).  Ablations (wrong
  // results, timing only): without the second barrier 173 us, without the LDS fragment reads 175 us, without the global
  // loads of the loop 143 us — but that variant also multiplies constant operands, and on this part a lower-toggle
  // operand stream clocks higher (MI355X_MICROARCH.md, DVFS), so it bounds the load path's share from above.
  __shared__ __attribute__((aligned(16))) float sA[64 * GV_SK];
  __shared__ __attribute__((aligned(16))) float sB[64 * GV_SK];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1, half = lane >> 5, l31 = lane & 31;
  // block -> (row tile, column tile, K split): XCD x (= block % 8, observed dispatch order) gets a contiguous range of tiles
  const int nb = p.gx * p.gy;
  const int bz = blockIdx.x / nb, t = blockIdx.x - bz * nb;
  const int per = nb >> 3, rem = nb & 7, xcd = t & 7, idx = t >> 3;
  const int L = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
  const int by = L / p.gx, bx = L - by * p.gx;
  const int m0 = by * 64, n0 = bx * 64;
  const int k_lo = bz * p.k_per_split, k_hi = min(p.K, k_lo + p.k_per_split);
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float4 ra[2], rb[2];
  const float* qa[2];
  const float* qb[2];
  gv_ptrs<A_KC>(p.A, p.lda, p.M, m0, k_lo + 32, tid, qa);   // running pointers: the slab AFTER the first
  gv_ptrs<B_KC>(p.B, p.ldb, p.N, n0, k_lo + 32, tid, qb);
  const int64_t a_step = A_KC ? 32 : 32 * p.lda, b_step = B_KC ? 32 : 32 * p.ldb;
  gv_fetch<A_KC>(p.A, p.lda, p.M, m0, k_lo, p.K, tid, ra);
  gv_fetch<B_KC>(p.B, p.ldb, p.N, n0, k_lo, p.K, tid, rb);
  for (int k0 = k_lo; k0 < k_hi; k0 += 32) {
    gv_stage<A_KC>(sA, tid, ra, k0, k_hi);
    gv_stage<B_KC>(sB, tid, rb, k0, k_hi);
    __syncthreads();
    if (k0 + 32 < k_hi) {  // the next slab is in flight during the MFMAs below
      if (k0 + 64 <= p.K) {   // wholly inside the matrix (block-uniform)
        gv_fetch_fast<A_KC>(qa, a_step, ra);
        gv_fetch_fast<B_KC>(qb, b_step, rb);
      } else {                // the slab that crosses the K edge: clamped addresses
        gv_fetch<A_KC>(p.A, p.lda, p.M, m0, k0 + 32, p.K, tid, ra);
        gv_fetch<B_KC>(p.B, p.ldb, p.N, n0, k0 + 32, p.K, tid, rb);
      }
    }
    float fa[16], fb[16];
    gv_frag<A_KC>(sA, wm * 32, l31, half, fa);
    gv_frag<B_KC>(sB, wn * 32, l31, half, fb);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
    __syncthreads();
  }
  gemm_epilogue_ext(p.e, acc, p.C, p.ldc, p.M, p.N, m0 + wm * 32, n0 + wn * 32, lane, p.atomic, blockIdx.x);
}

int launch_gemm_f32_ex(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                       int M, int N, int K, int splits, int accumulate, hipStream_t st, const GemmEpilogue* epi) {
  if (M <= 0 || N <= 0 || K <= 0) return MKE_OK;
  GemmEpilogue e = epi ? *epi : GemmEpilogue{};
  const bool single_only = e.bias || e.act != MKE_ACT_NONE || e.dact_y || e.target || e.sumsq || e.dot || e.colsum;
  const int gx = (N + 63) / 64, gy = (M + 63) / 64;
  if (splits <= 0) {  // enough blocks for ~4 per CU, slices of >= 128 k
    splits = 1;
    if (!single_only) {
      const int want = (4 * 256 + gx * gy - 1) / (gx * gy);
      splits = max(1, min(want, K / 128));
    }
  }
  if (single_only && (splits > 1 || accumulate)) { set_error("gemm epilogue needs a single K split and a plain store"); return MKE_E_SHAPE; }
  if ((e.target && !e.sumsq) || (e.dot_with && !e.dot)) { set_error("gemm epilogue: target needs sumsq, dot_with needs dot"); return MKE_E_NULL; }
  int kps = (K + splits - 1) / splits;
  kps = (kps + 31) / 32 * 32;
  const int gz = (K + kps - 1) / kps;
  const int atomic = (accumulate || gz > 1) ? 1 : 0;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const bool a_kc = a_cs == 1, a_mc = a_rs == 1 && !a_kc, b_nc = b_cs == 1, b_kc = b_rs == 1 && !b_nc;
  // 16-byte loads: base and leading dimension 16-byte aligned, and the contiguous extent rounded up to 4 inside the row
  auto p4 = [](int64_t v) { return (v + 3) & ~(int64_t)3; };
  const bool vec = (a_kc || a_mc) && (b_nc || b_kc) && al16(A) && al16(B) &&
                   (a_kc ? (a_rs % 4 == 0 && p4(K) <= a_rs) : (a_cs % 4 == 0 && p4(M) <= a_cs)) &&
                   (b_kc ? (b_cs % 4 == 0 && p4(K) <= b_cs) : (b_rs % 4 == 0 && p4(N) <= b_rs));
  if (vec) {
    GemmVParams p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = a_kc ? a_rs : a_cs; p.ldb = b_kc ? b_cs : b_rs; p.ldc = ldc;
    p.k_per_split = kps; p.atomic = atomic; p.gx = gx; p.gy = gy; p.gz = gz; p.e = e;
    const dim3 grid((unsigned)(gx * gy * gz));
    if (a_kc && b_kc) hipLaunchKernelGGL((k_gemm_vec<true, true>), grid, dim3(MKE_BLOCK), 0, st, p);
    else if (a_kc) hipLaunchKernelGGL((k_gemm_vec<true, false>), grid, dim3(MKE_BLOCK), 0, st, p);
    else if (b_kc) hipLaunchKernelGGL((k_gemm_vec<false, true>), grid, dim3(MKE_BLOCK), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_vec<false, false>), grid, dim3(MKE_BLOCK), 0, st, p);
    return check_launch("k_gemm_vec");
  }
  GemmParams p;  // any strides / alignment: the dword-load kernel with the same epilogue
  gemm_setup(p, A, a_rs, a_cs, B, b_rs, b_cs, C, ldc, M, N, K, gz, accumulate, nullptr);
  p.ext = 1;
  p.e = e;
  hipLaunchKernelGGL(k_gemm_f32, dim3(p.gx, p.gy, p.gz), dim3(MKE_BLOCK), 0, st, p);
  return check_launch("k_gemm_f32");
}

}  // namespace mke

extern "C" int mke_gemm_f32(const float* A, int64_t a_row_stride, int64_t a_col_stride, const float* B, int64_t b_row_stride,
                            int64_t b_col_stride, float* C, int64_t ldc, int M, int N, int K, int splits, int accumulate,
                            void* stream) {
  using namespace mke;
  if (M < 0 || N < 0 || K < 0) { set_error("mke_gemm_f32: negative size"); return MKE_E_SHAPE; }
  if (M == 0 || N == 0 || K == 0) return MKE_OK;
  if (!A || !B || !C) { set_error("mke_gemm_f32: NULL pointer"); return MKE_E_NULL; }
  if (ldc < N) { set_error("mke_gemm_f32: ldc < N"); return MKE_E_SHAPE; }
  if (splits > 1 && !accumulate) { set_error("mke_gemm_f32: split-K accumulates atomically: pass accumulate=1 and a zeroed (or to-be-added-to) C"); return MKE_E_SHAPE; }
  return launch_gemm_f32_ex(A, a_row_stride, a_col_stride, B, b_row_stride, b_col_stride, C, ldc, M, N, K, splits < 1 ? 1 : splits,
                            accumulate, (hipStream_t)stream, nullptr);
}
