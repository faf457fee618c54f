// mke_gemm.hip — small dense f32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fma
// chain), with arbitrary operand strides (so A^T / B^T need no copies) and optional split-K with atomic accumulation.
// It exists so that a whole attribute-view step (conv stack -> dense layer -> loss tail -> backward) can be enqueued by
// ONE native call: the dense layer's three products are [n,4d]x[4d,d], [4d,n]x[n,d] and [n,d]x[d,4d] with n = 5000,
// d = 75 — a few hundred MFLOP each, far below where a library call's launch + dispatch overhead is amortised.
//
// Block = 256 threads = 4 wavefronts, block tile 64 x 64, each wavefront one 32 x 32 MFMA accumulator; K is consumed
// in slabs of 16 staged through LDS (zero-filled at the edges).
#include "mke_common.h"

namespace mke {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
  const float* __restrict__ A;
  const float* __restrict__ B;
  float* __restrict__ C;
  int M, N, K;
  int64_t a_rs, a_cs, b_rs, b_cs;  // element strides: A(i,k) = A[i*a_rs + k*a_cs], B(k,j) = B[k*b_rs + j*b_cs]
  int64_t ldc;
  int k_per_split;
  int atomic;  // != 0: C += (atomicAdd), else C = (only with a single split)
};

#define GT 64
#define GK 16

__global__ __launch_bounds__(MKE_BLOCK) void k_gemm_f32(const GemmParams p) {
  __shared__ float As[GT][GK + 1];
  __shared__ float Bs[GK][GT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int k_lo = blockIdx.z * p.k_per_split;
  const int k_hi = min(p.K, k_lo + p.k_per_split);
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k0 = k_lo; k0 < k_hi; k0 += GK) {
    // stage A[64 x 16] and B[16 x 64]: 1024 elements each, 4 per thread
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * MKE_BLOCK;
      {  // A: walk k fastest when A is k-contiguous, m fastest otherwise (keeps the global reads coalesced)
        int mi, ki;
        if (p.a_cs == 1) { mi = idx / GK; ki = idx % GK; } else { mi = idx % GT; ki = idx / GT; }
        const int gm = m0 + mi, gk = k0 + ki;
        As[mi][ki] = (gm < p.M && gk < k_hi) ? p.A[gm * p.a_rs + gk * p.a_cs] : 0.f;
      }
      {
        int ki, ni;
        if (p.b_cs == 1) { ki = idx / GT; ni = idx % GT; } else { ki = idx % GK; ni = idx / GK; }
        const int gk = k0 + ki, gn = n0 + ni;
        Bs[ki][ni] = (gk < k_hi && gn < p.N) ? p.B[gk * p.b_rs + gn * p.b_cs] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK / 2; ++kk) {
      const float a = As[wm * 32 + l31][2 * kk + half];
      const float b = Bs[2 * kk + half][wn * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int col = n0 + wn * 32 + l31;
  if (col < p.N) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
      if (row < p.M) {
        float* c = p.C + row * p.ldc + col;
        if (p.atomic) atomic_add_f32(c, acc[reg]);
        else *c = acc[reg];
      }
    }
  }
}

int launch_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                    int M, int N, int K, int splits, int accumulate, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0) return MKE_OK;
  if (splits < 1) splits = 1;
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.a_rs = a_rs; p.a_cs = a_cs; p.b_rs = b_rs; p.b_cs = b_cs; p.ldc = ldc;
  int kps = (K + splits - 1) / splits;
  kps = (kps + GK - 1) / GK * GK;
  p.k_per_split = kps;
  const int nz = (K + kps - 1) / kps;
  p.atomic = (accumulate || nz > 1) ? 1 : 0;
  dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, nz);
  hipLaunchKernelGGL(k_gemm_f32, grid, dim3(MKE_BLOCK), 0, st, p);
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
  return launch_gemm_f32(A, a_row_stride, a_col_stride, B, b_row_stride, b_col_stride, C, ldc, M, N, K, splits, accumulate,
                         (hipStream_t)stream);
}
