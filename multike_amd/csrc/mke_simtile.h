// mke_simtile.h — the f32 MFMA sweep shared by the k-NN refresh (mke_knn.hip) and the alignment evaluator (mke_eval.hip):
// similarities of a block's 128 rows against a range of 64-column tiles, handed tile by tile to an epilogue that folds
// them into whatever the caller wants (candidate lists, rank counters) — the similarity matrix itself never exists.
//
// A block = 4 wavefronts x one 32-row strip.  The strip is the MFMA A operand and stays in VGPRs for the whole sweep; the
// columns are rows of a row-major [n][ld] matrix, so a 64-column tile is 64 consecutive rows = one contiguous copy, staged
// global -> registers -> LDS (double-buffered when two tiles fit under 64 KB: one barrier per tile) and read back as the B
// operand with two ds_read_b128 per eight v_mfma_f32_32x32x2_f32.  MFMA j of slab s multiplies k = 16 s + 8 (lane >> 5) + j
// for both operands, i.e. every similarity is the SAME k-ordered fma chain wherever it is computed — two evaluations of
// one (row, column) pair are bit-identical, which the evaluator's `sim > gold` comparison relies on.
// -DSIMT_ABLATE=1|2|3 (tools/sweep_ablate.sh, timing only, results wrong): 1 drops the epilogue, 2 also the staging of the next
// tile, 3 also the tile barrier — what is left is the bare ds_read + MFMA loop.
#pragma once
#include "mke_common.h"

namespace mke {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SIMT_BM 128                              // rows per block
#define SIMT_BN_FOR(KS) ((KS) <= 13 ? 64 : 32)   // columns per tile: BN x (kpad + 4) floats must stay under 64 KB

// MFMA A/B operand fragment of one row: frag[s * 8 + j] = row[16 s + 8 half + j]
template <int KS>
__device__ __forceinline__ void simt_load_fragment(const float* __restrict__ row, bool ok, int half, float (&frag)[KS * 8]) {
  const float* ap = row + half * 8;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float4 x = ok ? *reinterpret_cast<const float4*>(ap + s * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 y = ok ? *reinterpret_cast<const float4*>(ap + s * 16 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    frag[s * 8 + 0] = x.x; frag[s * 8 + 1] = x.y; frag[s * 8 + 2] = x.z; frag[s * 8 + 3] = x.w;
    frag[s * 8 + 4] = y.x; frag[s * 8 + 5] = y.y; frag[s * 8 + 6] = y.z; frag[s * 8 + 7] = y.w;
  }
}

// 32 x 32 similarities of two fragments (C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
template <int KS>
__device__ __forceinline__ f32x16 simt_fragment_product(const float (&a)[KS * 8], const float (&b)[KS * 8]) {
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < KS * 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
  return acc;
}

// Sweep tiles [t0, t1) of `cols` ([n_cols][ld] row-major, columns >= dim zero up to KS*16) with the strip fragment `a`.
// epi(acc, col, col_ok) is called once per 32-column group with this lane's column; all 256 threads must call sweep.
template <int KS, class Epilogue>
__device__ __forceinline__ void simt_sweep(const float (&a)[KS * 8], const float* __restrict__ cols, int ld, int n_cols, int t0, int t1,
                                           Epilogue&& epi) {
  constexpr int KP = KS * 16;
  constexpr int BN = SIMT_BN_FOR(KS);
  constexpr int NQ = BN * KP / (4 * MKE_BLOCK);  // float4 per thread per tile
  static_assert(BN * KP % (4 * MKE_BLOCK) == 0, "tile must split evenly into float4 per thread");
  constexpr int NBUF = 2 * BN * (KP + 4) * 4 <= 65536 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][BN][KP + 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  float4 pre[NQ];
  auto fetch = [&](int t) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = (tid + q * MKE_BLOCK) * 4;
      const int c = e / KP, k = e % KP;
      const int col = t * BN + c;
      pre[q] = col < n_cols ? *reinterpret_cast<const float4*>(cols + (int64_t)col * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = (tid + q * MKE_BLOCK) * 4;
      *reinterpret_cast<float4*>(&Bs[buf][e / KP][e % KP]) = pre[q];
    }
  };
  if (t0 < t1) {
    fetch(t0);
    stage(0);
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) % NBUF;
#if !defined(SIMT_ABLATE) || SIMT_ABLATE < 2
    if (t + 1 < t1) fetch(t + 1);  // in flight during the MFMAs below
#endif
    f32x16 acc[BN / 32];
#pragma unroll
    for (int cg = 0; cg < BN / 32; ++cg) {
      acc[cg] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const float* bp = &Bs[buf][cg * 32 + l31][half * 8];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(bp + s * 16);
        const float4 y = *reinterpret_cast<const float4*>(bp + s * 16 + 4);
        acc[cg] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 8 + 0], x.x, acc[cg], 0, 0, 0);
        acc[cg] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 8 + 1], x.y, acc[cg], 0, 0, 0);
        acc[cg] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 8 + 2], x.z, acc[cg], 0, 0, 0);
        acc[cg] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 8 + 3], x.w, acc[cg], 0, 0, 0);
        acc[cg] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 8 + 4], y.x, acc[cg], 0, 0, 0);
        acc[cg] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 8 + 5], y.y, acc[cg], 0, 0, 0);
        acc[cg] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 8 + 6], y.z, acc[cg], 0, 0, 0);
        acc[cg] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 8 + 7], y.w, acc[cg], 0, 0, 0);
      }
    }
#if !defined(SIMT_ABLATE) || SIMT_ABLATE < 2
    if (NBUF == 2 && t + 1 < t1) stage((t + 1 - t0) % NBUF);
#endif
#if !defined(SIMT_ABLATE) || SIMT_ABLATE < 1
#pragma unroll
    for (int cg = 0; cg < BN / 32; ++cg) {
      const int col = t * BN + cg * 32 + l31;
      epi(acc[cg], col, col < n_cols);
    }
#else
    {
      f32x16 z = acc[0];
      for (int cg = 1; cg < BN / 32; ++cg) z += acc[cg];
      if (z[0] == 12345.678f) epi(z, t * BN + l31, true);      // keeps the MFMAs alive; never taken
    }
#endif
    if (NBUF == 1) {
      __syncthreads();  // every wave is done reading the only buffer
      if (t + 1 < t1) stage(0);
    }
#if !defined(SIMT_ABLATE) || SIMT_ABLATE < 3
    __syncthreads();
#endif
  }
}

}  // namespace mke
