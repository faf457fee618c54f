// mke_eval.hip — alignment evaluator on the matrix cores (gfx950): rank of the gold counterpart under the
// normalised inner-product similarity, WITHOUT materialising the n1 x n2 similarity matrix.
//
// What it computes = what code/base/alignment.py:141-163 `calculate_rank` extracts from
// code/base/similarity.py:30-34 `sim` (normalised E1 . E2^T; gold column = row index): for row i,
//   rank_i  = #{ j : sim[i][j] > sim[i][i] }   (position of the gold in the descending order, ties aside)
//   best_i  = argmax_j sim[i][j]               (the `hits1_rest` pair)
// Hits@k = mean(rank < k), MR = mean(rank + 1), MRR = mean(1 / (rank + 1)).  The reference materialises a 60K x 60K
// fp32 matrix (14 GB) and argsorts its rows in 8 worker processes; here a wavefront keeps a 32-row strip of E1 in
// registers, streams 32-column tiles of E2^T, multiplies them with v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered fma
// chain) and folds each 32x32 tile of similarities into per-row counters in the epilogue.
//
// The gold similarity is taken from the SAME MFMA computation (the diagonal tile), so `sim > gold` is an exact
// comparison of identically rounded numbers and a row never counts itself.
#include "mke_common.h"

namespace mke {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define EV_TILES_PER_CHUNK 64  // 2048 columns per (strip, chunk) work item

template <int KH>  // KH = kpad / 2 (k-pairs)
__global__ __launch_bounds__(MKE_BLOCK) void k_align_rank(const float* __restrict__ A, int lda, const float* __restrict__ Bt,
                                                          int64_t ldb, int n1, int n2, int32_t* __restrict__ rank,
                                                          unsigned long long* __restrict__ best) {
  __shared__ float s_gold[MKE_BLOCK / 64][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int strip = blockIdx.x * (MKE_BLOCK / 64) + wv;
  const int row0 = strip * 32;
  const bool strip_live = row0 < n1;
  // A strip fragment: a[kk] = A[row0 + l31][2*kk + half]
  float a[KH];
  {
    const int r = row0 + l31;
    const float* ap = A + (int64_t)(r < n1 ? r : 0) * lda + half;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) a[kk] = (strip_live && r < n1) ? ap[2 * kk] : 0.f;
  }
  auto tile = [&](int col0) -> f32x16 {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* bp = Bt + (int64_t)half * ldb + col0 + l31;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
      const float b = bp[(int64_t)(2 * kk) * ldb];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b, acc, 0, 0, 0);
    }
    return acc;
  };
  // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float gold[16];
  {
    f32x16 d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (strip_live) d = tile(row0);  // diagonal tile: columns row0..row0+31 exist in the padded Bt
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      if (m == l31) s_gold[wv][m] = d[reg];
    }
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) gold[reg] = s_gold[wv][(reg & 3) + 8 * (reg >> 2) + 4 * half];
  }
  int cnt[16];
  float bestv[16];
  int bestc[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) { cnt[reg] = 0; bestv[reg] = -3.0e38f; bestc[reg] = 0; }
  const int ntiles = (n2 + 31) / 32;
  const int t0 = blockIdx.y * EV_TILES_PER_CHUNK;
  const int t1 = min(ntiles, t0 + EV_TILES_PER_CHUNK);
  if (strip_live) {
    for (int t = t0; t < t1; ++t) {
      const int col0 = t * 32;
      const f32x16 acc = tile(col0);
      const bool col_ok = col0 + l31 < n2;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float s = acc[reg];
        cnt[reg] += (col_ok && s > gold[reg]) ? 1 : 0;
        if (col_ok && s > bestv[reg]) { bestv[reg] = s; bestc[reg] = col0 + l31; }
      }
    }
    // fold the 32 lanes of each half (they hold different columns of the same 16 rows)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      int c = cnt[reg];
      float bv = bestv[reg];
      int bc = bestc[reg];
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) {
        c += __shfl_xor(c, off, 64);
        const float ov = __shfl_xor(bv, off, 64);
        const int oc = __shfl_xor(bc, off, 64);
        if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
      }
      const int row = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
      if (l31 == 0 && row < n1) {
        atomicAdd(&rank[row], c);
        // order-preserving key: similarity (monotone uint) in the high word, lowest column wins ties
        unsigned u = __float_as_uint(bv);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)bc);
        atomicMax(&best[row], key);
      }
    }
  }
}

}  // namespace mke

extern "C" int mke_align_rank(const float* emb1, int ld1, const float* emb2_t, int64_t ld2t, int kpad, int64_t n1, int64_t n2,
                              int32_t* rank, uint64_t* best, void* stream) {
  using namespace mke;
  if (n1 < 0 || n2 < 0 || n1 > 0x7FFFFFF0 || n2 > 0x7FFFFFF0) { set_error("bad n1/n2"); return MKE_E_SHAPE; }
  if (n1 == 0) return MKE_OK;
  if (!emb1 || !emb2_t || !rank || !best) { set_error("mke_align_rank: NULL pointer"); return MKE_E_NULL; }
  if (kpad <= 0 || kpad % 16 != 0 || kpad > MKE_MAX_STRIDE || ld1 < kpad) { set_error("kpad must be a multiple of 16 <= %d and <= ld1", MKE_MAX_STRIDE); return MKE_E_SHAPE; }
  const int64_t need = ((n1 > n2 ? n1 : n2) + 31) / 32 * 32;
  if (ld2t < need) { set_error("emb2_t row length %lld < %lld (columns must be zero-padded to a multiple of 32 covering max(n1,n2))", (long long)ld2t, (long long)need); return MKE_E_SHAPE; }
  if (n2 < n1) { set_error("gold column = row index needs n2 >= n1"); return MKE_E_SHAPE; }
  const int strips = (int)((n1 + 31) / 32);
  const int ntiles = (int)((n2 + 31) / 32);
  dim3 grid((strips + MKE_BLOCK / 64 - 1) / (MKE_BLOCK / 64), (ntiles + EV_TILES_PER_CHUNK - 1) / EV_TILES_PER_CHUNK);
  hipStream_t st = (hipStream_t)stream;
#define EV_CASE(K)                                                                                                    \
  case K:                                                                                                             \
    hipLaunchKernelGGL((k_align_rank<K / 2>), grid, dim3(MKE_BLOCK), 0, st, emb1, ld1, emb2_t, ld2t, (int)n1, (int)n2, \
                       rank, (unsigned long long*)best);                                                              \
    break;
  switch (kpad) {
    EV_CASE(16) EV_CASE(32) EV_CASE(48) EV_CASE(64) EV_CASE(80) EV_CASE(96) EV_CASE(112) EV_CASE(128) EV_CASE(160)
    EV_CASE(192) EV_CASE(208) EV_CASE(256) EV_CASE(320)
    default:
      set_error("unsupported kpad %d", kpad);
      return MKE_E_UNSUPPORTED;
  }
#undef EV_CASE
  return check_launch("k_align_rank");
}
