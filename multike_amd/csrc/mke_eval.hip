// mke_eval.hip — alignment evaluator on the matrix cores (gfx950): rank of the gold counterpart under the
// normalised inner-product similarity, WITHOUT materialising the n1 x n2 similarity matrix.
//
// What it computes = what code/base/alignment.py:141-163 `calculate_rank` extracts from
// code/base/similarity.py:30-34 `sim` (normalised E1 . E2^T; gold column = row index): for row i,
//   rank_i  = #{ j : sim[i][j] > sim[i][i] }   (position of the gold in the descending order, ties aside)
//   best_i  = argmax_j sim[i][j]               (the `hits1_rest` pair)
// Hits@k = mean(rank < k), MR = mean(rank + 1), MRR = mean(1 / (rank + 1)).  The reference materialises a 60K x 60K
// fp32 matrix (14 GB) and argsorts its rows in 8 worker processes; here a wavefront keeps a 32-row strip of E1 in
// registers, the block streams 64-column tiles of E2 through LDS (mke_simtile.h), multiplies them with
// v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered fma chain) and folds each tile of similarities into per-row counters.
//
// The gold similarity is taken from the SAME MFMA computation (the diagonal tile), so `sim > gold` is an exact
// comparison of identically rounded numbers and a row never counts itself.
#include "mke_simtile.h"

namespace mke {

struct AlignRankParams {
  const float* __restrict__ emb1;  // [n1][ld1]
  int ld1;
  const float* __restrict__ emb2;  // [n2][ld2]
  int ld2;
  int n1, n2;
  int tiles_per_chunk;
  int32_t* __restrict__ rank;
  int32_t* __restrict__ ties;   // nullable: #{ j : sim[i][j] == sim[i][i] } including j = i
  unsigned long long* __restrict__ best;
};

template <int KS, bool TIES>  // kpad / 16; TIES: also count the columns that tie with the gold
__global__ __launch_bounds__(MKE_BLOCK) void k_align_rank(const AlignRankParams p) {
  __shared__ float s_gold[MKE_BLOCK / 64][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int strip0 = blockIdx.x * SIMT_BM + wv * 32;
  float a[KS * 8];
  float gold[16];
  {
    const int r = strip0 + l31;
    const bool ok = r < p.n1;
    simt_load_fragment<KS>(p.emb1 + (int64_t)(ok ? r : 0) * p.ld1, ok, half, a);
    // gold similarity of row i = column i, from the SAME fma chain as the sweep computes it (bit-identical, so the row
    // never counts itself): the strip's 32 gold columns as a B fragment, diagonal of the 32 x 32 product
    float b[KS * 8];
    simt_load_fragment<KS>(p.emb2 + (int64_t)(ok ? r : 0) * p.ld2, ok, half, b);  // n2 >= n1: the row exists
    const f32x16 d = simt_fragment_product<KS>(a, b);
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
  int eq[TIES ? 16 : 1];
  float bestv[16];
  int bestc[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) { cnt[reg] = 0; bestv[reg] = -3.0e38f; bestc[reg] = 0; if (TIES) eq[reg] = 0; }
  const int ntiles = (p.n2 + SIMT_BN_FOR(KS) - 1) / SIMT_BN_FOR(KS);
  const int t0 = blockIdx.y * p.tiles_per_chunk;
  const int t1 = min(ntiles, t0 + p.tiles_per_chunk);
  simt_sweep<KS>(a, p.emb2, p.ld2, p.n2, t0, t1, [&](const f32x16& acc, int col, bool col_ok) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const float s = acc[reg];
      cnt[reg] += (col_ok && s > gold[reg]) ? 1 : 0;
      if (TIES) eq[reg] += (col_ok && s == gold[reg]) ? 1 : 0;
      if (col_ok && s > bestv[reg]) { bestv[reg] = s; bestc[reg] = col; }  // columns ascend: the lowest column wins a tie
    }
  });
  // fold the 32 lanes of each half (they hold different columns of the same 16 rows)
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    int c = cnt[reg];
    int ce = TIES ? eq[reg] : 0;
    float bv = bestv[reg];
    int bc = bestc[reg];
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      c += __shfl_xor(c, off, 64);
      if (TIES) ce += __shfl_xor(ce, off, 64);
      const float ov = __shfl_xor(bv, off, 64);
      const int oc = __shfl_xor(bc, off, 64);
      if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
    }
    const int row = strip0 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
    if (l31 == 0 && row < p.n1 && t0 < t1) {
      atomicAdd(&p.rank[row], c);
      if (TIES) atomicAdd(&p.ties[row], ce);
      // order-preserving key: similarity (monotone uint) in the high word, lowest column wins ties
      unsigned u = __float_as_uint(bv);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)bc);
      atomicMax(&p.best[row], key);
    }
  }
}

}  // namespace mke

extern "C" int mke_align_rank(const float* emb1, int ld1, const float* emb2, int ld2, int kpad, int64_t n1, int64_t n2,
                              int32_t* rank, int32_t* ties, uint64_t* best, void* stream) {
  using namespace mke;
  if (n1 < 0 || n2 < 0 || n1 > 0x7FFFFF00 || n2 > 0x7FFFFF00) { set_error("bad n1/n2"); return MKE_E_SHAPE; }
  if (n1 == 0) return MKE_OK;
  if (!emb1 || !emb2 || !rank || !best) { set_error("mke_align_rank: NULL pointer"); return MKE_E_NULL; }
  if (kpad <= 0 || kpad % 16 != 0 || kpad > MKE_MAX_STRIDE || ld1 < kpad || ld2 < kpad || ld1 % 4 != 0 || ld2 % 4 != 0) {
    set_error("kpad must be a multiple of 16 <= %d and <= ld1, ld2 (both multiples of 4)", MKE_MAX_STRIDE);
    return MKE_E_SHAPE;
  }
  if (n2 < n1) { set_error("gold column = row index needs n2 >= n1"); return MKE_E_SHAPE; }
  AlignRankParams p;
  p.emb1 = emb1; p.ld1 = ld1; p.emb2 = emb2; p.ld2 = ld2; p.n1 = (int)n1; p.n2 = (int)n2; p.rank = rank; p.ties = ties;
  p.best = (unsigned long long*)best;
  const int bn = SIMT_BN_FOR(kpad / 16);
  const int ntiles = (int)((n2 + bn - 1) / bn);
  const int row_blocks = (int)((n1 + SIMT_BM - 1) / SIMT_BM);
  // enough (row block, column chunk) items to fill the chip several times over; a chunk is at least 16 tiles
  int chunks = (6144 + row_blocks - 1) / row_blocks;
  if (chunks > (ntiles + 15) / 16) chunks = (ntiles + 15) / 16;
  if (chunks < 1) chunks = 1;
  p.tiles_per_chunk = (ntiles + chunks - 1) / chunks;
  dim3 grid((unsigned)row_blocks, (unsigned)((ntiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk));
  hipStream_t st = (hipStream_t)stream;
#define EV_CASE(K)                                                                \
  case K:                                                                         \
    if (ties) hipLaunchKernelGGL((k_align_rank<K / 16, true>), grid, dim3(MKE_BLOCK), 0, st, p);   \
    else hipLaunchKernelGGL((k_align_rank<K / 16, false>), grid, dim3(MKE_BLOCK), 0, st, p);       \
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
