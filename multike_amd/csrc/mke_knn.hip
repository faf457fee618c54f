// mke_knn.hip — truncated-sampling k-NN refresh without the similarity matrix (gfx950).
//
// What it computes = code/base/batch.py:119-150 (`generate_neighbours` / `find_neighbours`): for every useful entity of
// one KG the k = 2 % columns of `sim = E . E^T` (already row-normalised relation-view rows, so cosine) with the largest
// values, the entity itself included, as an unordered set.  The reference materialises the n x n matrix and
// argpartitions every row; at n = 100K that is 40 GB written and read back.  Here:
//
//   k_sim_select   the similarity tile never leaves the registers: a block owns 128 rows (4 wavefronts x one 32-row strip
//                  held as MFMA A operands in VGPRs), streams 64-column tiles of E (64 consecutive rows = one contiguous
//                  copy) through LDS, multiplies with v_mfma_f32_32x32x2_f32 (plain f32 fma chains) and, in the
//                  epilogue, appends the columns whose similarity exceeds the row's threshold tau to the row's candidate
//                  list: a ballot per accumulator register gives every hit its slot (the 32 lanes of a half-wave hold 32
//                  columns of ONE row), so a row's candidates of one block are written in column order with no atomics.
//                  The columns are split over gridDim.y segments (a row-owner block sweeping all columns alone would
//                  leave a 2x tail: 782 blocks on 768 slots); segment s of a row has its own counter and seg_cap slots.
//   k_topk_rows    exact k-th largest of a short list held in LDS (<= 4096 values: the candidates of a row, or a row of
//                  sample similarities when only the threshold is wanted): 4-pass byte-wise radix select on the
//                  order-preserving integer image of the floats, then an ordered compaction (block scan) of the values
//                  above the k-th plus the first ties — the output is deterministic and in candidate (= column) order.
//
// The threshold of a row is the m-th largest of its similarities to a fixed column sample (caller: a small library GEMM
// + k_topk_rows), chosen so that ~1.4 k columns pass; a row whose estimate came out too tight (< k hits) or whose
// segment overflowed is flagged and redone by the caller at full width.  The result is the exact top-k set.
#include "mke_simtile.h"

namespace mke {

#define KNN_MAX_LIST 4096

struct SimSelectParams {
  const float* __restrict__ emb;  // [n][ld], columns >= dim zero up to kpad
  int ld;
  int n_cols;             // columns = rows 0..n_cols-1 of emb
  int row_lo, row_hi;     // rows handled by this launch
  const float* __restrict__ tau;  // [row_hi - row_lo]
  int n_seg, seg_cap, tiles_per_seg;
  mke_candidate* __restrict__ cand;  // [rows][n_seg][seg_cap] (column, similarity) pairs: one 8-byte store per hit
  int32_t* __restrict__ seg_count;   // [rows][n_seg]
};

// KS <= 5 (dim <= 80): three blocks per CU (the 43 KB of LDS allow it): the compiler's own choice was 180 + 32 registers = two —
// 18.3 -> 16.3 ms for 100K x 100K (98 TFLOP/s); the extra wavefront per SIMD covers the epilogue's compare / append between
// MFMA bursts.  Wider rows would only fit by spilling (10 .. 330 registers at KS 6 .. 16): left to the compiler.
template <int KS>  // kpad / 16
__global__ __launch_bounds__(MKE_BLOCK, KS <= 5 ? 3 : 1) void k_sim_select(const SimSelectParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int strip0 = p.row_lo + blockIdx.x * SIMT_BM + wv * 32;
  float a[KS * 8];
  {
    const int r = strip0 + l31;
    const bool ok = r < p.row_hi;
    simt_load_fragment<KS>(p.emb + (int64_t)(ok ? r : p.row_lo) * p.ld, ok, half, a);
  }
  float tauR[16];
  int cnt[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int r = strip0 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
    tauR[reg] = r < p.row_hi ? p.tau[r - p.row_lo] : 3.0e38f;
    cnt[reg] = 0;
  }
  const int seg = blockIdx.y;
  const int ntiles = (p.n_cols + SIMT_BN_FOR(KS) - 1) / SIMT_BN_FOR(KS);
  const int t0 = seg * p.tiles_per_seg;
  const int t1 = min(ntiles, t0 + p.tiles_per_seg);
  const unsigned lt = (1u << l31) - 1u;
  // candidate slot of (row of accumulator register reg, position) as a 32-bit BYTE offset from p.cand (the launcher
  // bounds the slot count of a launch to 2^29): the store then takes the scalar base + a 32-bit vector offset
  const unsigned row_stride_b = (unsigned)(p.n_seg * p.seg_cap) * 8u;
  const unsigned base0_b = (unsigned)(((strip0 - p.row_lo + 4 * half) * p.n_seg + seg) * p.seg_cap) * 8u;
  char* const cand_b = reinterpret_cast<char*>(p.cand);
  simt_sweep<KS>(a, p.emb, p.ld, p.n_cols, t0, t1, [&](const f32x16& acc, int col, bool col_ok) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const float v = col_ok ? acc[reg] : -3.0e38f;  // one select, so that the compare's mask IS the ballot
      const bool hit = v > tauR[reg];
      const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
      if (m == 0) continue;  // wave-uniform
      const unsigned mh = half ? (unsigned)(m >> 32) : (unsigned)m;  // the 32 lanes of a half hold 32 columns of ONE row
      const int pos = cnt[reg] + __popc(mh & lt);
      if (hit && pos < p.seg_cap) {
        mke_candidate c;
        c.idx = col;
        c.sim = acc[reg];
        *reinterpret_cast<mke_candidate*>(cand_b + (base0_b + (unsigned)((reg & 3) + 8 * (reg >> 2)) * row_stride_b + (unsigned)pos * 8u)) = c;
      }
      cnt[reg] += __popc(mh);
    }
  });
  if (l31 == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = strip0 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
      if (r < p.row_hi) p.seg_count[(int64_t)(r - p.row_lo) * p.n_seg + seg] = cnt[reg];
    }
  }
}

// Similarities of rows [row_lo, row_hi) to a short list of sample rows, written out ([rows][n_samp]): the input of the
// threshold estimate.  Same sweep, same fma chains as k_sim_select, so the thresholds are taken from numbers the main
// pass would reproduce bit for bit — and the refresh does not depend on a library GEMM's choice of algorithm.
struct SimSampleParams {
  const float* __restrict__ emb;   // [n][ld]
  int ld;
  int row_lo, row_hi;
  const float* __restrict__ samp;  // [n_samp][ld_s] sample rows (same padding as emb)
  int ld_s, n_samp;
  int tiles_per_chunk;
  float* __restrict__ out;  // [row_hi - row_lo][n_samp]
};

template <int KS>
__global__ __launch_bounds__(MKE_BLOCK) void k_sim_sample(const SimSampleParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int strip0 = p.row_lo + blockIdx.x * SIMT_BM + wv * 32;
  float a[KS * 8];
  {
    const int r = strip0 + l31;
    const bool ok = r < p.row_hi;
    simt_load_fragment<KS>(p.emb + (int64_t)(ok ? r : p.row_lo) * p.ld, ok, half, a);
  }
  const int ntiles = (p.n_samp + SIMT_BN_FOR(KS) - 1) / SIMT_BN_FOR(KS);
  const int t0 = blockIdx.y * p.tiles_per_chunk;
  const int t1 = min(ntiles, t0 + p.tiles_per_chunk);
  float* o = p.out + (int64_t)(strip0 - p.row_lo + 4 * half) * p.n_samp;
  simt_sweep<KS>(a, p.samp, p.ld_s, p.n_samp, t0, t1, [&](const f32x16& acc, int col, bool col_ok) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int dr = (reg & 3) + 8 * (reg >> 2);
      if (col_ok && strip0 + 4 * half + dr < p.row_hi) o[(int64_t)dr * p.n_samp + col] = acc[reg];
    }
  });
}

// order-preserving integer image of a float: larger float <=> larger unsigned
__device__ __forceinline__ unsigned float_key(float v) {
  unsigned u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0u;  // -0 and +0 compare equal as floats: one key
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct TopkParams {
  const mke_candidate* __restrict__ cand;  // [rows][n_seg][seg_cap] pairs; or NULL and:
  const float* __restrict__ vals;     // [rows][n_seg][seg_cap]
  const int32_t* __restrict__ idx;    // nullable: same shape; NULL => the position in the list is the index
  const int32_t* __restrict__ seg_count;  // nullable: [rows][n_seg] valid entries per segment; NULL => seg_cap each
  int n_seg, seg_cap, k;
  const int32_t* __restrict__ id_map;  // nullable: out = id_map[index]
  int32_t* __restrict__ out_idx;       // nullable: [rows][k]
  float* __restrict__ out_kth;         // nullable: [rows] k-th largest value
  int32_t* __restrict__ status;        // nullable: [rows] 0 ok, 1 fewer than k entries, 2 a segment overflowed
};

// WITH_IDX = false: threshold mode (out_idx == NULL), the indices are not staged: 20 KB of LDS per block instead of 36,
// twice the resident blocks of this latency-bound kernel (0.31 -> 0.17 ms per 16384 x 4096 threshold block)
template <bool WITH_IDX>
__global__ __launch_bounds__(MKE_BLOCK) void k_topk_rows(const TopkParams p) {
  static_assert(MKE_BLOCK == 256, "one histogram bin per thread");
  __shared__ unsigned s_key[KNN_MAX_LIST];
  __shared__ int s_idx[WITH_IDX ? KNN_MAX_LIST : 1];
  __shared__ int s_hist[MKE_BLOCK / 64][256];
  __shared__ unsigned s_red[8];
  __shared__ int s_off[17];
  __shared__ int s_wave[MKE_BLOCK / 64];
  __shared__ unsigned s_prefix;
  __shared__ int s_need, s_bad;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int tot = 0, bad = 0;
    for (int s = 0; s < p.n_seg; ++s) {
      int c = p.seg_count ? p.seg_count[row * p.n_seg + s] : p.seg_cap;
      if (c > p.seg_cap) { bad = 2; c = p.seg_cap; }
      s_off[s] = tot;
      tot += c;
    }
    s_off[p.n_seg] = tot;
    if (!bad && tot < p.k) bad = 1;
    s_bad = bad;
  }
  __syncthreads();
  if (s_bad) {  // block-uniform
    if (tid == 0) {
      if (p.status) p.status[row] = s_bad;
      if (p.out_kth) p.out_kth[row] = -3.0e38f;
    }
    return;
  }
  const int total = s_off[p.n_seg];
  for (int s = 0; s < p.n_seg; ++s) {
    const int n = s_off[s + 1] - s_off[s];
    const int64_t base = (row * p.n_seg + s) * (int64_t)p.seg_cap;
    if (p.cand) {
      for (int i = tid; i < n; i += MKE_BLOCK) {
        const mke_candidate c = p.cand[base + i];
        s_key[s_off[s] + i] = float_key(c.sim);
        if (WITH_IDX) s_idx[s_off[s] + i] = c.idx;
      }
    } else {
      for (int i = tid; i < n; i += MKE_BLOCK) {
        s_key[s_off[s] + i] = float_key(p.vals[base + i]);
        if (WITH_IDX) s_idx[s_off[s] + i] = p.idx ? p.idx[base + i] : s * p.seg_cap + i;
      }
    }
  }
  __syncthreads();  // list position j was written by thread (j - s_off[seg]) % 256, not j % 256
  // The keys of one list share their leading bits (similarities above a threshold: same sign, one or two exponents), and
  // a byte-wise pass over shared bits would pile every LDS atomic onto one bin: find the highest bit in which the largest
  // and the smallest key differ and select on the bits below it only.
  unsigned kmin = 0xFFFFFFFFu, kmax = 0u;
  for (int i = tid; i < total; i += MKE_BLOCK) {
    const unsigned kx = s_key[i];
    kmin = min(kmin, kx);
    kmax = max(kmax, kx);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, off, 64));
    kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, off, 64));
  }
  const int lane = tid & 63, wv = tid >> 6;
  if (lane == 0) { s_red[wv] = kmin; s_red[4 + wv] = kmax; }
  __syncthreads();
  kmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
  kmax = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
  int hi = kmin == kmax ? 0 : 32 - __clz(kmin ^ kmax);  // number of low bits still undecided
  if (tid == 0) { s_prefix = hi >= 32 ? 0u : (kmax >> hi) << hi; s_need = p.k; }
  __syncthreads();
  // radix select on the undecided bits, most significant digit first; a private histogram per wavefront (4x fewer
  // collisions), bins summed and scanned from the top by all 256 threads
  while (hi > 0) {
    const int w = min(8, hi), shift = hi - w;
#pragma unroll
    for (int q = 0; q < MKE_BLOCK / 64; ++q) s_hist[q][tid] = 0;
    __syncthreads();
    const unsigned pre = s_prefix;
    const int need = s_need;
    for (int i = tid; i < total; i += MKE_BLOCK) {
      const unsigned kx = s_key[i];
      if (hi >= 32 || (kx >> hi) == (pre >> hi)) atomicAdd(&s_hist[wv][(kx >> shift) & ((1u << w) - 1u)], 1);
    }
    __syncthreads();
    // thread t owns digit 255 - t: inclusive scan from the largest digit down
    const int dgt = 255 - tid;
    const int h = s_hist[0][dgt] + s_hist[1][dgt] + s_hist[2][dgt] + s_hist[3][dgt];
    int incl = h;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    for (int q = 0; q < wv; ++q) incl += s_wave[q];
    if (incl >= need && incl - h < need) {  // exactly one thread: the digit where the count from the top reaches `need`
      s_prefix = pre | ((unsigned)dgt << shift);
      s_need = need - (incl - h);
    }
    hi = shift;
    __syncthreads();
  }
  const unsigned kth = s_prefix;
  const int ties = s_need;  // how many of the keys equal to kth belong to the top k
  if (tid == 0) {
    if (p.out_kth) p.out_kth[row] = key_float(kth);
    if (p.status) p.status[row] = 0;
  }
  if (!WITH_IDX || !p.out_idx) return;
  // ordered compaction: thread t owns a contiguous run; packed (greater | equal << 16) counts scanned over the block
  const int per = (total + MKE_BLOCK - 1) / MKE_BLOCK;
  const int lo = min(total, tid * per), up = min(total, lo + per);
  int mine = 0;
  for (int i = lo; i < up; ++i) {
    const unsigned kx = s_key[i];
    mine += (kx > kth ? 1 : 0) + (kx == kth ? 65536 : 0);
  }
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  __syncthreads();  // s_wave is reused
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  int before = incl - mine;
  for (int w = 0; w < wv; ++w) before += s_wave[w];
  int gt = before & 0xFFFF, eq = before >> 16;
  int32_t* o = p.out_idx + row * (int64_t)p.k;
  for (int i = lo; i < up; ++i) {
    const unsigned kx = s_key[i];
    if (kx > kth || (kx == kth && eq < ties)) {
      const int id = s_idx[i];
      o[gt + min(eq, ties)] = p.id_map ? p.id_map[id] : id;
    }
    gt += kx > kth ? 1 : 0;
    eq += kx == kth ? 1 : 0;
  }
}


// Exact top k of LONG rows (whole similarity rows: a short KG, or a row of the main pass whose threshold estimate was off —
// up to n = 100K+ values, far beyond the 4096 keys k_topk_rows holds in LDS).  One block per row; the row stays in global
// memory (L2) and is streamed five times: four byte-wise radix-select passes (private histogram per wavefront, the 256
// threads scan the bins from the top) fix the k-th largest key and how many of its ties belong to the top k, a fifth pass
// compacts the columns above it plus the first ties in COLUMN order (block scan per 256-column chunk) — the output is a
// deterministic function of the row.  Replaces torch.topk (a library call: its first use in a run cost a 170 ms
// compilation / initialisation on a fresh box).
struct TopkLongParams {
  const float* __restrict__ vals;   // [rows][ld]
  int64_t ld;
  int n, k;
  const int32_t* __restrict__ id_map;  // nullable
  int32_t* __restrict__ out_idx;       // [rows][k], column order
};

__global__ __launch_bounds__(MKE_BLOCK) void k_topk_long(const TopkLongParams p) {
  static_assert(MKE_BLOCK == 256, "one histogram bin per thread");
  __shared__ int s_hist[MKE_BLOCK / 64][256];
  __shared__ int s_wave[MKE_BLOCK / 64];
  __shared__ unsigned s_prefix;
  __shared__ int s_need;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* __restrict__ v = p.vals + (int64_t)blockIdx.x * p.ld;
  if (tid == 0) { s_prefix = 0u; s_need = p.k; }
  __syncthreads();
  for (int hi = 32; hi > 0; hi -= 8) {
    const int shift = hi - 8;
#pragma unroll
    for (int q = 0; q < MKE_BLOCK / 64; ++q) s_hist[q][tid] = 0;
    __syncthreads();
    const unsigned pre = s_prefix;
    const int need = s_need;
    for (int i = tid; i < p.n; i += MKE_BLOCK) {
      const unsigned kx = float_key(v[i]);
      if (hi >= 32 || (kx >> hi) == (pre >> hi)) atomicAdd(&s_hist[wv][(kx >> shift) & 255u], 1);
    }
    __syncthreads();
    const int dgt = 255 - tid;                       // thread t owns digit 255 - t: inclusive scan from the largest digit down
    const int h = s_hist[0][dgt] + s_hist[1][dgt] + s_hist[2][dgt] + s_hist[3][dgt];
    int incl = h;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    for (int q = 0; q < wv; ++q) incl += s_wave[q];
    if (incl >= need && incl - h < need) {           // exactly one thread: the digit where the count from the top reaches `need`
      s_prefix = pre | ((unsigned)dgt << shift);
      s_need = need - (incl - h);
    }
    __syncthreads();
  }
  const unsigned kth = s_prefix;
  const int ties = s_need;                           // how many of the keys equal to kth belong to the top k
  int32_t* __restrict__ o = p.out_idx + (int64_t)blockIdx.x * p.k;
  int gt_run = 0, eq_run = 0;                        // block-uniform running counts of the chunks before this one
  for (int base = 0; base < p.n; base += MKE_BLOCK) {
    const int i = base + tid;
    const unsigned kx = i < p.n ? float_key(v[i]) : 0u;
    const bool is_gt = i < p.n && kx > kth, is_eq = i < p.n && kx == kth;
    const uint64_t mg = __ballot(is_gt), me = __ballot(is_eq);
    const uint64_t lt = (1ull << lane) - 1ull;
    if (lane == 0) s_wave[wv] = __popcll(mg) | (__popcll(me) << 16);
    __syncthreads();
    int gt = gt_run + __popcll(mg & lt), eq = eq_run + __popcll(me & lt), tot = 0;
    for (int q = 0; q < MKE_BLOCK / 64; ++q) {
      const int c = s_wave[q];
      if (q < wv) { gt += c & 0xFFFF; eq += c >> 16; }
      tot += c;                                      // packed sums: <= 256 per half, no carry between the halves
    }
    if (is_gt || (is_eq && eq < ties)) o[gt + min(eq, ties)] = p.id_map ? p.id_map[i] : i;
    gt_run += tot & 0xFFFF;
    eq_run += tot >> 16;
    __syncthreads();
  }
}

}  // namespace mke

extern "C" int mke_sim_select(const float* emb, int ld, int kpad, int64_t n_cols, int64_t row_lo, int64_t row_hi, const float* tau,
                              int n_seg, int seg_cap, mke_candidate* cand, int32_t* seg_count, void* stream) {
  using namespace mke;
  if (n_cols < 0 || row_lo < 0 || row_hi < row_lo || row_hi > n_cols || n_cols > 0x7FFFFF00LL) { set_error("mke_sim_select: bad row/column range"); return MKE_E_SHAPE; }
  if (row_hi == row_lo) return MKE_OK;
  if (!emb || !tau || !cand || !seg_count) { set_error("mke_sim_select: NULL pointer"); return MKE_E_NULL; }
  if (kpad <= 0 || kpad % 16 != 0 || kpad > MKE_MAX_STRIDE || ld < kpad || ld % 4 != 0) { set_error("mke_sim_select: kpad must be a multiple of 16 <= %d and <= ld (ld a multiple of 4)", MKE_MAX_STRIDE); return MKE_E_SHAPE; }
  if (n_seg < 1 || n_seg > 16 || seg_cap < 1 || (int64_t)n_seg * seg_cap > KNN_MAX_LIST) { set_error("mke_sim_select: need 1 <= n_seg <= 16 and n_seg * seg_cap <= %d", KNN_MAX_LIST); return MKE_E_SHAPE; }
  if ((row_hi - row_lo + SIMT_BM) * (int64_t)n_seg * seg_cap > 0x1FFFFFFFLL) { set_error("mke_sim_select: more than 2^29 candidate slots in one launch (split the row range)"); return MKE_E_RANGE; }
  SimSelectParams p;
  p.emb = emb; p.ld = ld; p.n_cols = (int)n_cols; p.row_lo = (int)row_lo; p.row_hi = (int)row_hi; p.tau = tau;
  p.n_seg = n_seg; p.seg_cap = seg_cap;
  const int bn = SIMT_BN_FOR(kpad / 16);
  const int ntiles = (int)((n_cols + bn - 1) / bn);
  p.tiles_per_seg = (ntiles + n_seg - 1) / n_seg;
  p.cand = cand; p.seg_count = seg_count;
  dim3 grid((unsigned)((row_hi - row_lo + SIMT_BM - 1) / SIMT_BM), (unsigned)n_seg);
  hipStream_t st = (hipStream_t)stream;
#define KNN_CASE(K)                                                                 \
  case K:                                                                           \
    hipLaunchKernelGGL((k_sim_select<K / 16>), grid, dim3(MKE_BLOCK), 0, st, p);     \
    break;
  switch (kpad) {
    KNN_CASE(16) KNN_CASE(32) KNN_CASE(48) KNN_CASE(64) KNN_CASE(80) KNN_CASE(96) KNN_CASE(112) KNN_CASE(128) KNN_CASE(160)
    KNN_CASE(192) KNN_CASE(208) KNN_CASE(256) KNN_CASE(320)
    default:
      set_error("mke_sim_select: unsupported kpad %d", kpad);
      return MKE_E_UNSUPPORTED;
  }
#undef KNN_CASE
  return check_launch("k_sim_select");
}

extern "C" int mke_sim_sample(const float* emb, int ld, int kpad, int64_t n_rows, int64_t row_lo, int64_t row_hi, const float* samp,
                              int ld_samp, int n_samp, float* out, void* stream) {
  using namespace mke;
  if (n_rows < 0 || row_lo < 0 || row_hi < row_lo || row_hi > n_rows || n_rows > 0x7FFFFF00LL || n_samp < 1) { set_error("mke_sim_sample: bad row range / sample size"); return MKE_E_SHAPE; }
  if (row_hi == row_lo) return MKE_OK;
  if (!emb || !samp || !out) { set_error("mke_sim_sample: NULL pointer"); return MKE_E_NULL; }
  if (kpad <= 0 || kpad % 16 != 0 || kpad > MKE_MAX_STRIDE || ld < kpad || ld_samp < kpad || ld % 4 != 0 || ld_samp % 4 != 0) { set_error("mke_sim_sample: kpad must be a multiple of 16 <= %d and <= ld, ld_samp (multiples of 4)", MKE_MAX_STRIDE); return MKE_E_SHAPE; }
  SimSampleParams p;
  p.emb = emb; p.ld = ld; p.row_lo = (int)row_lo; p.row_hi = (int)row_hi; p.samp = samp; p.ld_s = ld_samp; p.n_samp = n_samp; p.out = out;
  const int bn = SIMT_BN_FOR(kpad / 16);
  const int ntiles = (n_samp + bn - 1) / bn;
  const int row_blocks = (int)((row_hi - row_lo + SIMT_BM - 1) / SIMT_BM);
  int chunks = (4096 + row_blocks - 1) / row_blocks;
  if (chunks > (ntiles + 7) / 8) chunks = (ntiles + 7) / 8;
  if (chunks < 1) chunks = 1;
  p.tiles_per_chunk = (ntiles + chunks - 1) / chunks;
  dim3 grid((unsigned)row_blocks, (unsigned)((ntiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk));
  hipStream_t st = (hipStream_t)stream;
#define KNN_CASE(K)                                                                 \
  case K:                                                                           \
    hipLaunchKernelGGL((k_sim_sample<K / 16>), grid, dim3(MKE_BLOCK), 0, st, p);     \
    break;
  switch (kpad) {
    KNN_CASE(16) KNN_CASE(32) KNN_CASE(48) KNN_CASE(64) KNN_CASE(80) KNN_CASE(96) KNN_CASE(112) KNN_CASE(128) KNN_CASE(160)
    KNN_CASE(192) KNN_CASE(208) KNN_CASE(256) KNN_CASE(320)
    default:
      set_error("mke_sim_sample: unsupported kpad %d", kpad);
      return MKE_E_UNSUPPORTED;
  }
#undef KNN_CASE
  return check_launch("k_sim_sample");
}

static int topk_launch(const mke_candidate* cand, const float* vals, const int32_t* idx, const int32_t* seg_count, int64_t rows,
                       int n_seg, int seg_cap, int k, const int32_t* id_map, int32_t* out_idx, float* out_kth, int32_t* status,
                       void* stream, const char* who) {
  using namespace mke;
  if (rows < 0 || rows > 0x7FFFFFFFLL) { set_error("%s: bad row count", who); return MKE_E_SHAPE; }
  if (rows == 0) return MKE_OK;
  if ((!vals && !cand) || (!out_idx && !out_kth)) { set_error("%s: NULL pointer", who); return MKE_E_NULL; }
  if (n_seg < 1 || n_seg > 16 || seg_cap < 1 || (int64_t)n_seg * seg_cap > KNN_MAX_LIST || k < 1 || k > n_seg * seg_cap) {
    set_error("%s: need 1 <= n_seg <= 16, n_seg * seg_cap <= %d and 1 <= k <= n_seg * seg_cap", who, KNN_MAX_LIST);
    return MKE_E_SHAPE;
  }
  TopkParams p;
  p.cand = cand; p.vals = vals; p.idx = idx; p.seg_count = seg_count; p.n_seg = n_seg; p.seg_cap = seg_cap; p.k = k; p.id_map = id_map;
  p.out_idx = out_idx; p.out_kth = out_kth; p.status = status;
  if (out_idx) hipLaunchKernelGGL(k_topk_rows<true>, dim3((unsigned)rows), dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(k_topk_rows<false>, dim3((unsigned)rows), dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  return check_launch("k_topk_rows");
}

extern "C" int mke_topk_rows(const float* vals, const int32_t* idx, const int32_t* seg_count, int64_t rows, int n_seg, int seg_cap,
                             int k, const int32_t* id_map, int32_t* out_idx, float* out_kth, int32_t* status, void* stream) {
  return topk_launch(nullptr, vals, idx, seg_count, rows, n_seg, seg_cap, k, id_map, out_idx, out_kth, status, stream, "mke_topk_rows");
}

extern "C" int mke_topk_candidates(const mke_candidate* cand, const int32_t* seg_count, int64_t rows, int n_seg, int seg_cap, int k,
                                   const int32_t* id_map, int32_t* out_idx, float* out_kth, int32_t* status, void* stream) {
  return topk_launch(cand, nullptr, nullptr, seg_count, rows, n_seg, seg_cap, k, id_map, out_idx, out_kth, status, stream,
                     "mke_topk_candidates");
}

extern "C" int mke_topk_long(const float* vals, int64_t rows, int64_t n, int64_t ld, int k, const int32_t* id_map, int32_t* out_idx,
                             void* stream) {
  using namespace mke;
  if (rows < 0 || rows > 0x7FFFFFFFLL || n < 1 || n > 0x7FFFFF00LL || ld < n || k < 1 || k > n) { set_error("mke_topk_long: need rows >= 0, 1 <= k <= n <= ld"); return MKE_E_SHAPE; }
  if (rows == 0) return MKE_OK;
  if (!vals || !out_idx) { set_error("mke_topk_long: NULL pointer"); return MKE_E_NULL; }
  TopkLongParams p;
  p.vals = vals; p.ld = ld; p.n = (int)n; p.k = k; p.id_map = id_map; p.out_idx = out_idx;
  hipLaunchKernelGGL(k_topk_long, dim3((unsigned)rows), dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  return check_launch("k_topk_long");
}
