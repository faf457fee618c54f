// mke_attr_cnn.hip — the attribute-view CNN scorer (code/MultiKE_model.py:34-63 `conv`) for gfx950.
//
// TF1 semantics restated: stack (attribute row, literal row) -> [2, d, 1]; batch-norm in
// inference mode along the width axis (gamma[w] * x / sqrt(1 + 1e-3) + beta[w]); two conv2d(2 filters, 2x4, SAME,
// tanh); l2-normalise over the width axis per (row, channel); flatten (index h*2d + w*2 + c); dense 4d -> d with
// tanh (the GEMM is a plain library GEMM on the host side); l2-normalise over the WHOLE [B, d] batch; score =
// -||h - out||^2; loss = scale * sum w * log(1 + exp(-score)).
//
// Shape: the conv stack is one 64-lane wavefront per triple, lane = width position (WPL positions per lane), rows
// staged through per-wave LDS strips for the +-2 neighbour reads.  The backward kernel recomputes the (cheap)
// forward instead of storing activations, back-propagates through both convolutions, the width normalisation and
// the batch-norm affine, scatters the attribute-row gradient with atomics and block-reduces the parameter gradients.
// The batch-global normalisation needs two batch-wide sums (sum z^2, sum g.z): kernels write per-block partials
// and the next kernel's blocks add them up themselves — no extra reduction launches, no host round trip.
#include "mke_gemm.h"

namespace mke {

#define CNN_BN_EPS 1e-3f
#define CNN_NCONV 52  // K1 16 + b1 2 + K2 32 + b2 2
#define CNN_WS_COPIES 32
#define CNN_WS_STRIDE(d) (2 * (d) + 64)  // MKE_CNN_WORKSPACE_FLOATS(dim) = copies * stride

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_sum(float v) {
  v = sub16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// The LDS strips of k_attr_conv are private to a wavefront: ordering its own LDS writes before its own later reads needs
// no block barrier, only that the writes have been issued to the LDS (in-order per wave) and that the compiler does not
// move the reads up.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two arrays of partials at once (one pass over both, one pair of barriers)
__device__ __forceinline__ void totals_of_partials2(const double* __restrict__ pa, const double* __restrict__ pb, double& ta, double& tb) {
  __shared__ double s_t2[2];
  __shared__ double s_w2[2][MKE_BLOCK / 64];
  double va = 0.0, vb = 0.0;
  for (int i = threadIdx.x; i < MKE_LOSS_PARTIALS; i += MKE_BLOCK) { va += pa[i]; vb += pb[i]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { va += __shfl_down(va, off, 64); vb += __shfl_down(vb, off, 64); }
  if ((threadIdx.x & 63) == 0) { s_w2[0][threadIdx.x >> 6] = va; s_w2[1][threadIdx.x >> 6] = vb; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < MKE_BLOCK / 64; ++w) t += s_w2[threadIdx.x][w];
    s_t2[threadIdx.x] = t;
  }
  __syncthreads();
  ta = s_t2[0];
  tb = s_t2[1];
}

struct ConvParams {
  const float* __restrict__ attr;
  int attr_stride, attr_norm;
  const float* __restrict__ lit;
  int lit_stride;
  int dim;
  const int32_t* __restrict__ ia;
  const int32_t* __restrict__ iv;
  int64_t n;
  const float* __restrict__ params;  // packed: gamma[d] beta[d] K1[16] b1[2] K2[32] b2[2] ...
  float* __restrict__ flat;          // fwd out [n][flat_stride]; column 4d = 1 when flat_stride > 4d
  int flat_stride;
  const float* __restrict__ dflat;   // bwd in  [n][4d]
  float* __restrict__ gparams;       // bwd out, same packing, atomically accumulated
  float* __restrict__ gattr;         // bwd out: attribute-table gradient scratch (nullable)
  int gattr_copies;                  // >= 1: the scratch is [copies][n_attr][attr_stride], triple t adds to copy t % copies (a few hundred
  int64_t gattr_copy_elems;          // attribute rows take 5,000 triples per step and their frequencies are heavy-tailed)
  int32_t* __restrict__ tattr;
  int32_t tag;
  float* __restrict__ ws;            // bwd: MKE_CNN_WORKSPACE_FLOATS(dim) zero-invariant floats (nullable)
  // fwd with the dense layer in the same launch (k_attr_conv<WPL, 16, false, true>): z = tanh([flat, 1] W), W [4 dim + 1][dim]
  const float* __restrict__ W;
  float* __restrict__ z;
  double* __restrict__ ssq;          // per-block sums of z^2 (MKE_LOSS_PARTIALS slots; the ones no block owns are cleared)
  // bwd with the dflat product in the block (k_attr_conv<WPL, 16, true, false, 4, true>): dflat tile = 16 rows of dz x W^T; the blocks
  // behind the first conv_blocks of the grid are riders that compute the weight-gradient product `tall` ([flat, 1]^T dz over K splits)
  const float* __restrict__ dz;      // [n][dim]: g = dL/dout-side gradient of the loss tail (dz = dz_of(g, z, S, T) is formed on load)
  const float* __restrict__ zmat;    // [n][dim] z
  const double* __restrict__ dotp;   // per-block sums of g . z (with ssq: the two batch-wide scalars of the transform)
  int conv_blocks;
  GemmParams tall;
};

// K1[kh][kw][0][f] at k1[(kh*4+kw)*2+f]; K2[kh][kw][c][f] at k2[((kh*4+kw)*2+c)*2+f]  (TF HWIO order)
// LPT lanes per triple (64: a wavefront per triple; 32: one per HALF — dim 75 fills 75 of 128 lane slots in two passes of 64
// lanes but 75 of 96 in three passes of 32, and a wavefront then carries two triples: a quarter fewer instructions per
// triple), WPL width positions per lane: position w = tl + LPT * i of lane tl of the group.
// DENSE (forward, LPT = 16: a quarter-wave per triple, 16 triples per block = the rows of one 16 x 16 MFMA tile): the dense
// layer follows in the same block — flat rows stay in LDS as the A operand of z = tanh([flat, 1] W) (dim <= 80: five column
// tiles, K = 4 dim + 1 <= 321 split over the four wavefronts as in k_gemm_tall), W fragments requested before the
// convolution starts; flat still goes to memory once (the weight-gradient product reads it).
// NW wavefronts per block (4; the two-triples-per-wavefront backward: 2 — each convolution kernel costs the latency chain of
// one block plus ~2 us per further 4-wavefront block on the same CU (batch-size scan in profiles/r02_attr_step.md); 5000
// triples are 625 such blocks = three on 113 of the 256 CUs, but 1250 half blocks = at most five halves per CU)
// DFL (backward, LPT = 16, four wavefronts: the forward's shape): round 4 — the block first computes ITS 16 rows of dflat = dz W^T on
// the matrix cores (K = dim <= 80: 20 k-steps; 4 dim <= 320 columns: five 16-column tiles per wavefront, no exchange between the
// wavefronts) straight into the d1 strips of its 16 triples (a triple's 4 dim values fit its own strip, which the backward only
// writes after it has taken them out), so dflat never goes to memory and the product needs no blocks of its own; the
// weight-gradient product rides on extra blocks at the END of the grid (dispatched after the convolution blocks, which are the
// long ones), its LDS exchange area aliased onto the c1 strips.
template <int WPL, int LPT, bool BWD, bool DENSE = false, int NW = MKE_BLOCK / 64, bool DFL = false>
__global__ __launch_bounds__(NW * 64, DFL ? 2 : 1) void k_attr_conv(const ConvParams p) {
  static_assert(NW == MKE_BLOCK / 64 || (BWD && !DENSE), "short blocks: backward only");
  static_assert(!DFL || (BWD && !DENSE && LPT == 16 && NW == MKE_BLOCK / 64), "in-block dflat: backward, a quarter-wave per triple");
  constexpr int NT = NW * 64;                      // threads per block
  static_assert(!DENSE || (!BWD && LPT == 16), "dense layer: forward, a quarter-wave per triple");
  constexpr int TPW = 64 / LPT;                    // triples per wavefront
  constexpr int NSLOT = NW * TPW;                  // triples in flight per block, each with its own LDS strips
  constexpr int DP = LPT * WPL + 4;
  constexpr int FS = 4 * LPT * WPL + 5;            // DENSE: row stride of the flat tile (odd: conflict-free column reads)
  constexpr int KS = 21;                           // DENSE: k-steps of 4 per wavefront (K <= 336)
  constexpr int NCT = 5;                           // DENSE: column tiles of 16
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float s_flat[DENSE ? NSLOT : 1][DENSE ? FS : 1];
  __shared__ float s_acc[DENSE ? MKE_BLOCK / 64 : 1][DENSE ? NCT : 1][4][DENSE ? 64 : 1];
  float wfrag[DENSE ? KS : 1][DENSE ? NCT : 1];
  if constexpr (DENSE) {
    const int r16 = threadIdx.x & 15, kq = (threadIdx.x & 63) >> 4, wvv = threadIdx.x >> 6;
    const int K = 4 * p.dim + 1;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const float* __restrict__ bp = p.W + (int64_t)min(4 * KS * wvv + kq + 4 * i, K - 1) * p.dim;
#pragma unroll
      for (int c = 0; c < NCT; ++c) wfrag[i][c] = bp[min(16 * c + r16, p.dim - 1)];
    }
  }
  // LPT = 16: a half-wave reads two triples' strips at once — their distance must be 16 banks (mod 32) for the two 16-lane
  // runs not to overlap: 4 DP already is (the c1 strips); the x strips get their own row length (2 DPX = 16 mod 32)
  constexpr int DPX = LPT == 16 ? DP + ((8 - DP % 16) + 16) % 16 : DP;
  __shared__ float s_x[NSLOT][2][DPX];
  // the c1 / d2 / d1 strips as ONE array: the rider blocks' exchange area (20 KB) is aliased onto its start whatever the width
  // (aliased onto the c1 strips alone it fitted only above dim 64, and at dim 64 itself the block then took 82 KB of LDS = one
  // per compute unit: 62 us per step against 52 at dim 75)
  __shared__ float s_strips[BWD ? 3 : 1][NSLOT][2][2][DP];
  float (&s_c1)[NSLOT][2][2][DP] = s_strips[0];
  float (&s_d2)[NSLOT][2][2][DP] = s_strips[BWD ? 1 : 0];
  float (&s_d1)[NSLOT][2][2][DP] = s_strips[BWD ? 2 : 0];
  if constexpr (DFL) {
    constexpr bool ALIAS = sizeof(s_strips) >= sizeof(float) * (MKE_BLOCK / 64) * 5 * 4 * 64;
    __shared__ float s_rider[ALIAS ? 1 : (MKE_BLOCK / 64) * 5 * 4 * 64];
    static_assert(2 * 2 * DP >= 4 * LPT * WPL, "a triple's dflat row must fit its own d1 strip");
    if ((int)blockIdx.x >= p.conv_blocks) {      // block-uniform: a rider of the weight-gradient product
      const int r = (int)blockIdx.x - p.conv_blocks;
      double S, T;
      totals_of_partials2(p.ssq, p.dotp, S, T);
      const float inv = rsqrtf(fmaxf((float)S, MKE_L2_EPS));
      const float coef = (float)S > MKE_L2_EPS ? (float)T * inv * inv : 0.f;
      gemm_tall_block_dz<5, 32, 4>(p.tall, r % p.tall.gx, r / p.tall.gx,
                                   reinterpret_cast<float (*)[5][4][64]>(ALIAS ? &s_strips[0][0][0][0][0] : &s_rider[0]), p.zmat, inv, coef);
      return;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tl = lane & (LPT - 1);               // lane inside the triple's group
  const int slot = wv * TPW + lane / LPT;         // which of the block's triples
  auto group_sum = [](float v) {                  // over the LPT lanes of a triple
    v = sub16_sum(v);
    if (LPT >= 32) v += __shfl_xor(v, 16, 64);
    if (LPT == 64) v += __shfl_xor(v, 32, 64);
    return v;
  };
  if constexpr (DFL) {
    // ---- dflat rows of the block's 16 triples: [16 x dim] (dz) x [dim x 4 dim] (W^T), f32 MFMA 16x16x4.  BEFORE the convolution's
    // constants are loaded: the 100 B fragments and the 50 filter weights are never live together (together: 256 registers and
    // 232 spilled).  The host launches one block per 16 triples (a single pass of the loop below).
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int KSB = 20, NTB = 5;             // k-steps of 4 (dim <= 80), column tiles per wavefront (4 dim <= 320 = 20 tiles)
    const int r16 = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * NSLOT;
    const int d = p.dim;
    const int ncol = 4 * d;
    // B = W^T [dim][4 dim] (p.W: the transposed copy k_attr_tail_bwd left): lane (r16, kq) of k-step i reads wt[4 i + kq][16 ct + r16]
    // — a quarter-wave reads 16 consecutive floats (from W itself the same operand is 16 rows 300 bytes apart: 16 cache lines per
    // load instead of 1, and the phase took 14 us instead of 3).  A = the block's 16 rows of dz: 1200 consecutive floats, copied
    // into the (still unused) x strips with coalesced loads and read back per lane.
    float bw[KSB][NTB];
#pragma unroll
    for (int i = 0; i < KSB; ++i) {
      const int kc = min(4 * i + kq, d - 1);       // unconditional loads from clamped addresses; k >= dim contributes a = 0
#pragma unroll
      for (int c = 0; c < NTB; ++c) bw[i][c] = p.W[kc * ncol + min(16 * (wv + 4 * c) + r16, ncol - 1)];
    }
    float* s_dz = &s_x[0][0][0];
    static_assert(sizeof(s_x) >= sizeof(float) * NSLOT * (LPT * WPL + 1), "the dz tile (16 rows of dim <= 16 WPL floats, odd row stride) must fit the x strips");
    {
      // the tile's g and z are requested before the two batch-wide sums are added up (one round trip for all of it)
      constexpr int NEL = (NSLOT * LPT * WPL + NT - 1) / NT;
      const int64_t e0 = m0 * d, e_end = min(p.n, m0 + NSLOT) * d;
      float gv[NEL], zv[NEL];
#pragma unroll
      for (int q = 0; q < NEL; ++q) {
        const int64_t e = e0 + threadIdx.x + q * NT;
        const bool ok = threadIdx.x + q * NT < NSLOT * d && e < e_end;
        gv[q] = ok ? p.dz[e] : 0.f;
        zv[q] = ok ? p.zmat[e] : 0.f;
      }
      double S, T;
      totals_of_partials2(p.ssq, p.dotp, S, T);
      const float inv = rsqrtf(fmaxf((float)S, MKE_L2_EPS));
      const float coef = (float)S > MKE_L2_EPS ? (float)T * inv * inv : 0.f;
      // row stride of the tile in LDS: odd (dim 64 with stride 64 put the 16 rows of a fragment read on ONE bank: 62.5 us per
      // step against 52.3 at dim 75)
      const int sd = d | 1;
#pragma unroll
      for (int q = 0; q < NEL; ++q) {
        const int e = threadIdx.x + q * NT;
        if (e < NSLOT * d) s_dz[(e / d) * sd + e % d] = dz_of(gv[q], zv[q], inv, coef);
      }
    }
    __syncthreads();
    float av[KSB];
#pragma unroll
    for (int i = 0; i < KSB; ++i) av[i] = s_dz[r16 * (d | 1) + min(4 * i + kq, d - 1)];
    f32x4 acc[NTB];
#pragma unroll
    for (int c = 0; c < NTB; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KSB; ++i) {
      const float ai = (4 * i + kq < d) ? av[i] : 0.f;   // rows past n are zero in the tile
#pragma unroll
      for (int c = 0; c < NTB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, bw[i][c], acc[c], 0, 0, 0);
    }
    // C/D map of the 16x16 forms: col = lane & 15, row = 4 * (lane >> 4) + reg  ->  triple slot = row, dflat column = col
#pragma unroll
    for (int c = 0; c < NTB; ++c) {
      const int col = 16 * (wv + 4 * c) + r16;
      if (col < ncol) {
#pragma unroll
        for (int r = 0; r < 4; ++r) (&s_d1[4 * kq + r][0][0][0])[col] = acc[c][r];
      }
    }
    __syncthreads();
  }
  const int d = p.dim;
  const float bn_s = rsqrtf(1.0f + CNN_BN_EPS);
  const float* __restrict__ gamma = p.params;
  const float* __restrict__ beta = p.params + d;
  float k1[16], b1[2], k2[32], b2[2];
  {
    const float* cp = p.params + 2 * d;
#pragma unroll
    for (int i = 0; i < 16; ++i) k1[i] = cp[i];
    b1[0] = cp[16]; b1[1] = cp[17];
#pragma unroll
    for (int i = 0; i < 32; ++i) k2[i] = cp[18 + i];
    b2[0] = cp[50]; b2[1] = cp[51];
  }
  float gam[WPL], bet[WPL];
#pragma unroll
  for (int i = 0; i < WPL; ++i) {
    const int w = tl + LPT * i;
    gam[i] = w < d ? gamma[w] : 0.f;
    bet[i] = w < d ? beta[w] : 0.f;
  }
  // parameter-gradient accumulators (BWD)
  // The 52 conv-parameter gradients are NOT kept in registers: each is reduced over its quarter-wave as soon as it is
  // formed and added to the quarter's own LDS slot (s_part).  48 + 4 long-lived accumulators pushed the backward kernel to
  // 208 unified registers = 2 waves per SIMD = three rounds of waves for 5000 triples.
  __shared__ float s_part[BWD ? CNN_NCONV : 1][4 * NW + 1];
  const int pq = (lane >> 4) + 4 * wv;
  const bool plead = (lane & 15) == 0;
  auto park = [&](int idx, float v) {
    v = sub16_sum(v);
    if (plead) atomicAdd(&s_part[idx][pq], v);   // the slot belongs to this quarter-wave: no contention
  };
  float a_gam[WPL], a_bet[WPL];
  if constexpr (BWD) {
    for (int i = threadIdx.x; i < CNN_NCONV * (4 * NW + 1); i += NT) (&s_part[0][0])[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WPL; ++i) a_gam[i] = a_bet[i] = 0.f;
  }
  float(*xs)[DPX] = s_x[slot];
  float(*c1s)[2][DP] = s_c1[slot];

  const int64_t wave0 = (int64_t)blockIdx.x * NSLOT + slot;
  const int64_t nwaves = (int64_t)(DFL ? p.conv_blocks : (int)gridDim.x) * NSLOT;
  const int64_t iters = (p.n + nwaves - 1) / nwaves;  // block-uniform trip count (barriers inside)
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t t = wave0 + it * nwaves;
    const bool live = t < p.n;
    const int ra = live ? p.ia[t] : 0, rv = live ? p.iv[t] : 0;
    float raw[2][WPL];
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
      const int w = tl + LPT * i;
      raw[0][i] = (live && w < d) ? p.attr[(int64_t)ra * p.attr_stride + w] : 0.f;
      raw[1][i] = (live && w < d) ? p.lit[(int64_t)rv * p.lit_stride + w] : 0.f;
    }
    if (p.attr_norm) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < WPL; ++i) s = fmaf(raw[0][i], raw[0][i], s);
      s = group_sum(s);
      const float inv = rsqrtf(fmaxf(s, MKE_L2_EPS));
#pragma unroll
      for (int i = 0; i < WPL; ++i) raw[0][i] *= inv;
    }
    // ---- batch-norm affine, stage x with zero pads: index w+1 <-> width w ---------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        const int w = tl + LPT * i;
        xs[h][w + 1] = w < d ? fmaf(gam[i] * bn_s, raw[h][i], bet[i]) : 0.f;
      }
      if (tl < 4) xs[h][tl == 0 ? 0 : LPT * WPL + tl] = 0.f;
    }
    wave_lds_sync();
    // ---- conv1 ---------------------------------------------------------------------------------------------
    float c1[2][2][WPL];
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
      const int w = tl + LPT * i;
#pragma unroll
      for (int h = 0; h < 2; ++h)
      {
        // the two filters of a tap are adjacent in the packed weights: one v_pk_fma_f32 per tap for both (the input broadcast)
        v2f acc = {b1[0], b1[1]};
#pragma unroll
        for (int kh = 0; kh + h < 2; ++kh)
#pragma unroll
          for (int kw = 0; kw < 4; ++kw) {
            const float x = xs[h + kh][w + kw];
            acc = __builtin_elementwise_fma(v2f{k1[(kh * 4 + kw) * 2], k1[(kh * 4 + kw) * 2 + 1]}, v2f{x, x}, acc);
          }
        c1[h][0][i] = w < d ? tanh_f(acc.x) : 0.f;
        c1[h][1][i] = w < d ? tanh_f(acc.y) : 0.f;
        c1s[h][0][w + 1] = c1[h][0][i];
        c1s[h][1][w + 1] = c1[h][1][i];
      }
    }
    if (tl < 4) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < 2; ++f) c1s[h][f][tl == 0 ? 0 : LPT * WPL + tl] = 0.f;
    }
    wave_lds_sync();
    // ---- conv2 + width normalisation -----------------------------------------------------------------------
    float c2[2][2][WPL], nrm[2][2], ssq[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int f = 0; f < 2; ++f) ssq[h][f] = 0.f;
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
      const int w = tl + LPT * i;
#pragma unroll
      for (int h = 0; h < 2; ++h)
      {
        v2f acc = {b2[0], b2[1]};
#pragma unroll
        for (int kh = 0; kh + h < 2; ++kh)
#pragma unroll
          for (int kw = 0; kw < 4; ++kw)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const float x = c1s[h + kh][c][w + kw];
              const int ki = ((kh * 4 + kw) * 2 + c) * 2;
              acc = __builtin_elementwise_fma(v2f{k2[ki], k2[ki + 1]}, v2f{x, x}, acc);
            }
        c2[h][0][i] = w < d ? tanh_f(acc.x) : 0.f;
        c2[h][1][i] = w < d ? tanh_f(acc.y) : 0.f;
        ssq[h][0] = fmaf(c2[h][0][i], c2[h][0][i], ssq[h][0]);
        ssq[h][1] = fmaf(c2[h][1][i], c2[h][1][i], ssq[h][1]);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        ssq[h][f] = group_sum(ssq[h][f]);
        nrm[h][f] = rsqrtf(fmaxf(ssq[h][f], MKE_L2_EPS));
      }
    if constexpr (!BWD) {
      if constexpr (DENSE) {
        if (tl == 0) s_flat[slot][4 * d] = live ? 1.0f : 0.f;
#pragma unroll
        for (int i = 0; i < WPL; ++i) {
          const int w = tl + LPT * i;
          if (w < d) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              s_flat[slot][h * 2 * d + w * 2] = live ? c2[h][0][i] * nrm[h][0] : 0.f;
              s_flat[slot][h * 2 * d + w * 2 + 1] = live ? c2[h][1][i] * nrm[h][1] : 0.f;
            }
          }
        }
      }
      if (live) {
        float* o = p.flat + t * (int64_t)p.flat_stride;
        if (tl == 0 && p.flat_stride > 4 * d) o[4 * d] = 1.0f;  // bias column: [flat, 1] @ [W; bias]
#pragma unroll
        for (int i = 0; i < WPL; ++i) {
          const int w = tl + LPT * i;
          if (w < d) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float2 v = make_float2(c2[h][0][i] * nrm[h][0], c2[h][1][i] * nrm[h][1]);
              *reinterpret_cast<float2*>(o + h * 2 * d + w * 2) = v;
            }
          }
        }
      }
      wave_lds_sync();  // LDS strips are rewritten by the next iteration
      continue;
    }
    if constexpr (BWD) {
      float(*d2s)[2][DP] = s_d2[slot];
      float(*d1s)[2][DP] = s_d1[slot];
      // ---- width-normalisation backward, tanh', parameter gradients of conv2 -------------------------------
      float dy[2][2][WPL], dot[2][2];
      const float* gi = DFL ? &s_d1[slot][0][0][0] : p.dflat + t * (int64_t)(4 * d);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < 2; ++f) dot[h][f] = 0.f;
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        const int w = tl + LPT * i;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float2 v = make_float2(0.f, 0.f);
          if (live && w < d) v = *reinterpret_cast<const float2*>(gi + h * 2 * d + w * 2);
          dy[h][0][i] = v.x; dy[h][1][i] = v.y;
#pragma unroll
          for (int f = 0; f < 2; ++f) dot[h][f] = fmaf(dy[h][f][i], c2[h][f][i] * nrm[h][f], dot[h][f]);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < 2; ++f) dot[h][f] = group_sum(dot[h][f]);
      float dp2[2][2][WPL];
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        const int w = tl + LPT * i;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const float y = c2[h][f][i] * nrm[h][f];
            const float dc2 = ssq[h][f] > MKE_L2_EPS ? nrm[h][f] * (dy[h][f][i] - y * dot[h][f]) : nrm[h][f] * dy[h][f][i];
            dp2[h][f][i] = (w < d) ? dc2 * (1.0f - c2[h][f][i] * c2[h][f][i]) : 0.f;
            d2s[h][f][w + 2] = dp2[h][f][i];  // index w+2 <-> width w
          }
      }
      // parameter gradients of conv2, one scalar at a time: db2[f] = sum dp2[.][f][.],
      // dK2[kh][kw][c][f] = sum_{h, w} dp2[h][f][w] * c1[h + kh][c][w + kw - 1]
      {
        // both filters of a tap at once (v_pk_fma_f32: the pair (dp2[.][0], dp2[.][1]) times the broadcast c1 value)
        v2f v = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WPL; ++i) v += v2f{dp2[0][0][i], dp2[0][1][i]} + v2f{dp2[1][0][i], dp2[1][1][i]};
        park(50, v.x);
        park(51, v.y);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int kw = 0; kw < 4; ++kw)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              v2f a = {0.f, 0.f};
#pragma unroll
              for (int h = 0; h + kh < 2; ++h)
#pragma unroll
                for (int i = 0; i < WPL; ++i) {
                  const float x = c1s[h + kh][c][tl + LPT * i + kw];
                  a = __builtin_elementwise_fma(v2f{dp2[h][0][i], dp2[h][1][i]}, v2f{x, x}, a);
                }
              park(18 + ((kh * 4 + kw) * 2 + c) * 2, a.x);
              park(18 + ((kh * 4 + kw) * 2 + c) * 2 + 1, a.y);
            }
      }
      if (tl < 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int f = 0; f < 2; ++f) d2s[h][f][tl < 2 ? tl : LPT * WPL + tl] = 0.f;
      }
      wave_lds_sync();
      // ---- conv2 transposed -> dc1, tanh', parameter gradients of conv1 -----------------------------------
      float dp1[2][2][WPL];
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        const int w = tl + LPT * i;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            v2f acc2 = {0.f, 0.f};   // the two filters' terms side by side, added at the end
#pragma unroll
            for (int kh = 0; kh <= hh; ++kh)
#pragma unroll
              for (int kw = 0; kw < 4; ++kw) {
                const int ki = ((kh * 4 + kw) * 2 + c) * 2;
                acc2 = __builtin_elementwise_fma(v2f{k2[ki], k2[ki + 1]},
                                                 v2f{d2s[hh - kh][0][w + 3 - kw], d2s[hh - kh][1][w + 3 - kw]}, acc2);
              }
            const float acc = acc2.x + acc2.y;
            dp1[hh][c][i] = (w < d) ? acc * (1.0f - c1[hh][c][i] * c1[hh][c][i]) : 0.f;
            d1s[hh][c][w + 2] = dp1[hh][c][i];
          }
      }
      // parameter gradients of conv1: db1[c], dK1[kh][kw][0][c] = sum_{h, w} dp1[h][c][w] * x[h + kh][w + kw - 1]
      {
        v2f v = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WPL; ++i) v += v2f{dp1[0][0][i], dp1[0][1][i]} + v2f{dp1[1][0][i], dp1[1][1][i]};
        park(16, v.x);
        park(17, v.y);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int kw = 0; kw < 4; ++kw) {
            v2f a = {0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh + kh < 2; ++hh)
#pragma unroll
              for (int i = 0; i < WPL; ++i) {
                const float x = xs[hh + kh][tl + LPT * i + kw];
                a = __builtin_elementwise_fma(v2f{dp1[hh][0][i], dp1[hh][1][i]}, v2f{x, x}, a);
              }
            park((kh * 4 + kw) * 2, a.x);
            park((kh * 4 + kw) * 2 + 1, a.y);
          }
      }
      if (tl < 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int f = 0; f < 2; ++f) d1s[h][f][tl < 2 ? tl : LPT * WPL + tl] = 0.f;
      }
      wave_lds_sync();
      // ---- conv1 transposed -> dx, batch-norm affine backward, attribute-row gradient ---------------------
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        const int w = tl + LPT * i;
        float dx[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          v2f acc2 = {0.f, 0.f};
#pragma unroll
          for (int kh = 0; kh <= hh; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw)
              acc2 = __builtin_elementwise_fma(v2f{k1[(kh * 4 + kw) * 2], k1[(kh * 4 + kw) * 2 + 1]},
                                               v2f{d1s[hh - kh][0][w + 3 - kw], d1s[hh - kh][1][w + 3 - kw]}, acc2);
          dx[hh] = (live && w < d) ? acc2.x + acc2.y : 0.f;
        }
        a_gam[i] += (dx[0] * raw[0][i] + dx[1] * raw[1][i]) * bn_s;
        a_bet[i] += dx[0] + dx[1];
        if (live && w < d && p.gattr) atomic_add_f32(p.gattr + (t % p.gattr_copies) * p.gattr_copy_elems + (int64_t)ra * p.attr_stride + w, dx[0] * gam[i] * bn_s);
      }
      if (live && tl == 0 && p.gattr) p.tattr[ra] = p.tag;
      wave_lds_sync();
    }
  }

  __syncthreads();  // the strips are reused by the block-level reduction below
  if constexpr (DENSE) {
    // ---- z tile = tanh([flat, 1] W): the block's 16 flat rows (LDS) x W [K][d], K split over the four wavefronts ----
    const int r16 = lane & 15, kq = lane >> 4;
    const int K = 4 * d + 1, k0 = 4 * KS * wv + kq;
    f32x4 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // all of the wavefront's A operands out of LDS first, unconditionally from clamped columns (a guarded read compiles to an
    // exec-masked branch + a full LDS wait in front of EVERY k-step's five MFMAs: 21 exposed LDS round trips, ~2,300 of the
    // phase's 5,800 cycles; s_memtime stamps, tools/attr_stamps.py), then the 105 MFMAs back to back
    float av[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) av[i] = s_flat[r16][min(k0 + 4 * i, K - 1)];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const float ai = (k0 + 4 * i < K) ? av[i] : 0.f;
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, wfrag[i][c], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_acc[wv][c][r][lane] = acc[c][r];
    __syncthreads();
    // wave w finishes column tiles w, w + 4: C/D map of the 16x16 forms: col = lane & 15, row = 4 * (lane >> 4) + reg
    float ssq = 0.f;
    const int64_t m0 = (int64_t)blockIdx.x * NSLOT;
    for (int c = wv; c < NCT; c += MKE_BLOCK / 64) {
      const int col = 16 * c + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s_acc[0][c][r][lane] + s_acc[1][c][r][lane] + s_acc[2][c][r][lane] + s_acc[3][c][r][lane];
        const int64_t orow = m0 + 4 * kq + r;
        if (col < d && orow < p.n) {
          v = tanh_f(v);
          p.z[orow * d + col] = v;
          ssq = fmaf(v, v, ssq);
        }
      }
    }
    // <= 8 squares per lane: 16-lane rows in f32 on DPP, the block's 16 row sums in double (block_sum_double's six 64-bit
    // shuffle steps were 1,900 cycles at the very end of the launch's critical path)
    __shared__ double s_rows[MKE_BLOCK / 16];
    ssq = sub16_sum(ssq);
    if ((threadIdx.x & 15) == 0) s_rows[threadIdx.x >> 4] = (double)ssq;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < MKE_BLOCK / 16; ++k) tot += s_rows[k];
      p.ssq[blockIdx.x] = tot;
      for (int k = blockIdx.x + gridDim.x; k < MKE_LOSS_PARTIALS; k += gridDim.x) p.ssq[k] = 0.0;
    }
  }
  if constexpr (BWD) {
    // ---- block-reduce the parameter gradients, one atomic per block per scalar --------------------------------
    // the 16 quarter-wave slots of every scalar were filled during the loop (`park`); 52 threads add them up.  (First
    // version: 52 full wave reductions + LDS atomics per wave = 20 of the kernel's 45 us.)  gamma / beta are per width
    // position: the four waves' vectors go through the (now free) x / c1 strips.
    float* gsum = &s_x[0][0][0];
    float* bsum = &s_c1[0][0][0][0];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
      gsum[slot * LPT * WPL + tl + LPT * i] = a_gam[i];
      bsum[slot * LPT * WPL + tl + LPT * i] = a_bet[i];
    }
    __syncthreads();
    // Every block adding its 2d + 52 sums straight into grad_params is a chain of gridDim.x same-address atomics per
    // scalar (measured: 45 of the kernel's 66 us at 1024 blocks).  With a workspace the blocks spread over
    // CNN_WS_COPIES privatised copies (chain length gridDim.x / copies); whoever consumes the gradient next (the dense
    // update, or k_cnn_ws_fold) adds the copies up and zeroes them.  (A last-block-done ticket was tried first: the
    // ticket is itself a gridDim.x-long same-address chain and cost more than it saved, 95 us.)
    float* dst = p.ws ? p.ws + (size_t)(blockIdx.x % CNN_WS_COPIES) * CNN_WS_STRIDE(d) : p.gparams;
    for (int w = threadIdx.x; w < d; w += NT) {
      float g = 0.f, b = 0.f;
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) { g += gsum[k * LPT * WPL + w]; b += bsum[k * LPT * WPL + w]; }
      atomic_add_f32(dst + w, g);
      atomic_add_f32(dst + d + w, b);
    }
    if (threadIdx.x < CNN_NCONV) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 4 * NW; ++k) v += s_part[threadIdx.x][k];
      atomic_add_f32(dst + 2 * d + threadIdx.x, v);
    }
  }
}

// grad_params[i] += sum over the CNN_WS_COPIES privatised copies, copies zeroed (the standalone entry's epilogue; inside
// mke_attr_step the dense update does this itself)
__global__ __launch_bounds__(MKE_BLOCK) void k_cnn_ws_fold(float* __restrict__ ws, float* __restrict__ gparams, int np, int stride) {
  const int i = blockIdx.x * MKE_BLOCK + threadIdx.x;
  if (i >= np) return;
  float v = 0.f;
#pragma unroll 8
  for (int c = 0; c < CNN_WS_COPIES; ++c) v += ws[(size_t)c * stride + i];
  for (int c = 0; c < CNN_WS_COPIES; ++c) ws[(size_t)c * stride + i] = 0.f;
  gparams[i] += v;
}

// ---- tail: z = tanh(zpre + bias), partial sums of z^2 -------------------------------------------------------------
__global__ __launch_bounds__(MKE_BLOCK) void k_attr_tail_z(float* __restrict__ z, const float* __restrict__ bias, int64_t n,
                                                           int dim, double* __restrict__ partials) {
  const int64_t total = n * dim;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * MKE_BLOCK) {
    const float v = tanhf(bias ? z[i] + bias[i % dim] : z[i]);
    z[i] = v;
    s = fmaf(v, v, s);
  }
  const double tot = block_sum_double(s);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// every block adds up the MKE_LOSS_PARTIALS partials of the previous kernel itself
__device__ __forceinline__ double total_of_partials(const double* __restrict__ partials) {
  __shared__ double s_tot;
  double v = 0.0;
  for (int i = threadIdx.x; i < MKE_LOSS_PARTIALS; i += MKE_BLOCK) v += partials[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __shared__ double s_w[MKE_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < MKE_BLOCK / 64; ++w) t += s_w[w];
    s_tot = t;
  }
  __syncthreads();
  return s_tot;
}

struct TailParams {
  const float* __restrict__ z;
  const double* __restrict__ sumsq;
  const float* __restrict__ ent;
  int ent_stride, ent_norm;
  const int32_t* __restrict__ ih;
  const float* __restrict__ ws;
  float scale;
  int64_t n;
  int dim;
  float* __restrict__ gout;
  double* __restrict__ dotp;
  float* __restrict__ gent;
  int32_t* __restrict__ tent;
  int32_t tag;
  double* __restrict__ lossp;
  const float* __restrict__ w;   // nullable pair: side job wt[k][f] = w[f][k], f < 4 dim, k < dim (the dense layer's weights transposed,
  float* __restrict__ wt;        // for the dflat product inside the convolution-backward launch)
};

template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_attr_tail_loss(const TailParams p) {
  if (p.wt) {
    const int total_w = 4 * p.dim * p.dim;
    for (int e = blockIdx.x * MKE_BLOCK + threadIdx.x; e < total_w; e += gridDim.x * MKE_BLOCK) {
      const int k = e / (4 * p.dim), f = e - k * (4 * p.dim);
      p.wt[e] = p.w[f * p.dim + k];
    }
  }
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  // a row's id, z values and weight are requested one row ahead — the first row's before the partial sums are added up, so
  // that its memory round trip and theirs are the same one
  int row_n = 0;
  float Zn[FPL], w_n = 1.0f;
  auto request = [&](int64_t i) {
    const bool ok = i < p.n;
    row_n = ok ? p.ih[i] : 0;
    w_n = (ok && p.ws) ? p.ws[i] : 1.0f;
    const float* zp = p.z + (ok ? i : 0) * (int64_t)p.dim + j;
#pragma unroll
    for (int k = 0; k < FPL; ++k) Zn[k] = (ok && k * 16 + j < p.dim) ? zp[k * 16] : 0.f;
  };
  request(sub0);
  const double S = total_of_partials(p.sumsq);
  const float inv = rsqrtf(fmaxf((float)S, MKE_L2_EPS));
  float loss = 0.f, dotacc = 0.f;
  for (int64_t i = sub0; i < p.n; i += nsub) {
    const int row = row_n;
    const float w = w_n;
    float H[FPL], Z[FPL];
    load_row<FPL>(p.ent, row, p.ent_stride, j, H);
#pragma unroll
    for (int k = 0; k < FPL; ++k) Z[k] = Zn[k];
    request(i + nsub);
    l2_normalize_row<FPL>(H, p.ent_norm);
    float x = 0.f;
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      H[k] = H[k] - Z[k] * inv;  // diff = h - out
      x = fmaf(H[k], H[k], x);
    }
    x = sub16_sum(x);
    loss += w * softplus_f(x);
    const float c = 2.0f * p.scale * w * sigmoid_f(x);
    float dz = 0.f;
    float* go = p.gout + i * (int64_t)p.dim + j;
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      H[k] *= c;  // g_h = 2 c diff ; g_out = -g_h
      if (k * 16 + j < p.dim) go[k * 16] = -H[k];
      dz = fmaf(-H[k], Z[k], dz);
    }
    dotacc += dz;  // per-lane partial of sum g_out . z
    if (p.gent) {
      atomic_add_row<FPL>(p.gent, row, p.ent_stride, p.dim, j, H, 1.0f);
      if (j == 0) p.tent[row] = p.tag;
    }
  }
  const double lt = block_sum_double(j == 0 ? loss : 0.f);
  __syncthreads();
  const double dt = block_sum_double(dotacc);
  if (threadIdx.x == 0) {
    p.lossp[blockIdx.x] = lt * (double)p.scale;
    p.dotp[blockIdx.x] = dt;
    // one pass over the rows takes n / 16 blocks; the partial slots no block owns are cleared here (every block of this and
    // of the next kernel re-adds all MKE_LOSS_PARTIALS slots: 2048 blocks doing that was 32 MB of L2 reads)
    for (int k = blockIdx.x + gridDim.x; k < MKE_LOSS_PARTIALS; k += gridDim.x) { p.lossp[k] = 0.0; p.dotp[k] = 0.0; }
  }
}

// dzpre = inv * (g_out - out * (g_out . out)) * (1 - z^2), in place over gout
__global__ __launch_bounds__(MKE_BLOCK) void k_attr_tail_bwd(const float* __restrict__ z, float* __restrict__ g,
                                                             const double* __restrict__ sumsq, const double* __restrict__ dotp,
                                                             int64_t n, int dim) {
  // the first four elements of every thread (all of them when the grid is sized by the launcher) are requested before the
  // partial sums: one memory round trip in front of the arithmetic instead of three
  const int64_t total = n * dim;
  const int64_t i0 = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x, step = (int64_t)gridDim.x * MKE_BLOCK;
  float zv[4], gv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = i0 + j * step;
    zv[j] = i < total ? z[i] : 0.f;
    gv[j] = i < total ? g[i] : 0.f;
  }
  double S, T;
  totals_of_partials2(sumsq, dotp, S, T);
  const float inv = rsqrtf(fmaxf((float)S, MKE_L2_EPS));
  const float coef = (float)S > MKE_L2_EPS ? (float)T * inv * inv : 0.f;  // out * (g.out) = z * inv^2 * T
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = i0 + j * step;
    if (i < total) g[i] = inv * (gv[j] - zv[j] * coef) * (1.0f - zv[j] * zv[j]);
  }
  for (int64_t i = i0 + 4 * step; i < total; i += step) {
    const float zz = z[i];
    g[i] = inv * (g[i] - zz * coef) * (1.0f - zz * zz);
  }
}

// out[j] += sum_i x[i][j]: each block folds its rows, one atomic per column per block (standalone bias gradient; inside
// mke_attr_step the bias is a row of the extended weight matrix and its gradient a row of dW)
__global__ __launch_bounds__(MKE_BLOCK) void k_colsum_add(const float* __restrict__ x, int64_t n, int dim, float* __restrict__ out) {
  for (int j = threadIdx.x; j < dim; j += MKE_BLOCK) {
    float s = 0.f;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) s += x[i * dim + j];
    atomic_add_f32(out + j, s);
  }
}

__global__ __launch_bounds__(MKE_BLOCK) void k_dense_update(const DenseJob j) { dense_update_range(j, blockIdx.x, gridDim.x); }

int launch_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                    int M, int N, int K, int splits, int accumulate, hipStream_t st, double* tanh_sumsq_partials, int epi_plain);
bool launch_gemm_tallsplit_plus(const float* A0, int64_t a0_rs, int64_t a0_cs, const float* B0, int64_t b0_rs, float* C0, int64_t ldc0,
                                int M0, int N0, int K0, const float* A1, int64_t a1_rs, int64_t a1_cs, const float* B1, int64_t b1_rs,
                                int64_t b1_cs, float* C1, int64_t ldc1, int M1, int N1, int K1, hipStream_t st, int* rc);
int launch_gemm_f32_pair(const float* A0, int64_t a0_rs, int64_t a0_cs, const float* B0, int64_t b0_rs, int64_t b0_cs, float* C0,
                         int64_t ldc0, int M0, int N0, int K0, int splits0, int acc0, const float* A1, int64_t a1_rs, int64_t a1_cs,
                         const float* B1, int64_t b1_rs, int64_t b1_cs, float* C1, int64_t ldc1, int M1, int N1, int K1, int splits1,
                         int acc1, hipStream_t st);
int launch_rows_update_multi(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim, int optimizer,
                             float lr, hipStream_t st, const mke_count_job* count, const DenseJob* dense);


static int conv_dispatch(const ConvParams& p, bool bwd, hipStream_t st) {
  // dim <= 96: two triples per wavefront, 32 lanes each (dim 75: three passes of 32 lanes instead of two of 64)
  // (the backward only: it is bound by instruction issue; the forward is a latency chain and prefers twice the wavefronts)
  if (!bwd && p.W) {   // conv stack + dense layer, 16 triples per block (the caller checked dim <= 80, n <= 16 MKE_LOSS_PARTIALS)
    const unsigned nb = (unsigned)((p.n + 15) / 16);
    switch ((p.dim + 15) / 16) {
      case 1: hipLaunchKernelGGL((k_attr_conv<1, 16, false, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_attr_conv<2, 16, false, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
      case 3: hipLaunchKernelGGL((k_attr_conv<3, 16, false, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
      case 4: hipLaunchKernelGGL((k_attr_conv<4, 16, false, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
      default: hipLaunchKernelGGL((k_attr_conv<5, 16, false, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
    }
    return check_launch("k_attr_conv");
  }
  if (bwd && p.dz) {   // the backward with its dflat rows computed in the block and the weight-gradient product on rider blocks
    const unsigned nb = (unsigned)p.conv_blocks + (unsigned)(p.tall.gx * p.tall.gz);
    switch ((p.dim + 15) / 16) {
      case 1: hipLaunchKernelGGL((k_attr_conv<1, 16, true, false, 4, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_attr_conv<2, 16, true, false, 4, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
      case 3: hipLaunchKernelGGL((k_attr_conv<3, 16, true, false, 4, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
      case 4: hipLaunchKernelGGL((k_attr_conv<4, 16, true, false, 4, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
      default: hipLaunchKernelGGL((k_attr_conv<5, 16, true, false, 4, true>), dim3(nb), dim3(MKE_BLOCK), 0, st, p); break;
    }
    return check_launch("k_attr_conv");
  }
  const bool half = bwd && p.dim <= 96;
  const int wpl = half ? (p.dim + 31) / 32 : (p.dim + 63) / 64;
  const int nw = half ? 2 : MKE_BLOCK / 64;   // the two-triples-per-wavefront backward runs in blocks of two wavefronts
  const int per_block = nw * (half ? 2 : 1);
  int64_t blocks = (p.n + per_block - 1) / per_block;
  if (blocks > 4096) blocks = 4096;  // one triple per group up to 16K (32K) triples: no half-idle second pass
  if (blocks < 1) blocks = 1;
#define MKE_CONV_CASE(W, L)                                                                                        \
    if (bwd) hipLaunchKernelGGL((k_attr_conv<W, L, true>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, st, p);      \
    else hipLaunchKernelGGL((k_attr_conv<W, L, false>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, st, p);        \
    break;
  if (half) {
    switch (wpl) {
      case 1: hipLaunchKernelGGL((k_attr_conv<1, 32, true, false, 2>), dim3((unsigned)blocks), dim3(128), 0, st, p); break;
      case 2: hipLaunchKernelGGL((k_attr_conv<2, 32, true, false, 2>), dim3((unsigned)blocks), dim3(128), 0, st, p); break;
      case 3: hipLaunchKernelGGL((k_attr_conv<3, 32, true, false, 2>), dim3((unsigned)blocks), dim3(128), 0, st, p); break;
      default: set_error("attribute CNN: dim %d not supported", p.dim); return MKE_E_UNSUPPORTED;
    }
  } else {
    switch (wpl) {
      case 1: MKE_CONV_CASE(1, 64) case 2: MKE_CONV_CASE(2, 64) case 3: MKE_CONV_CASE(3, 64) case 4: MKE_CONV_CASE(4, 64) case 5: MKE_CONV_CASE(5, 64)
      default: set_error("attribute CNN: dim %d not supported (<= 320)", p.dim); return MKE_E_UNSUPPORTED;
    }
  }
#undef MKE_CONV_CASE
  return check_launch("k_attr_conv");
}

static int ws_fold(float* ws, float* gparams, int dim, hipStream_t st) {
  const int np = 2 * dim + CNN_NCONV;
  hipLaunchKernelGGL(k_cnn_ws_fold, dim3((np + MKE_BLOCK - 1) / MKE_BLOCK), dim3(MKE_BLOCK), 0, st, ws, gparams, np, CNN_WS_STRIDE(dim));
  return check_launch("k_cnn_ws_fold");
}

static int dense_update_impl(float* param, float* acc, float* grad, int64_t n, int optimizer, float lr, float* ws, int ws_n,
                             int ws_stride, hipStream_t st) {
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!param || !grad) { set_error("mke_dense_update: NULL pointer"); return MKE_E_NULL; }
  if (optimizer != MKE_OPT_ADAGRAD && optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", optimizer); return MKE_E_UNSUPPORTED; }
  if (optimizer == MKE_OPT_ADAGRAD && !acc) { set_error("Adagrad needs an accumulator"); return MKE_E_NULL; }
  int64_t blocks = (n + MKE_BLOCK - 1) / MKE_BLOCK;
  if (blocks > 1024) blocks = 1024;
  DenseJob j{param, acc, grad, n, optimizer, lr, ws, ws_n, ws_stride, CNN_WS_COPIES};
  hipLaunchKernelGGL(k_dense_update, dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, st, j);
  return check_launch("k_dense_update");
}

}  // namespace mke

extern "C" int mke_attr_conv_fwd(const float* attr_table, int attr_stride, int attr_normalize, const float* lit_table,
                                 int lit_stride, int dim, const int32_t* ia, const int32_t* iv, int64_t n,
                                 const float* params, float* flat, int flat_stride, void* stream) {
  using namespace mke;
  if (n < 0 || dim <= 0 || dim > MKE_MAX_STRIDE || attr_stride < dim || lit_stride < dim) { set_error("mke_attr_conv_fwd: bad n/dim/stride"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!attr_table || !lit_table || !ia || !iv || !params || !flat) { set_error("mke_attr_conv_fwd: NULL pointer"); return MKE_E_NULL; }
  if (flat_stride < 4 * dim || (flat_stride & 1)) { set_error("mke_attr_conv_fwd: flat_stride must be even and >= 4*dim"); return MKE_E_SHAPE; }
  ConvParams p{};
  p.attr = attr_table; p.attr_stride = attr_stride; p.attr_norm = attr_normalize; p.lit = lit_table; p.lit_stride = lit_stride;
  p.dim = dim; p.ia = ia; p.iv = iv; p.n = n; p.params = params; p.flat = flat; p.flat_stride = flat_stride;
  return conv_dispatch(p, false, (hipStream_t)stream);
}

extern "C" int mke_attr_conv_bwd(const float* attr_table, int attr_stride, int attr_normalize, const float* lit_table,
                                 int lit_stride, int dim, const int32_t* ia, const int32_t* iv, int64_t n,
                                 const float* params, const float* dflat, float* grad_params, float* grad_attr,
                                 int32_t* touched_attr, int32_t tag, float* workspace, void* stream) {
  using namespace mke;
  if (n < 0 || dim <= 0 || dim > MKE_MAX_STRIDE || attr_stride < dim || lit_stride < dim) { set_error("mke_attr_conv_bwd: bad n/dim/stride"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!attr_table || !lit_table || !ia || !iv || !params || !dflat || !grad_params) { set_error("mke_attr_conv_bwd: NULL pointer"); return MKE_E_NULL; }
  if (grad_attr && !touched_attr) { set_error("NULL touched array"); return MKE_E_NULL; }
  ConvParams p{};
  p.attr = attr_table; p.attr_stride = attr_stride; p.attr_norm = attr_normalize; p.lit = lit_table; p.lit_stride = lit_stride;
  p.dim = dim; p.ia = ia; p.iv = iv; p.n = n; p.params = params; p.dflat = dflat; p.gparams = grad_params;
  p.gattr = grad_attr; p.gattr_copies = 1; p.gattr_copy_elems = 0; p.tattr = touched_attr; p.tag = tag; p.ws = workspace;
  int rc = conv_dispatch(p, true, (hipStream_t)stream);
  if (rc || !workspace) return rc;
  return ws_fold(workspace, grad_params, dim, (hipStream_t)stream);
}

extern "C" int mke_attr_tail_z(float* z, const float* bias, int64_t n, int dim, double* sumsq_partials, void* stream) {
  using namespace mke;
  if (n < 0 || dim <= 0) { set_error("bad n/dim"); return MKE_E_SHAPE; }
  if (!z || !sumsq_partials) { set_error("mke_attr_tail_z: NULL pointer"); return MKE_E_NULL; }
  hipLaunchKernelGGL(k_attr_tail_z, dim3(MKE_LOSS_PARTIALS), dim3(MKE_BLOCK), 0, (hipStream_t)stream, z, bias, n, dim,
                     sumsq_partials);
  return check_launch("k_attr_tail_z");
}

namespace mke {
static int tail_loss_impl(const float* z, const double* sumsq_partials, const float* ent_table, int ent_stride,
                          int ent_normalize, const int32_t* ih, const float* weights, float scale, int64_t n, int dim,
                          float* gout, double* dot_partials, float* grad_ent, int32_t* touched_ent, int32_t tag,
                          double* loss_partials, const float* w, float* wt, void* stream);
}
extern "C" int mke_attr_tail_loss(const float* z, const double* sumsq_partials, const float* ent_table, int ent_stride,
                                  int ent_normalize, const int32_t* ih, const float* weights, float scale, int64_t n, int dim,
                                  float* gout, double* dot_partials, float* grad_ent, int32_t* touched_ent, int32_t tag,
                                  double* loss_partials, void* stream) {
  return mke::tail_loss_impl(z, sumsq_partials, ent_table, ent_stride, ent_normalize, ih, weights, scale, n, dim, gout, dot_partials,
                             grad_ent, touched_ent, tag, loss_partials, nullptr, nullptr, stream);
}
namespace mke {
static int tail_loss_impl(const float* z, const double* sumsq_partials, const float* ent_table, int ent_stride,
                          int ent_normalize, const int32_t* ih, const float* weights, float scale, int64_t n, int dim,
                          float* gout, double* dot_partials, float* grad_ent, int32_t* touched_ent, int32_t tag,
                          double* loss_partials, const float* w, float* wt, void* stream) {
  if (n < 0 || dim <= 0 || ent_stride % 16 != 0 || dim > ent_stride || ent_stride > MKE_MAX_STRIDE) { set_error("bad n/dim/stride"); return MKE_E_SHAPE; }
  if (!z || !sumsq_partials || !ent_table || !gout || !dot_partials || !loss_partials || (n > 0 && !ih)) { set_error("mke_attr_tail_loss: NULL pointer"); return MKE_E_NULL; }
  if (grad_ent && !touched_ent) { set_error("NULL touched array"); return MKE_E_NULL; }
  TailParams p;
  p.z = z; p.sumsq = sumsq_partials; p.ent = ent_table; p.ent_stride = ent_stride; p.ent_norm = ent_normalize; p.ih = ih;
  p.ws = weights; p.scale = scale; p.n = n; p.dim = dim; p.gout = gout; p.dotp = dot_partials; p.gent = grad_ent;
  p.tent = touched_ent; p.tag = tag; p.lossp = loss_partials; p.w = w; p.wt = wt;
  const int fpl = ent_stride / 16;
  int64_t blocks = (n + MKE_SUBS_PER_BLOCK - 1) / MKE_SUBS_PER_BLOCK;
  blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, MKE_LOSS_PARTIALS));
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_attr_tail_loss<FPL>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  });
  return check_launch("k_attr_tail_loss");
}
}  // namespace mke

extern "C" int mke_attr_tail_bwd(const float* z, float* gout, const double* sumsq_partials, const double* dot_partials,
                                 int64_t n, int dim, float* grad_bias, void* stream) {
  using namespace mke;
  if (n < 0 || dim <= 0) { set_error("bad n/dim"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!z || !gout || !sumsq_partials || !dot_partials) { set_error("mke_attr_tail_bwd: NULL pointer"); return MKE_E_NULL; }
  int64_t blocks = (n * dim + MKE_BLOCK * 4 - 1) / (MKE_BLOCK * 4);
  blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, 1024));
  hipLaunchKernelGGL(k_attr_tail_bwd, dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, (hipStream_t)stream, z, gout, sumsq_partials,
                     dot_partials, n, dim);
  int rc = check_launch("k_attr_tail_bwd");
  if (rc || !grad_bias) return rc;
  hipLaunchKernelGGL(k_colsum_add, dim3(64), dim3(MKE_BLOCK), 0, (hipStream_t)stream, gout, n, dim, grad_bias);
  return check_launch("k_colsum_add");
}

extern "C" int mke_dense_update(float* param, float* acc, float* grad, int64_t n, int optimizer, float lr, void* stream) {
  return mke::dense_update_impl(param, acc, grad, n, optimizer, lr, nullptr, 0, 0, (hipStream_t)stream);
}

extern "C" int64_t mke_attr_scratch_floats(int64_t n, int dim) {
  if (n < 0 || dim <= 0) return 0;
  return n * ((int64_t)dim * 10 + 4);  // flat n*(4d+4) | dflat n*4d | z n*d | gout n*d
}

static int attr_step_impl(const mke_attr_step_args* a, double* lossp, double* ssq, double* dot, void* stream, int phases);

extern "C" int mke_attr_step(const mke_attr_step_args* a, void* stream) {
  mke::TuningScope scope(a ? a->tuning : nullptr);   // the arguments' knobs for the duration of this call
  if (!a) { mke::set_error("mke_attr_step: NULL args"); return MKE_E_NULL; }
  if (!a->partials) { mke::set_error("mke_attr_step: NULL pointer"); return MKE_E_NULL; }
  return attr_step_impl(a, a->partials, a->partials + MKE_LOSS_PARTIALS, a->partials + 2 * MKE_LOSS_PARTIALS, stream, MKE_ATTR_ALL);
}

// The same step in phases (bit mask), for the data-parallel attribute view (multike_amd/distributed_views.py): the
// batch-wide normalisation couples the ranks through two scalars, so the caller all-reduces
//   args->partials[MKE_LOSS_PARTIALS ...)      (sum z^2)  between MKE_ATTR_FWD and MKE_ATTR_TAIL,
//   args->partials[2 MKE_LOSS_PARTIALS ...)    (sum g.z)  between MKE_ATTR_TAIL and MKE_ATTR_BWD
// (replace each array by [global sum, 0, 0, ...]: the consuming kernel adds the entries up itself), and the parameter /
// attribute-table gradients between MKE_ATTR_BWD and MKE_ATTR_UPD.
extern "C" int mke_attr_step_phases(const mke_attr_step_args* a, int phases, void* stream) {
  mke::TuningScope scope(a ? a->tuning : nullptr);   // the arguments' knobs for the duration of this call
  using namespace mke;
  if (!a) { set_error("mke_attr_step_phases: NULL args"); return MKE_E_NULL; }
  if (phases <= 0 || phases > MKE_ATTR_ALL) { set_error("mke_attr_step_phases: bad phase mask"); return MKE_E_SHAPE; }
  if (!a->partials) { set_error("mke_attr_step_phases: NULL partials"); return MKE_E_NULL; }
  return attr_step_impl(a, a->partials, a->partials + MKE_LOSS_PARTIALS, a->partials + 2 * MKE_LOSS_PARTIALS, stream, phases);
}

extern "C" int mke_attr_steps(const mke_attr_step_args* args, const int64_t* step_off, int n_steps, double* loss_ring, int ring,
                              void* stream) {
  mke::TuningScope scope(args ? args->tuning : nullptr);   // the arguments' knobs for the duration of this call
  using namespace mke;
  if (!args || !step_off || !loss_ring) { set_error("mke_attr_steps: NULL pointer"); return MKE_E_NULL; }
  if (n_steps < 0 || ring < 1) { set_error("mke_attr_steps: bad n_steps/ring"); return MKE_E_SHAPE; }
  if (!args->partials || !args->ih || !args->ia || !args->iv) { set_error("mke_attr_steps: NULL pointer in args"); return MKE_E_NULL; }
  if ((int64_t)args->tag + n_steps >= 0x7FFFFFFFLL) { set_error("tag overflow"); return MKE_E_RANGE; }
  for (int s = 0; s < n_steps; ++s) {
    const int64_t lo = step_off[s], hi = step_off[s + 1];
    if (lo < 0 || hi < lo) { set_error("mke_attr_steps: step_off must be non-decreasing"); return MKE_E_SHAPE; }
    mke_attr_step_args a = *args;
    a.ih += lo; a.ia += lo; a.iv += lo;
    if (a.weights) a.weights += lo;
    a.n = hi - lo;
    a.tag = args->tag + s;
    const int rc = attr_step_impl(&a, loss_ring + (int64_t)(s % ring) * MKE_LOSS_PARTIALS, args->partials + MKE_LOSS_PARTIALS,
                                  args->partials + 2 * MKE_LOSS_PARTIALS, stream, MKE_ATTR_ALL);
    if (rc) return rc;
  }
  return MKE_OK;
}

static int attr_step_impl(const mke_attr_step_args* a, double* lossp, double* ssq, double* dot, void* stream, int phases) {
  using namespace mke;
  if (!a) { set_error("mke_attr_step: NULL args"); return MKE_E_NULL; }
  if (a->n < 0 || a->dim <= 0) { set_error("mke_attr_step: bad n/dim"); return MKE_E_SHAPE; }
  if (!a->ent_table || !a->attr_table || !a->lit_table || !a->params || !a->param_grads || !a->scratch || !a->partials) { set_error("mke_attr_step: NULL pointer"); return MKE_E_NULL; }
  if (a->n == 0 && phases == MKE_ATTR_ALL) {
    hipError_t e = hipMemsetAsync(lossp, 0, sizeof(double) * MKE_LOSS_PARTIALS, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("mke_attr_step: memset failed"); return (int)e; }
    return MKE_OK;
  }
  if (a->n == 0) {  // a rank that owns none of the step's triples still takes part in the reductions: its sums are zero
    hipError_t e = hipSuccess;
    if (phases & MKE_ATTR_FWD) e = hipMemsetAsync(ssq, 0, sizeof(double) * MKE_LOSS_PARTIALS, (hipStream_t)stream);
    if (e == hipSuccess && (phases & MKE_ATTR_TAIL)) e = hipMemsetAsync(lossp, 0, sizeof(double) * MKE_LOSS_PARTIALS, (hipStream_t)stream);
    if (e == hipSuccess && (phases & MKE_ATTR_TAIL)) e = hipMemsetAsync(dot, 0, sizeof(double) * MKE_LOSS_PARTIALS, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("mke_attr_step: memset failed"); return (int)e; }
    phases &= MKE_ATTR_UPD;
    if (!phases) return MKE_OK;
  }
  const int d = a->dim;
  const int64_t n = a->n;
  hipStream_t st = (hipStream_t)stream;
  const int fs = 4 * d + 4;  // row stride of flat: column 4d is the constant 1 that carries the bias through the GEMMs
  float* flat = a->scratch;
  float* dflat = flat + n * fs;
  float* z = dflat + n * 4 * d;
  float* gout = z + n * d;
  float* W = a->params + MKE_CNN_CONV_PARAMS(d);
  float* gW = a->param_grads + MKE_CNN_CONV_PARAMS(d);  // bias / its gradient are row 4d of W / gW (packed right behind)
  int rc;
  // forward: conv stack -> dense -> tanh -> batch-global normalisation -> loss
  if (phases & MKE_ATTR_FWD) {
    if (d <= 80 && (n + 15) / 16 <= MKE_LOSS_PARTIALS) {
      // one launch: conv stack, then z = tanh([flat, 1] [W; bias]) on the block's 16 flat rows, per-block sums of z^2
      ConvParams p{};
      p.attr = a->attr_table; p.attr_stride = a->attr_stride; p.attr_norm = a->attr_normalize; p.lit = a->lit_table;
      p.lit_stride = a->lit_stride; p.dim = d; p.ia = a->ia; p.iv = a->iv; p.n = n; p.params = a->params; p.flat = flat;
      p.flat_stride = fs; p.W = W; p.z = z; p.ssq = ssq;
      if ((rc = conv_dispatch(p, false, st))) return rc;
    } else {
      if ((rc = mke_attr_conv_fwd(a->attr_table, a->attr_stride, a->attr_normalize, a->lit_table, a->lit_stride, d, a->ia, a->iv, n,
                                  a->params, flat, fs, stream))) return rc;
      // z = tanh([flat, 1] [W; bias]) with the per-block sums of z^2 written by the GEMM's epilogue
      if ((rc = launch_gemm_f32(flat, fs, 1, W, d, 1, z, d, (int)n, d, 4 * d + 1, 1, 0, st, ssq, 0))) return rc;
    }
  }
  const bool upd = a->update != 0 && (phases & MKE_ATTR_UPD);
  // dim <= 80 (round 4: 64 < dim <= 80; round 5: every narrower width too): the rest of the backward is ONE launch — every convolution-backward block forms its 16 rows of
  // dz = dL/dzpre from g and z with the two batch-wide sums, multiplies them with W^T on the matrix cores straight into its LDS strips
  // (dflat never goes to memory), and [dW; dbias] = [flat, 1]^T dz (split over K, atomic) rides on extra blocks of the same grid,
  // forming dz on the way into its MFMAs.  W^T (the B operand read 16 consecutive floats per quarter-wave) is left in the unused dflat
  // scratch by the loss-tail launch.  4 launches per step: forward, loss tail, backward, updates.
  const bool fused_tail = tune_attr_fused_bwd() && d <= 80 && (int64_t)n * fs < (1LL << 31) && n >= d;
  if ((phases & MKE_ATTR_TAIL) &&
      (rc = tail_loss_impl(z, ssq, a->ent_table, a->ent_stride, a->ent_normalize, a->ih, a->weights, a->scale, n, d, gout, dot,
                           a->ent_grad, a->ent_touched, a->tag, lossp, fused_tail ? W : nullptr, fused_tail ? dflat : nullptr, stream))) return rc;
  // backward.  The fused launch reads W^T from the scratch where the step's loss tail left it.  A call that runs the backward
  // WITHOUT the tail (mke_attr_step_phases: the sharded view all-reduces a scalar between the two) takes the fused path only
  // when the last tail enqueued on this scratch was this step's (same tag, same parameters) and did leave W^T — otherwise (the
  // option toggled in between, a BWD-only call out of the blue) the unfused path, which needs nothing but g in `gout`
  // (round-4 advice).  Host-side bookkeeping of what was ENQUEUED; stream order makes it true on the device.
  // Per host thread (two threads driving two attribute graphs do not see each other's note: round-5 advice), single use: the
  // backward that consumes it, an update of these parameters (W changes) or any other tail on this thread clears it, so a
  // BWD-only call that re-uses a tag later never finds a stale W^T.
  static thread_local const float* wt_scratch = nullptr;
  static thread_local const float* wt_params = nullptr;
  static thread_local int32_t wt_tag = 0;
  if (phases & MKE_ATTR_TAIL) { wt_scratch = fused_tail ? dflat : nullptr; wt_params = a->params; wt_tag = a->tag; }
  if ((phases & MKE_ATTR_UPD) && !(phases & MKE_ATTR_BWD) && wt_params == a->params) wt_scratch = nullptr;   // the parameters move: W^T is stale
  if (phases & MKE_ATTR_BWD) {
  const bool fused = fused_tail && wt_scratch == dflat && wt_params == a->params && wt_tag == a->tag;
  wt_scratch = nullptr;            // consumed (or not applicable): one backward per tail
  if (!fused && (rc = mke_attr_tail_bwd(z, gout, ssq, dot, n, d, nullptr, stream))) return rc;   // gout = dL/dzpre (the fused launch forms it on load)
  if (a->attr_grad && !a->attr_touched) { set_error("mke_attr_step: NULL touched array"); return MKE_E_NULL; }
  ConvParams p{};
  p.attr = a->attr_table; p.attr_stride = a->attr_stride; p.attr_norm = a->attr_normalize; p.lit = a->lit_table;
  p.lit_stride = a->lit_stride; p.dim = d; p.ia = a->ia; p.iv = a->iv; p.n = n; p.params = a->params; p.dflat = dflat;
  p.gparams = a->param_grads; p.gattr = a->attr_grad; p.tattr = a->attr_touched; p.tag = a->tag; p.ws = a->workspace;
  p.gattr_copies = a->attr_grad_copies > 1 ? a->attr_grad_copies : 1; p.gattr_copy_elems = a->n_attr * (int64_t)a->attr_stride;
  if (fused) {
    p.dz = gout /* g */; p.zmat = z; p.ssq = ssq; p.dotp = dot; p.W = dflat /* W^T [dim][4 dim] */; p.conv_blocks = (int)((n + 15) / 16);
    GemmParams& t = p.tall;
    t = GemmParams{};
    t.A = flat; t.B = gout; t.C = gW; t.M = 4 * d + 1; t.N = d; t.K = (int)n; t.a_rs = 1; t.a_cs = fs; t.b_rs = d; t.b_cs = 1; t.ldc = d;
    t.k_per_split = 512;   // 16 x KS(32) k per rider: 190 riders at n = 5000 — with the 313 convolution blocks they are all resident at once
    t.gx = (t.M + 15) / 16; t.gy = 1; t.gz = (t.K + t.k_per_split - 1) / t.k_per_split; t.atomic = 1;
    if ((rc = conv_dispatch(p, true, st))) return rc;
  } else {
    // [dW; dbias] = [flat, 1]^T dz (split-K, atomic)  and  dflat = dz W^T, one launch; then the convolution backward
    if (!launch_gemm_tallsplit_plus(flat, 1, fs, gout, d, gW, d, 4 * d + 1, d, (int)n,
                                    gout, d, 1, W, 1, d, dflat, 4 * d, (int)n, 4 * d, d, st, &rc))
      rc = launch_gemm_f32_pair(flat, 1, fs, gout, d, 1, gW, d, 4 * d + 1, d, (int)n, 32, 1,
                                gout, d, 1, W, 1, d, dflat, 4 * d, (int)n, 4 * d, d, 1, 0, st);
    if (rc) return rc;
    if ((rc = conv_dispatch(p, true, st))) return rc;
  }
  if (!upd && a->workspace && (rc = ws_fold(a->workspace, a->param_grads, d, st))) return rc;   // gradients complete in param_grads
  }
  if (upd) {
    // one launch: entity rows, attribute rows, and (rider blocks) the dense update of the packed parameters, which also
    // adds up the privatised copies of the conv / BN gradients
    if (a->optimizer != MKE_OPT_ADAGRAD && a->optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", a->optimizer); return MKE_E_UNSUPPORTED; }
    if (a->optimizer == MKE_OPT_ADAGRAD && (!a->param_acc || (a->ent_grad && !a->ent_acc) || (a->attr_grad && !a->attr_acc))) { set_error("mke_attr_step: Adagrad needs accumulators"); return MKE_E_NULL; }
    if (a->ent_stride != a->attr_stride && a->ent_grad && a->attr_grad) { set_error("mke_attr_step: entity / attribute strides differ"); return MKE_E_SHAPE; }
    mke_update_table tabs[2];
    int nt = 0;
    if (a->ent_grad) tabs[nt++] = mke_update_table{a->ent_table, a->ent_acc, a->ent_grad, a->ent_touched, a->n_ent, a->ent_normalize, 1, nullptr};
    if (a->attr_grad) tabs[nt++] = mke_update_table{a->attr_table, a->attr_acc, a->attr_grad, a->attr_touched, a->n_attr, a->attr_normalize,
                                                    a->attr_grad_copies > 1 ? a->attr_grad_copies : 1, nullptr};
    DenseJob dj{a->params, a->param_acc, a->param_grads, (int64_t)MKE_CNN_PARAMS(d), a->optimizer, a->lr, a->workspace,
                MKE_CNN_CONV_PARAMS(d), CNN_WS_STRIDE(d), CNN_WS_COPIES};
    UpdateTouchedHint hint(a->n);   // at most one head row per triple
    if ((rc = launch_rows_update_multi(tabs, nt, a->tag, a->ent_grad ? a->ent_stride : a->attr_stride, d, a->optimizer, a->lr, st,
                                       nullptr, &dj))) return rc;
  }
  return MKE_OK;
}
