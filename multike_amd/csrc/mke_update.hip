// mke_update.hip — per-row optimizer step on the rows touched in this step (gfx950).
//
// Semantics: TF1's dense apply_gradients on a variable read through tf.nn.l2_normalize(var, 1)
// (code/base/initializers.py:26, code/MultiKE_model.py:15-31).  TF back-propagates through the
// normalisation (Jacobian below) and runs ApplyAdagrad over the whole [n,dim] variable; a row whose
// gradient is exactly zero is left bit-identical (acc += 0, w -= 0), so visiting only the rows flagged
// by the scatter kernels is the same function at a fraction of the traffic.
//
// Shape: a wavefront takes 16 consecutive rows; lanes read touched[base + (lane & 15)] (one 64-byte read), the
// wave ballots the flags and deals the set bits round-robin to its four 16-lane quarter-waves (a dense chunk — the
// relation table, where every row is hit every step — is 4 rows deep per quarter instead of 16); each visited row is
// 3 row reads (grad, w, acc) + 3 row writes (0, w, acc).
#include "mke_common.h"

namespace mke {

struct UpdateParams {
  float* __restrict__ table;
  float* __restrict__ acc;
  float* __restrict__ grad;
  int copies;
  const int32_t* __restrict__ touched;
  int32_t* __restrict__ refcount;  // nullable
  int32_t tag;
  int64_t n_rows;
  int stride, dim;
  int normalize, optimizer;
  float lr;
  int chunk;  // rows per wavefront: 16, or 4 for small dense tables (every quarter-wave gets its own row at once)
  // owner side of the sharded step: gradient = sum over ranks of the returned rows (mke_update_table.slot_of)
  const float* __restrict__ src_rows;
  int32_t* __restrict__ slot_of;
  int n_ranks;
  int64_t capacity;
  // hub rows (mke_hot_rows): private copies of a row's gradient behind the table's own rows in `grad`
  const int32_t* __restrict__ hot_slot;
  int32_t n_hot, hot_copies;
  int64_t hot_row0;
};

// SHARD: the owner side of the sharded step (gradient gathered through slot_of); a compile-time switch so that the
// single-GPU step does not carry its registers and branches
template <int FPL, bool SHARD, bool HOT = false>   // HOT: hub rows compiled in (their batched copy loads cost the plain instantiation 40 registers)
__device__ __forceinline__ void update_one_row(const UpdateParams& p, int64_t row, int j) {
  constexpr int64_t STRIDE = FPL * 16;   // == p.stride (the dispatch picks FPL from it): a shift-add, not a 64-bit multiply
  float* gp = p.grad + row * STRIDE + j;
  float* wp = p.table + row * STRIDE + j;
  float g[FPL], w[FPL], a[FPL];
  if (SHARD && p.slot_of) {  // rank-ordered sum of the rows sent back for this row (deterministic, no atomics)
    int32_t* so = p.slot_of + row * (int64_t)p.n_ranks;
#pragma unroll
    for (int k = 0; k < FPL; ++k) g[k] = 0.f;
    for (int r = 0; r < p.n_ranks; ++r) {
      const int s = so[r];  // uniform over the quarter-wave
      if (s >= 0) {
        const float* src = p.src_rows + ((int64_t)r * p.capacity + s) * p.stride + j;
#pragma unroll
        for (int k = 0; k < FPL; ++k) g[k] += src[k * 16];
      }
    }
    for (int r = j; r < p.n_ranks; r += 16) so[r] = -1;
  } else {
#pragma unroll
    for (int k = 0; k < FPL; ++k) g[k] = gp[k * 16];
  }
  if (p.copies > 1) {  // privatised gradient: sum (and consume) the other copies
    const int64_t ce = p.n_rows * (int64_t)p.stride;
    for (int c = 1; c < p.copies; ++c) {
      float* gc = gp + c * ce;
#pragma unroll
      for (int k = 0; k < FPL; ++k) { g[k] += gc[k * 16]; gc[k * 16] = 0.f; }
    }
  }
  if (HOT && !SHARD && p.hot_slot) {  // a hub row: the groups' flushes went to its private copies
    const int hs = p.hot_slot[row];
    if (hs >= 0) {
      // HB copies in flight together (one at a time they were a 16-deep chain of round trips: +4.7 us on the launch)
      constexpr int HB = FPL <= 5 ? 4 : 2;
      for (int c0 = 0; c0 < p.hot_copies; c0 += HB) {
        float t[HB][FPL];
#pragma unroll
        for (int c = 0; c < HB; ++c) {
          float* gc = p.grad + (p.hot_row0 + (int64_t)min(c0 + c, p.hot_copies - 1) * p.n_hot + hs) * STRIDE + j;
#pragma unroll
          for (int k = 0; k < FPL; ++k) t[c][k] = gc[k * 16];
        }
#pragma unroll
        for (int c = 0; c < HB; ++c) {
          if (c0 + c < p.hot_copies) {
            float* gc = p.grad + (p.hot_row0 + (int64_t)(c0 + c) * p.n_hot + hs) * STRIDE + j;
#pragma unroll
            for (int k = 0; k < FPL; ++k) { g[k] += t[c][k]; gc[k * 16] = 0.f; }
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < FPL; ++k) w[k] = wp[k * 16];
  const bool adagrad = p.optimizer == MKE_OPT_ADAGRAD;
  float* ap = adagrad ? p.acc + row * STRIDE + j : nullptr;
  if (adagrad) {
#pragma unroll
    for (int k = 0; k < FPL; ++k) a[k] = ap[k * 16];
  }
  if (!(SHARD && p.slot_of)) {
#pragma unroll
    for (int k = 0; k < FPL; ++k) gp[k * 16] = 0.f;  // consume: restore the all-zero invariant
  }
  if (p.refcount && j == 0) p.refcount[row] = 0;

  if (p.normalize) {
    float s = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      s = fmaf(w[k], w[k], s);
      dot = fmaf(w[k], g[k], dot);
    }
    s = sub16_sum(s);
    dot = sub16_sum(dot);
    const float inv = rsqrtf(fmaxf(s, MKE_L2_EPS));
    // g = (ghat - what*(what.ghat)) * inv ; what.ghat = inv*dot ; what = w*inv
    const float coef = (s > MKE_L2_EPS) ? dot * inv * inv : 0.f;
#pragma unroll
    for (int k = 0; k < FPL; ++k) g[k] = (g[k] - w[k] * coef) * inv;
  }
  if (adagrad) {
#pragma unroll
    for (int k = 0; k < FPL; ++k) {
      a[k] = fmaf(g[k], g[k], a[k]);
      w[k] -= p.lr * g[k] * adagrad_scale(a[k]);
    }
#pragma unroll
    for (int k = 0; k < FPL; ++k) ap[k * 16] = a[k];
  } else {
#pragma unroll
    for (int k = 0; k < FPL; ++k) w[k] -= p.lr * g[k];
  }
#pragma unroll
  for (int k = 0; k < FPL; ++k) wp[k * 16] = w[k];
}

// Wide rows on sparse tables (stride a multiple of 64 floats, one flag per lane): a WHOLE wavefront per row, lane l holding
// columns [l V, (l + 1) V).  With the quarter-wave shape a wavefront that finds one touched row among its 64 flags (the C5
// shape: 1.1 on average) issues 96 quarter-populated dword loads / stores for it — the row update was bound by vector-memory
// instruction issue (3.3 M wave-instructions per launch), not by the 204 MB it moves; here a row is 3 loads + 3 stores of
// 16 bytes per lane.  Same arithmetic per element; the two row sums are reduced over 64 lanes instead of 16.
template <int V, bool HOT = false>
__device__ __forceinline__ void update_one_row_wave(const UpdateParams& p, int64_t row, int lane) {
  const int64_t off = row * (int64_t)(V * 64) + lane * V;
  float* gp = p.grad + off;
  float* wp = p.table + off;
  const bool adagrad = p.optimizer == MKE_OPT_ADAGRAD;
  float* ap = adagrad ? p.acc + off : nullptr;
  float g[V], w[V], a[V];
  auto ld = [&](const float* q, float (&v)[V]) {
    if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(q); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = q[k];
    }
  };
  auto st = [&](float* q, const float (&v)[V]) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
      for (int k = 0; k < V; ++k) q[k] = v[k];
    }
  };
  ld(gp, g);
  ld(wp, w);
  if (adagrad) ld(ap, a);
  if (p.copies > 1) {  // privatised gradient: sum (and consume) the other copies
    const int64_t ce = p.n_rows * (int64_t)(V * 64);
    for (int c = 1; c < p.copies; ++c) {
      float t[V];
      ld(gp + c * ce, t);
#pragma unroll
      for (int k = 0; k < V; ++k) { g[k] += t[k]; t[k] = 0.f; }
      st(gp + c * ce, t);
    }
  }
  if (HOT && p.hot_slot) {  // a hub row: the groups' flushes went to its private copies
    const int hs = p.hot_slot[row];
    if (hs >= 0) {
      for (int c = 0; c < p.hot_copies; ++c) {
        float* gc = p.grad + (p.hot_row0 + (int64_t)c * p.n_hot + hs) * (int64_t)(V * 64) + lane * V;
        float t[V];
        ld(gc, t);
#pragma unroll
        for (int k = 0; k < V; ++k) { g[k] += t[k]; t[k] = 0.f; }
        st(gc, t);
      }
    }
  }
  {
    float z[V];
#pragma unroll
    for (int k = 0; k < V; ++k) z[k] = 0.f;
    st(gp, z);  // consume: restore the all-zero invariant
  }
  if (p.refcount && lane == 0) p.refcount[row] = 0;
  auto wave_sum = [](float v) {
    v = sub16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  };
  if (p.normalize) {
    float s = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      s = fmaf(w[k], w[k], s);
      dot = fmaf(w[k], g[k], dot);
    }
    s = wave_sum(s);
    dot = wave_sum(dot);
    const float inv = rsqrtf(fmaxf(s, MKE_L2_EPS));
    const float coef = (s > MKE_L2_EPS) ? dot * inv * inv : 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = (g[k] - w[k] * coef) * inv;
  }
  if (adagrad) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      a[k] = fmaf(g[k], g[k], a[k]);
      w[k] -= p.lr * g[k] * adagrad_scale(a[k]);
    }
    st(ap, a);
  } else {
#pragma unroll
    for (int k = 0; k < V; ++k) w[k] -= p.lr * g[k];
  }
  st(wp, w);
}

// One wavefront, one 16-row chunk: ballot the flags, deal the set bits round-robin to the four quarter-waves.  Every
// quarter looks up ITS row of the round (the (4*round + q)-th set bit) so that the four row visits of a round execute
// together in one instruction stream — a branch per set bit would serialise them.
template <int FPL, bool SHARD, bool HOT = false>
__device__ __forceinline__ void walk_chunk(const UpdateParams& p, int64_t wave, int j, int q) {
  const int64_t base = wave * p.chunk;
  if (base >= p.n_rows) return;  // wave-uniform
  // chunk <= 16: the four quarter-waves hold the same flags (lane j <-> row base + j); chunk == 64: one flag per lane
  const int lane = q * 16 + j;
  const int64_t r = base + (p.chunk == 64 ? lane : (j & (p.chunk - 1)));
  bool mine = r < p.n_rows;
  if (SHARD && mine && p.slot_of) {
    const int32_t* so = p.slot_of + r * (int64_t)p.n_ranks;
    int any = -1;
    for (int g = 0; g < p.n_ranks; ++g) any = max(any, so[g]);
    mine = any >= 0;
  } else if (mine) {
    mine = p.touched == nullptr || p.touched[r] == p.tag;
  }
  uint64_t m = __ballot(mine);
  if constexpr (!SHARD && FPL % 4 == 0) {
    if (p.chunk == 64) {   // one flag per lane, wide rows: the whole wavefront visits the set rows one after the other
      while (m) {
        update_one_row_wave<FPL / 4, HOT>(p, base + __builtin_ctzll(m), lane);
        m &= m - 1;
      }
      return;
    }
  }
  if (p.chunk < 64) m &= (1ull << p.chunk) - 1ull;
  // quarter q takes the q-th, (q+4)-th, ... set bit: a running copy of the mask with the bits already dealt removed
  for (int k = 0; k < q; ++k) m &= m - 1;
  while (__ballot(m != 0)) {   // wave-uniform trip count: the quarters visit their rows of a round together
    if (m) update_one_row<FPL, SHARD, HOT>(p, base + __builtin_ctzll(m), j);
    m &= m - 1; m &= m - 1; m &= m - 1; m &= m - 1;
  }
}

template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_rows_update(const UpdateParams p) {
  const int j = threadIdx.x & 15;
  const int64_t wave = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 6;
  walk_chunk<FPL, false>(p, wave, j, (threadIdx.x & 63) >> 4);
}

struct MultiUpdateParams {
  UpdateParams t[MKE_MAX_UPDATE_TABLES];
  int64_t block_end[MKE_MAX_UPDATE_TABLES];  // exclusive prefix of blocks per table
  int n_tables;
  // optional rider: reference counting of the next step in blocks [block_end[n_tables-1], gridDim.x)
  mke_count_job cj;
  int count_blocks;
  // optional rider: dense parameter update in the blocks after those
  DenseJob dj;
  int dense_blocks;
};

template <int FPL, bool SHARD, bool HOT = false>
__global__ __launch_bounds__(MKE_BLOCK) void k_rows_update_multi(const MultiUpdateParams mp) {
  // rider blocks come FIRST in the grid: they are dispatched with the first update blocks and finish under them (at the end
  // of the grid they were a tail of their own: 13.6 -> 16.7 us at the C2 shape)
  const int64_t riders = (int64_t)mp.count_blocks + mp.dense_blocks;
  if ((int64_t)blockIdx.x < mp.count_blocks) {  // count the next step's entity references
    count_refs_range(mp.cj, blockIdx.x, mp.count_blocks);
    return;
  }
  if ((int64_t)blockIdx.x < riders) {  // dense parameter update
    dense_update_range(mp.dj, (int64_t)blockIdx.x - mp.count_blocks, mp.dense_blocks);
    return;
  }
  const int64_t ub = (int64_t)blockIdx.x - riders;   // index among the update blocks
  int ti = 0;
  int64_t first = 0;
#pragma unroll
  for (int k = 0; k < MKE_MAX_UPDATE_TABLES - 1; ++k) {
    if (ti == k && k + 1 < mp.n_tables && ub >= mp.block_end[k]) { first = mp.block_end[k]; ti = k + 1; }
  }
  const UpdateParams& p = mp.t[ti];
  const int j = threadIdx.x & 15;
  const int64_t wave = ((ub - first) * MKE_BLOCK + threadIdx.x) >> 6;
  walk_chunk<FPL, SHARD, HOT>(p, wave, j, (threadIdx.x & 63) >> 4);
}

// 4 rows for small dense tables (every quarter-wave gets a row at once); 16 for tables where a step touches a good share of the
// rows (C2: 17 % of 200K: 2.8 flags set per 16); 64 (one flag per lane) for large tables touched sparsely (C5: 1.7 % of 2M rows —
// a quarter as many wavefronts, each still finding about one row: 77 -> 59 us).  "update_chunk" = 16 / 64 forces one.
thread_local int64_t g_update_touched_hint = 0;
static inline int chunk_for(int64_t n_rows) {
  if (n_rows <= 16384) return 4;
  if (tune_update_chunk()) return tune_update_chunk();
  if (n_rows > 500000) return 64;
  return (g_update_touched_hint > 0 && g_update_touched_hint * 16 <= n_rows) ? 64 : 16;   // sparse step on a mid-sized table
}

int launch_rows_update_multi(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim, int optimizer,
                             float lr, hipStream_t st, const mke_count_job* count = nullptr, const DenseJob* dense = nullptr) {
  MultiUpdateParams mp;
  mp.n_tables = n_tables;
  mp.count_blocks = 0;
  mp.dense_blocks = 0;
  mp.dj = DenseJob{};
  mp.cj = mke_count_job{};
  int64_t blocks = 0;
  for (int k = 0; k < n_tables; ++k) {
    const int64_t rows_per_block = (int64_t)(MKE_BLOCK / 64) * chunk_for(tables[k].n_rows);
    UpdateParams& p = mp.t[k];
    p.table = tables[k].table; p.acc = tables[k].acc; p.grad = tables[k].grad; p.touched = tables[k].touched;
    p.copies = tables[k].grad_copies < 1 ? 1 : tables[k].grad_copies;
    p.refcount = tables[k].ref_count;
    p.src_rows = tables[k].src_rows; p.slot_of = tables[k].slot_of; p.n_ranks = tables[k].n_ranks; p.capacity = tables[k].capacity;
    if (p.slot_of) p.copies = 1;
    const bool hot = tables[k].hot.slot && tables[k].hot.n_hot > 0 && !tables[k].slot_of;
    p.hot_slot = hot ? tables[k].hot.slot : nullptr; p.n_hot = hot ? tables[k].hot.n_hot : 0;
    p.hot_copies = hot ? tables[k].hot.copies : 1; p.hot_row0 = hot ? tables[k].hot.row0 : 0;
    p.tag = tag; p.n_rows = tables[k].n_rows; p.stride = stride; p.dim = dim; p.normalize = tables[k].normalize;
    p.optimizer = optimizer; p.lr = lr; p.chunk = chunk_for(tables[k].n_rows);
    blocks += (tables[k].n_rows + rows_per_block - 1) / rows_per_block;
    mp.block_end[k] = blocks;
  }
  if (count && count->n_pos + count->n_neg > 0) {
    int64_t cb = (count->n_pos + count->n_neg + MKE_BLOCK - 1) / MKE_BLOCK;
    if (cb > 1024) cb = 1024;
    mp.cj = *count;
    if (mp.cj.neg_per_pos < 1) mp.cj.neg_per_pos = 1;
    mp.count_blocks = (int)cb;
    blocks += cb;
  }
  if (dense && dense->n > 0) {
    int64_t db = (dense->n + MKE_BLOCK - 1) / MKE_BLOCK;
    if (db > 256) db = 256;
    mp.dj = *dense;
    mp.dense_blocks = (int)db;
    blocks += db;
  }
  if (blocks == 0) return MKE_OK;
  const int fpl = stride / 16;
  bool shard = false;
  for (int k = 0; k < n_tables; ++k) shard = shard || tables[k].slot_of != nullptr;
  bool hot = false;
  for (int k = 0; k < n_tables; ++k) hot = hot || mp.t[k].hot_slot != nullptr;
  if (shard) {
    MKE_DISPATCH_FPL(fpl, { hipLaunchKernelGGL((k_rows_update_multi<FPL, true>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, st, mp); });
  } else if (hot) {
    MKE_DISPATCH_FPL(fpl, { hipLaunchKernelGGL((k_rows_update_multi<FPL, false, true>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, st, mp); });
  } else {
    MKE_DISPATCH_FPL(fpl, { hipLaunchKernelGGL((k_rows_update_multi<FPL, false>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, st, mp); });
  }
  return check_launch("k_rows_update_multi");
}

}  // namespace mke

extern "C" int mke_rows_update(float* table, float* acc, float* grad, int grad_copies, const int32_t* touched, int32_t tag,
                               int64_t n_rows, int stride, int dim, int normalize, int optimizer, float lr,
                               void* stream) {
  using namespace mke;
  if (!table || !grad) { set_error("mke_rows_update: NULL table/grad"); return MKE_E_NULL; }
  if (optimizer != MKE_OPT_ADAGRAD && optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", optimizer); return MKE_E_UNSUPPORTED; }
  if (optimizer == MKE_OPT_ADAGRAD && !acc) { set_error("Adagrad needs an accumulator"); return MKE_E_NULL; }
  if (n_rows < 0) { set_error("negative n_rows"); return MKE_E_SHAPE; }
  if (stride <= 0 || stride % 16 != 0 || dim <= 0 || dim > stride || stride > MKE_MAX_STRIDE) {
    set_error("bad stride/dim: stride=%d dim=%d", stride, dim);
    return MKE_E_SHAPE;
  }
  if (n_rows == 0) return MKE_OK;
  UpdateParams p;
  p.table = table; p.acc = acc; p.grad = grad; p.copies = grad_copies < 1 ? 1 : grad_copies; p.touched = touched; p.tag = tag; p.n_rows = n_rows;
  p.refcount = nullptr;
  p.src_rows = nullptr; p.slot_of = nullptr; p.n_ranks = 0; p.capacity = 0;
  p.hot_slot = nullptr; p.n_hot = 0; p.hot_copies = 1; p.hot_row0 = 0;
  p.stride = stride; p.dim = dim; p.normalize = normalize; p.optimizer = optimizer; p.lr = lr;
  p.chunk = chunk_for(n_rows);
  const int64_t rows_per_block = (int64_t)(MKE_BLOCK / 64) * p.chunk;
  const int64_t blocks = (n_rows + rows_per_block - 1) / rows_per_block;
  hipStream_t st = (hipStream_t)stream;
  const int fpl = stride / 16;
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_rows_update<FPL>), dim3((unsigned)blocks), dim3(MKE_BLOCK), 0, st, p);
  });
  return check_launch("k_rows_update");
}

extern "C" int mke_rows_update_multi_count(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim,
                                           int optimizer, float lr, const mke_count_job* count, void* stream);

extern "C" int mke_rows_update_multi_t(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim, int optimizer,
                                       float lr, const mke_count_job* count, const mke_tuning* tuning, void* stream) {
  mke::TuningScope scope(tuning);      // this call's knobs (NULL: the process defaults)
  return mke_rows_update_multi_count(tables, n_tables, tag, stride, dim, optimizer, lr, count, stream);
}

extern "C" int mke_rows_update_multi(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim,
                                     int optimizer, float lr, void* stream) {
  return mke_rows_update_multi_count(tables, n_tables, tag, stride, dim, optimizer, lr, nullptr, stream);
}

extern "C" int mke_rows_update_multi_count(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim,
                                           int optimizer, float lr, const mke_count_job* count, void* stream) {
  using namespace mke;
  if (count) {
    if (count->n_pos < 0 || count->n_neg < 0) { set_error("count job: negative count"); return MKE_E_SHAPE; }
    if (count->n_pos + count->n_neg > 0 && (!count->ref_count || !count->pos_h || !count->pos_t || (count->n_neg > 0 && (!count->neg_h || !count->neg_t)))) { set_error("count job: NULL pointer"); return MKE_E_NULL; }
    if (count->n_neg > 0 && (count->neg_per_pos < 1 || count->n_neg != count->n_pos * (int64_t)count->neg_per_pos)) { set_error("count job needs grouped negatives"); return MKE_E_SHAPE; }
  }
  if (!tables) { set_error("mke_rows_update_multi: NULL tables"); return MKE_E_NULL; }
  if (n_tables < 1 || n_tables > MKE_MAX_UPDATE_TABLES) { set_error("n_tables must be in [1,%d]", MKE_MAX_UPDATE_TABLES); return MKE_E_SHAPE; }
  if (optimizer != MKE_OPT_ADAGRAD && optimizer != MKE_OPT_SGD) { set_error("unsupported optimizer %d", optimizer); return MKE_E_UNSUPPORTED; }
  if (stride <= 0 || stride % 16 != 0 || dim <= 0 || dim > stride || stride > MKE_MAX_STRIDE) { set_error("bad stride/dim: stride=%d dim=%d", stride, dim); return MKE_E_SHAPE; }
  bool any_shard = false, any_hot = false;
  for (int k = 0; k < n_tables; ++k) {
    any_shard = any_shard || tables[k].slot_of != nullptr;
    any_hot = any_hot || (tables[k].hot.slot != nullptr && tables[k].hot.n_hot > 0);
    if (tables[k].hot.slot && tables[k].hot.n_hot > 0) {   // hub rows: private copies behind the table's own rows (round-5 advice: validated here)
      if (tables[k].hot.copies < 1 || tables[k].hot.copies > 64 || tables[k].hot.row0 < tables[k].n_rows) { set_error("table %d: hub rows need 1 <= copies <= 64 and row0 >= n_rows", k); return MKE_E_SHAPE; }
      if (tables[k].grad_copies > 1) { set_error("table %d: hub rows and a wholly privatised gradient (grad_copies > 1) exclude each other", k); return MKE_E_UNSUPPORTED; }
      if (tables[k].slot_of) { set_error("table %d: hub rows on a table whose gradient comes through slot_of", k); return MKE_E_UNSUPPORTED; }
    }
    if (!tables[k].table || (!tables[k].grad && !tables[k].slot_of)) { set_error("table %d: NULL table/grad", k); return MKE_E_NULL; }
    if (tables[k].slot_of && (!tables[k].src_rows || tables[k].n_ranks < 1 || tables[k].n_ranks > 64 || tables[k].capacity < 1)) {
      set_error("table %d: slot_of needs src_rows, 1 <= n_ranks <= 64 and capacity >= 1", k);
      return MKE_E_SHAPE;
    }
    if (optimizer == MKE_OPT_ADAGRAD && !tables[k].acc) { set_error("table %d: Adagrad needs an accumulator", k); return MKE_E_NULL; }
    if (tables[k].n_rows < 0) { set_error("negative n_rows"); return MKE_E_SHAPE; }
  }
  // one launch instantiates EITHER the sharded-gradient form OR the hub-row form: a call that mixed them dropped the hub rows'
  // copies silently (never added, never re-zeroed)
  if (any_shard && any_hot) { set_error("mke_rows_update_multi: a slot_of table and a hub-row table cannot share one call"); return MKE_E_UNSUPPORTED; }
  return launch_rows_update_multi(tables, n_tables, tag, stride, dim, optimizer, lr, (hipStream_t)stream, count);
}
