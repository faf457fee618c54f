// mke_shard.hip — device-side bookkeeping of the entity-row sharded (multi-GPU) step (gfx950).
//
// New design (the reference has no multi-device code).  Entity rows are sharded by id % G.  Per step a
// rank needs the set of distinct entity rows its triples reference, grouped by owner, in a FIXED-CAPACITY layout
// [G][C] so that every exchange is an equal-split all-to-all with no size negotiation and no host synchronisation:
//   k_rowset_build : ids -> first-touch detection (atomicExch on a flag per entity) -> slot in the owner's segment
//                    (block-aggregated counters) -> req[owner][slot] = id / G, id_map[id] = owner*C + slot
//   k_rowset_remap : index streams -> compact indices through id_map; clears the flags it used
//   k_rows_gather_padded : owner side, req (local rows, -1 = pad) -> raw rows [n][stride] (pad -> zero row)
//   k_rows_scatter_add   : owner side, returned gradient rows -> atomic add into the shard's gradient scratch
#include "mke_common.h"

namespace mke {

#define MKE_MAX_RANKS 64

struct RowsetParams {
  const int32_t* ids[4];
  int64_t len[4];
  int64_t total;
  int32_t* flags;    // [n_ent] 0 on entry
  int32_t* counts;   // [G] 0 on entry
  int32_t* req;      // [G][C] pre-filled with -1
  int32_t* id_map;   // [n_ent]
  int32_t* overflow; // [1] set to 1 when an owner's segment is full (those ids map to the pad row G*C)
  int G, C;
};

__device__ __forceinline__ int32_t stream_at(const RowsetParams& p, int64_t i) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (i < p.len[s]) return p.ids[s][i];
    i -= p.len[s];
  }
  return -1;
}

// Each thread claims ROWSET_IPT ids per pass: their first-touch atomics are in flight together, and a block issues ONE
// global atomic per owner for MKE_BLOCK * ROWSET_IPT ids — the per-owner counters are same-address atomic chains
// (~20-45 ns per link), so the number of blocks x passes is what bounds this kernel, not the id traffic.
#define ROWSET_IPT 4
__global__ __launch_bounds__(MKE_BLOCK) void k_rowset_build(const RowsetParams p) {
  __shared__ int s_cnt[MKE_MAX_RANKS];
  __shared__ int s_base[MKE_MAX_RANKS];
  const int64_t per_pass = (int64_t)MKE_BLOCK * ROWSET_IPT;
  const int64_t stride = (int64_t)gridDim.x * per_pass;
  for (int64_t base = (int64_t)blockIdx.x * per_pass; base < p.total; base += stride) {  // block-uniform trip count
    if (threadIdx.x < p.G) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int id[ROWSET_IPT], local[ROWSET_IPT];
    bool first[ROWSET_IPT];
#pragma unroll
    for (int k = 0; k < ROWSET_IPT; ++k) {
      const int64_t i = base + k * MKE_BLOCK + threadIdx.x;
      id[k] = i < p.total ? stream_at(p, i) : -1;
    }
#pragma unroll
    for (int k = 0; k < ROWSET_IPT; ++k) first[k] = id[k] >= 0 && atomicExch(&p.flags[id[k]], 1) == 0;
#pragma unroll
    for (int k = 0; k < ROWSET_IPT; ++k) local[k] = first[k] ? atomicAdd(&s_cnt[id[k] % p.G], 1) : 0;
    __syncthreads();
    if (threadIdx.x < p.G) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&p.counts[threadIdx.x], s_cnt[threadIdx.x]) : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ROWSET_IPT; ++k) {
      if (!first[k]) continue;
      const int owner = id[k] % p.G;
      const int slot = s_base[owner] + local[k];
      if (slot < p.C) {
        p.req[(int64_t)owner * p.C + slot] = id[k] / p.G;
        p.id_map[id[k]] = owner * p.C + slot;
      } else {
        // the owner's segment is full: the id goes to the PAD row behind the compact row set (index G*C: an all-zero row
        // on the reading side, a gradient row nobody collects on the writing side), so no other entity's row is read or
        // updated in its place; the step's result is incomplete and flagged
        *p.overflow = 1;
        p.id_map[id[k]] = p.G * p.C;
      }
    }
  }
}

struct RemapParams {
  const int32_t* ids[4];
  int32_t* out[4];
  int64_t len[4];
  int64_t total;
  const int32_t* id_map;
  int32_t* flags;
  int32_t* reset_req;     // nullable: req[i] = -1 for i < reset_req_len (re-initialise for the next build)
  int64_t reset_req_len;
  int32_t* reset_counts;  // nullable
  int n_counts;
  const int32_t* want;    // nullable: [G][C] local rows the other ranks requested (-1 = pad) ...
  int32_t* slot_of;       // ... inverted into slot_of[row * G + g] = slot (entries of rows nobody wants stay -1)
  int G;
  int64_t C;
};

__global__ __launch_bounds__(MKE_BLOCK) void k_rowset_remap(const RemapParams p) {
  if (p.reset_counts && blockIdx.x == 0 && threadIdx.x < p.n_counts) p.reset_counts[threadIdx.x] = 0;
  if (p.reset_req)
    for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < p.reset_req_len; i += (int64_t)gridDim.x * MKE_BLOCK)
      p.reset_req[i] = -1;
  if (p.want)
    for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < p.G * p.C; i += (int64_t)gridDim.x * MKE_BLOCK) {
      const int row = p.want[i];
      if (row >= 0) p.slot_of[(int64_t)row * p.G + i / p.C] = (int32_t)(i % p.C);
    }
  for (int64_t i = (int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x; i < p.total; i += (int64_t)gridDim.x * MKE_BLOCK) {
    int64_t k = i;
    int s = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (s == q && k >= p.len[q]) { k -= p.len[q]; s = q + 1; }
    const int id = p.ids[s][k];
    p.out[s][k] = p.id_map[id];
    p.flags[id] = 0;
  }
}

// pure row copy, one float4 per lane: a block pass covers 256 / (stride / 4) whole rows (12 rows of 320 bytes: 240 of the
// 256 lanes busy; a lane-per-row-quarter mapping leaves 44 of 64 lanes idle on its second pass), four independent
// rows in flight per thread
__global__ __launch_bounds__(MKE_BLOCK) void k_rows_gather_padded(const float* __restrict__ table, int stride,
                                                                  const int32_t* __restrict__ idx, int64_t n,
                                                                  float* __restrict__ out, float* __restrict__ zero_rows) {
  const int q = stride >> 2;        // float4 per row (<= 256 / 4)
  const int rpb = MKE_BLOCK / q;    // rows per block pass
  const int r = threadIdx.x / q, c = threadIdx.x - r * q;
  if (r >= rpb) return;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t step = (int64_t)gridDim.x * rpb;
  for (int64_t i0 = (int64_t)blockIdx.x * rpb + r; i0 < n; i0 += 4 * step) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * step;
      v[u] = z;
      if (i < n) {
        const int row = idx[i];
        if (row >= 0) v[u] = *reinterpret_cast<const float4*>(table + (int64_t)row * stride + 4 * c);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * step;
      if (i < n) {
        *reinterpret_cast<float4*>(out + i * stride + 4 * c) = v[u];
        if (zero_rows) *reinterpret_cast<float4*>(zero_rows + i * stride + 4 * c) = z;
      }
    }
  }
}

template <int FPL>
__global__ __launch_bounds__(MKE_BLOCK) void k_rows_scatter_add(const int32_t* __restrict__ idx, const float* __restrict__ rows,
                                                                int64_t n, int stride, int dim, float* __restrict__ grad,
                                                                int32_t* __restrict__ touched, int32_t tag,
                                                                int32_t* __restrict__ reset_req, int32_t* __restrict__ reset_counts,
                                                                int n_counts) {
  if (reset_counts && blockIdx.x == 0 && threadIdx.x < n_counts) reset_counts[threadIdx.x] = 0;
  const int j = threadIdx.x & 15;
  const int64_t sub0 = ((int64_t)blockIdx.x * MKE_BLOCK + threadIdx.x) >> 4;
  const int64_t nsub = ((int64_t)gridDim.x * MKE_BLOCK) >> 4;
  for (int64_t i = sub0; i < n; i += nsub) {
    const int row = idx[i];
    if (reset_req && j == 0) reset_req[i] = -1;
    if (row < 0) continue;
    float v[FPL];
    load_row<FPL>(rows, i, stride, j, v);
    atomic_add_row<FPL>(grad, row, stride, dim, j, v, 1.0f);
    if (j == 0) touched[row] = tag;
  }
}

static inline unsigned blocks_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (unsigned)b;
}

}  // namespace mke

extern "C" int mke_rowset_build(const int32_t* ids0, int64_t n0, const int32_t* ids1, int64_t n1, const int32_t* ids2,
                                int64_t n2, const int32_t* ids3, int64_t n3, int32_t* flags, int32_t* counts, int32_t* req,
                                int32_t* id_map, int32_t* overflow, int n_ranks, int capacity, void* stream) {
  using namespace mke;
  if (n0 < 0 || n1 < 0 || n2 < 0 || n3 < 0) { set_error("negative length"); return MKE_E_SHAPE; }
  if (n_ranks < 1 || n_ranks > MKE_MAX_RANKS || capacity < 1) { set_error("n_ranks must be in [1,%d], capacity >= 1", MKE_MAX_RANKS); return MKE_E_SHAPE; }
  if (!flags || !counts || !req || !id_map || !overflow) { set_error("mke_rowset_build: NULL pointer"); return MKE_E_NULL; }
  RowsetParams p;
  p.ids[0] = ids0; p.ids[1] = ids1; p.ids[2] = ids2; p.ids[3] = ids3;
  p.len[0] = n0; p.len[1] = n1; p.len[2] = n2; p.len[3] = n3;
  p.total = n0 + n1 + n2 + n3;
  for (int s = 0; s < 4; ++s)
    if (p.len[s] > 0 && !p.ids[s]) { set_error("NULL id stream %d", s); return MKE_E_NULL; }
  if (p.total == 0) return MKE_OK;
  p.flags = flags; p.counts = counts; p.req = req; p.id_map = id_map; p.overflow = overflow; p.G = n_ranks; p.C = capacity;
  // few, fat blocks: one global atomic per (block iteration, owner) on n_ranks counters
  hipLaunchKernelGGL(k_rowset_build, dim3(blocks_for(p.total, MKE_BLOCK * ROWSET_IPT) > 512 ? 512 : blocks_for(p.total, MKE_BLOCK * ROWSET_IPT)),
                     dim3(MKE_BLOCK), 0, (hipStream_t)stream, p);
  return check_launch("k_rowset_build");
}

extern "C" int mke_rowset_remap(const int32_t* ids0, int32_t* out0, int64_t n0, const int32_t* ids1, int32_t* out1, int64_t n1,
                                const int32_t* ids2, int32_t* out2, int64_t n2, const int32_t* ids3, int32_t* out3, int64_t n3,
                                const int32_t* id_map, int32_t* flags, int32_t* reset_req, int64_t reset_req_len,
                                int32_t* reset_counts, int n_counts, const int32_t* want, int32_t* slot_of, int n_ranks,
                                int capacity, void* stream) {
  using namespace mke;
  RemapParams p;
  p.reset_req = reset_req; p.reset_req_len = reset_req ? reset_req_len : 0; p.reset_counts = reset_counts;
  p.n_counts = reset_counts ? n_counts : 0;
  if (reset_req_len < 0 || n_counts < 0 || n_counts > MKE_MAX_RANKS) { set_error("bad reset lengths"); return MKE_E_SHAPE; }
  p.ids[0] = ids0; p.ids[1] = ids1; p.ids[2] = ids2; p.ids[3] = ids3;
  p.out[0] = out0; p.out[1] = out1; p.out[2] = out2; p.out[3] = out3;
  p.len[0] = n0; p.len[1] = n1; p.len[2] = n2; p.len[3] = n3;
  p.total = 0;
  for (int s = 0; s < 4; ++s) {
    if (p.len[s] < 0) { set_error("negative length"); return MKE_E_SHAPE; }
    if (p.len[s] > 0 && (!p.ids[s] || !p.out[s])) { set_error("NULL stream %d", s); return MKE_E_NULL; }
    p.total += p.len[s];
  }
  if (want && (!slot_of || n_ranks < 1 || n_ranks > MKE_MAX_RANKS || capacity < 1)) { set_error("mke_rowset_remap: bad inversion arguments"); return MKE_E_SHAPE; }
  p.want = want; p.slot_of = slot_of; p.G = n_ranks; p.C = capacity;
  if (p.total == 0 && !reset_req && !reset_counts && !want) return MKE_OK;
  if (p.total > 0 && (!id_map || !flags)) { set_error("mke_rowset_remap: NULL pointer"); return MKE_E_NULL; }
  p.id_map = id_map; p.flags = flags;
  int64_t work = p.total > p.reset_req_len ? p.total : p.reset_req_len;
  if (want && (int64_t)n_ranks * capacity > work) work = (int64_t)n_ranks * capacity;
  hipLaunchKernelGGL(k_rowset_remap, dim3(blocks_for(work, MKE_BLOCK)), dim3(MKE_BLOCK), 0,
                     (hipStream_t)stream, p);
  return check_launch("k_rowset_remap");
}

extern "C" int mke_rows_gather_padded(const float* table, int stride, const int32_t* idx, int64_t n, float* out,
                                      float* zero_rows, void* stream) {
  using namespace mke;
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!table || !idx || !out) { set_error("mke_rows_gather_padded: NULL pointer"); return MKE_E_NULL; }
  if (stride <= 0 || stride % 16 != 0 || stride > MKE_MAX_STRIDE) { set_error("bad stride %d", stride); return MKE_E_SHAPE; }
  hipLaunchKernelGGL(k_rows_gather_padded, dim3(blocks_for(n, (MKE_BLOCK / (stride / 4)) * 4)), dim3(MKE_BLOCK), 0, (hipStream_t)stream,
                     table, stride, idx, n, out, zero_rows);
  return check_launch("k_rows_gather_padded");
}

extern "C" int mke_rows_scatter_add(const int32_t* idx, const float* rows, int64_t n, int stride, int dim, float* grad,
                                    int32_t* touched, int32_t tag, int32_t* reset_req, int32_t* reset_counts, int n_counts,
                                    void* stream) {
  using namespace mke;
  if (n < 0) { set_error("negative n"); return MKE_E_SHAPE; }
  if (n == 0) return MKE_OK;
  if (!idx || !rows || !grad || !touched) { set_error("mke_rows_scatter_add: NULL pointer"); return MKE_E_NULL; }
  if (reset_counts && (n_counts < 0 || n_counts > MKE_MAX_RANKS)) { set_error("bad n_counts"); return MKE_E_SHAPE; }
  if (stride <= 0 || stride % 16 != 0 || dim <= 0 || dim > stride || stride > MKE_MAX_STRIDE) { set_error("bad stride/dim"); return MKE_E_SHAPE; }
  const int fpl = stride / 16;
  MKE_DISPATCH_FPL(fpl, {
    hipLaunchKernelGGL((k_rows_scatter_add<FPL>), dim3(blocks_for(n, MKE_SUBS_PER_BLOCK)), dim3(MKE_BLOCK), 0,
                       (hipStream_t)stream, idx, rows, n, stride, dim, grad, touched, tag, reset_req, reset_counts, n_counts);
  });
  return check_launch("k_rows_scatter_add");
}
