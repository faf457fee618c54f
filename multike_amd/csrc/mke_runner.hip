// mke_runner.hip — native step runner of the relation view: the host-side loop the reference runs in Python
// (code/MultiKE_model.py:302-312: batch_queue.get() -> session.run) expressed as a C++ loop that only enqueues
// kernels.  Per step: [sampler for the next `sample_chunk` steps when the previous chunk is used up] ->
// fused triple step -> one update launch covering the relation and the entity table.
#include "mke_common.h"

namespace mke {
#define RUN_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("mke_relation_steps: %s: %s", #x, hipGetErrorString(e_)); rc = (int)e_; goto done; } } while (0)
#define RUN_MKE(x) do { rc = (x); if (rc) goto done; } while (0)

// Overlapped schedule: a second stream runs the work that does not depend on the tables — reference counting of step
// s+1 and negative sampling of chunk c+1 — while the main stream scores and updates step s.
static int run_overlapped(const mke_relation_plan* pl, int step_begin, int step_end, hipStream_t mainS, const mke_update_table* ut_in) {
  int rc = MKE_OK;
  const int N = pl->neg_per_pos, SC = pl->sample_chunk;
  hipStream_t side = nullptr;
  hipEvent_t evC[2] = {nullptr, nullptr}, evU[2] = {nullptr, nullptr}, evS[2] = {nullptr, nullptr}, evFree[2] = {nullptr, nullptr},
             evStart = nullptr;
  bool uRecorded[2] = {false, false}, freeRecorded[2] = {false, false};
  int sampled_hi = -1;  // highest chunk index whose sampler launch has been enqueued
  mke_update_table ut[2] = {ut_in[0], ut_in[1]};
  const int first_chunk = step_begin / SC;
  auto chunk_lo = [&](int c) { int s = c * SC; if (s < step_begin) s = step_begin; return s; };
  auto chunk_hi = [&](int c) { int s = (c + 1) * SC; return s < step_end ? s : step_end; };
  auto sample_chunk = [&](int c) -> int {   // on the side stream; chunks are sampled in order, each once
    if (c <= sampled_hi) return MKE_OK;
    sampled_hi = c;
    const int s0 = chunk_lo(c), s1 = chunk_hi(c);
    if (s0 >= s1) return MKE_OK;
    const int b = c & 1;
    if (freeRecorded[b]) { hipError_t e = hipStreamWaitEvent(side, evFree[b], 0); if (e != hipSuccess) return (int)e; }
    const int64_t lo = pl->step_off[s0], n = pl->step_off[s1] - lo;
    if (n * N > pl->neg_chunk_capacity) { set_error("negative chunk buffer too small: %lld > %lld", (long long)(n * N), (long long)pl->neg_chunk_capacity); return MKE_E_SHAPE; }
    const int64_t o = (int64_t)b * pl->neg_chunk_capacity;
    int r = mke_neg_sample(pl->pos_h + lo, pl->pos_r + lo, pl->pos_t + lo, n, lo, pl->pos_kg ? pl->pos_kg + lo : nullptr, pl->sides, N,
                           pl->max_try, pl->seed_lo, pl->seed_hi, pl->stream_id, pl->neg_h + o, pl->neg_r + o, pl->neg_t + o, side);
    if (r) return r;
    hipError_t e = hipEventRecord(evS[b], side);
    return e == hipSuccess ? MKE_OK : (int)e;
  };
  auto neg_off = [&](int s) {  // element offset of step s's negatives inside neg_*
    const int c = s / SC;
    return (int64_t)(c & 1) * pl->neg_chunk_capacity + (pl->step_off[s] - pl->step_off[chunk_lo(c)]) * N;
  };
  auto side_step = [&](int s) -> int {      // reference counts of step s
    const int b = s & 1;
    { const int r0 = sample_chunk(s / SC); if (r0) return r0; }  // normally already done one chunk ahead (below)
    if (uRecorded[b]) { hipError_t e = hipStreamWaitEvent(side, evU[b], 0); if (e != hipSuccess) return (int)e; }
    const int64_t lo = pl->step_off[s], hi = pl->step_off[s + 1], no = neg_off(s);
    int r = mke_count_entity_refs(pl->pos_h + lo, pl->pos_t + lo, hi - lo, pl->neg_h + no, pl->neg_t + no, (hi - lo) * N, N,
                                  pl->ent_ref_count + (int64_t)b * pl->n_ent, side);
    if (r) return r;
    hipError_t e = hipEventRecord(evC[b], side);
    return e == hipSuccess ? MKE_OK : (int)e;
  };

  RUN_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    RUN_HIP(hipEventCreateWithFlags(&evC[i], hipEventDisableTiming));
    RUN_HIP(hipEventCreateWithFlags(&evU[i], hipEventDisableTiming));
    RUN_HIP(hipEventCreateWithFlags(&evS[i], hipEventDisableTiming));
    RUN_HIP(hipEventCreateWithFlags(&evFree[i], hipEventDisableTiming));
  }
  RUN_HIP(hipEventCreateWithFlags(&evStart, hipEventDisableTiming));
  RUN_HIP(hipEventRecord(evStart, mainS));           // everything the caller enqueued before (epoch shuffle) is visible
  RUN_HIP(hipStreamWaitEvent(side, evStart, 0));
  sampled_hi = first_chunk - 1;
  RUN_MKE(side_step(step_begin));
  for (int s = step_begin; s < step_end; ++s) {
    if (s + 1 < step_end) RUN_MKE(side_step(s + 1));   // look ahead: overlaps with this step's kernels
    const int b = s & 1, c = s / SC;
    if (s == chunk_lo(c)) RUN_HIP(hipStreamWaitEvent(mainS, evS[c & 1], 0));
    RUN_HIP(hipStreamWaitEvent(mainS, evC[b], 0));
    const int64_t lo = pl->step_off[s], hi = pl->step_off[s + 1], no = neg_off(s);
    const int32_t tag = pl->tag_base + s;
    int32_t* refc = pl->ent_ref_count + (int64_t)b * pl->n_ent;
    RUN_MKE(mke_triple_score_fwd_bwd_x(pl->ent_table, pl->n_ent, pl->ent_normalize, pl->rel_table, pl->n_rel, pl->rel_normalize,
                                       pl->stride, pl->dim, pl->pos_h + lo, pl->pos_r + lo, pl->pos_t + lo, pl->pos_w ? pl->pos_w + lo : nullptr, hi - lo,
                                       pl->neg_h + no, pl->neg_r + no, pl->neg_t + no, nullptr, (hi - lo) * N, N, pl->scale,
                                       pl->ent_grad, pl->rel_grad, pl->rel_grad_copies, pl->ent_touched, pl->rel_touched, tag, refc,
                                       pl->ent_acc, pl->optimizer, pl->lr,
                                       pl->loss_partials + (int64_t)(s % pl->loss_ring) * MKE_LOSS_PARTIALS, mainS));
    ut[1].ref_count = refc;
    RUN_MKE(mke_rows_update_multi(ut, 2, tag, pl->stride, pl->dim, pl->optimizer, pl->lr, mainS));
    RUN_HIP(hipEventRecord(evU[b], mainS));
    uRecorded[b] = true;
    if (s + 1 == chunk_hi(c)) { RUN_HIP(hipEventRecord(evFree[c & 1], mainS)); freeRecorded[c & 1] = true; }
    // one chunk ahead: chunk c+1's negatives go to the buffer chunk c-1 used; its last step is enqueued by now
    if (s == chunk_lo(c) && chunk_lo(c + 1) < step_end) RUN_MKE(sample_chunk(c + 1));
  }
  // the caller's stream must not run ahead of the side stream's last work (buffers are the caller's)
  RUN_HIP(hipEventRecord(evStart, side));
  RUN_HIP(hipStreamWaitEvent(mainS, evStart, 0));
done:
  for (int i = 0; i < 2; ++i) {
    if (evC[i]) (void)hipEventDestroy(evC[i]);
    if (evU[i]) (void)hipEventDestroy(evU[i]);
    if (evS[i]) (void)hipEventDestroy(evS[i]);
    if (evFree[i]) (void)hipEventDestroy(evFree[i]);
  }
  if (evStart) (void)hipEventDestroy(evStart);
  if (side) (void)hipStreamDestroy(side);
  return rc;
}
}  // namespace mke

extern "C" int mke_relation_steps(const mke_relation_plan* pl, int step_begin, int step_end, void* stream) {
  using namespace mke;
  if (!pl) { set_error("mke_relation_steps: NULL plan"); return MKE_E_NULL; }
  TuningScope scope(pl->tuning);       // the plan's knobs for the duration of this call
  if (!pl->step_off || !pl->pos_h || !pl->pos_r || !pl->pos_t || !pl->loss_partials) { set_error("mke_relation_steps: NULL pointer in plan"); return MKE_E_NULL; }
  if (step_begin < 0 || step_end > pl->n_steps || step_begin > step_end) { set_error("step range [%d,%d) outside [0,%d)", step_begin, step_end, pl->n_steps); return MKE_E_SHAPE; }
  if (pl->loss_ring < 1 || pl->sample_chunk < 1) { set_error("loss_ring and sample_chunk must be >= 1"); return MKE_E_SHAPE; }
  const int N = pl->neg_per_pos;
  if (N > 0 && (!pl->neg_h || !pl->neg_r || !pl->neg_t)) { set_error("NULL negative scratch"); return MKE_E_NULL; }
  if ((int64_t)pl->tag_base + step_end >= 0x7FFFFFFFLL) { set_error("tag overflow"); return MKE_E_RANGE; }

  mke_update_table ut[2] = {};
  ut[0].table = pl->rel_table; ut[0].acc = pl->rel_acc; ut[0].grad = pl->rel_grad; ut[0].touched = pl->rel_touched;
  ut[0].n_rows = pl->n_rel; ut[0].normalize = pl->rel_normalize; ut[0].grad_copies = pl->rel_grad_copies; ut[0].ref_count = nullptr;
  ut[1].table = pl->ent_table; ut[1].acc = pl->ent_acc; ut[1].grad = pl->ent_grad; ut[1].touched = pl->ent_touched;
  ut[1].n_rows = pl->n_ent; ut[1].normalize = pl->ent_normalize; ut[1].grad_copies = 1; ut[1].ref_count = nullptr;
  ut[1].hot = pl->hot;     // hub rows of the entity table (slot == NULL: none)
  const mke_hot_rows* hot = (pl->hot.slot && pl->hot.n_hot > 0) ? &pl->hot : nullptr;

  if (pl->overlap) {
    if (N <= 0 || !pl->ent_ref_count) { set_error("overlap mode needs negatives and the reference-count scratch"); return MKE_E_SHAPE; }
    if (step_begin == step_end) return MKE_OK;
    return run_overlapped(pl, step_begin, step_end, (hipStream_t)stream, ut);
  }

  int64_t chunk_lo = 0;  // first positive (epoch position) whose negatives sit at neg_*[0]
  int chunk_end = step_begin;  // steps < chunk_end are sampled
  bool counted_ahead = false;  // the current step's references were counted by the previous step's update launch
  for (int s = step_begin; s < step_end; ++s) {
    const int64_t lo = pl->step_off[s], hi = pl->step_off[s + 1];
    if (N > 0 && s >= chunk_end && pl->negatives_ready) {  // sampled by the caller ahead of time
      chunk_end = step_end;
      chunk_lo = pl->step_off[step_begin];
    }
    if (N > 0 && s >= chunk_end) {
      chunk_end = s + pl->sample_chunk < step_end ? s + pl->sample_chunk : step_end;
      chunk_lo = lo;
      const int64_t n = pl->step_off[chunk_end] - lo;
      const int rc = mke_neg_sample(pl->pos_h + lo, pl->pos_r + lo, pl->pos_t + lo, n, lo, pl->pos_kg ? pl->pos_kg + lo : nullptr,
                                    pl->sides, N, pl->max_try, pl->seed_lo, pl->seed_hi, pl->stream_id, pl->neg_h, pl->neg_r,
                                    pl->neg_t, stream);
      if (rc) return rc;
    }
    const int64_t no = (lo - chunk_lo) * N;
    const int32_t tag = pl->tag_base + s;
    // exclusive-row fast path: steps alternate between the two halves of ent_ref_count; the counting of step s+1
    // rides in the update launch of step s when its negatives are already sampled (same chunk)
    int32_t* refc = (N > 0 && pl->ent_ref_count) ? pl->ent_ref_count + (int64_t)(s & 1) * pl->n_ent : nullptr;
    int rc = MKE_OK;
    if (refc && !counted_ahead) {
      rc = mke_count_entity_refs(pl->pos_h + lo, pl->pos_t + lo, hi - lo, pl->neg_h + no, pl->neg_t + no, (hi - lo) * N, N, refc, stream);
      if (rc) return rc;
    }
    counted_ahead = false;
    // the next step's negatives exist already (same sample chunk): its reference counts are taken by rider blocks of THIS
    // step's score launch (into the other half of ent_ref_count)
    mke_count_job cj{};
    const mke_count_job* cjp = nullptr;
    if (refc && s + 1 < step_end && s + 1 < chunk_end) {
      const int64_t lo1 = pl->step_off[s + 1], hi1 = pl->step_off[s + 2], no1 = (lo1 - chunk_lo) * N;
      cj.pos_h = pl->pos_h + lo1; cj.pos_t = pl->pos_t + lo1; cj.n_pos = hi1 - lo1;
      cj.neg_h = pl->neg_h + no1; cj.neg_t = pl->neg_t + no1; cj.n_neg = (hi1 - lo1) * N; cj.neg_per_pos = N;
      cj.ref_count = pl->ent_ref_count + (int64_t)((s + 1) & 1) * pl->n_ent;
      cjp = &cj;
      counted_ahead = true;
    }
    const bool in_score = tune_count_in_score() != 0;
    rc = mke_triple_score_fwd_bwd_xch(pl->ent_table, pl->n_ent, pl->ent_normalize, pl->rel_table, pl->n_rel, pl->rel_normalize,
                                     pl->stride, pl->dim, pl->pos_h + lo, pl->pos_r + lo, pl->pos_t + lo, pl->pos_w ? pl->pos_w + lo : nullptr, hi - lo,
                                     N ? pl->neg_h + no : nullptr, N ? pl->neg_r + no : nullptr, N ? pl->neg_t + no : nullptr,
                                     nullptr, (hi - lo) * N, N, pl->scale, pl->ent_grad, pl->rel_grad, pl->rel_grad_copies,
                                     pl->ent_touched, pl->rel_touched, tag, refc, pl->ent_acc, pl->optimizer, pl->lr,
                                     in_score ? cjp : nullptr, hot,
                                     pl->loss_partials + (int64_t)(s % pl->loss_ring) * MKE_LOSS_PARTIALS, stream);
    if (rc) return rc;
    ut[1].ref_count = refc;
    if (in_score) cjp = nullptr;
    UpdateTouchedHint hint(N == 0 ? 2 * (hi - lo) : 0);   // positives only (the cross-KG loops): head + tail per triple
    rc = mke_rows_update_multi_count(ut, 2, tag, pl->stride, pl->dim, pl->optimizer, pl->lr, cjp, stream);
    if (rc) return rc;
  }
  return MKE_OK;
}
