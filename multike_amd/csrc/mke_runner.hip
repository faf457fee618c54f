// mke_runner.hip — native step runner of the relation view: the host-side loop the reference runs in Python
// (code/MultiKE_model.py:302-312: batch_queue.get() -> session.run) expressed as a C++ loop that only enqueues
// kernels.  Per step: [sampler for the next `sample_chunk` steps when the previous chunk is used up] ->
// fused triple step -> one update launch covering the relation and the entity table.
#include "mke_common.h"

extern "C" int mke_relation_steps(const mke_relation_plan* pl, int step_begin, int step_end, void* stream) {
  using namespace mke;
  if (!pl) { set_error("mke_relation_steps: NULL plan"); return MKE_E_NULL; }
  if (!pl->step_off || !pl->pos_h || !pl->pos_r || !pl->pos_t || !pl->loss_partials) { set_error("mke_relation_steps: NULL pointer in plan"); return MKE_E_NULL; }
  if (step_begin < 0 || step_end > pl->n_steps || step_begin > step_end) { set_error("step range [%d,%d) outside [0,%d)", step_begin, step_end, pl->n_steps); return MKE_E_SHAPE; }
  if (pl->loss_ring < 1 || pl->sample_chunk < 1) { set_error("loss_ring and sample_chunk must be >= 1"); return MKE_E_SHAPE; }
  const int N = pl->neg_per_pos;
  if (N > 0 && (!pl->neg_h || !pl->neg_r || !pl->neg_t)) { set_error("NULL negative scratch"); return MKE_E_NULL; }
  if ((int64_t)pl->tag_base + step_end >= 0x7FFFFFFFLL) { set_error("tag overflow"); return MKE_E_RANGE; }

  mke_update_table ut[2];
  ut[0].table = pl->rel_table; ut[0].acc = pl->rel_acc; ut[0].grad = pl->rel_grad; ut[0].touched = pl->rel_touched;
  ut[0].n_rows = pl->n_rel; ut[0].normalize = pl->rel_normalize; ut[0].grad_copies = pl->rel_grad_copies; ut[0].ref_count = nullptr;
  ut[1].table = pl->ent_table; ut[1].acc = pl->ent_acc; ut[1].grad = pl->ent_grad; ut[1].touched = pl->ent_touched;
  ut[1].n_rows = pl->n_ent; ut[1].normalize = pl->ent_normalize; ut[1].grad_copies = 1; ut[1].ref_count = pl->ent_ref_count;

  int64_t chunk_lo = 0;  // first positive (epoch position) whose negatives sit at neg_*[0]
  int chunk_end = step_begin;  // steps < chunk_end are sampled
  for (int s = step_begin; s < step_end; ++s) {
    const int64_t lo = pl->step_off[s], hi = pl->step_off[s + 1];
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
    int32_t* refc = (N > 0) ? pl->ent_ref_count : nullptr;  // exclusive-row fast path
    int rc = MKE_OK;
    if (refc) {
      rc = mke_count_entity_refs(pl->pos_h + lo, pl->pos_t + lo, hi - lo, pl->neg_h + no, pl->neg_t + no, (hi - lo) * N, N, refc, stream);
      if (rc) return rc;
    }
    rc = mke_triple_score_fwd_bwd_x(pl->ent_table, pl->n_ent, pl->ent_normalize, pl->rel_table, pl->n_rel, pl->rel_normalize,
                                    pl->stride, pl->dim, pl->pos_h + lo, pl->pos_r + lo, pl->pos_t + lo, nullptr, hi - lo,
                                    N ? pl->neg_h + no : nullptr, N ? pl->neg_r + no : nullptr, N ? pl->neg_t + no : nullptr,
                                    nullptr, (hi - lo) * N, N, pl->scale, pl->ent_grad, pl->rel_grad, pl->rel_grad_copies,
                                    pl->ent_touched, pl->rel_touched, tag, refc, pl->ent_acc, pl->optimizer, pl->lr,
                                    pl->loss_partials + (int64_t)(s % pl->loss_ring) * MKE_LOSS_PARTIALS, stream);
    if (rc) return rc;
    rc = mke_rows_update_multi(ut, 2, tag, pl->stride, pl->dim, pl->optimizer, pl->lr, stream);
    if (rc) return rc;
  }
  return MKE_OK;
}
