/*
 * multike_hip.h — C-ABI of libmultike_hip.so: the MI355X (gfx950) hot path of MultiKE training.
 *
 * Boundary contract:
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Every pointer is a DEVICE pointer owned by the caller (PyTorch on the Python side) unless the
 *     parameter comment says "host". The library never allocates, frees or synchronises; every entry
 *     point only enqueues kernels on the given hipStream_t (passed as void* so that the header does not
 *     need the HIP headers; NULL = the legacy default stream).
 *   - Return value: 0 = ok, <0 = bad argument (MKE_E_*), >0 = a hipError_t from the launch.
 *     mke_last_error() returns a thread-local human-readable message for the last non-zero return.
 *   - No global mutable state inside a call: the library is re-entrant, two host threads may enqueue on two streams, and
 *     every performance knob travels WITH the plan / arguments of a call (mke_tuning, version 105: two trainers in one
 *     process may hold different settings).  mke_set_option only sets the process-wide DEFAULTS that a tuning field left
 *     at MKE_TUNE_DEFAULT (or a NULL mke_tuning) falls back to.  mke_relation_steps in overlap mode creates (and
 *     destroys) a private stream and events for the duration of the call.
 *
 * The reference (nju-websoft/MultiKE) has NO native code; each entry point below replaces a group of
 * TensorFlow-1.x graph ops that the reference builds in Python.  The "replaces" lines cite the
 * reference Python that constructs those ops (paths relative to the reference root).
 *
 * Table layout in HBM (all tables, Adagrad slots and gradient scratch share it):
 *   float32 [n_rows][stride], row-major, stride % 16 == 0 and stride >= dim; columns [dim, stride) are
 *   ZERO and stay zero (their gradients are exactly zero), so kernels may run over the padded width.
 *   One 16-lane quarter-wavefront owns one row: lane j holds columns {j, j+16, j+32, ...}.
 */
#ifndef MULTIKE_HIP_H
#define MULTIKE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKE_VERSION 105 /* 0.1.5: + entity-major second pass of the owner-computes step (mke_oc_step.em_*, mke_oc_em_plan, MKE_OC_PASS2: additions only), + mke_oc_steps / mke_oc_comm (the G > 1 step loop as ONE native call), per-plan tuning (mke_tuning); 0.1.4: mke_oc_* exchange ONE vector per positive (the side its negatives corrupt): group flags in the codes, slot -1, mke_oc_plan takes the codes, mke_oc_step.hot (hub rows of the shard), mke_attr_step_args.attr_grad_copies, + mke_probe_rows; 0.1.3: + hub rows (mke_hot_rows: mke_triple_score_fwd_bwd_xch, mke_update_table.hot, mke_relation_plan.hot; additions only); 0.1.2: + mke_oc_plan, mke_topk_long, options "attr_fused_bwd" / "oc_score_quarter" (additions only); 0.1.1: mke_align_rank gained `ties` */

/* error codes (negative = argument errors) */
#define MKE_OK 0
#define MKE_E_NULL (-1)     /* a required pointer is NULL */
#define MKE_E_SHAPE (-2)    /* bad dim / stride / count */
#define MKE_E_UNSUPPORTED (-3)
#define MKE_E_RANGE (-4)    /* id does not fit the packed key / table */

/* Number of per-block loss partials every loss-producing kernel writes (doubles).  The caller passes a
 * double[MKE_LOSS_PARTIALS] scratch; the kernel OVERWRITES all entries; the loss is their sum. */
#define MKE_LOSS_PARTIALS 2048

/* Largest supported stride (floats). */
#define MKE_MAX_STRIDE 320

/* optimizer kinds for mke_rows_update — code/MultiKE_model.py:15-25 get_optimizer */
#define MKE_OPT_ADAGRAD 0 /* tf.train.AdagradOptimizer: acc += g*g; w -= lr*g/sqrt(acc); acc0 = 0.1, no eps */
#define MKE_OPT_SGD 1     /* tf.train.GradientDescentOptimizer: w -= lr*g */
#define MKE_OPT_ADAM 2     /* tf.train.AdamOptimizer     -- dense entry points only (mke_rows_update_dense, mke_dense_update_opt) */
#define MKE_OPT_ADADELTA 3 /* tf.train.AdadeltaOptimizer -- dense entry points only */

int mke_version(void);
const char* mke_last_error(void);

/* Per-call tuning (version 105; performance only, never results): a field at MKE_TUNE_DEFAULT follows the process default
 * (mke_set_option).  Carried by pointer (NULL = all defaults) in mke_relation_plan, mke_attr_step_args and mke_oc_step, and
 * taken as an argument by the *_t entry points; the library reads it for the duration of that one call only. */
#define MKE_TUNE_DEFAULT (-2)
typedef struct mke_tuning {
  int score_splits, score_half_groups, score_offsets32, score_lane_ids, count_in_score, update_chunk, oc_score_quarter,
      attr_fused_bwd, sampler_fast;
  int reserved[7];   /* MKE_TUNE_DEFAULT */
} mke_tuning;
int mke_tuning_init(mke_tuning* t);   /* every field = MKE_TUNE_DEFAULT */

/* Process-wide DEFAULTS of the tuning knobs (performance only, never results).  Unknown name -> MKE_E_UNSUPPORTED.
 *   "score_splits"  : wavefronts sharing one positive's negatives in mke_triple_score_fwd_bwd (0 = auto)
 *   "update_chunk"  : rows per wavefront of the row-update kernels on large tables: 0 = by table size (default), 16, 64
 *   "attr_fused_bwd" : attribute step, 64 < dim <= 80: 1 (default) = the dflat product inside the convolution-backward launch and
 *                     the weight-gradient product on rider blocks of it, both forming dz = dL/dzpre on load (4 launches per step);
 *                     0 = tail backward, the two products and the convolution backward as launches of their own (6)
 *   "oc_score_quarter" : mke_oc_score with a quarter-wave per positive (four positives per wavefront) instead of a wavefront:
 *                     -1 = by shape (default: n_ranks >= 4, neg_per_pos <= 8 n_ranks, stride <= 128), 0 = never, 1 = always
 *   "oc_em_keys64"  : mke_oc_em_plan sorts its (step, row) keys as 64-bit words even when n_steps * (n_local + n_rel) < 2^32 (0, the
 *                     default: 32-bit words whenever they fit) — the instantiation a KG with >= 2^32 (step, row) pairs per epoch takes;
 *                     same lists bit for bit
 *   "sampler_fast"  : mke_neg_sample*: 1 (default) = a round's coin block evaluated in the idle last lane of the group's draw
 *                     evaluation and duplicates among first draws found through an LDS table; 0 = separate coin evaluation and
 *                     a shuffle loop.  Same Philox stream, same output bit for bit
 *   "deterministic" : 1 = the host side (tables.StepEngine) takes the deterministic path below (read by the caller; the
 *                     kernels themselves are selected by which entry point is called)
 *   "score_half_groups" : largest neg_per_pos for which mke_triple_score_fwd_bwd scores TWO groups per wavefront (one per
 *                     half: half as many wavefronts, all resident at once); -1 (default) = by row width — rows up to 128
 *                     floats: always, wider rows: up to 31 negatives; 0 = never
 *   "score_offsets32"   : 1 (default) = the training kernel addresses rows with 32-bit byte offsets when the tables are
 *                     below 4 GB; 0 = 64-bit addresses always
 *   "score_lane_ids"    : 1 (default) = the training kernel fetches a group's negative ids and reference counts once, one
 *                     negative per lane, and gathers accumulator rows only for rows it finishes in place; 0 = per round
 *   "count_in_score"    : 1 (default) = mke_relation_steps lets the NEXT step's reference counting ride in the score
 *                     launch (mke_triple_score_fwd_bwd_xc); 0 = in the update launch (mke_rows_update_multi_count)
 * Returns the previous value through *old_value when it is not NULL. */
int mke_set_option(const char* name, int value, int* old_value);


/* ------------------------------------------------------------------------------------------------
 * (1) Fused relation-view triple step: gather + normalise-on-read + translation score + logistic loss
 *     + gradient scatter-add (normalised space).
 *
 * replaces: code/MultiKE_model.py:122-130 (six embedding_lookup + relation_logistic_loss),
 *           code/losses.py:4-12 (a1), :30-34 (a2, n_neg = 0), :44-50 (a3, pos_w != NULL, n_neg = 0),
 *           code/base/initializers.py:26 (l2_normalize on read), and the backward half of
 *           optimizer.compute_gradients at code/MultiKE_model.py:28-31 for those graphs.
 *
 *   loss = scale * ( sum_p pw_p * log(1+exp(||h+r-t||^2)) + sum_n nw_n * log(1+exp(-||h'+r'-t'||^2)) )
 *   rows are read through x * rsqrt(max(sum x^2, 1e-12)) when the table's normalise flag is set.
 *
 *   Gradients w.r.t. the NORMALISED rows are atomically added into grad_ent / grad_rel (same layout as
 *   the tables, all-zero on entry by invariant — mke_rows_update re-zeroes what it consumes), and
 *   touched_*[row] = tag is stored for every row that received a contribution.
 *   grad_ent == NULL means forward only (loss only).
 *
 *   grad_rel_copies = K >= 1: grad_rel is [K][n_rel][stride]; work item i adds into copy i % K.  The few
 *   hundred relation rows each receive thousands of row-adds per step and same-address atomics serialise in
 *   the memory-side atomic units, so the hot table is privatised K ways; mke_rows_update sums the copies.
 *
 *   neg_per_pos > 0: negatives are grouped, negatives [i*neg_per_pos, (i+1)*neg_per_pos) belong to
 *   positive i (the layout code/base/batch.py:86-116 produces) and n_neg must equal n_pos*neg_per_pos;
 *   rows a negative shares with its positive are loaded once and their gradients pre-reduced in
 *   registers.  neg_per_pos == 0 with n_neg > 0: arbitrary negatives (scored independently).
 * ------------------------------------------------------------------------------------------------ */
int mke_triple_score_fwd_bwd(
    const float* ent_table, int64_t n_ent, int ent_normalize,
    const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim,
    const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w /*nullable*/,
    int64_t n_pos,
    const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w /*nullable*/,
    int64_t n_neg, int neg_per_pos,
    float scale,
    float* grad_ent /*nullable*/, float* grad_rel /*nullable iff grad_ent is*/, int grad_rel_copies,
    int32_t* touched_ent, int32_t* touched_rel, int32_t tag,
    double* loss_partials /* [MKE_LOSS_PARTIALS] */,
    void* stream);

/* (1b) Exclusive-row fast path.  Most entity rows a step touches are the corrupted entity of exactly ONE negative
 *   (uniform sampling: ~70 % of the touched rows at the DBP-WD shape).  Such a row is read by one quarter-wave and
 *   receives one gradient contribution, so that quarter-wave can apply the Jacobian + optimizer update itself — the
 *   scatter (atomic read-modify-write of the gradient row) and the later visit by mke_rows_update (3 row reads, 3 row
 *   writes) collapse into 1 accumulator read + 2 row writes.  Result: identical to the two-kernel path (same formulas,
 *   one contribution => no summation-order freedom).
 *
 *   mke_count_entity_refs: ref_count[e] += occurrences of e in (pos_h, pos_t, neg_h, neg_t), not counting a negative's
 *     entry that repeats its own positive's entity on that side (grouped negatives, neg_per_pos >= 1).  ref_count is
 *     int32 [n_ent], zero on entry by invariant.
 *   mke_triple_score_fwd_bwd_x: as (1), plus: a negative that differs from its positive in exactly one entity e with
 *     ref_count[e] == 1 is applied in place on (ent_table, ent_acc) with (optimizer, lr) and ref_count[e] is reset to 0;
 *     every other row goes through grad_ent / touched_ent as in (1).  mke_rows_update* reset ref_count for the rows they
 *     visit when given the array (mke_update_table.ref_count / the ref_count argument), restoring the invariant. */
/* Deterministic mode (parity / debugging; SURVEY.md §7 "hard parts"): the same step with every gradient-row contribution
 * STORED into a slot of its own instead of added atomically — slot ((g * (neg_per_pos + 1) + n) * 3 + c) for contribution c
 * (0 head, 1 relation, 2 tail row) of triple n of group g (n = neg_per_pos: the group's pre-reduced flush), key
 * (is_relation << 40) | row — then summed per row in slot order:
 *     fill stage_keys with 0x7F bytes -> mke_triple_score_fwd_bwd_det -> stable sort of the keys (-> sorted_keys, order)
 *     -> mke_stage_reduce -> mke_rows_update_multi as usual.
 * stage_rows: [stage_slots][stride] floats, stage_slots >= 3 * n_pos * (neg_per_pos + 1) (ungrouped: 3 * (n_pos + n_neg)).
 * Results are bit-identical from run to run; rows referenced once are still updated in place (one contribution: no order). */
int mke_triple_score_fwd_bwd_det(
    float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w,
    int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w, int64_t n_neg,
    int neg_per_pos, float scale, float* grad_ent, float* grad_rel, int32_t* touched_ent, int32_t* touched_rel, int32_t tag,
    int32_t* ref_count /*nullable*/, float* ent_acc, int optimizer, float lr, float* stage_rows, int64_t* stage_keys,
    int64_t stage_slots, double* loss_partials, void* stream);
int mke_stage_reduce(const float* stage_rows, const int64_t* sorted_keys, const int64_t* order, int64_t n_slots, int stride,
                     float* grad_ent, float* grad_rel, int32_t* touched_ent, int32_t* touched_rel, int32_t tag, void* stream);
int mke_count_entity_refs(const int32_t* pos_h, const int32_t* pos_t, int64_t n_pos, const int32_t* neg_h,
                          const int32_t* neg_t, int64_t n_neg, int neg_per_pos, int32_t* ref_count, void* stream);
int mke_triple_score_fwd_bwd_x(
    float* ent_table, int64_t n_ent, int ent_normalize,
    const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim,
    const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w /*nullable*/, int64_t n_pos,
    const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w /*nullable*/, int64_t n_neg,
    int neg_per_pos, float scale,
    float* grad_ent, float* grad_rel, int grad_rel_copies,
    int32_t* touched_ent, int32_t* touched_rel, int32_t tag,
    int32_t* ref_count, float* ent_acc /*nullable for SGD*/, int optimizer, float lr,
    double* loss_partials, void* stream);

/* The same, and the first blocks of the launch also count the entity references of the NEXT step into next_count->ref_count
 * (mke_count_entity_refs semantics; a different buffer than `ref_count`; NULL = plain mke_triple_score_fwd_bwd_x): the counting
 * finishes under the scoring blocks instead of paying a launch or a tail of its own.  mke_count_job: section (2). */
struct mke_count_job;
/* Hub rows (version 103).  A KG's degree distribution is heavy-tailed (code/base/batch.py:45-54 feeds real triples): a few
 * hundred entities are head or tail of several positives of EVERY step, and the flushes of all those groups' shared-row
 * gradients serialise on the same few cache lines of the gradient scratch.  For the rows listed here the flush of group g
 * goes to one of `copies` private copies instead — extra rows behind the table's own in the SAME scratch:
 *     copy k of hot row i = scratch row  row0 + k * n_hot + i      (k = g % copies; the scratch has row0 + copies * n_hot rows)
 * touched[entity] is set as usual; mke_rows_update_multi adds the copies to the row's own gradient (and re-zeroes them) when
 * it visits the row (mke_update_table.hot).  Contributions of corrupted triples keep going to the row itself. */
typedef struct mke_hot_rows {
  const int32_t* slot;   /* device int32 [n_ent]: index of the row among the n_hot hub rows, or -1 */
  int32_t n_hot, copies;
  int64_t row0;          /* first copy row (>= n_ent) */
} mke_hot_rows;
int mke_triple_score_fwd_bwd_xc(
    float* ent_table, int64_t n_ent, int ent_normalize,
    const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim,
    const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w /*nullable*/, int64_t n_pos,
    const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w /*nullable*/, int64_t n_neg,
    int neg_per_pos, float scale,
    float* grad_ent, float* grad_rel, int grad_rel_copies,
    int32_t* touched_ent, int32_t* touched_rel, int32_t tag,
    int32_t* ref_count, float* ent_acc /*nullable for SGD*/, int optimizer, float lr,
    const struct mke_count_job* next_count /*nullable*/, double* loss_partials, void* stream);
/* The same with hub rows (hot == NULL or hot->n_hot == 0: identical to mke_triple_score_fwd_bwd_xc). */
int mke_triple_score_fwd_bwd_xch(
    float* ent_table, int64_t n_ent, int ent_normalize,
    const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim,
    const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w /*nullable*/, int64_t n_pos,
    const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w /*nullable*/, int64_t n_neg,
    int neg_per_pos, float scale,
    float* grad_ent, float* grad_rel, int grad_rel_copies,
    int32_t* touched_ent, int32_t* touched_rel, int32_t tag,
    int32_t* ref_count, float* ent_acc /*nullable for SGD*/, int optimizer, float lr,
    const struct mke_count_job* next_count /*nullable*/, const mke_hot_rows* hot /*nullable*/, double* loss_partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (2) Per-row optimizer step on the rows touched in this step: Jacobian of normalise-on-read, then
 *     the optimizer update; consumes (re-zeroes) the gradient rows.
 *
 * replaces: the gradient of tf.nn.l2_normalize at code/base/initializers.py:26 and
 *           optimizer.apply_gradients at code/MultiKE_model.py:31 (tf.train.AdagradOptimizer,
 *           code/MultiKE_model.py:17).  TF applies it densely to the whole [n,dim] variable; rows
 *           with a zero gradient are left bit-identical by TF too, so touching only rows with
 *           touched[row] == tag gives the same result.
 *
 *   touched == NULL: every row of the table is visited (replicated tables after an all-reduce).
 *   for every row with touched[row] == tag:
 *     ghat = grad[row];  grad[row] = 0
 *     normalize: s = sum w^2; inv = rsqrt(max(s,1e-12)); what = w*inv;
 *                g = (s > 1e-12) ? (ghat - what*(what.ghat))*inv : ghat*inv     else g = ghat
 *     ADAGRAD: acc += g*g; w -= lr*g/sqrt(acc)        SGD: w -= lr*g   (acc may be NULL)
 * ------------------------------------------------------------------------------------------------ */
int mke_rows_update(
    float* table, float* acc /*nullable for SGD*/, float* grad, int grad_copies /* grad is [copies][n_rows][stride] */,
    const int32_t* touched, int32_t tag,
    int64_t n_rows, int stride, int dim,
    int normalize, int optimizer, float lr,
    void* stream);

/* Same update over up to MKE_MAX_UPDATE_TABLES tables of one stride/dim in ONE launch (e.g. the entity and the
 * relation table of the relation-view graph).  tables is a HOST array. */
#define MKE_MAX_UPDATE_TABLES 4
typedef struct mke_update_table {
  float* table;
  float* acc;   /* nullable for SGD */
  float* grad;
  const int32_t* touched;
  int64_t n_rows;
  int normalize;
  int grad_copies; /* grad is [grad_copies][n_rows][stride] */
  int32_t* ref_count; /* nullable: reset to 0 for every visited row (exclusive-row fast path bookkeeping) */
  /* Owner side of the sharded step (section 7): when slot_of != NULL the row's gradient is not taken from grad/touched
   * (both may be NULL) but summed, in rank order, from the rows the ranks sent back: for every g with
   * s = slot_of[row * n_ranks + g] >= 0, ghat += src_rows[g * capacity + s][:].  Rows without a slot are not visited;
   * the slots a visit consumed are reset to -1.  One pass replaces scatter-add + touched-row update. */
  const float* src_rows;
  int32_t* slot_of;
  int n_ranks;
  int64_t capacity;
  /* version 103: hub rows of this table (see mke_hot_rows; slot == NULL: none) */
  mke_hot_rows hot;
} mke_update_table;
int mke_rows_update_multi(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim,
                          int optimizer, float lr, void* stream);

/* The same launch can also carry the reference counting of the NEXT step (mke_count_entity_refs semantics into
 * next_ref_count, a different buffer than the ones the tables reset): the counting blocks ride along with the update
 * blocks instead of paying their own kernel boundary.  count == NULL: plain mke_rows_update_multi. */
typedef struct mke_count_job {
  const int32_t* pos_h; const int32_t* pos_t; int64_t n_pos;
  const int32_t* neg_h; const int32_t* neg_t; int64_t n_neg;
  int neg_per_pos;
  int32_t* ref_count;
} mke_count_job;
int mke_rows_update_multi_count(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim,
                                int optimizer, float lr, const mke_count_job* count /*nullable*/, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (3) Uniform / truncated negative sampler (counter-based Philox4x32-10; specification:
 *     oracle/sampler_oracle.py philox_negatives).
 *
 * replaces: code/base/batch.py:86-116 generate_neg_triples_fast as called from
 *           code/base/batch.py:40-41 (one call per KG per step).
 *
 *   One KG's sampling context (the arguments batch.py:40-41 passes per KG):
 * ------------------------------------------------------------------------------------------------ */
typedef struct mke_kg_side {
  const int32_t* ent_list;     /* entities_list of the KG, or NULL = contiguous range [ent_lo, ent_lo+n_ent) */
  int32_t ent_lo;
  int32_t n_ent;               /* size of the candidate population when no neighbour list applies */
  const int32_t* cand_table;   /* nullable [n_ent_total][cand_k]: truncated-sampling neighbours, indexed by entity id */
  const uint8_t* cand_valid;   /* nullable [n_ent_total]: entity has a neighbour list (dict membership, batch.py:94-95) */
  int32_t cand_k;
  const uint64_t* known_keys;  /* nullable: hash set built by mke_tripleset_build */
  uint64_t known_capacity;     /* power of two */
} mke_kg_side;

/*   For positive i (of n_pos), up to max_try rounds: one fair coin per round picks the corrupted side;
 *   `need` distinct candidates are drawn without replacement from the candidate list of the positive's
 *   head (or tail): cand_table row of that entity if it has one, else the KG's entity list.  In rounds
 *   0..max_try-2 candidates forming a known triple are dropped; the last round keeps everything.
 *   Exactly neg_per_pos (<= 64) negatives per positive are written at neg_*[i*neg_per_pos ...]; neg_r is a
 *   copy of the positive's relation.
 *
 *   sides: HOST pointer to 2 contexts; pos_kg (device, nullable = all KG 0) selects the context per
 *   positive, so one launch can cover many steps of [KG1 part | KG2 part] batches.
 *   RNG stream of positive i = Philox key (seed_lo, seed_hi), counter (i + pos_offset, round | blk<<8,
 *   slot, stream_id + kg).  The output does not depend on how a range of positives is split into calls. */
int mke_neg_sample(
    const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, int64_t n_pos, int64_t pos_offset,
    const uint8_t* pos_kg /*nullable*/, const mke_kg_side* sides /* host, [2] */,
    int neg_per_pos, int max_try,
    uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id,
    int32_t* neg_h, int32_t* neg_r, int32_t* neg_t,
    void* stream);
/* Same, for positives that are NOT consecutive in the epoch order (a rank's share of every step of a sharded epoch,
 * gathered into one array): pos_index[i] = epoch position of positive i (takes the place of i + pos_offset in the RNG
 * counter), so a whole epoch of a rank's negatives is one launch and equals what per-step calls would have drawn. */
int mke_neg_sample_at(
    const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, int64_t n_pos, const int32_t* pos_index,
    const uint8_t* pos_kg /*nullable*/, const mke_kg_side* sides /* host, [2] */,
    int neg_per_pos, int max_try,
    uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id,
    int32_t* neg_h, int32_t* neg_r, int32_t* neg_t,
    void* stream);

/* Known-triple hash set.  key = h<<38 | t<<12 | r  (h,t < 2^26, r < 2^12); empty slot = ~0.
 * keys must be filled with 0xFF bytes by the caller before the first build call; capacity is a power
 * of two >= 2 * (number of triples).  Several build calls may add to the same set.
 * replaces: the Python set `all_triples_set` membership test at code/base/batch.py:109. */
int mke_tripleset_build(
    const int32_t* h, const int32_t* r, const int32_t* t, int64_t n,
    uint64_t* keys, uint64_t capacity, void* stream);

/* Membership query (used by tests and by the host-side mirror of the filter): out[i] = 1 if present. */
int mke_tripleset_query(
    const int32_t* h, const int32_t* r, const int32_t* t, int64_t n,
    const uint64_t* keys, uint64_t capacity, uint8_t* out, void* stream);

/* random.sample(list, batch) for every step of an epoch in one launch -- the batching of the cross-KG inference and
 * common-space loops (code/MultiKE_model.py:358,380,402,425,446: `batch` distinct list positions per step, steps
 * independent).  out[s * batch + i] = pi_s(i), i < batch, s < n_steps, where pi_s is a keyed pseudo-random permutation of
 * [0, n): 6-round Feistel network on the smallest domain 4^k >= n, cycle-walked into [0, n); round keys from
 * Philox4x32-10(step, {0,1}, 0x5A4D504C, stream_id; seed).  Deterministic in (seed, stream_id, step, i);
 * oracle/sampler_oracle.py:distinct_sample restates it. */
int mke_sample_distinct(int64_t n, int batch, int n_steps, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id,
                        int32_t* out /* [n_steps][batch] */, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (4) Loss ops over ALREADY GATHERED rows — the losses.py surface itself (forward + gradient w.r.t.
 *     every gathered row in one pass; rows are dense [n][ld] with ld >= dim, no padding rule).
 *
 * replaces: code/losses.py:4-12 relation_logistic_loss, :15-27 attribute_logistic_loss,
 *           :30-34 / :37-41 *_wo_negs, :44-50 logistic_loss_wo_negs (weights != NULL).
 *   sign = +1: sum w * log(1+exp(+||h+r-t||^2))   (positives)
 *   sign = -1: sum w * log(1+exp(-||h+r-t||^2))   (negatives)
 *   gh/gr/gt (nullable, all or none): gradient rows, OVERWRITTEN.  gt = -gh.
 * ------------------------------------------------------------------------------------------------ */
int mke_gathered_logistic_fwd_bwd(
    const float* hs, const float* rs, const float* ts, const float* ws /*nullable*/,
    int64_t n, int dim, int ld, int sign,
    float* gh, float* gr, float* gt,
    double* loss_partials /* [MKE_LOSS_PARTIALS] */,
    void* stream);

/* replaces: code/losses.py:66-69 alignment_loss — sum ||a-b||^2 ; ga = 2(a-b), gb = -ga (nullable). */
int mke_gathered_alignment_fwd_bwd(
    const float* a, const float* b, int64_t n, int dim, int ld,
    float* ga, float* gb,
    double* loss_partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (5) Fused alignment step over table rows (common-space learning, one term):
 *       loss = weight * sum_i || A^[ia_i] - B^[ib_i] ||^2
 *     with normalise-on-read per table flag; grads (normalised space) atomically added to grad_a /
 *     grad_b (either may be NULL = constant table), touched flags stored.
 * replaces: code/MultiKE_model.py:229-236 (_define_common_space_learning_graph lookups + the three
 *           alignment_loss terms, one call per term).
 * ------------------------------------------------------------------------------------------------ */
int mke_align_fwd_bwd(
    const float* table_a, int a_normalize, const float* table_b, int b_normalize,
    int stride, int dim,
    const int32_t* ia, const int32_t* ib, int64_t n,
    float weight,
    float* grad_a /*nullable*/, int32_t* touched_a, float* grad_b /*nullable*/, int32_t* touched_b,
    int32_t tag,
    double* loss_partials, void* stream);

/* A whole epoch of common-space steps as one native call (the step loop of code/MultiKE_model.py:458-473 around the graph
 * :225-239): per step every term runs mke_align_fwd_bwd on positions [step_off[s], step_off[s+1]) of ia / ib, then ONE
 * update launch covers every trainable table.  Tables share stride/dim; a table with grad == NULL is constant (the name
 * view).  loss_partials: [n_steps][n_terms][MKE_LOSS_PARTIALS]; tag of step s = tag_base + s. */
#define MKE_ALIGN_MAX_TABLES 4
#define MKE_ALIGN_MAX_TERMS 4
typedef struct mke_align_table {
  float* table; float* acc /*nullable for SGD / constant*/; float* grad /*NULL = constant*/; int32_t* touched;
  int64_t n_rows; int normalize;
} mke_align_table;
typedef struct mke_align_term { int a, b; float weight; } mke_align_term;   /* indices into tables[] */
typedef struct mke_align_plan {
  mke_align_table tables[MKE_ALIGN_MAX_TABLES]; int n_tables;
  mke_align_term terms[MKE_ALIGN_MAX_TERMS]; int n_terms;
  int stride, dim;
  const int32_t* ia; const int32_t* ib;          /* device, epoch order */
  const int64_t* step_off; int n_steps;          /* HOST, n_steps + 1 offsets */
  int optimizer; float lr; int32_t tag_base;
  double* loss_partials;
} mke_align_plan;
int mke_align_steps(const mke_align_plan* plan, void* stream);

/* Gather normalised rows into a dense [n][dim] matrix (the `.eval()` / embedding_lookup read path).
 * replaces: code/MultiKE_model.py:263-277 eval_kg*_ent_embeddings. */
int mke_gather_rows(
    const float* table, int normalize, int stride, int dim,
    const int32_t* idx /*nullable = identity*/, int64_t n,
    float* out /* [n][dim] */, void* stream);

/* Placement probe (no reference counterpart; read-only): rows idx[0..n) of a and (nullable) b, c — arrays of the same
 * [rows][stride] shape — read TOGETHER, the way the relation step reads a row of the table, its accumulator and its gradient;
 * out[i] = the sum of what was read of row idx[i].  The host side times it on candidate allocations of a >= 1 GB table's
 * companion arrays: on MI355X the step's time depends on how the three arrays' physical pages sit relative to each other
 * (+-12 % at the 2M x 256 shape), which no virtual-address choice controls. */
int mke_probe_rows(const float* a, const float* b /*nullable*/, const float* c /*nullable*/, int stride, const int32_t* idx,
                   int64_t n, float* out /* [n] */, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (6) Native step runner of the relation view: enqueues steps [step_begin, step_end) of an epoch —
 *     negative sampling (batched `sample_chunk` steps per launch), the fused triple step and the row
 *     update of both tables — without returning to the host language between steps.
 *
 * replaces: the loop body of code/MultiKE_model.py:302-312 (batch_queue.get + session.run) together with
 *           the producer side code/base/batch.py:22-42; the epoch shuffle (code/MultiKE_model.py:314-315)
 *           stays with the caller, who rewrites pos_* in place between epochs.
 *
 *   Positives are epoch-ordered and step-contiguous: step s owns [step_off[s], step_off[s+1]) and the first
 *   pos_kg==0 part of it is KG1's slice, the rest KG2's (code/base/batch.py:36-42).
 *   Loss of step s: loss_partials + (s % loss_ring) * MKE_LOSS_PARTIALS.
 *   tag of step s = tag_base + s (must stay < 2^31 and never repeat for these touched arrays).
 * ------------------------------------------------------------------------------------------------ */
typedef struct mke_relation_plan {
  float* ent_table; int64_t n_ent; int ent_normalize;
  float* rel_table; int64_t n_rel; int rel_normalize;
  float* ent_acc; float* rel_acc;            /* this optimizer's Adagrad slots (NULL for SGD) */
  float* ent_grad; float* rel_grad;          /* zero-invariant gradient scratch; rel_grad is [rel_grad_copies][n_rel][stride] */
  int rel_grad_copies;
  int32_t* ent_touched; int32_t* rel_touched;
  int32_t* ent_ref_count;                    /* nullable: enables the exclusive-row fast path (zero-invariant scratch),
                                                int32 [2][n_ent]: steps alternate between the two halves so that the
                                                counting of step s+1 can ride in the update launch of step s */
  int overlap;                               /* != 0: the table-independent work of the NEXT step (reference counting) and of
                                                the NEXT sample chunk (negative sampling) is enqueued on a second stream and
                                                overlaps with scoring / updating the current step.  Needs neg_* to hold TWO
                                                chunk buffers of neg_chunk_capacity elements each and ent_ref_count [2][n_ent].
                                                The call creates and destroys its own stream and events. */
  int64_t neg_chunk_capacity;                /* elements per negative chunk buffer (overlap mode) */
  int stride, dim;
  const int32_t* pos_h; const int32_t* pos_r; const int32_t* pos_t;  /* device, epoch order */
  const uint8_t* pos_kg;                     /* device, [n positives] 0/1 */
  const int64_t* step_off;                   /* HOST [n_steps+1] */
  int n_steps;
  mke_kg_side sides[2];
  int neg_per_pos, max_try;
  int sample_chunk;                          /* steps sampled per sampler launch (>=1) */
  int negatives_ready;                       /* != 0: neg_* already hold the negatives of steps [step_begin, step_end) laid out
                                                from step_begin (sampled earlier, e.g. on another stream during the previous
                                                epoch): no sampler launch in this call */
  int32_t* neg_h; int32_t* neg_r; int32_t* neg_t;  /* device scratch, >= max positives of any sample_chunk consecutive steps * neg_per_pos */
  uint32_t seed_lo, seed_hi, stream_id;
  int optimizer; float lr; float scale;
  double* loss_partials; int loss_ring;      /* device [loss_ring][MKE_LOSS_PARTIALS] */
  int32_t tag_base;
  const float* pos_w;                        /* nullable: per-positive weights, epoch order like pos_* (weighted positives-only
                                                loops, code/MultiKE_model.py:393-414); neg_per_pos == 0 runs positives only */
  mke_hot_rows hot;                          /* version 103: hub rows of the entity table (slot == NULL: none); ent_grad then has
                                                hot.row0 + hot.copies * hot.n_hot rows */
  const mke_tuning* tuning;                  /* version 105: host pointer, NULL = the process defaults */
} mke_relation_plan;

int mke_relation_steps(const mke_relation_plan* plan, int step_begin, int step_end, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (6c) Truncated-sampling k-NN refresh without the similarity matrix.
 *
 * replaces: `generate_neighbours` / `find_neighbours` at code/base/batch.py:119-150 (sim = E . E^T over one KG's useful
 *           entities, per row the k largest columns via argpartition, self included, unordered).
 *
 *   mke_sim_select: for rows [row_lo, row_hi) of emb ([n_cols][ld] float32, row-normalised by the caller, columns
 *     dim..kpad zero, kpad a multiple of 16) against ALL n_cols rows: every column j with emb[i] . emb[j] > tau[i - row_lo]
 *     is appended to row i's candidate list, similarity alongside.  The columns are split into n_seg contiguous ranges;
 *     range s of row i owns cand[i - row_lo][s][0 .. seg_cap) and seg_count[i - row_lo][s] (the number
 *     of hits, which may exceed seg_cap: only the first seg_cap are stored).  Inside a segment the candidates are in
 *     column order.  n_seg <= 16, n_seg * seg_cap <= 4096.  The similarity is an f32 MFMA fma chain over k.
 *   mke_sim_sample: out[i - row_lo][c] = emb[i] . samp[c] for rows [row_lo, row_hi) and the n_samp rows of samp (same padding
 *     as emb): the similarities the thresholds are estimated from, made of the same fma chains as mke_sim_select's.
 *   mke_topk_rows: per row of a short list (n_seg segments of seg_cap slots, seg_count valid entries each; NULL
 *     seg_count = all slots valid): the exact k largest values.  out_idx[row][0..k) = their idx entries (NULL idx = the
 *     slot number), mapped through id_map when given, in list order; ties at the k-th value are broken by list order.
 *     out_kth[row] = the k-th largest value.  status[row] = 0 ok, 1 = fewer than k entries, 2 = a segment overflowed
 *     (seg_count > seg_cap); for status != 0 out_idx[row] is not written and out_kth[row] = -3e38.
 *   mke_topk_candidates: the same over mke_sim_select's (column, similarity) pairs.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mke_candidate {
  int32_t idx;
  float sim;
} mke_candidate;
int mke_sim_select(const float* emb, int ld, int kpad, int64_t n_cols, int64_t row_lo, int64_t row_hi, const float* tau,
                   int n_seg, int seg_cap, mke_candidate* cand, int32_t* seg_count, void* stream);
int mke_sim_sample(const float* emb, int ld, int kpad, int64_t n_rows, int64_t row_lo, int64_t row_hi, const float* samp,
                   int ld_samp, int n_samp, float* out /* [row_hi - row_lo][n_samp] */, void* stream);
int mke_topk_candidates(const mke_candidate* cand, const int32_t* seg_count, int64_t rows, int n_seg, int seg_cap, int k,
                        const int32_t* id_map /*nullable*/, int32_t* out_idx /*nullable*/, float* out_kth /*nullable*/,
                        int32_t* status /*nullable*/, void* stream);
/* exact top k of LONG rows (vals [rows][ld], n <= ld values each, any n): the k columns with the largest values, the first
 * ties of the k-th value in column order, written in column order as out_idx[row][0..k) (through id_map when given).
 * replaces: np.argpartition over a whole similarity row at code/base/batch.py:143-150 (short KGs, and the rows of the
 * thresholded pass that have to be redone at full width) */
int mke_topk_long(const float* vals, int64_t rows, int64_t n, int64_t ld, int k, const int32_t* id_map /*nullable*/,
                  int32_t* out_idx /* [rows][k] */, void* stream);
int mke_topk_rows(const float* vals, const int32_t* idx /*nullable*/, const int32_t* seg_count /*nullable*/, int64_t rows,
                  int n_seg, int seg_cap, int k, const int32_t* id_map /*nullable*/, int32_t* out_idx /*nullable*/,
                  float* out_kth /*nullable*/, int32_t* status /*nullable*/, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (7) Bookkeeping of the entity-row sharded (multi-GPU) step.  New design — the reference has no multi-device
 *     code.  Entity rows are owned by rank id % n_ranks (local row id / n_ranks).  Everything
 *     is fixed-capacity so that the exchanges are equal-split all-to-alls with no host synchronisation.
 *
 *   mke_rowset_build: distinct ids of up to four index streams -> req[owner][slot] = id / n_ranks (req is
 *     [n_ranks][capacity], pre-filled with -1 by the caller), id_map[id] = owner*capacity + slot, counts[owner] =
 *     rows requested of that owner.  flags ([n_ent], zero) and counts ([n_ranks], zero) are scratch; flags are left
 *     dirty and are cleared by mke_rowset_remap.  *overflow is set to 1 when a segment is full (result invalid).
 *     Slot order inside a segment is unspecified.
 *   mke_rowset_remap: for up to four streams, out_s[i] = id_map[ids_s[i]] and flags[ids_s[i]] = 0; optionally
 *     re-initialises a req / counts pair for the next build (req[:] = -1, counts[:] = 0); and, when want != NULL
 *     (the [n_ranks][capacity] local rows the other ranks asked this owner for, -1 = pad), inverts it for the owner's
 *     reduce-and-update launch: slot_of[row * n_ranks + g] = slot of `row` in rank g's segment.  slot_of is
 *     [n_local_rows * n_ranks], all -1 on entry by invariant (mke_rows_update_multi resets what it consumes).
 *   mke_rows_gather_padded: out[i][:] = idx[i] >= 0 ? table[idx[i]][:] : 0   (raw padded rows, no normalisation);
 *     when zero_rows != NULL, zero_rows[i][:] = 0 as well (clears the compact gradient scratch in the same pass).
 *   mke_rows_scatter_add: grad[idx[i]][:] += rows[i][:] (atomic), touched[idx[i]] = tag, for idx[i] >= 0; when
 *     reset_req / reset_counts are given they are re-initialised for the next step (req[i] = -1, counts[:] = 0).
 * ------------------------------------------------------------------------------------------------ */
int mke_rowset_build(const int32_t* ids0, int64_t n0, const int32_t* ids1, int64_t n1, const int32_t* ids2, int64_t n2,
                     const int32_t* ids3, int64_t n3, int32_t* flags, int32_t* counts, int32_t* req, int32_t* id_map,
                     int32_t* overflow, int n_ranks, int capacity, void* stream);
int mke_rowset_remap(const int32_t* ids0, int32_t* out0, int64_t n0, const int32_t* ids1, int32_t* out1, int64_t n1,
                     const int32_t* ids2, int32_t* out2, int64_t n2, const int32_t* ids3, int32_t* out3, int64_t n3,
                     const int32_t* id_map, int32_t* flags, int32_t* reset_req /*nullable*/, int64_t reset_req_len,
                     int32_t* reset_counts /*nullable*/, int n_counts,
                     const int32_t* want /*nullable*/, int32_t* slot_of, int n_ranks, int capacity, void* stream);
int mke_rows_gather_padded(const float* table, int stride, const int32_t* idx, int64_t n, float* out,
                           float* zero_rows /*nullable*/, void* stream);
int mke_rows_scatter_add(const int32_t* idx, const float* rows, int64_t n, int stride, int dim, float* grad,
                         int32_t* touched, int32_t tag, int32_t* reset_req /*nullable*/, int32_t* reset_counts /*nullable*/,
                         int n_counts, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (8) Attribute-view CNN scorer.
 *
 * replaces: `conv()` at code/MultiKE_model.py:34-63 and the loss lines of the three graphs that call it
 *           (:145-148 attribute view, :182-183 ckge_attr x2, :214-218 ckga_attr), forward and backward.
 *
 *   Parameters of one CNN are ONE packed float buffer (so that one update launch covers them):
 *     gamma[dim] | beta[dim] | K1[2][4][1][2] | b1[2] | K2[2][4][2][2] | b2[2] | W[4*dim][dim] | bias[dim]
 *     (conv kernels in TF HWIO order, W row index = h*2*dim + w*2 + c).  MKE_CNN_CONV_PARAMS(dim) floats belong to
 *     the conv stack; W / bias follow.
 *   Pipeline of one step (all enqueue-only):
 *     mke_attr_conv_fwd   : gather attribute (table flag) + literal rows, BN affine, 2 x conv+tanh, width l2-norm -> flat [n][4*dim]
 *     (library GEMM)      : zpre = flat @ W        (mke_attr_step: [flat, 1] @ [W; bias] -- W and bias are adjacent in the pack)
 *     mke_attr_tail_z     : z = tanh(zpre + bias) in place; per-block partial sums of z^2
 *     mke_attr_tail_loss  : out = z / ||z||_F (whole batch); loss = scale * sum w log(1+exp(||h-out||^2)); scatter of the
 *                           entity-row gradient; gout = dL/dout; per-block partials of sum gout.z
 *     mke_attr_tail_bwd   : gout <- dL/dzpre (through the batch-global normalisation and tanh), in place; dbias += column sums
 *     (library GEMMs)     : dW = flat^T @ dzpre ; dflat = dzpre @ W^T
 *     mke_attr_conv_bwd   : recomputes the conv stack, back-propagates it, scatters the attribute-row gradient and
 *                           accumulates the conv / BN parameter gradients into grad_params (atomic)
 *     mke_dense_update    : Adagrad / SGD over the packed parameter buffer (consumes = zeroes the gradient)
 * ------------------------------------------------------------------------------------------------ */
#define MKE_CNN_CONV_PARAMS(dim) (2 * (dim) + 52)
#define MKE_CNN_PARAMS(dim) (MKE_CNN_CONV_PARAMS(dim) + 4 * (dim) * (dim) + (dim))
#define MKE_CNN_WORKSPACE_FLOATS(dim) (32 * (2 * (dim) + 64))
int mke_attr_conv_fwd(const float* attr_table, int attr_stride, int attr_normalize, const float* lit_table, int lit_stride,
                      int dim, const int32_t* ia, const int32_t* iv, int64_t n, const float* params, float* flat,
                      int flat_stride /* even, >= 4*dim; when > 4*dim, flat[t][4*dim] = 1 (bias column: [flat,1] @ [W;bias]) */,
                      void* stream);
int mke_attr_conv_bwd(const float* attr_table, int attr_stride, int attr_normalize, const float* lit_table, int lit_stride,
                      int dim, const int32_t* ia, const int32_t* iv, int64_t n, const float* params, const float* dflat,
                      float* grad_params, float* grad_attr /*nullable*/, int32_t* touched_attr, int32_t tag,
                      float* workspace /* nullable: MKE_CNN_WORKSPACE_FLOATS(dim) floats, all-zero before the first call;
                                          left all-zero by every call.  Without it every block accumulates straight into
                                          grad_params: same result, ~3x slower at 5000 triples */,
                      void* stream);
int mke_attr_tail_z(float* z /* in: zpre, out: z */, const float* bias /* nullable: already in zpre */, int64_t n, int dim,
                    double* sumsq_partials /* [MKE_LOSS_PARTIALS] */, void* stream);
int mke_attr_tail_loss(const float* z, const double* sumsq_partials, const float* ent_table, int ent_stride,
                       int ent_normalize, const int32_t* ih, const float* weights /*nullable*/, float scale, int64_t n,
                       int dim, float* gout /* [n][dim] */, double* dot_partials, float* grad_ent /*nullable*/,
                       int32_t* touched_ent, int32_t tag, double* loss_partials, void* stream);
int mke_attr_tail_bwd(const float* z, float* gout, const double* sumsq_partials, const double* dot_partials, int64_t n,
                      int dim, float* grad_bias /* nullable: += column sums of dL/dzpre (the dense layer's bias gradient) */,
                      void* stream);
/* The whole pipeline above as ONE native call (the dense layer's three products run on mke_gemm_f32):
 * conv_fwd -> GEMM -> tail_z -> tail_loss -> tail_bwd -> GEMM ([dW; dbias], split-K) -> GEMM (dflat) -> conv_bwd
 * -> [update != 0] row updates of the entity / attribute tables and the dense update of the packed parameters.
 * ent_grad / attr_grad NULL = that table is constant.  param_grads must be all-zero on entry (the dense update restores
 * it; with update == 0 the caller inspects and clears it).  scratch: mke_attr_scratch_floats(n, dim) floats.
 * partials: double[3 * MKE_LOSS_PARTIALS] (loss | sum z^2 | sum g.z); the loss is the sum of the first block. */
typedef struct mke_attr_step_args {
  float* ent_table; int64_t n_ent; int ent_stride; int ent_normalize; float* ent_acc; float* ent_grad; int32_t* ent_touched;
  float* attr_table; int64_t n_attr; int attr_stride; int attr_normalize; float* attr_acc; float* attr_grad; int32_t* attr_touched;
  const float* lit_table; int lit_stride;
  int dim;
  const int32_t* ih; const int32_t* ia; const int32_t* iv; const float* weights /*nullable*/; int64_t n;
  float scale;
  float* params; float* param_grads; float* param_acc /*nullable for SGD*/;
  float* scratch; double* partials;
  int optimizer; float lr; int32_t tag; int update;
  float* workspace;   /* nullable: MKE_CNN_WORKSPACE_FLOATS(dim) floats, zero before first use, left zero (see mke_attr_conv_bwd) */
  int attr_grad_copies;   /* version 104: 0 / 1 = attr_grad is [n_attr][attr_stride]; c > 1 = [c][n_attr][attr_stride], all zero between steps:
                             triple t adds to copy t % c and the update sums them (a few hundred attribute rows take a step's 5,000
                             triples, and real attribute frequencies are heavy-tailed: same-address atomics serialise) */
  const mke_tuning* tuning;   /* version 105: host pointer, NULL = the process defaults */
} mke_attr_step_args;
int64_t mke_attr_scratch_floats(int64_t n, int dim);
int mke_attr_step(const mke_attr_step_args* args, void* stream);
/* The same step cut into phases for data-parallel training over ranks (each rank trains the triples whose head entity it
 * owns): the batch-wide l2_normalize (code/MultiKE_model.py:60) couples the ranks through two scalars and the replicated
 * parameters through their gradients, so between the phases the caller all-reduces
 *   partials[MKE_LOSS_PARTIALS ..)    (sum z^2)  after MKE_ATTR_FWD,   partials[2 MKE_LOSS_PARTIALS ..) (sum g.z) after MKE_ATTR_TAIL
 *   (replace each array by [global sum, 0, 0, ...]; the consuming kernel adds the entries up),
 *   param_grads and attr_grad after MKE_ATTR_BWD (then MKE_ATTR_UPD with attr_touched = NULL: every attribute row).
 * n == 0 is legal (a rank owning none of the step's triples contributes zero sums). */
#define MKE_ATTR_FWD 1
#define MKE_ATTR_TAIL 2
#define MKE_ATTR_BWD 4
#define MKE_ATTR_UPD 8
#define MKE_ATTR_ALL 15
int mke_attr_step_phases(const mke_attr_step_args* args, int phases, void* stream);
/* n_steps consecutive mke_attr_step calls without a host round trip between them (the per-step loop of
 * code/MultiKE_model.py:319-345,371-391,416-437 as one native call): step s uses positions [step_off[s], step_off[s+1])
 * of args->ih / ia / iv / weights (epoch order), tag args->tag + s, and writes its loss partials to
 * loss_ring[s % ring][MKE_LOSS_PARTIALS]; args->n is ignored, args->scratch must hold the largest step,
 * args->partials is used as scratch.  step_off is a HOST array of n_steps + 1 offsets. */
int mke_attr_steps(const mke_attr_step_args* args, const int64_t* step_off, int n_steps, double* loss_ring, int ring,
                   void* stream);

/* Adam / Adadelta (selectable through args.optimizer, code/MultiKE_model.py:15-25; TF1 defaults beta1 0.9, beta2 0.999,
 * epsilon 1e-8, rho 0.95).  These rules move weights whose gradient is zero, and TF applies them to the whole variable
 * (the gradient through tf.nn.l2_normalize(table, 1) is dense), so both entry points stream EVERY element: grad (consumed =
 * zeroed), the parameter and two slot arrays of the same layout -- Adam: slot1 = m, slot2 = v (both start at 0), `step` =
 * 1, 2, ... the number of this update;  Adadelta: slot1 = accum, slot2 = accum_update (both start at 0).  The row form
 * applies the l2_normalize Jacobian first when `normalize` is set, exactly as mke_rows_update does. */
typedef struct mke_optimizer {
  int kind;          /* MKE_OPT_ADAM | MKE_OPT_ADADELTA */
  float lr, beta1, beta2, epsilon, rho;
  int64_t step;      /* Adam bias correction */
} mke_optimizer;
int mke_rows_update_dense(float* table, float* slot1, float* slot2, float* grad, int64_t n_rows, int stride, int dim,
                          int normalize, const mke_optimizer* opt, void* stream);
int mke_dense_update_opt(float* param, float* slot1, float* slot2, float* grad, int64_t n, const mke_optimizer* opt,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * Space-mapping step of the SSL driver (code/losses.py:53-63; graph code/MultiKE_model.py:241-261; loop :439-454):
 *     loss = sum_k [ sum (F - l2n(V_k[idx] @ M_k))^2 + orthogonal_weight * ||M_k M_k^T - I||^2 + norm_w * ||M_k||^2 ]
 *   F = rows idx of the shared table (trainable, normalise-on-read), V_k = rows idx of view k's table (constant here),
 *   l2n = tf.nn.l2_normalize with no axis (the whole [n, dim] batch), M = [n_views][dim][dim] packed, row-major.
 *   One call = forward, backward, the row update of the shared table and the dense update of the matrices (update != 0).
 *   gM must be all-zero on entry (the dense update restores it; with update == 0 the caller reads and clears it).
 *   scratch: mke_mapping_scratch_floats(n, dim) floats.  partials: double[2 * MKE_MAPPING_MAX_VIEWS * MKE_LOSS_PARTIALS]
 *   scratch (per view k: block 2k = partial sums of P_k^2, block 2k+1 = partial sums of G_k . out_k).
 *   loss_partials: double[(MKE_MAPPING_MAX_VIEWS + 1) * MKE_LOSS_PARTIALS], overwritten: one block per view (map loss) and one
 *   for the orthogonality + norm terms; the loss is the sum of all of it.  dim <= 88.
 * ------------------------------------------------------------------------------------------------ */
#define MKE_MAPPING_MAX_VIEWS 3
typedef struct mke_mapping_view { const float* table; int normalize; } mke_mapping_view;
typedef struct mke_mapping_step_args {
  float* ent_table; int64_t n_ent; int ent_normalize; float* ent_acc; float* ent_grad /*NULL = constant*/; int32_t* ent_touched;
  mke_mapping_view views[MKE_MAPPING_MAX_VIEWS]; int n_views;
  int stride, dim;
  const int32_t* idx; int64_t n;
  float* M; float* gM; float* accM /*nullable for SGD*/;
  float orthogonal_weight, norm_w;
  float* scratch; double* partials;
  int optimizer; float lr; int32_t tag; int update;
} mke_mapping_step_args;
int64_t mke_mapping_scratch_floats(int64_t n, int dim);
int mke_mapping_step(const mke_mapping_step_args* args, double* loss_partials, void* stream);
/* The same step cut where the batch-wide sums live, for the row-sharded trainer (every rank maps the entities of the step it
 * owns; multike_amd/distributed_views.py): FWD (gathers, P_k = V_k M_k, partials block 2k) | all-reduce of sum P_k^2 — the
 * caller replaces block 2k by the total | TAIL (losses, G_k, partials block 2k+1) | all-reduce of sum G_k . out_k, block 2k+1
 * replaced | BWD (gM += V_k^T dP_k) | all-reduce of gM | UPD (orthogonality / norm terms added to gM, row scatter, updates).
 * A part with n == 0 contributes zeros.  MKE_MAP_ALL in one call is mke_mapping_step. */
#define MKE_MAP_FWD 1
#define MKE_MAP_TAIL 2
#define MKE_MAP_BWD 4
#define MKE_MAP_UPD 8
#define MKE_MAP_ALL 15
int mke_mapping_step_phases(const mke_mapping_step_args* args, double* loss_partials, int phases, void* stream);
/* n_steps consecutive steps: step s uses idx[step_off[s] .. step_off[s+1]) (step_off: HOST array), tag args->tag + s, and writes
 * its loss partials to loss_ring[s % ring][(MKE_MAPPING_MAX_VIEWS + 1) * MKE_LOSS_PARTIALS]; args->n is ignored. */
int mke_mapping_steps(const mke_mapping_step_args* args, const int64_t* step_off, int n_steps, double* loss_ring, int ring,
                      void* stream);

/* dense Adagrad / SGD over n contiguous floats; grad is zeroed — tf.train.AdagradOptimizer on the CNN variables */
int mke_dense_update(float* param, float* acc /*nullable for SGD*/, float* grad, int64_t n, int optimizer, float lr,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * (9) Alignment evaluator ("next" row §8f-2): rank of the gold counterpart and arg-max column under the inner
 *     product of (already normalised) embeddings, on the f32 matrix cores, without materialising the matrix.
 *
 * replaces: code/base/similarity.py:30-34 (sim = E1 . E2^T) + code/base/alignment.py:141-163 calculate_rank
 *           (argsort / argpartition per row, gold located by position) as used by greedy_alignment (:8-79).
 *
 *   emb1 : [n1][ld1], emb2 : [n2][ld2], both row-major with columns [dim, kpad) zero, kpad a multiple of 16, ld1 and
 *   ld2 multiples of 4 (16-byte rows).  Gold column of row i is i (so n2 >= n1).
 *   rank[i] += #{j < n2 : sim[i][j] > sim[i][i]}  (rank zeroed by the caller); with `ties` also the columns that tie with
 *   the gold (the reference's argsort puts the gold at an arbitrary place among them: the host reports the mid-rank);
 *   best[i]  = max over j of (ordered(sim[i][j]) << 32 | 0xFFFFFFFF - j)  (best zeroed by the caller; arg-max column =
 *              0xFFFFFFFF - low word, lowest column on ties).
 * ------------------------------------------------------------------------------------------------ */
int mke_align_rank(const float* emb1, int ld1, const float* emb2, int ld2, int kpad, int64_t n1, int64_t n2,
                   int32_t* rank, int32_t* ties /* nullable: ties[i] += #{j : sim[i][j] == sim[i][i]} (j = i included) */,
                   uint64_t* best, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (10) Small dense f32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32, exact f32) with arbitrary operand strides:
 *        C[M][N] (=|+=) A[M][K] . B[K][N],  A(i,k) = A[i*a_row_stride + k*a_col_stride], likewise B.
 *      splits > 1: split-K, partial products are added atomically (accumulate must be 1; C zeroed or holding the
 *      value to add to).  Used by mke_attr_step for the dense layer of the attribute CNN (tf.layers.dense,
 *      code/MultiKE_model.py:59) and its two gradient products.
 * ------------------------------------------------------------------------------------------------ */
int mke_gemm_f32(const float* A, int64_t a_row_stride, int64_t a_col_stride, const float* B, int64_t b_row_stride,
                 int64_t b_col_stride, float* C, int64_t ldc, int M, int N, int K, int splits, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (12) Literal auto-encoder: training steps and the final encoding as native calls on the
 *      hand-written f32 MFMA GEMMs with fused epilogues (mke_autoenc.hip, mke_gemm.hip).
 *
 * replaces: AutoEncoderModel's graph and training loop, code/literal_encoder.py:41-112 (`_init_graph`, `encoder`,
 *           `decoder`, `train_one_epoch`: session.run([loss, optimizer]) per batch) and `encoder_multi_batches` (:114-144).
 *
 *   dims[0..n_layers] = input width, hidden widths, code width; the decoder mirrors them.  Parameters are ONE packed
 *   float buffer; tensor t (t < n_layers: encoder layer t; t >= n_layers: decoder layer t - n_layers) has its weight
 *   [in][out] row-major WITH ROW STRIDE (out + 3) & ~3 at params + w_off[t] and its bias [out] at params + b_off[t];
 *   offsets must be multiples of 4 floats (pad entries are zero and have a zero gradient for ever), so that every
 *   operand of every product is readable with aligned 16-byte loads.  grads is all-zero on entry and on exit when
 *   update != 0 (the update consumes it); acc is the Adagrad accumulator (filled with 0.1 by the caller, TF1 default).
 *   act: 0 none (what the shipped "thah" selects, :75-78), 1 tanh, 2 sigmoid.  normalize: tf.nn.l2_normalize over the
 *   WHOLE code matrix of the batch (:65-66).
 *   mke_ae_train_steps: rows [0, n_rows) of x in batches of batch_rows (the last may be short); loss_out[b] (device
 *   double) = mean((decoded - x)^2) of batch b.  scratch: mke_ae_scratch_floats(plan, min(batch_rows, n_rows)) floats;
 *   partials: double[3 * MKE_LOSS_PARTIALS], all-zero on entry and on exit; scalars: float[4].
 * ------------------------------------------------------------------------------------------------ */
#define MKE_AE_MAX_LAYERS 4
typedef struct mke_ae_plan {
  int n_layers;
  int dims[MKE_AE_MAX_LAYERS + 1];
  int act, normalize;
  float* params; float* grads; float* acc /*nullable for SGD*/;
  int64_t n_params;
  int64_t w_off[2 * MKE_AE_MAX_LAYERS], b_off[2 * MKE_AE_MAX_LAYERS];
  int optimizer; float lr; int update;
  float* scratch; int64_t scratch_floats;
  double* partials; float* scalars;
} mke_ae_plan;
int64_t mke_ae_scratch_floats(const mke_ae_plan* plan, int64_t rows);
int mke_ae_train_steps(const mke_ae_plan* plan, const float* x, int64_t n_rows, int64_t ldx, int64_t batch_rows,
                       double* loss_out /* device [ceil(n_rows / batch_rows)] */, void* stream);
/* ONE batch cut at its batch-wide sums, for data-parallel training of the auto-encoder (the code matrix is
 * normalised as a whole, code/literal_encoder.py:65-66): every rank holds the replicated parameters and `rows` of the
 * `global_rows` rows of the batch (rows may be 0).  Between the phases the caller all-reduces
 *   plan->partials[0 ..)                      (sum code^2)      after MKE_AE_ENC,
 *   plan->partials[MKE_LOSS_PARTIALS ..)      (sum dcn . code)  after MKE_AE_DEC,
 * (replace each array by [global sum, 0, 0, ...]: the consuming phase adds the entries up itself and re-zeroes them), and
 * plan->grads after MKE_AE_BWD (then identical updates on every rank).  loss_out[0] = this rank's share of the batch's
 * loss, sum over its rows / (global_rows * dims[0]): the shares add up to mean((decoded - x)^2).  MKE_AE_ALL with
 * rows == global_rows is one step of mke_ae_train_steps. */
#define MKE_AE_ENC 1
#define MKE_AE_DEC 2
#define MKE_AE_BWD 4
#define MKE_AE_UPD 8
#define MKE_AE_ALL 15
int mke_ae_step_phases(const mke_ae_plan* plan, const float* x, int64_t rows, int64_t ldx, int64_t global_rows, int phases,
                       double* loss_out, void* stream);
/* out [n_rows][ld_out] = encoder(x): code/literal_encoder.py:114-144 (no normalisation of input or output) */
int mke_ae_encode(const mke_ae_plan* plan, const float* x, int64_t n_rows, int64_t ldx, float* out, int64_t ld_out, void* stream);
/* one dense layer, out [M][ld_out] = act(x [M][ldx] @ w [K][ldw] + b [N]) (b nullable): the `encoder` / `decoder` methods of
 * code/literal_encoder.py:71-91 on arbitrary inputs */
int mke_dense_layer_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* b /*nullable*/, int act, float* out,
                        int64_t ld_out, int M, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (13) Multi-GPU relation-view step, "owner computes" form (mke_oc.hip; new design, the reference is
 *      single-device).  Entity rows are sharded id % n_ranks (local row = id / n_ranks); the relation table is replicated.
 *      A global step = n_pos positives in the reference's epoch order; rank g is HOME of positives [g*per, (g+1)*per).
 *      Instead of moving entity rows to the triples, each rank scores the negatives whose corrupted entity it owns; what
 *      crosses the links per positive p is HR_p = h^ + r^ (built by the owner of h) when a negative of p corrupts the tail,
 *      RT_p = r^ - t^ (built by the owner of t) when one corrupts the head, and the gradients w.r.t. them.  The reference's
 *      sampler tosses one coin per ROUND (code/base/batch.py:97-105), so almost every positive needs only ONE of the two; the
 *      positive's own term is scored like a negative, by the owner of the entity the travelling vector lacks
 *      (d = HR_p - t^ at the owner of t, or h^ + RT_p at the owner of h).  Per step, on every rank:
 *          mke_oc_bases  -> all-gather of the send blocks (mke_oc_count meanwhile) -> mke_oc_score -> reduce-scatter of
 *          g_all -> mke_oc_apply -> all-reduce of rel_grad -> mke_rows_update_multi(relation table, shard)
 *      Block of rank o (mke_oc_block_floats floats): [capacity] HR vectors (slot order) | [capacity] RT vectors.
 *      codes: the negatives as (corrupt entity << 1) | corrupted-head (entity ids < 2^29), packed by every home rank for its
 *      own positives (mke_oc_pack_codes; table-independent, exchanged once per epoch); the codes of home rank g's positives
 *      of this step are codes[code_off[g] + j * neg_per_pos + n] (j-th positive of its slice).  The FIRST code of a positive
 *      also carries which vectors its group needs: MKE_OC_NEED_RT (some negative corrupts the head), MKE_OC_NEED_HR (some
 *      negative corrupts the tail, or none corrupts the head: the positive's own term needs one of the two).
 *      slot_h[i] / slot_t[i]: slot of positive i's HR / RT vector in its owner's block (i-th positive of the step; any
 *      numbering the ranks agree on), -1 when that vector is not needed; own_h / own_t: the positives whose head / tail this
 *      rank owns AND whose HR / RT vector is needed, in slot order.
 *      g_all: [n_ranks][2 * capacity][stride] — every slot is overwritten each step; gv: this rank's block after the
 *      reduce-scatter.  ref_count (nullable): zero-invariant counters of the exclusive-row fast path (a corrupt row
 *      referenced once in the whole global step is updated in place by mke_oc_score).  Same arithmetic as
 *      mke_triple_score_fwd_bwd_x; both tables are read through l2_normalize (the relation view's tables).
 * ------------------------------------------------------------------------------------------------ */
#define MKE_OC_MAX_RANKS 16
#define MKE_OC_NEED_HR 0x40000000u
#define MKE_OC_NEED_RT 0x80000000u
typedef struct mke_oc_step {
  float* ent; float* ent_acc /*nullable: SGD*/; float* ent_grad; int32_t* ent_touched; int32_t* ref_count /*nullable*/;
  int64_t n_local;
  const float* rel; float* rel_acc /*mke_oc_run's update only*/; float* rel_grad; int rel_grad_copies; int32_t* rel_touched;
  int64_t n_rel;
  int stride, dim, rank, n_ranks;
  const int32_t* pos_h; const int32_t* pos_r; const int32_t* pos_t;   /* [n_pos], GLOBAL entity ids */
  int64_t n_pos, per;
  const int32_t* slot_h; const int32_t* slot_t;                       /* [n_pos] */
  const int32_t* own_h; int64_t n_own_h; const int32_t* own_t; int64_t n_own_t;
  int neg_per_pos; int64_t capacity;
  const int32_t* codes; int64_t code_off[MKE_OC_MAX_RANKS];
  int optimizer; float lr, scale; int32_t tag;
  /* peer-direct mode (n_peers == n_ranks; 0 = the blocks are exchanged by collectives): peer_v[o] = rank o's SEND block,
   * peer_g[o] = this rank's slice [2 * capacity][stride] of rank o's gradient inbox — both mapped into this process
   * (hipIpc): mke_oc_score reads the vectors and writes its partial gradient vectors straight over xGMI, mke_oc_apply sums
   * the n_ranks slices of its own inbox in rank order.  The caller places a cross-GPU barrier after mke_oc_bases and after
   * mke_oc_score. */
  int n_peers; const float* peer_v[MKE_OC_MAX_RANKS]; float* peer_g[MKE_OC_MAX_RANKS];
  const float* pos_w;   /* nullable: [n_pos] weights of the positives (the weighted cross-KG loops, code/losses.py:44-50) */
  /* version 104: hub rows of this rank's shard (mke_hot_rows over LOCAL rows; slot == NULL: none) — rows that are head / tail of
   * many positives of every global step: mke_oc_apply and the positives' own terms add to their private copies (ent_grad then
   * has hot.row0 + hot.copies * hot.n_hot rows, hot.row0 >= n_local), they are not reference-counted and never finished in
   * place; mke_oc_run's update adds the copies */
  mke_hot_rows hot;
  /* version 105: ENTITY-MAJOR second pass (em_coef != NULL selects it; new design).  mke_oc_score then writes
   * no row gradient at all: per (positive, owned negative) — and for the positive's own term — it stores ONE coefficient,
   * em_coef[(em_pos0 + i) * (neg_per_pos + 1) + n] (n == neg_per_pos: the own term), and mke_oc_pass2 finishes every touched
   * owned row of the GLOBAL STEP in place from the row's reference list (mke_oc_em_plan, sorted by row: a fixed summation
   * order, so results are bit-reproducible run to run): ghat = sum coef (c^ + sg V) + sum (+-) gv, then the Jacobian of the
   * normalisation and the optimizer — no gradient scratch, no touched flags, no reference counts, no hub-row copies, no
   * atomics on entity rows, and the entity table takes no part in mke_oc_apply / the update launch.
   * em_refs: pairs (locator, coefficient index) of the whole epoch; em_rows / em_off: the work items of THIS step (a touched owned
   * row, or one 32-reference segment of a long row's list) and their offsets into em_refs (em_off[em_n_rows] valid); em_v[c] / em_gv[c]: chunk c's all-gathered vectors [n_ranks][block] and
   * reduce-scattered gradient block (the step's parts, at most MKE_OC_EM_MAX_CHUNKS). */
  float* em_coef; int64_t em_pos0;
  const uint32_t* em_refs; const int32_t* em_rows; const int32_t* em_off; int64_t em_n_rows;
  int em_chunks; int64_t em_block_floats; const float* em_v[4]; const float* em_gv[4];
  /* long rows (lists of more than 32 references: hub entities, frequent relations): em_rows / em_off are the plan's work ITEMS
   * (item_row / item_off of this step); em_part[w] the partial slot of item w and em_long_rows / em_long_part0 this step's long
   * rows, slots as the plan numbered them — slot p of this step lives at em_partials + (p - em_part0) * (stride + 16) floats
   * (em_part0 = the plan's step_part0 of this step; the buffer holds the step's slots only); mke_oc_pass2 adds a combine launch
   * when em_n_long > 0. */
  const int32_t* em_part; const int32_t* em_long_rows; const int32_t* em_long_part0; int64_t em_n_long; int64_t em_part0; float* em_partials;
  /* which work items a mke_oc_pass2 call takes: 0 all; 1 those WITHOUT a gradient-vector reference (bit 30 of the plan's item_row
   * clear: they need only the coefficients and the all-gathered vectors, so they may run while the step's reduce-scatter is on the
   * wire); 2 those with one (after the reduce-scatter), followed by the long rows' combine launch. */
  int em_mode;
  const mke_tuning* tuning;   /* version 105: host pointer, NULL = the process defaults */
} mke_oc_step;
#define MKE_OC_EM_MAX_CHUNKS 4
#define MKE_OC_EM_WAVES 32768
int64_t mke_oc_block_floats(int64_t capacity, int stride);
/* codes[e] of negative e = (p, n) of positives pos_h[0..n_pos): neg_h / neg_t are mke_neg_sample's output; the group flags
 * (MKE_OC_NEED_*) go into codes[p * neg_per_pos] */
int mke_oc_pack_codes(const int32_t* pos_h, const int32_t* neg_h, const int32_t* neg_t, int64_t n_pos, int neg_per_pos,
                      int32_t* codes, void* stream);
/* Per-epoch plan (table-independent; new design, no reference counterpart): part k of the epoch = epoch positions
 * [part_lo[k], part_lo[k+1]) (device array of n_parts + 1 offsets); codes = the WHOLE epoch's codes by epoch position
 * (codes[i * neg_per_pos + n]; nullable when neg_per_pos == 0: every positive then needs HR).  slot_h[i] / slot_t[i] = rank of
 * positive i among the positives of its part that need HR / RT and whose head / tail has the same owner (id % n_ranks), in
 * epoch order, -1 when positive i does not need that vector; own_h / own_t[part_lo[k] + s] = position inside part k of the
 * positive holding slot s of THIS rank's block (s < counts[...][rank]);
 * counts[(x * n_parts + k) * n_ranks + g] = positives of part k needing HR (x = 0) / RT (x = 1) whose head / tail rank g owns. */
int mke_oc_plan(const int32_t* pos_h, const int32_t* pos_t, const int32_t* codes, int neg_per_pos, const int64_t* part_lo,
                int n_parts, int n_ranks, int rank, int32_t* slot_h, int32_t* slot_t, int32_t* own_h, int32_t* own_t,
                int32_t* counts, void* stream);
int mke_oc_bases(const mke_oc_step* step, float* send_block, void* stream);
int mke_oc_count(const mke_oc_step* step, void* stream);
int mke_oc_score(const mke_oc_step* step, const float* v_all, int64_t block_floats, float* g_all,
                 double* loss_partials /* [MKE_LOSS_PARTIALS] */, void* stream);
int mke_oc_apply(const mke_oc_step* step, const float* gv, void* stream);
/* the phases selected by the bit mask, in the order above, in one call (what lies between two collectives);
 * MKE_OC_BASES | MKE_OC_COUNT is ONE launch (the counting needs only the codes: it runs on rider blocks beside the bases): */
#define MKE_OC_BASES 1
#define MKE_OC_COUNT 2
#define MKE_OC_SCORE 4
#define MKE_OC_APPLY 8
#define MKE_OC_UPDATE 16   /* mke_rows_update_multi: relation table (every row) + the shard's touched rows (entity-major: relation table only) */
#define MKE_OC_PASS2 32    /* entity-major second pass over the touched owned rows of the global step (after the LAST part's reduce-scatter) */
int mke_oc_run(const mke_oc_step* step, int phases, float* send_block, const float* v_all, int64_t block_floats, float* g_all,
               const float* gv, double* loss_partials, void* stream);

/* Entity-major second pass of the owner-computes step (version 105; new design, no reference counterpart; semantics matched:
 * code/MultiKE_model.py:304-310 — ONE update per row per step from the sum of all its contributions).
 * mke_oc_em_plan (per epoch, table-independent, after mke_oc_plan): the references of every global step to the rows THIS rank
 * owns, sorted by (step, local row, positive, kind) —
 *     negative n of positive i whose corrupt entity is owned         -> (vector of i's group, coefficient (i, n))
 *     own term of positive i (owner of t when HR travels, else of h) -> (vector, coefficient (i, neg_per_pos))
 *     head / tail of positive i owned and its HR / RT vector travels -> (+ / - the reduce-scattered gradient vector), and the
 *         same vector (+) into the positive's RELATION row, listed as local row n_local + r: mke_oc_pass2 STORES this rank's
 *         partial relation gradient into rel_grad (copy 0; one writer per row and step) for the all-reduce — no mke_oc_apply
 * Epoch positions [step_lo[s], step_lo[s+1]) are global step s; its parts are the `chunks` ceil-split slices of the step
 * (chunk of positive i = i / ceil(size / chunks)).  Outputs: refs[2 k], refs[2 k + 1] = locator and coefficient index of
 * reference k; rows[u] / off[u] = local row (>= n_local: relation row) and first reference of the u-th touched (step, row); step_row0[s] = first u of
 * step s (n_steps + 1 entries); n_refs[0] = references owned (> capacity: the plan is INVALID, enlarge and re-plan).
 * keys / keys_alt: capacity + 1 scratch keys each; flags / scan: capacity + 1 scratch ints each; temp: mke_oc_em_plan_temp_bytes. */
typedef struct mke_oc_em_plan_args {
  const int32_t* pos_h; const int32_t* pos_r; const int32_t* pos_t; const int32_t* codes; int neg_per_pos;
  const int32_t* slot_h; const int32_t* slot_t;
  const int64_t* step_lo; int n_steps; int chunks; int64_t n_all; int64_t max_step /* host: most positives of a step */;
  int n_ranks, rank; int64_t n_local, n_rel;
  uint64_t* keys; uint64_t* keys_alt; int64_t capacity;
  uint32_t* vals_alt;       /* capacity + 1 scratch ints (the sorted positions) */
  uint64_t* scratch8;       /* capacity + 1 scratch 8-byte words (the references at their unsorted positions) */
  int32_t* wave_scratch;    /* 2 * (MKE_OC_EM_WAVES + 1) scratch ints */
  uint32_t* refs; int32_t* rows; int32_t* off; int32_t* flags; int32_t* scan;
  int64_t* step_row0; int64_t* n_refs;
  /* the second pass's WORK ITEMS: a touched row's list in segments of at most 32 references.  item_row[w] = local row (bit 31:
   * a segment of a LONG row — more than one segment), item_off[w] = its first reference (item_off[w + 1] ends it), item_part[w]
   * = the partial slot a long row's segment writes (-1: the item finishes its row); long_row[l] / long_part0[l] = the long rows
   * and their first partial slot (long_part0[l + 1] ends them); step_item0 / step_long0 / step_part0[s] = first item / long row /
   * partial slot of global step s (n_steps + 1 entries each).  item_* : capacity + 1 ints each; long_*: capacity / 32 + 2.  Bit 30 of
   * item_row[w]: the item's references include a gradient vector (mke_oc_step.em_mode). */
  int32_t* item_row; int32_t* item_off; int32_t* item_part; int32_t* long_row; int32_t* long_part0;
  int64_t* step_item0; int64_t* step_long0; int64_t* step_part0;
  void* temp; int64_t temp_bytes;
} mke_oc_em_plan_args;
int64_t mke_oc_em_plan_temp_bytes(int64_t capacity);
int mke_oc_em_plan(const mke_oc_em_plan_args* args, void* stream);
/* the second pass of the step `step` describes (its em_* fields; launched once per global step) */
int mke_oc_pass2(const mke_oc_step* step, void* stream);

/* The step loop of the owner-computes relation view as ONE native call (version 105; new design — the single-device
 * counterpart is mke_relation_steps; semantics: code/MultiKE_model.py:304-322, one optimizer step per global batch, in epoch
 * order).  mke_oc_comm: the three collectives of the step.  kind NCCL: `all_gather` / `reduce_scatter` / `all_reduce` are the
 * addresses of ncclAllGather / ncclReduceScatter / ncclAllReduce of the RCCL the caller created `ctx` (an ncclComm_t) with —
 * this library neither links nor loads RCCL; kind CALLBACK: int fn(void* ctx, const float* send, float* recv, int64_t count,
 * void* stream) for all_gather (count = floats SENT) and reduce_scatter (count = floats RECEIVED), int fn(void* ctx, float*
 * buf, int64_t count, void* stream) for all_reduce, each returning 0 after enqueueing (or completing) the collective in
 * stream order; kind LOOPBACK: a one-GPU measurement stand-in — the bytes a rank of `world` would receive are written in HBM
 * and the stream is held for bytes / wire_gbps + latency_us (what tools/oc_rank_compute.py models the links with). */
#define MKE_OC_COMM_NCCL 0
#define MKE_OC_COMM_CALLBACK 1
#define MKE_OC_COMM_LOOPBACK 2
typedef struct mke_oc_comm {
  int kind; void* ctx; void* all_gather; void* reduce_scatter; void* all_reduce;
  int world, rank; float wire_gbps, latency_us;   /* LOOPBACK only */
} mke_oc_comm;
/* parts: HOST array of the epoch's parts in order (mke_oc_step; at most `chunks` per global step), step_part0: HOST array of
 * n_steps + 1 first-part indices; chunk c's exchange buffers send[c] (this rank's block), v_all[c] ([n_ranks] blocks), g_all[c]
 * ([n_ranks][2 capacity][stride]) and gv[c]; the losses of part c of step s go to loss_ring + (s * chunks + c) * loss_stride
 * doubles (MKE_LOSS_PARTIALS each); step s of the call carries tag tag_base + (s - step_begin) + 1.  comm == NULL: one rank, no
 * collectives.  comm_stream (nullable): with chunks > 1, the all-gather / reduce-scatter of a part run there, ordered against
 * `stream` by events created for the duration of the call, so that they overlap the other parts' scoring; NULL or == stream:
 * everything in stream order on `stream`. */
typedef struct mke_oc_loop {
  const mke_oc_step* parts; const int32_t* step_part0; int n_steps;
  int chunks; float* send[4]; float* v_all[4]; float* g_all[4]; float* gv[4]; int64_t block_floats;
  double* loss_ring; int64_t loss_stride; int32_t tag_base;
  const mke_oc_comm* comm; void* comm_stream;
  int overlap_rs;   /* entity-major steps with one part and a comm_stream: the reduce-scatter goes to comm_stream and the work items
                       without gradient-vector references (em_mode 1) run under it; the rest (em_mode 2) after it */
} mke_oc_loop;
int mke_oc_steps(const mke_oc_loop* loop, int step_begin, int step_end, void* stream);

/* The per-step entry points with their tuning as an argument (version 105): mke_triple_score_fwd_bwd_xch and
 * mke_rows_update_multi_count, `tuning` = NULL being exactly those. */
int mke_triple_score_fwd_bwd_t(
    float* ent_table, int64_t n_ent, int ent_normalize, const float* rel_table, int64_t n_rel, int rel_normalize,
    int stride, int dim, const int32_t* pos_h, const int32_t* pos_r, const int32_t* pos_t, const float* pos_w,
    int64_t n_pos, const int32_t* neg_h, const int32_t* neg_r, const int32_t* neg_t, const float* neg_w, int64_t n_neg,
    int neg_per_pos, float scale, float* grad_ent, float* grad_rel, int grad_rel_copies, int32_t* touched_ent,
    int32_t* touched_rel, int32_t tag, int32_t* ref_count, float* ent_acc, int optimizer, float lr,
    const mke_count_job* next_count, const mke_hot_rows* hot, const mke_tuning* tuning, double* loss_partials, void* stream);
int mke_rows_update_multi_t(const mke_update_table* tables, int n_tables, int32_t tag, int stride, int dim, int optimizer,
                            float lr, const mke_count_job* count, const mke_tuning* tuning, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MULTIKE_HIP_H */
