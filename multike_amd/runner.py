"""RelationViewRunner — drives `mke_relation_steps`, the native (C++) step loop of the relation view.

One call enqueues many consecutive steps (sampler chunk -> fused triple step -> row update) on the current
stream with no Python between the steps; this is the replacement of the reference's per-step
`batch_queue.get()` + `session.run()` loop (code/MultiKE_model.py:302-312).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .sampling import RelationBatcher
from .tables import EmbeddingTable

_OPT = {"Adagrad": _lib.OPT_ADAGRAD, "SGD": _lib.OPT_SGD}


class RelationViewRunner:
    def __init__(self, ent: EmbeddingTable, rel: EmbeddingTable, batcher: RelationBatcher, opt_name: str = "relation",
                 lr: float = 0.001, optimizer: str = "Adagrad", scale: float = 1.0, sample_chunk: int | None = None,
                 max_try: int = 10, exclusive_rows: bool = True, overlap: bool | None = None, hot_rows: bool | None = None,
                 tuning: dict | None = None, deterministic: bool | None = None):
        """tuning: performance knobs of THIS runner's plan (names of `_lib.TUNING_FIELDS`; mke_relation_plan.tuning) — every knob
        not named follows the process default; deterministic: the fixed-order mode for this runner (None: the process default)."""
        self.tuning = _lib.tuning(**tuning) if tuning else None
        self.deterministic = deterministic
        if optimizer not in _OPT:
            raise _lib.MultiKEHipError(f"optimizer {optimizer!r} not supported by the HIP path (Adagrad, SGD)")
        self.ent, self.rel, self.bat = ent, rel, batcher
        self.opt_name, self.lr, self.optimizer, self.scale, self.max_try = opt_name, lr, optimizer, scale, max_try
        self.steps = batcher.steps
        self.exclusive_rows = exclusive_rows
        N = batcher.neg_per_pos
        # Hub rows: entities that are head / tail of HOT_MIN (20) or more positives of an average step get private copies of their
        # gradient row for the groups' flushes (the degree distribution is static: the positives are the KGs' triples in a new
        # order every epoch).  hot_rows=None: when the table has no hub declaration yet and the KG has any; False: never.
        if hot_rows is not False and ent.n_hot == 0 and ent.grad_copies == 1 and ent._grad is None and N > 0 and self.steps > 0:
            self._declare_hot_rows(ent, batcher)
        # Default: negatives of a whole chunk are sampled by one launch on the same stream — by default the whole epoch
        # (910K positives x 25 x 12 B = 273 MB at the DBP-WD shape: nothing next to 288 GB).
        # overlap=True (opt-in): the next step's reference counts and the next chunk's negatives are produced on a
        # second stream while the current step is scored and updated (chunks of 8 steps, two chunk buffers).  Measured
        # on MI355X at the C2 shape it is SLOWER (87 vs 75 us/step): three cross-stream event waits per step cost more
        # than the ~13 us of table-independent work they hide.
        self.overlap = False if overlap is None else bool(overlap)
        if self.overlap and not (exclusive_rows and N > 0):
            raise _lib.MultiKEHipError("overlap mode needs negatives and exclusive_rows")
        default_chunk = 8 if self.overlap else self.steps
        self.sample_chunk = default_chunk if sample_chunk is None else max(1, min(int(sample_chunk), max(self.steps, 1)))
        off = batcher.off
        span = max(int(off[min(s + self.sample_chunk, self.steps)] - off[s]) for s in range(self.steps)) if self.steps else 0
        self.neg_chunk_capacity = max(1, span * N)
        nbuf = 2 if self.overlap else 1
        self.neg = tuple(torch.empty(nbuf * self.neg_chunk_capacity, dtype=torch.int32, device=ent.device) for _ in range(3))
        # two halves: steps alternate, so that step s+1 can be counted while step s is being updated
        self.refcount = torch.zeros(2 * ent.n_rows, dtype=torch.int32, device=ent.device) if exclusive_rows else None
        self.loss = torch.zeros(max(1, self.steps), _lib.LOSS_PARTIALS, dtype=torch.float64, device=ent.device)
        self._step_off = np.ascontiguousarray(off, dtype=np.int64)
        self.tag = 0  # tags handed out so far; each epoch consumes `steps` of them
        self._epoch_tag_base = None
        self.plan = _lib.RelationPlanStruct()
        self._fill_static()

    # measured on the Zipf(1.0) C2 shape (EXPERIMENTS R5.2): 4 copies already take the score launch from 44 to 37-38 us (uniform: 35.5);
    # more copies and a lower threshold only lengthen the update launch (each hub row sums its copies: 32 copies +4.5 us)
    HOT_MIN, HOT_MAX, HOT_COPIES = 20.0, 1024, 8

    def _declare_hot_rows(self, ent: EmbeddingTable, b: RelationBatcher):
        deg = torch.bincount(torch.cat([b.pos_h, b.pos_t]).long(), minlength=ent.n_rows).float() / max(1, self.steps)
        hot = torch.nonzero(deg >= self.HOT_MIN).reshape(-1)
        if hot.numel() > self.HOT_MAX:
            hot = torch.topk(deg, self.HOT_MAX).indices
        if hot.numel():
            ent.set_hot_rows(hot.cpu().numpy(), self.HOT_COPIES)

    def _fill_static(self):
        p, e, r, b = self.plan, self.ent, self.rel, self.bat
        if e.stride != r.stride or e.dim != r.dim:
            raise _lib.MultiKEHipError("entity and relation tables must share dim/stride")
        adagrad = self.optimizer == "Adagrad"
        p.ent_table, p.n_ent, p.ent_normalize = _lib.ptr(e.data, torch.float32, "ent"), e.n_rows, int(e.normalize)
        p.rel_table, p.n_rel, p.rel_normalize = _lib.ptr(r.data, torch.float32, "rel"), r.n_rows, int(r.normalize)
        p.ent_acc = _lib.ptr(e.slot(self.opt_name), torch.float32, "acc") if adagrad else None
        p.rel_acc = _lib.ptr(r.slot(self.opt_name), torch.float32, "acc") if adagrad else None
        p.ent_grad, p.rel_grad = _lib.ptr(e.grad, torch.float32, "g"), _lib.ptr(r.grad, torch.float32, "g")
        p.rel_grad_copies = r.grad_copies
        if e.grad_copies != 1:
            raise _lib.MultiKEHipError("the entity table's gradient scratch cannot be privatised")
        p.ent_touched, p.rel_touched = _lib.ptr(e.touched, torch.int32, "t"), _lib.ptr(r.touched, torch.int32, "t")
        p.ent_ref_count = _lib.ptr(self.refcount, torch.int32, "refcount") if self.exclusive_rows else None
        p.overlap = int(self.overlap)
        p.neg_chunk_capacity = self.neg_chunk_capacity
        p.stride, p.dim = e.stride, e.dim
        p.pos_kg = _lib.ptr(b.pos_kg, torch.uint8, "pos_kg")
        p.step_off = self._step_off.ctypes.data_as(C.POINTER(C.c_int64))
        p.n_steps = self.steps
        p.neg_per_pos, p.max_try, p.sample_chunk = b.neg_per_pos, self.max_try, self.sample_chunk
        p.neg_h, p.neg_r, p.neg_t = (_lib.ptr(x, torch.int32, "neg") for x in self.neg)
        p.optimizer, p.lr, p.scale = _OPT[self.optimizer], self.lr, self.scale
        p.loss_partials, p.loss_ring = _lib.ptr(self.loss, torch.float64, "loss"), self.loss.shape[0]
        p.hot = e.hot_struct()
        p.tuning = _lib.tuning_ptr(self.tuning)

    def _fill_epoch(self):
        p, b = self.plan, self.bat
        b.side1.fill(p.sides[0])
        b.side2.fill(p.sides[1])
        # positives are re-materialised (new tensors) by every shuffle
        p.pos_h, p.pos_r, p.pos_t = (_lib.ptr(x, torch.int32, "pos") for x in (b.pos_h, b.pos_r, b.pos_t))
        p.seed_lo, p.seed_hi = b.rng_seed
        p.stream_id = b.rng_stream

    def run(self, step_begin: int = 0, step_end: int | None = None):
        """Enqueue steps [step_begin, step_end) of the current epoch (default: all).  A call with step_begin == 0
        starts a new epoch (fresh tag range); other calls continue the current one."""
        step_end = self.steps if step_end is None else step_end
        if (self.deterministic if self.deterministic is not None else _lib.get_option("deterministic")):
            # parity / debugging mode: step by step through StepEngine's deterministic path (fixed-order gradient sums)
            from .tables import StepEngine
            if getattr(self, "_det_engine", None) is None:
                self._det_engine = StepEngine(self.ent.device, loss_ring=max(2, self.steps), deterministic=True)
                self._det_engine.tag = 1 << 29
            N = self.bat.neg_per_pos
            for s in range(step_begin, step_end):
                pos, neg = self.bat.batch(s)
                if N and self.sample_chunk >= self.steps:       # the epoch's negatives where the native loop leaves them
                    lo = int(self.bat.off[s])
                    for dst, src in zip(self.neg, neg):
                        dst[lo * N:lo * N + src.numel()].copy_(src)
                lp = self._det_engine.relation_step(self.ent, self.rel, self.opt_name, pos, neg if self.bat.neg_per_pos else None,
                                                    neg_per_pos=self.bat.neg_per_pos, lr=self.lr, scale=self.scale,
                                                    optimizer=self.optimizer, exclusive_rows=self.exclusive_rows)
                self.loss[s].copy_(lp)
            return
        if step_begin == 0 or self._epoch_tag_base is None:
            self._epoch_tag_base = self.tag
            self.tag += self.steps
        self._fill_epoch()
        # StepEngine and this runner may share touched arrays: keep the tag spaces apart (runner: upper half)
        self.plan.tag_base = (1 << 30) + 1 + self._epoch_tag_base
        _lib.relation_steps(self.plan, step_begin, step_end)

    # ------------------------------------------------------------------------------------------------
    def run_epochs(self, n_epochs: int, on_epoch_end=None, prefetch: bool = False):
        """Train `n_epochs` whole epochs; `on_epoch_end(epoch_index, runner)` is called after each epoch's work is
        enqueued (its losses are in `self.loss`; reading them synchronises).
        prefetch=True: while epoch e is trained on the current stream, epoch e+1 is permuted and its negatives are sampled
        on a second stream (one cross-stream hand-over per epoch).  Measured on MI355X at the C2 shape this is SLOWER
        (79 vs 71 us/step): the 1 ms epoch sampler fills the chip with 910K latency-bound waves and starves the
        dependent score/update chain for longer than it would have taken in line.  Off by default."""
        if not prefetch or self.overlap or self.bat.neg_per_pos == 0 or self.steps == 0:
            for e in range(n_epochs):
                if e > 0:
                    self.bat.shuffle()
                self.run()
                if on_epoch_end:
                    on_epoch_end(e, self)
            return
        b = self.bat
        N = b.neg_per_pos
        total = int(b.off[-1]) * N
        if getattr(self, "_neg_sets", None) is None:
            self._neg_sets = [tuple(x[:total] for x in self.neg) if self.neg[0].numel() >= total else
                              tuple(torch.empty(total, dtype=torch.int32, device=self.ent.device) for _ in range(3)),
                              tuple(torch.empty(total, dtype=torch.int32, device=self.ent.device) for _ in range(3))]
            self._side = torch.cuda.Stream(device=self.ent.device)
            self._ev_staged, self._ev_sampled = torch.cuda.Event(), torch.cuda.Event()
        main = torch.cuda.current_stream()
        cur = 0
        from .sampling import sample_negatives

        def sample_epoch(pos, stream_id, out):
            sample_negatives(pos, b.side1, N, seed=b.rng_seed, stream_id=stream_id, pos_offset=0, max_try=self.max_try, out=out,
                             side1=b.side2, pos_kg=b.pos_kg)

        sample_epoch((b.pos_h, b.pos_r, b.pos_t), b.rng_stream, self._neg_sets[cur])      # first epoch: inline
        for e in range(n_epochs):
            if e + 1 < n_epochs:
                staged = b.stage_next_epoch()                       # tiny torch ops on the main stream
                self._ev_staged.record(main)
                with torch.cuda.stream(self._side):
                    self._side.wait_event(self._ev_staged)
                    old = _lib.pin_stream(self._side.cuda_stream)
                    try:
                        sample_epoch(staged, ((b.epoch + 1) * 2) & 0xFFFFFFFF, self._neg_sets[1 - cur])
                    finally:
                        _lib.pin_stream(old)
                    self._ev_sampled.record(self._side)
            # this epoch's steps: negatives are ready in set `cur`
            p = self.plan
            p.neg_h, p.neg_r, p.neg_t = (_lib.ptr(x, torch.int32, "neg") for x in self._neg_sets[cur])
            p.negatives_ready, p.sample_chunk = 1, self.steps
            self._epoch_tag_base = self.tag
            self.tag += self.steps
            self._fill_epoch()
            p.tag_base = (1 << 30) + 1 + self._epoch_tag_base
            try:
                _lib.relation_steps(p, 0, self.steps)
            finally:
                p.negatives_ready, p.sample_chunk = 0, self.sample_chunk
                p.neg_h, p.neg_r, p.neg_t = (_lib.ptr(x, torch.int32, "neg") for x in self.neg)
            if on_epoch_end:
                on_epoch_end(e, self)
            if e + 1 < n_epochs:
                main.wait_event(self._ev_sampled)                   # the one hand-over of the epoch
                b.commit_staged()
                cur = 1 - cur

    def epoch_loss_sum(self) -> torch.Tensor:
        """Sum of the batch losses of the steps run in this epoch (device scalar, float64)."""
        return self.loss.sum()

    def step_losses(self) -> torch.Tensor:
        return self.loss.sum(dim=1)


def run_positive_steps(ent: EmbeddingTable, rel: EmbeddingTable, opt_name: str, cols, weights, step_off, tag_base: int,
                       lr: float = 0.001, scale: float = 1.0, optimizer: str = "Adagrad") -> torch.Tensor:
    """Positives-only relation-style steps (the cross-KG inference loops, code/MultiKE_model.py:349-369,393-414) as ONE
    native call: step s scores positions [step_off[s], step_off[s+1]) of `cols` = (h, r, t) int32 (and `weights`),
    then updates both tables; tag of step s = tag_base + s.  Returns the loss partials [n_steps, LOSS_PARTIALS]."""
    if optimizer not in _OPT:
        raise _lib.MultiKEHipError(f"optimizer {optimizer!r} not supported by the HIP path (Adagrad, SGD)")
    if ent.stride != rel.stride or ent.dim != rel.dim:
        raise _lib.MultiKEHipError("entity and relation tables must share dim/stride")
    off = np.ascontiguousarray(step_off, dtype=np.int64)
    steps = len(off) - 1
    loss = torch.zeros(max(1, steps), _lib.LOSS_PARTIALS, dtype=torch.float64, device=ent.device)
    if steps <= 0:
        return loss[:0]
    adagrad = optimizer == "Adagrad"
    f32, i32 = torch.float32, torch.int32
    p = _lib.RelationPlanStruct()
    p.ent_table, p.n_ent, p.ent_normalize = _lib.ptr(ent.data, f32, "ent"), ent.n_rows, int(ent.normalize)
    p.rel_table, p.n_rel, p.rel_normalize = _lib.ptr(rel.data, f32, "rel"), rel.n_rows, int(rel.normalize)
    p.ent_acc = _lib.ptr(ent.slot(opt_name), f32, "acc") if adagrad else None
    p.rel_acc = _lib.ptr(rel.slot(opt_name), f32, "acc") if adagrad else None
    p.ent_grad, p.rel_grad, p.rel_grad_copies = _lib.ptr(ent.grad, f32, "g"), _lib.ptr(rel.grad, f32, "g"), rel.grad_copies
    p.ent_touched, p.rel_touched = _lib.ptr(ent.touched, i32, "t"), _lib.ptr(rel.touched, i32, "t")
    p.ent_ref_count, p.overlap, p.neg_chunk_capacity = None, 0, 0
    p.stride, p.dim = ent.stride, ent.dim
    p.pos_h, p.pos_r, p.pos_t = (_lib.ptr(c, i32, "pos") for c in cols)
    p.pos_w = _lib.ptr(weights, f32, "pos_w") if weights is not None else None
    p.pos_kg = None
    p.step_off, p.n_steps = off.ctypes.data_as(C.POINTER(C.c_int64)), steps
    p.neg_per_pos, p.max_try, p.sample_chunk, p.negatives_ready = 0, 0, 1, 0
    p.neg_h = p.neg_r = p.neg_t = None
    p.optimizer, p.lr, p.scale = _OPT[optimizer], lr, scale
    p.loss_partials, p.loss_ring, p.tag_base = _lib.ptr(loss, torch.float64, "loss"), steps, tag_base
    p.hot = ent.hot_struct()          # hub rows the relation view declared on this table: the same entities lead the supervision triples
    _lib.relation_steps(p, 0, steps)
    return loss


def run_alignment_steps(tables, terms, idx_a, idx_b, step_off, opt_name: str, tag_base: int, lr: float,
                        optimizer: str = "Adagrad") -> torch.Tensor:
    """A whole epoch of common-space steps as ONE native call (`mke_align_steps`; code/MultiKE_model.py:458-473).
    tables: list of EmbeddingTable (constant tables are left untouched); terms: [(index_a, index_b, weight)].
    Returns the loss partials [n_steps, n_terms, LOSS_PARTIALS]."""
    if optimizer not in _OPT:
        raise _lib.MultiKEHipError(f"optimizer {optimizer!r} not supported by the native step loops (Adagrad, SGD)")
    off = np.ascontiguousarray(step_off, dtype=np.int64)
    steps = len(off) - 1
    dev = tables[0].device
    loss = torch.zeros(max(1, steps), len(terms), _lib.LOSS_PARTIALS, dtype=torch.float64, device=dev)
    if steps <= 0:
        return loss[:0]
    f32, i32 = torch.float32, torch.int32
    p = _lib.AlignPlanStruct()
    p.n_tables, p.n_terms = len(tables), len(terms)
    for k, t in enumerate(tables):
        if t.stride != tables[0].stride or t.dim != tables[0].dim:
            raise _lib.MultiKEHipError("common-space tables must share dim/stride")
        s = p.tables[k]
        s.table, s.n_rows, s.normalize = _lib.ptr(t.data, f32, "table"), t.n_rows, int(t.normalize)
        if t.trainable:
            s.grad, s.touched = _lib.ptr(t.grad, f32, "grad"), _lib.ptr(t.touched, i32, "touched")
            s.acc = _lib.ptr(t.slot(opt_name), f32, "acc") if optimizer == "Adagrad" else None
        else:
            s.grad = s.touched = s.acc = None
    for k, (a, b, w) in enumerate(terms):
        p.terms[k].a, p.terms[k].b, p.terms[k].weight = int(a), int(b), float(w)
    p.stride, p.dim = tables[0].stride, tables[0].dim
    p.ia, p.ib = _lib.ptr(idx_a, i32, "ia"), _lib.ptr(idx_b, i32, "ib")
    p.step_off, p.n_steps = off.ctypes.data_as(C.POINTER(C.c_int64)), steps
    p.optimizer, p.lr, p.tag_base = _OPT[optimizer], float(lr), int(tag_base)
    p.loss_partials = _lib.ptr(loss, torch.float64, "loss")
    _lib.align_steps(p)
    return loss


class SpaceMappingState:
    """The three mapping matrices of the SSL driver's space-mapping graph (code/MultiKE_model.py:241-261), packed
    [n_views, d, d] with their gradient scratch and Adagrad accumulator, plus the scratch of the native step."""

    def __init__(self, matrices, device):
        self.n_views, self.dim = len(matrices), matrices[0].shape[0]
        self.M = torch.stack([m.detach().to(device=device, dtype=torch.float32) for m in matrices]).contiguous()
        self.gM = torch.zeros_like(self.M)
        self.accM = torch.full_like(self.M, 0.1)                 # tf.train.AdagradOptimizer initial accumulator
        self.partials = torch.zeros(2 * _lib.MAPPING_MAX_VIEWS * _lib.LOSS_PARTIALS, dtype=torch.float64, device=device)
        self._scratch = None

    def scratch(self, n):
        need = _lib.mapping_scratch_floats(n, self.dim)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(max(need, 1), dtype=torch.float32, device=self.M.device)
        return self._scratch


def run_space_mapping_steps(state: SpaceMappingState, ent: EmbeddingTable, views, idx, step_off, opt_name: str, tag_base: int,
                            lr: float, orthogonal_weight: float, optimizer: str = "Adagrad", norm_w: float = 0.0001,
                            update: bool = True) -> torch.Tensor:
    """A whole epoch of space-mapping steps as ONE native call (`mke_mapping_steps`).  `views`: the EmbeddingTables mapped
    onto the shared table `ent` (constants here).  Returns the loss partials [n_steps, 4, LOSS_PARTIALS]."""
    if optimizer not in _OPT:
        raise _lib.MultiKEHipError(f"optimizer {optimizer!r} not supported by the native step loops (Adagrad, SGD)")
    off = np.ascontiguousarray(step_off, dtype=np.int64)
    steps = len(off) - 1
    ring = torch.zeros(max(1, steps), 4, _lib.LOSS_PARTIALS, dtype=torch.float64, device=ent.device)
    if steps <= 0:
        return ring[:0]
    f32, i32 = torch.float32, torch.int32
    a = _lib.MappingStepArgs()
    a.ent_table, a.n_ent, a.ent_normalize = _lib.ptr(ent.data, f32, "ent"), ent.n_rows, int(ent.normalize)
    a.ent_acc = _lib.ptr(ent.slot(opt_name), f32, "acc") if optimizer == "Adagrad" else None
    a.ent_grad, a.ent_touched = _lib.ptr(ent.grad, f32, "grad"), _lib.ptr(ent.touched, i32, "touched")
    a.n_views = len(views)
    for k, t in enumerate(views):
        if t.stride != ent.stride or t.dim != ent.dim:
            raise _lib.MultiKEHipError("space-mapping tables must share dim/stride")
        a.views[k].table, a.views[k].normalize = _lib.ptr(t.data, f32, "view"), int(t.normalize)
    a.stride, a.dim = ent.stride, ent.dim
    a.idx, a.n = _lib.ptr(idx, i32, "idx"), int(np.diff(off).max())
    a.M, a.gM = _lib.ptr(state.M, f32, "M"), _lib.ptr(state.gM, f32, "gM")
    a.accM = _lib.ptr(state.accM, f32, "accM") if optimizer == "Adagrad" else None
    a.orthogonal_weight, a.norm_w = float(orthogonal_weight), float(norm_w)
    a.scratch = _lib.ptr(state.scratch(int(np.diff(off).max())), f32, "scratch")
    a.partials = _lib.ptr(state.partials, torch.float64, "partials")
    a.optimizer, a.lr, a.tag, a.update = _OPT[optimizer], float(lr), int(tag_base), int(update)
    _lib.mapping_steps(a, off, ring)
    return ring
