"""Embedding tables in HBM and the step executor that drives the HIP kernels.

`EmbeddingTable` plays the role of a TF variable created by `xavier_init(shape, name, is_l2_norm)`
(reference code/base/initializers.py:22-26, code/MultiKE_model.py:86-99): raw trainable rows plus the
"read through l2_normalize" flag.  Layout: float32 [n_rows, stride], stride = multiple of 16 >= dim, pad
columns are zero and stay zero.  Each optimizer that touches the table owns its own Adagrad accumulator
("slot") exactly as each `generate_optimizer` call does in the reference (code/MultiKE_model.py:28-31;
SURVEY.md §9.3-4).

`StepEngine` runs one "session.run([loss, optimizer])" worth of work as kernel launches on the current
stream: scatter kernel(s) -> per-table row update.  No host synchronisation happens here.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

ADAGRAD_INIT_ACC = 0.1  # tf.train.AdagradOptimizer default initial_accumulator_value (TF1)

# ---- placement by trial of the big row arrays ----------------------------------------------------------------------------
# Measured on MI355X (tools/hbm_map.py, profiles/r05_hbm_map.log; EXPERIMENTS R5.10): HBM allocations come in two classes — a
# read-only gather of random 1 KB rows takes 322-332 us on most 2 GB allocations and 349-357 us (+9 %) on others, the slow ones
# lying in runs of 12-26 GB in allocation order (~30 % of a fresh device), stable for the life of the allocation — and the
# relation step at the 2M x 256 shape runs 295-323 us on arrays of the fast class and 367-369 us on arrays of the slow class
# (tools/c5_probe.py): the "lottery" of +-12 % between runs that rounds 3-4 could only describe.  No virtual-address choice
# controls it, so arrays of >= MKE_PLACE_MIN_MB (default 1024) are placed by trial: candidates are allocated (all kept alive,
# so that they are different physical pages), each is timed with the probe (mke_probe_rows, 2M random rows, 0.3-1 ms) TOGETHER
# with the table's arrays that exist already, the fastest is kept and the rest returned to the driver.  MKE_PLACE=0 turns it off.
_PLACE_IDX = {}


def _probe_us(arr: torch.Tensor, companions=()) -> float:
    """us per launch of mke_probe_rows: 2M random rows of `arr` — and of up to two companion arrays of the same shape, read
    together the way the step reads a row of the table, its accumulator and its gradient."""
    import os
    n = arr.shape[0]
    key = (arr.device, n)
    if key not in _PLACE_IDX:
        g = torch.Generator(device=arr.device); g.manual_seed(12345)
        k = int(os.environ.get("MKE_PLACE_PROBE_ROWS", 1 << 21))
        _PLACE_IDX.clear()
        _PLACE_IDX[key] = (torch.randint(0, n, (k,), device=arr.device, generator=g, dtype=torch.int32),
                           torch.empty(k, dtype=torch.float32, device=arr.device))
    idx, out = _PLACE_IDX[key]
    comp = [c for c in companions if c is not None and c.dim() == 2 and c.shape[1] == arr.shape[1] and c.shape[0] >= n][:2]
    comp += [None] * (2 - len(comp))
    best = float("inf")
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.probe_rows(arr, comp[0], comp[1], idx, out)
        e1.record()
        e1.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def placed_rows(n_rows: int, stride: int, device, fill: float = 0.0, report: list | None = None, companions=()) -> torch.Tensor:
    """A float32 [n_rows][stride] array filled with `fill`; big ones (see above) on the fastest of MKE_PLACE_TRIES (default 8)
    candidate allocations, all alive while they are compared.  companions: the arrays of the same table that exist already (the
    table for its accumulator, both for the gradient scratch): the candidates are probed TOGETHER with them — single arrays of
    the fast class still made steps 12 % apart (tools/c5_probe.py: the combination has a term of its own, and the probe of
    the three arrays read together ranks the triples as the step does)."""
    import os
    device = torch.device(device)
    nbytes = n_rows * stride * 4
    big = device.type == "cuda" and stride % 16 == 0 and stride <= _lib.MAX_STRIDE and \
        nbytes >= int(os.environ.get("MKE_PLACE_MIN_MB", 1024)) << 20 and os.environ.get("MKE_PLACE", "1") != "0"
    if not big:
        return torch.full((n_rows, stride), fill, dtype=torch.float32, device=device) if fill else \
            torch.zeros(n_rows, stride, dtype=torch.float32, device=device)
    tries = int(os.environ.get("MKE_PLACE_TRIES", 8))
    budget = torch.cuda.mem_get_info(device)[0] // 4
    cands, times = [], []
    while len(cands) < tries and (len(cands) + 1) * nbytes <= budget:
        c = torch.empty(n_rows, stride, dtype=torch.float32, device=device)
        cands.append(c)
        times.append(_probe_us(c, companions))
    k = int(np.argmin(times)) if times else -1
    if report is not None:
        report.append({"bytes": nbytes, "read_with": len([c for c in companions if c is not None]), "probe_us": [round(t, 1) for t in times], "kept": k})
    if k < 0:
        return torch.full((n_rows, stride), fill, dtype=torch.float32, device=device)
    keep = cands[k]
    del cands, c
    torch.cuda.empty_cache()          # the rejected candidates go back to the driver (not into PyTorch's pool, which would hand them out again)
    keep.fill_(fill)
    return keep


PLACEMENT_LOG: list = []     # one entry per placed array of this process (bench.py prints it)


class EmbeddingTable:
    def __init__(self, n_rows: int, dim: int, name: str = "", normalize: bool = True, trainable: bool = True,
                 device="cuda", values=None, seed=None, grad_copies: int = 1):
        self.n_rows, self.dim, self.name = int(n_rows), int(dim), name
        self.normalize, self.trainable = bool(normalize), bool(trainable)
        self.stride = _lib.stride_for(self.dim)
        self.device = torch.device(device)
        self.data = placed_rows(self.n_rows, self.stride, self.device, 0.0, PLACEMENT_LOG)
        if values is not None:
            v = torch.as_tensor(np.asarray(values), dtype=torch.float32)
            assert v.shape == (self.n_rows, self.dim), (v.shape, (self.n_rows, self.dim))
            self.data[:, :self.dim] = v.to(self.device)
        elif trainable:
            self.data[:, :self.dim] = xavier_truncated_normal(self.n_rows, self.dim, self.device, seed)
        # hot tables (a few hundred relation rows hit thousands of times per step) privatise their gradient
        # scratch `grad_copies` ways so that same-address atomics do not serialise (mke_triple_score_fwd_bwd)
        self.grad_copies = int(grad_copies)
        self.slots: dict[str, torch.Tensor] = {}
        self._grad = None
        self._grad_handed = False        # the scratch's address has been given out (plans, argument structs)
        self._touched = None
        self._refcount = None
        # hub rows (include/multike_hip.h mke_hot_rows): rows that several positives of EVERY step have as head or tail; the
        # groups' flushes go to `hot_copies` private copies kept behind the table's own rows in the gradient scratch
        self.hot_slot = None          # int32 [n_rows]: index among the hub rows, or -1
        self.n_hot, self.hot_copies = 0, 1

    @property
    def refcount(self) -> torch.Tensor:
        """int32 [n_rows] zero-invariant scratch of the exclusive-row fast path (mke_count_entity_refs)."""
        if self._refcount is None:
            self._refcount = torch.zeros(self.n_rows, dtype=torch.int32, device=self.device)
        return self._refcount

    # -- scratch (shared by every optimizer of this table; steps are serial on the stream) --
    @property
    def grad(self) -> torch.Tensor:
        """[n_rows][stride] zero-invariant gradient scratch ([copies][n_rows][stride] when privatised as a whole)."""
        self._grad_handed = True         # from here on its address may sit in a plan: set_hot_rows refuses to re-allocate it
        if self._grad is None:
            if self.grad_copies == 1:
                self._grad_full = placed_rows(self.n_rows + self.hot_copies * self.n_hot, self.stride, self.device, 0.0, PLACEMENT_LOG,
                                              [self.data] + [v for v in self.slots.values() if torch.is_tensor(v)][:1])
                self._grad = self._grad_full[:self.n_rows]       # same storage: the hub rows' copies sit behind it
            else:
                self._grad = self._grad_full = torch.zeros((self.grad_copies,) + tuple(self.data.shape), dtype=torch.float32, device=self.device)
        return self._grad

    def set_hot_rows(self, rows, copies: int = 16):
        """Declare the hub rows of this table (ids, at most a few thousand): their gradient flushes are privatised `copies`
        ways (mke_triple_score_fwd_bwd_xch / mke_update_table.hot).  Must be called before the scratch is first used, or
        while it is all zero (between steps): the scratch is re-allocated with the copy rows behind the table's own."""
        rows = np.unique(np.asarray(rows, dtype=np.int64))
        if self.grad_copies != 1:
            raise _lib.MultiKEHipError("hub rows and a wholly privatised gradient scratch exclude each other")
        if len(rows) and (rows[0] < 0 or rows[-1] >= self.n_rows):
            raise _lib.MultiKEHipError("hub row id outside the table")
        copies = max(1, int(copies))
        if (self.n_rows + copies * len(rows)) * self.stride >= 2 ** 30:
            return                   # would leave the kernels' 32-bit row offsets: no declaration, the scratch stays as it is
        if self._grad is not None and self._grad_handed:
            # a plan / argument struct may hold the old scratch's address (round-5 advice): the declaration must come first
            raise _lib.MultiKEHipError("set_hot_rows after the gradient scratch was handed out: declare hub rows before building "
                                       "runners / trainers on the table")
        slot = np.full(self.n_rows, -1, dtype=np.int32)
        slot[rows] = np.arange(len(rows), dtype=np.int32)
        self.hot_slot = torch.as_tensor(slot, device=self.device)
        self.n_hot, self.hot_copies = int(len(rows)), copies
        self._grad = None            # re-created (all zero) with the copy rows on next use

    def hot_struct(self):
        """mke_hot_rows of this table (zeroed when it has none); touching `grad` first makes sure the scratch has the rows."""
        h = _lib.HotRowsStruct()
        if self.n_hot:
            _ = self.grad
            h.slot, h.n_hot, h.copies, h.row0 = _lib.ptr(self.hot_slot, torch.int32, "hot_slot"), self.n_hot, self.hot_copies, self.n_rows
        return h

    @property
    def touched(self) -> torch.Tensor:
        if self._touched is None:
            self._touched = torch.zeros(self.n_rows, dtype=torch.int32, device=self.device)
        return self._touched

    def slot(self, optimizer_name: str) -> torch.Tensor:
        """Adagrad accumulator of one optimizer (created on first use, filled with 0.1)."""
        s = self.slots.get(optimizer_name)
        if s is None:
            s = placed_rows(self.n_rows, self.stride, self.device, ADAGRAD_INIT_ACC, PLACEMENT_LOG, [self.data, self._grad_full if self._grad is not None else None])
            self.slots[optimizer_name] = s
        return s

    def dense_slots(self, optimizer_name: str):
        """(slot1, slot2, step) of an Adam / Adadelta optimizer instance: two zero-initialised arrays (m, v / accum,
        accum_update) and the number of the update about to be applied (1, 2, ...), advanced by this call."""
        key = optimizer_name + "/dense"
        st = self.slots.get(key)
        if st is None:
            st = [torch.zeros_like(self.data), torch.zeros_like(self.data), 0]
            self.slots[key] = st
        st[2] += 1
        return st[0], st[1], st[2]

    # -- read paths --
    def lookup(self, idx: torch.Tensor | None = None) -> torch.Tensor:
        """embedding_lookup on the normalised view -> dense [n, dim] float32 (HIP gather kernel)."""
        n = self.n_rows if idx is None else idx.numel()
        out = torch.empty(n, self.dim, dtype=torch.float32, device=self.device)
        if idx is not None and idx.dtype != torch.int32:
            idx = idx.to(torch.int32)
        _lib.gather_rows(self.data, self.normalize, self.dim, idx, out)
        return out

    def eval(self, session=None) -> np.ndarray:
        """`tensor.eval(session=...)` of the reference (code/MultiKE_model.py:280-285): the normalised view."""
        return self.lookup(None).cpu().numpy()

    def raw(self) -> torch.Tensor:
        return self.data[:, :self.dim]


def xavier_truncated_normal(n, d, device, seed=None) -> torch.Tensor:
    """TF1 xavier_initializer(uniform=False) (code/base/initializers.py:24-25): truncated
    normal (resample outside 2 sigma), sigma = sqrt(1.3 * 2 / (n + d))."""
    g = torch.Generator(device="cpu")
    if seed is not None:
        g.manual_seed(int(seed))
    x = torch.empty(n, d, dtype=torch.float32)
    torch.nn.init.trunc_normal_(x, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=g)
    return (x * float(np.sqrt(2.6 / (n + d)))).to(device)


_OPT = {"Adagrad": _lib.OPT_ADAGRAD, "SGD": _lib.OPT_SGD}


class StepEngine:
    """Enqueues the kernels of one training step.  One instance per model (per HIP stream)."""

    def __init__(self, device="cuda", loss_ring: int = 256, tuning: dict = None, deterministic: bool = None):
        """tuning: performance knobs of THIS engine's launches (names of `_lib.TUNING_FIELDS`, e.g. {"score_splits": 2}); every
        knob not named follows the process default (`mke_set_option`).  deterministic: the fixed-order gradient reduction for this
        engine (None: the process default, `mke_set_option("deterministic")`).  Two engines in one process may differ."""
        self.tuning = _lib.tuning(**tuning) if tuning else None
        self.deterministic = deterministic
        self.device = torch.device(device)
        self.tag = 0
        self.loss_ring = torch.zeros(loss_ring, _lib.LOSS_PARTIALS, dtype=torch.float64, device=self.device)
        self._ring_pos = 0

    def _next(self):
        self.tag += 1
        if self.tag >= 2 ** 31 - 1:
            raise _lib.MultiKEHipError("step tag overflow")
        slot = self.loss_ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % self.loss_ring.shape[0]
        return self.tag, slot

    def _apply(self, table: EmbeddingTable, opt_name: str, optimizer: str, lr: float, tag: int):
        if optimizer in _lib.DENSE_OPTS:      # Adam / Adadelta: every row moves, touched or not
            s1, s2, step = table.dense_slots(opt_name)
            _lib.rows_update_dense(table.data, s1, s2, table.grad, table.dim, table.normalize,
                                   _lib.optimizer_struct(optimizer, lr, step))
            return
        if optimizer not in _OPT:
            raise _lib.MultiKEHipError(f"optimizer {optimizer!r} not supported (Adagrad, SGD, Adam, Adadelta)")
        acc = table.slot(opt_name) if optimizer == "Adagrad" else None
        _lib.rows_update(table.data, acc, table.grad, table.touched, tag, table.dim, table.normalize, _OPT[optimizer], lr)

    def relation_step(self, ent: EmbeddingTable, rel: EmbeddingTable, opt_name: str, pos, neg=None, neg_per_pos=0,
                      lr=0.001, pos_w=None, neg_w=None, scale=1.0, optimizer="Adagrad", update=True,
                      exclusive_rows=True) -> torch.Tensor:
        """loss + optimizer of a relation-view style graph (a1/a2/a3).  Returns the loss partials (a view into the
        ring; `.sum()` is the loss) without synchronising.  With grouped negatives the exclusive-row fast path is
        used (rows referenced once in the step are updated by the scoring quarter-wave itself)."""
        det = self.deterministic if self.deterministic is not None else _lib.get_option("deterministic")
        if update and optimizer in _OPT and det:
            return self._relation_step_deterministic(ent, rel, opt_name, pos, neg, neg_per_pos, lr, pos_w, neg_w, scale,
                                                     optimizer, exclusive_rows)
        tag, lp = self._next()
        if update and exclusive_rows and neg is not None and neg_per_pos > 0 and optimizer in _OPT:
            _lib.count_entity_refs(pos[0], pos[2], neg[0], neg[2], neg_per_pos, ent.refcount)
            acc = ent.slot(opt_name) if optimizer == "Adagrad" else None
            _lib.triple_score_fwd_bwd_x(ent.data, ent.normalize, rel.data, rel.normalize, ent.dim, pos, pos_w, neg, neg_w,
                                        neg_per_pos, scale, ent.grad, rel.grad, ent.touched, rel.touched, tag, ent.refcount,
                                        acc, _OPT[optimizer], lr, lp, tuning=self.tuning)
            _lib.rows_update_multi([(rel.data, rel.slot(opt_name) if optimizer == "Adagrad" else None, rel.grad, rel.touched,
                                     rel.normalize),
                                    (ent.data, acc, ent.grad, ent.touched, ent.normalize, ent.refcount)], tag, ent.stride,
                                   ent.dim, _OPT[optimizer], lr, tuning=self.tuning)
            return lp
        _lib.triple_score_fwd_bwd(ent.data, ent.normalize, rel.data, rel.normalize, ent.dim, pos, pos_w, neg, neg_w,
                                  neg_per_pos, scale, ent.grad if update else None, rel.grad if update else None,
                                  ent.touched, rel.touched, tag, lp)
        if update:
            self._apply(ent, opt_name, optimizer, lr, tag)
            self._apply(rel, opt_name, optimizer, lr, tag)
        return lp

    def _relation_step_deterministic(self, ent, rel, opt_name, pos, neg, neg_per_pos, lr, pos_w, neg_w, scale, optimizer,
                                     exclusive_rows):
        """`mke_set_option("deterministic", 1)`: every gradient-row contribution is stored into a slot of its own
        (mke_triple_score_fwd_bwd_det), a stable sort of the slot keys brings a row's contributions together in slot
        order, mke_stage_reduce sums them front to back: bit-identical results from run to run, and hub rows whose
        gradient is a cancelling sum of hundreds of terms are summed in ONE order (the atomic path's order changes from
        run to run).  Rows referenced once are still finished in place (one contribution: no order to fix)."""
        tag, lp = self._next()
        n_pos = pos[0].numel()
        n_neg = 0 if neg is None else neg[0].numel()
        slots = 3 * (n_pos * (neg_per_pos + 1) if neg_per_pos > 0 else n_pos + n_neg)
        st = getattr(self, "_stage", None)
        if st is None or st[0].shape[0] < slots or st[0].shape[1] != ent.stride:
            st = self._stage = (torch.empty(max(slots, 1), ent.stride, dtype=torch.float32, device=self.device),
                                torch.empty(max(slots, 1), dtype=torch.int64, device=self.device))
        rows, keys = st[0][:max(slots, 1)], st[1][:max(slots, 1)]
        keys.fill_(0x7F7F7F7F7F7F7F7F)
        adagrad = optimizer == "Adagrad"
        acc = ent.slot(opt_name) if adagrad else None
        refc = None
        if exclusive_rows and neg is not None and neg_per_pos > 0:
            _lib.count_entity_refs(pos[0], pos[2], neg[0], neg[2], neg_per_pos, ent.refcount)
            refc = ent.refcount
        grel = rel.grad if rel.grad.dim() == 2 else rel.grad[0]
        _lib.triple_score_fwd_bwd_det(ent.data, ent.normalize, rel.data, rel.normalize, ent.dim, pos, pos_w, neg, neg_w,
                                      neg_per_pos, scale, ent.grad, grel, ent.touched, rel.touched, tag, refc, acc,
                                      _OPT[optimizer], lr, rows, keys, lp)
        sorted_keys, order = torch.sort(keys, stable=True)
        _lib.stage_reduce(rows, sorted_keys, order, ent.grad, grel, ent.touched, rel.touched, tag)
        _lib.rows_update_multi([(rel.data, rel.slot(opt_name) if adagrad else None, rel.grad, rel.touched, rel.normalize),
                                (ent.data, acc, ent.grad, ent.touched, ent.normalize, ent.refcount if refc is not None else None)]
                               if refc is not None else
                               [(rel.data, rel.slot(opt_name) if adagrad else None, rel.grad, rel.touched, rel.normalize),
                                (ent.data, acc, ent.grad, ent.touched, ent.normalize)], tag, ent.stride, ent.dim, _OPT[optimizer], lr)
        return lp

    def alignment_step(self, terms, opt_name: str, lr: float, optimizer="Adagrad") -> torch.Tensor:
        """terms: list of (table_a, idx_a, table_b, idx_b, weight).  One optimizer step over the sum of the
        terms (code/MultiKE_model.py:229-239).  Returns the summed loss as a device scalar."""
        tag = None
        total = None
        tables = []
        for (ta, ia, tb, ib, w) in terms:
            t, lp = self._next()
            if tag is None:
                tag = t
            _lib.align_fwd_bwd(ta.data, ta.normalize, tb.data, tb.normalize, ta.dim, ia, ib, float(w),
                               ta.grad if ta.trainable else None, ta.touched if ta.trainable else None,
                               tb.grad if tb.trainable else None, tb.touched if tb.trainable else None, tag, lp)
            s = lp.sum()
            total = s if total is None else total + s
            for tb_ in (ta, tb):
                if tb_.trainable and all(tb_ is not x for x in tables):
                    tables.append(tb_)
        for tb_ in tables:
            self._apply(tb_, opt_name, optimizer, lr, tag)
        return total
