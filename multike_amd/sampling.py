"""Device-resident relation-triple batches and negative sampling.

Host-side mirror of what the reference's producer processes do per step
(code/base/batch.py:22-54,86-116, driven from code/MultiKE_model.py:291-303), restructured for one GPU:
the positive lists of both KGs live in HBM as int32 columns, an epoch is one device permutation per KG,
a step's positives are a contiguous slice [KG1 part | KG2 part] (proportional split,
code/base/batch.py:36-37), and negatives come from `mke_neg_sample` (Philox stream, see
oracle/sampler_oracle.py for the specification the kernel is tested against bit for bit).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib


def kg_batch_split(n1: int, n2: int, batch_size: int):
    """code/base/batch.py:36-37."""
    b1 = int(n1 / (n1 + n2) * batch_size)
    return b1, batch_size - b1


class KnownTripleSet:
    """Open-addressing hash set of packed (h, r, t) keys in HBM — the `all_triples_set` membership test of
    code/base/batch.py:109."""

    # packed key = h<<38 | t<<12 | r (26 + 26 + 12 bits).  The largest entity id is excluded so that no triple packs to the
    # all-ones word, which is the EMPTY marker of the open-addressing table (MKE_EMPTY_KEY)
    MAX_ENT, MAX_REL = (1 << 26) - 1, 1 << 12

    def __init__(self, h: torch.Tensor, r: torch.Tensor, t: torch.Tensor):
        n = h.numel()
        if n and (int(h.max()) >= self.MAX_ENT or int(t.max()) >= self.MAX_ENT or int(r.max()) >= self.MAX_REL):
            raise _lib.MultiKEHipError("entity id >= 2^26 - 1 or relation id >= 2^12 does not fit the packed triple key")
        cap = 1
        while cap < 2 * n + 2:
            cap *= 2
        self.keys = torch.full((cap,), -1, dtype=torch.int64, device=h.device)
        self.add(h, r, t)

    def add(self, h, r, t):
        _lib.tripleset_build(h.contiguous(), r.contiguous(), t.contiguous(), self.keys)

    def contains(self, h, r, t) -> torch.Tensor:
        out = torch.empty(h.numel(), dtype=torch.uint8, device=h.device)
        _lib.tripleset_query(h.contiguous(), r.contiguous(), t.contiguous(), self.keys, out)
        return out.bool()


class KGSide:
    """Everything the sampler needs about one KG: candidate population, known triples, neighbour table."""

    def __init__(self, entities, known: KnownTripleSet | None, device="cuda"):
        ents = np.asarray(entities, dtype=np.int64)
        self.n_ent = len(ents)
        if self.n_ent and np.array_equal(ents, np.arange(ents[0], ents[0] + self.n_ent)):
            self.ent_lo, self.ent_list = int(ents[0]), None  # contiguous id range (code/base/read.py:75-84)
        else:
            self.ent_lo, self.ent_list = 0, torch.as_tensor(ents, dtype=torch.int32, device=device)
        self.known = known
        self.cand_table = None  # [n_ent_total, k] int32: truncated-sampling neighbours (code/base/batch.py:119-150)
        self.cand_valid = None  # [n_ent_total] uint8: entity has a neighbour list (dict membership)

    def set_neighbours(self, cand_table: torch.Tensor | None, cand_valid: torch.Tensor | None):
        self.cand_table, self.cand_valid = cand_table, cand_valid

    def fill(self, st: "_lib.KGSideStruct"):
        """Write this context into an mke_kg_side (the tensors must outlive the struct's use)."""
        st.ent_list = _lib.ptr(self.ent_list, torch.int32, "ent_list")
        st.ent_lo, st.n_ent = self.ent_lo, self.n_ent
        st.cand_table = _lib.ptr(self.cand_table, torch.int32, "cand_table")
        st.cand_valid = _lib.ptr(self.cand_valid, torch.uint8, "cand_valid")
        st.cand_k = 0 if self.cand_table is None else self.cand_table.shape[1]
        st.known_keys = None if self.known is None else _lib.ptr(self.known.keys, torch.int64, "known_keys")
        st.known_capacity = 0 if self.known is None else self.known.keys.numel()


def side_array(side0: KGSide, side1: KGSide | None = None):
    arr = (_lib.KGSideStruct * 2)()
    side0.fill(arr[0])
    (side1 or side0).fill(arr[1])
    return arr


def sample_negatives(pos, side: KGSide, neg_per_pos: int, seed=(0, 0), stream_id=0, pos_offset=0, max_try=10,
                     out=None, side1: KGSide | None = None, pos_kg: torch.Tensor | None = None):
    """generate_neg_triples_fast on device.  With `pos_kg` (uint8 per positive) and `side1`, one call covers
    positives of both KGs (KG k uses RNG stream `stream_id + k`)."""
    ph, pr, pt = pos
    n = ph.numel() * neg_per_pos
    if out is None:
        out = tuple(torch.empty(n, dtype=torch.int32, device=ph.device) for _ in range(3))
    if n:
        _lib.neg_sample(pos, pos_offset, pos_kg, side_array(side, side1), neg_per_pos, max_try, seed, stream_id, out)
    return out


def int_triples(triples) -> np.ndarray:
    """(h, r, t) id triples — an array, or the reference's list / set of tuples — as an int32 array [n, 3].  Tuples go through one
    flat iterator (np.asarray of 700K tuples builds 700K temporary rows: 0.23 s against 0.07)."""
    if isinstance(triples, np.ndarray) or hasattr(triples, "__array__"):
        return np.asarray(triples, dtype=np.int32).reshape(-1, 3)
    if hasattr(triples, "cols"):
        return np.asarray(triples.cols, dtype=np.int32).reshape(-1, 3)
    import itertools
    n = len(triples)
    if n == 0:
        return np.zeros((0, 3), dtype=np.int32)
    first = next(iter(triples))
    if len(first) != 3:
        return np.asarray(list(triples), dtype=np.int32).reshape(-1, 3)
    return np.fromiter(itertools.chain.from_iterable(triples), dtype=np.int64, count=3 * n).reshape(n, 3).astype(np.int32)


class RelationBatcher:
    """Epoch/step bookkeeping of the relation view on device.

    triples1/triples2: array-like [n, 3] (h, r, t) of each KG's `local_relation_triples_list`.
    """

    def __init__(self, triples1, triples2, side1: KGSide, side2: KGSide, batch_size: int, neg_per_pos: int,
                 device="cuda", seed: int = 0):
        self.device = torch.device(device)
        self.t1 = torch.as_tensor(int_triples(triples1), device=self.device)
        self.t2 = torch.as_tensor(int_triples(triples2), device=self.device)
        self.side1, self.side2 = side1, side2
        self.batch_size, self.neg_per_pos = int(batch_size), int(neg_per_pos)
        self.n1, self.n2 = self.t1.shape[0], self.t2.shape[0]
        self.b1, self.b2 = kg_batch_split(self.n1, self.n2, self.batch_size)
        # the reference's step count: ceil((n1+n2)/batch_size) (code/MultiKE_CSL.py:38-40)
        self.steps = int(math.ceil((self.n1 + self.n2) / self.batch_size))
        self.seed = int(seed)
        self.epoch = 0
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self.seed)
        self._layout()
        self._materialise(None, None)

    def _layout(self):
        """Per-step [lo, hi) of each KG's slice (code/base/batch.py:45-54: short or empty at the end)."""
        s = np.arange(self.steps, dtype=np.int64)
        self.lo1, self.hi1 = np.minimum(s * self.b1, self.n1), np.minimum((s + 1) * self.b1, self.n1)
        self.lo2, self.hi2 = np.minimum(s * self.b2, self.n2), np.minimum((s + 1) * self.b2, self.n2)
        c1, c2 = self.hi1 - self.lo1, self.hi2 - self.lo2
        self.off = np.zeros(self.steps + 1, dtype=np.int64)
        self.off[1:] = np.cumsum(c1 + c2)
        self.cnt1 = c1
        # gather map: epoch position -> (kg, index into that KG's (shuffled) list)
        src = np.empty(int(self.off[-1]), dtype=np.int64)
        for i in range(self.steps):
            o = self.off[i]
            src[o:o + c1[i]] = np.arange(self.lo1[i], self.hi1[i])
            src[o + c1[i]:self.off[i + 1]] = self.n1 + np.arange(self.lo2[i], self.hi2[i])
        self._src = torch.as_tensor(src, device=self.device)
        kg = np.zeros(int(self.off[-1]), dtype=np.uint8)
        for i in range(self.steps):
            kg[self.off[i] + c1[i]:self.off[i + 1]] = 1
        self.pos_kg = torch.as_tensor(kg, device=self.device)  # fixed across epochs: only the contents shuffle

    def _materialise(self, perm1, perm2):
        if self.device.type == "cuda":
            # the lists are replaced below; if this epoch runs on another stream than the one they were allocated on
            # (drivers run the relation group on a side stream), the allocator must not hand the old ones out again
            # before this stream has read them
            cur = torch.cuda.current_stream(self.device)
            for t in (self.t1, self.t2):
                t.record_stream(cur)
        t1 = self.t1 if perm1 is None else self.t1[perm1]
        t2 = self.t2 if perm2 is None else self.t2[perm2]
        allt = torch.cat([t1, t2], 0)[self._src]  # epoch order, step-contiguous
        if getattr(self, "pos_h", None) is None:
            self.pos_h = allt[:, 0].contiguous()
            self.pos_r = allt[:, 1].contiguous()
            self.pos_t = allt[:, 2].contiguous()
        else:  # persistent epoch buffers: addresses stay valid across epochs (native plans, side streams)
            self.pos_h.copy_(allt[:, 0])
            self.pos_r.copy_(allt[:, 1])
            self.pos_t.copy_(allt[:, 2])
        self.t1, self.t2 = t1, t2

    # -- next-epoch staging: lets a caller permute (and sample) epoch e+1 while epoch e is still being trained ------
    def stage_next_epoch(self):
        """Draw the next epoch's permutation into the ALTERNATE epoch buffers (the current ones are untouched).
        Returns the staged (pos_h, pos_r, pos_t); `commit_staged()` makes them current."""
        p1 = torch.randperm(self.n1, generator=self._gen, device=self.device)
        p2 = torch.randperm(self.n2, generator=self._gen, device=self.device)
        t1, t2 = self.t1[p1], self.t2[p2]
        allt = torch.cat([t1, t2], 0)[self._src]
        if getattr(self, "_alt", None) is None:
            self._alt = tuple(allt[:, k].contiguous() for k in range(3))
        else:
            for k in range(3):
                self._alt[k].copy_(allt[:, k])
        self._staged_lists = (t1, t2)
        return self._alt

    def commit_staged(self):
        """The staged epoch becomes the current one (buffer swap; same effect as `shuffle()`)."""
        cur = (self.pos_h, self.pos_r, self.pos_t)
        self.pos_h, self.pos_r, self.pos_t = self._alt
        self._alt = cur
        self.t1, self.t2 = self._staged_lists
        self._staged_lists = None
        self.epoch += 1

    def shuffle(self):
        """random.shuffle of both positive lists after an epoch (code/MultiKE_model.py:314-315)."""
        p1 = torch.randperm(self.n1, generator=self._gen, device=self.device)
        p2 = torch.randperm(self.n2, generator=self._gen, device=self.device)
        self._materialise(p1, p2)
        self.epoch += 1

    def positives(self, step: int):
        lo, hi = int(self.off[step]), int(self.off[step + 1])
        return self.pos_h[lo:hi], self.pos_r[lo:hi], self.pos_t[lo:hi]

    @property
    def rng_seed(self):
        return (self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)

    @property
    def rng_stream(self):
        return (self.epoch * 2) & 0xFFFFFFFF

    def batch(self, step: int, out=None):
        """(pos, neg) of one step; negatives grouped neg_per_pos per positive in positive order."""
        lo, hi = int(self.off[step]), int(self.off[step + 1])
        N = self.neg_per_pos
        if out is None:
            out = tuple(torch.empty((hi - lo) * N, dtype=torch.int32, device=self.device) for _ in range(3))
        pos = (self.pos_h[lo:hi], self.pos_r[lo:hi], self.pos_t[lo:hi])
        if hi > lo:
            sample_negatives(pos, self.side1, N, seed=self.rng_seed, stream_id=self.rng_stream, pos_offset=lo, out=out,
                             side1=self.side2, pos_kg=self.pos_kg[lo:hi])
        return pos, out
