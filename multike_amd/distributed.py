"""Entity-row sharded relation-view training over `torch.distributed` (RCCL over xGMI on MI355X).

The reference has no multi-device code; this is new design.  One process per GPU.
  * entity table + its Adagrad slot + gradient scratch are row-sharded by  id % world  (local row = id // world);
  * the relation table (|R| x dim, ~165 KB) is replicated;
  * a global step is `world` x batch_size positives in the reference's epoch order; rank r scores the r-th
    contiguous slice with its own negatives (the Philox stream is indexed by the GLOBAL epoch position, so the
    negatives of a positive do not depend on the world size);
  * exchange per step:  all-to-all(v) of the needed remote row ids  ->  all-to-all(v) of the raw rows  ->  local fused
    triple step on the compact [U, stride] row set  ->  all-to-all(v) of the gradient rows back to their owners
    (already pre-reduced per sender by the scatter kernel), owner adds them and runs the row update ONCE per row
    per step (dense-Adagrad-equivalent, SURVEY.md §8e "semantics note");  all-reduce of the replicated relation
    gradient, then every rank applies the identical relation update.
xGMI is point-to-point, so the row exchange is an all-to-all (every link busy), not a ring.

The compute steps go through a small backend object: `HipBackend` (the product, HIP kernels) — tests inject a
CPU backend built on the oracle to exercise this exchange logic under `gloo` with world_size 2.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .sampling import KGSide, KnownTripleSet, RelationBatcher, side_array
from .tables import ADAGRAD_INIT_ACC


class HipBackend:
    """Product backend: every compute step is a HIP kernel of libmultike_hip.so."""

    device_type = "cuda"

    def make_known(self, h, r, t):
        return KnownTripleSet(h, r, t)

    def sample(self, pos, pos_offset, pos_kg, side1, side2, neg_per_pos, seed, stream_id, out):
        _lib.neg_sample(pos, pos_offset, pos_kg, side_array(side1, side2), neg_per_pos, 10, seed, stream_id, out)

    def score(self, ent, ent_norm, rel, rel_norm, dim, pos, neg, neg_per_pos, grad_ent, grad_rel, touched_ent,
              touched_rel, tag, loss_partials):
        _lib.triple_score_fwd_bwd(ent, ent_norm, rel, rel_norm, dim, pos, None, neg, None, neg_per_pos, 1.0, grad_ent,
                                  grad_rel, touched_ent, touched_rel, tag, loss_partials)

    def update(self, table, acc, grad, touched, tag, dim, normalize, lr):
        _lib.rows_update(table, acc, grad, touched, tag, dim, normalize, _lib.OPT_ADAGRAD, lr)


class ShardedRelationTrainer:
    def __init__(self, kgs, ent0: np.ndarray, rel0: np.ndarray, batch_size: int, neg_per_pos: int, rank: int,
                 world: int, seed: int = 0, lr: float = 0.001, backend=None, device=None, dtype=torch.float32):
        self.backend = backend or HipBackend()
        self.device = torch.device(device or ("cuda" if self.backend.device_type == "cuda" else "cpu"))
        self.rank, self.world, self.lr = rank, world, lr
        self.dim = ent0.shape[1]
        self.stride = _lib.stride_for(self.dim)
        self.N = neg_per_pos
        self.n_ent = ent0.shape[0]
        dev, st = self.device, self.stride
        # --- row-sharded entity state -------------------------------------------------------------
        mine = np.arange(rank, self.n_ent, world)
        self.n_local = len(mine)
        self.ent = torch.zeros(self.n_local, st, dtype=dtype, device=dev)
        self.ent[:, :self.dim] = torch.as_tensor(ent0[mine], dtype=dtype, device=dev)
        self.ent_acc = torch.full_like(self.ent, ADAGRAD_INIT_ACC)
        self.ent_grad = torch.zeros_like(self.ent)
        self.ent_touched = torch.zeros(self.n_local, dtype=torch.int32, device=dev)
        # --- replicated relation state ------------------------------------------------------------
        self.rel = torch.zeros(rel0.shape[0], st, dtype=dtype, device=dev)
        self.rel[:, :self.dim] = torch.as_tensor(rel0, dtype=dtype, device=dev)
        self.rel_acc = torch.full_like(self.rel, ADAGRAD_INIT_ACC)
        self.rel_grad = torch.zeros_like(self.rel)
        self.rel_touched = torch.zeros(rel0.shape[0], dtype=torch.int32, device=dev)
        # --- global epoch order (identical on every rank: same seed) ---------------------------------
        sides = []
        for k in (0, 1):
            t = torch.as_tensor(np.asarray(kgs.triples[k], dtype=np.int32), device=dev)
            known = self.backend.make_known(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())
            sides.append(KGSide(kgs.entities(k), known, device=dev))
        self.bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], batch_size * world, neg_per_pos,
                                   device=dev, seed=seed)
        self.steps = self.bat.steps
        self.tag = 0
        self.loss_partials = torch.zeros(_lib.LOSS_PARTIALS, dtype=torch.float64, device=dev)
        self.loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
        self._id_map = torch.zeros(self.n_ent, dtype=torch.int32, device=dev)  # global id -> compact row (scratch)
        self.last_stats = {}

    # ------------------------------------------------------------------------------------------------
    def my_slice(self, s: int):
        """[a, b) epoch positions of this rank's share of global step s (contiguous, ceil split)."""
        lo, hi = int(self.bat.off[s]), int(self.bat.off[s + 1])
        per = int(math.ceil((hi - lo) / self.world)) if hi > lo else 0
        a = min(hi, lo + self.rank * per)
        return a, min(hi, a + per)

    def global_scored(self, i: int) -> int:
        s = i % self.steps
        return int(self.bat.off[s + 1] - self.bat.off[s]) * (1 + self.N)

    def _all_to_all(self, send: torch.Tensor, send_counts, recv_counts):
        out = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts))
        return out

    def step(self, i: int):
        s = i % self.steps
        if s == 0 and i > 0:
            self.bat.shuffle()
        b, N, G, dev = self.bat, self.N, self.world, self.device
        a, e = self.my_slice(s)
        pos = (b.pos_h[a:e], b.pos_r[a:e], b.pos_t[a:e])
        n_pos = e - a
        neg = tuple(torch.empty(n_pos * N, dtype=torch.int32, device=dev) for _ in range(3))
        if n_pos and N:
            self.backend.sample(pos, a, b.pos_kg[a:e], b.side1, b.side2, N, b.rng_seed, b.rng_stream, neg)
        # ---- which entity rows does this rank need, and who owns them -------------------------------
        ids = torch.cat([pos[0], pos[2], neg[0], neg[2]]).long()
        uniq = torch.unique(ids)                                   # sorted global ids
        owner = uniq % G
        order = torch.argsort(owner, stable=True)                  # compact order = grouped by owner
        req = uniq[order]
        send_counts = torch.bincount(owner, minlength=G)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts)
        sc, rc = send_counts.tolist(), recv_counts.tolist()         # host sync: sizes of the variable exchanges
        U = int(req.numel())
        self._id_map[req] = torch.arange(U, dtype=torch.int32, device=dev)
        # ---- ids out, raw rows back -------------------------------------------------------------------
        want = self._all_to_all((req // G).to(torch.int32), sc, rc).long()   # local rows other ranks ask of me
        rows = self._all_to_all(self.ent.index_select(0, want), rc, sc)       # [U, stride] in compact order
        # ---- local fused step on the compact row set ---------------------------------------------------
        cpos = (self._id_map[pos[0].long()], pos[1], self._id_map[pos[2].long()])
        cneg = (self._id_map[neg[0].long()], neg[1], self._id_map[neg[2].long()])
        cgrad = torch.zeros_like(rows)
        ctouched = torch.zeros(max(U, 1), dtype=torch.int32, device=dev)
        self.tag += 1
        tag = self.tag
        self.backend.score(rows, True, self.rel, True, self.dim, cpos, cneg, N, cgrad, self.rel_grad, ctouched,
                           self.rel_touched, tag, self.loss_partials)
        self.loss_sum += self.loss_partials.sum()
        # ---- gradient rows back to their owners; owner reduces and updates once per row ----------------
        got = self._all_to_all(cgrad, sc, rc)
        self.ent_grad.index_add_(0, want, got)
        self.ent_touched[want] = tag
        self.backend.update(self.ent, self.ent_acc, self.ent_grad, self.ent_touched, tag, self.dim, True, self.lr)
        # ---- replicated relation table: all-reduce the (tiny) dense gradient, identical update everywhere
        dist.all_reduce(self.rel_grad)
        self.rel_touched.fill_(tag)
        self.backend.update(self.rel, self.rel_acc, self.rel_grad, self.rel_touched, tag, self.dim, True, self.lr)
        self.last_stats = {"unique_rows": U, "remote_rows": U - sc[self.rank], "positives": n_pos}

    # ------------------------------------------------------------------------------------------------
    def gather_entity_table(self) -> torch.Tensor:
        """Reassemble the full [n_ent, dim] raw table on every rank (tests / checkpoint)."""
        pad = int(math.ceil(self.n_ent / self.world))
        mine = torch.zeros(pad, self.stride, dtype=self.ent.dtype, device=self.device)
        mine[:self.n_local] = self.ent
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        full = torch.zeros(self.n_ent, self.dim, dtype=self.ent.dtype, device=self.device)
        for r in range(self.world):
            n = len(range(r, self.n_ent, self.world))
            full[r::self.world] = parts[r][:n, :self.dim]
        return full

    def epoch_loss(self) -> float:
        t = self.loss_sum.clone()
        dist.all_reduce(t)
        self.loss_sum.zero_()
        return float(t)
