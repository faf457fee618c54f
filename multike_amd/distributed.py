"""Entity-row sharded relation-view training over `torch.distributed` (RCCL over xGMI on MI355X).

The reference has no multi-device code; this is new design.  One process per GPU.
  * entity table + its Adagrad slot + gradient scratch are row-sharded by  id % world  (local row = id // world);
  * the relation table (|R| x dim, ~165 KB) is replicated;
  * a global step is `world` x batch_size positives in the reference's epoch order; rank r scores the r-th
    contiguous slice with its own negatives (the Philox stream is indexed by the GLOBAL epoch position, so the
    negatives of a positive do not depend on the world size);
  * exchange per step, all FIXED-CAPACITY ([world][C] slots per rank, C sized from the batch) so that every
    collective is an equal-split all-to-all and nothing on the host ever waits for the device:
      row-set build (HIP: distinct ids grouped by owner)  ->  all-to-all of the requested local rows (ids)
      ->  owner gathers raw rows  ->  all-to-all of the rows  ->  local fused triple step on the compact
      [world*C, stride] row set  ->  all-to-all of the gradient rows back (pre-reduced per sender by the scatter
      kernel)  ->  owner scatter-adds them and runs the row update ONCE per row per step (dense-Adagrad-equivalent,
      SURVEY.md §8e "semantics note")  ->  all-reduce of the replicated relation gradient, identical relation
      update on every rank.
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
    """Product backend: every compute / bookkeeping step is a HIP kernel of libmultike_hip.so."""

    device_type = "cuda"

    def make_known(self, h, r, t):
        return KnownTripleSet(h, r, t)

    def sample(self, pos, pos_offset, pos_kg, side1, side2, neg_per_pos, seed, stream_id, out):
        _lib.neg_sample(pos, pos_offset, pos_kg, side_array(side1, side2), neg_per_pos, 10, seed, stream_id, out)

    def sample_at(self, pos, pos_index, pos_kg, side1, side2, neg_per_pos, seed, stream_id, out):
        """Negatives of positives given by explicit epoch positions (a rank's share of a whole epoch): one launch."""
        _lib.neg_sample_at(pos, pos_index, pos_kg, side_array(side1, side2), neg_per_pos, 10, seed, stream_id, out)

    def rowset_build(self, streams, flags, counts, req, id_map, overflow, n_ranks, capacity):
        _lib.rowset_build(streams, flags, counts, req, id_map, overflow, n_ranks, capacity)

    def rowset_remap(self, streams, outs, id_map, flags, reset_req=None, reset_counts=None, want=None, slot_of=None,
                     n_ranks=0, capacity=0):
        _lib.rowset_remap(streams, outs, id_map, flags, reset_req, reset_counts, want, slot_of, n_ranks, capacity)

    def gather_padded(self, table, idx, out, zero_rows=None):
        _lib.rows_gather_padded(table, idx, out, zero_rows)

    def scatter_add(self, idx, rows, dim, grad, touched, tag, reset_req=None, reset_counts=None):
        _lib.rows_scatter_add(idx, rows, dim, grad, touched, tag, reset_req, reset_counts)

    def score(self, ent, ent_norm, rel, rel_norm, dim, pos, neg, neg_per_pos, grad_ent, grad_rel, touched_ent,
              touched_rel, tag, loss_partials):
        _lib.triple_score_fwd_bwd(ent, ent_norm, rel, rel_norm, dim, pos, None, neg, None, neg_per_pos, 1.0, grad_ent,
                                  grad_rel, touched_ent, touched_rel, tag, loss_partials)

    def update(self, table, acc, grad, touched, tag, dim, normalize, lr):
        _lib.rows_update(table, acc, grad, touched, tag, dim, normalize, _lib.OPT_ADAGRAD, lr)

    def update_pair(self, t0, t1, tag, dim, lr):
        """One launch over two tables; each t = (table, acc, grad, touched, normalize); touched may be None."""
        _lib.rows_update_multi([t0, t1], tag, t0[0].shape[1], dim, _lib.OPT_ADAGRAD, lr)

    def reduce_update(self, rel, shard, tag, dim, lr):
        """One launch: the replicated table rel = (table, acc, grad, None, normalize), every row; and the owner's
        reduce-and-update of its shard = dict(table, acc, normalize, src_rows, slot_of, n_ranks, capacity): each row
        some rank sent a gradient for sums its contributions in rank order and is updated once."""
        _lib.rows_update_multi([rel, shard], tag, rel[0].shape[1], dim, _lib.OPT_ADAGRAD, lr)


class TorchComm:
    """The collectives of the sharded step, on torch.distributed (RCCL over xGMI on MI355X; gloo in the CPU tests)."""

    def all_to_all_single(self, out, inp, group=None):
        dist.all_to_all_single(out, inp, group=group)

    def all_reduce(self, t, op=None):
        dist.all_reduce(t) if op is None else dist.all_reduce(t, op=op)

    def all_gather(self, parts, mine):
        dist.all_gather(parts, mine)


class HostStagedComm(TorchComm):
    """Test vehicle: the same collectives on device tensors through a CPU backend (gloo), staged over the host.  Lets two
    ranks that SHARE one GPU exercise the device kernels of the sharded step with world_size 2 -- RCCL refuses two ranks
    on one device, and a single-GPU box is all the test environment offers."""

    def all_to_all_single(self, out, inp, group=None):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)

    def all_reduce(self, t, op=None):
        c = t.cpu()
        dist.all_reduce(c) if op is None else dist.all_reduce(c, op=op)
        t.copy_(c)

    def all_gather(self, parts, mine):
        cp = [torch.empty(p.shape, dtype=p.dtype) for p in parts]
        dist.all_gather(cp, mine.cpu())
        for p, c in zip(parts, cp):
            p.copy_(c)


class ShardedRelationTrainer:
    def __init__(self, kgs, ent0: np.ndarray, rel0: np.ndarray, batch_size: int, neg_per_pos: int, rank: int,
                 world: int, seed: int = 0, lr: float = 0.001, backend=None, device=None, dtype=torch.float32,
                 lookahead: int | None = None, comm=None):
        self.backend = backend or HipBackend()
        self.comm = comm or TorchComm()
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
        self.loss_ring = torch.zeros(max(1, self.steps), _lib.LOSS_PARTIALS, dtype=torch.float64, device=dev)
        # --- fixed-capacity exchange buffers ---------------------------------------------------------
        G = world
        max_pos = int(math.ceil(batch_size * world / world))                     # positives of one rank per step
        bound = max_pos * (2 + neg_per_pos)                                      # distinct rows a rank can need
        max_local = int(math.ceil(self.n_ent / G))
        C = int(min(max_local, math.ceil(bound / G * 1.25) + 256))              # worst case: every referenced row distinct
        C = self._calibrate_capacity(C, max_local)
        self.C = C
        i32 = dict(dtype=torch.int32, device=dev)
        self._flags = torch.zeros(self.n_ent, **i32)
        self._id_map = torch.zeros(self.n_ent, **i32)                            # global id -> compact row
        self._counts = torch.zeros(G, **i32)
        self._overflow = torch.zeros(1, **i32)
        self._req = torch.full((G * C,), -1, **i32)                              # re-initialised by rowset_remap each step
        self._rows_out = torch.empty(G * C, st, dtype=dtype, device=dev)
        # compact row set + one PAD row (index G*C): ids that overflowed an owner's segment read this all-zero row and
        # add their gradient to a row nobody collects (mke_rowset_build), instead of aliasing another entity's slot
        self._rows_in_all = torch.zeros(G * C + 1, st, dtype=dtype, device=dev)
        self._rows_in = self._rows_in_all[:G * C]
        self._cgrad_all = torch.zeros(G * C + 1, st, dtype=dtype, device=dev)
        self._cgrad = self._cgrad_all[:G * C]
        self._ggot = torch.empty(G * C, st, dtype=dtype, device=dev)
        self._ctouched = torch.zeros(G * C + 1, **i32)
        self._max_rows = torch.zeros(1, **i32)                                   # largest per-owner request count seen
        # --- plan phase (sampler -> row-set build -> id exchange -> remap) is table-independent: it runs one step
        #     ahead on its own stream + communicator, double-buffered, off the critical path of the step ----------
        # plans may be enqueued `lookahead` steps ahead of their use on a side stream with their own communicator.
        # Default 0 = inline on the main stream with the single default communicator: two communicators progressing
        # concurrently on side streams is the classic NCCL/RCCL ordering hazard and could not be exercised on >1 GPU
        # in this round's environment (1-GPU boxes only); opt in with lookahead=2 or MKE_SHARD_LOOKAHEAD=2.
        import os
        self.lookahead = int(os.environ.get("MKE_SHARD_LOOKAHEAD", "0")) if lookahead is None else int(lookahead)
        nslot = self.lookahead + 1
        self._nslot = nslot
        self._want2 = [torch.empty(G * C, **i32) for _ in range(nslot)]
        # want inverted per local row (slot_of[row * G + g]); all -1 between steps: the update launch resets what it reads
        self._slot_of = [torch.full((max(1, self.n_local) * G,), -1, **i32) for _ in range(nslot)]
        self._cidx2 = [[torch.empty(max_pos * (1 if k < 2 else neg_per_pos), **i32) for k in range(4)] for _ in range(nslot)]
        # this rank's share of every step of the epoch, as epoch positions: its negatives are sampled by ONE launch per
        # epoch (a per-step sampler launch is latency-bound: 38 us for 5000 positives vs 3.5 us/step amortised)
        sl = [self.my_slice(s) for s in range(self.steps)]
        self._loc_off = np.zeros(self.steps + 1, dtype=np.int64)
        self._loc_off[1:] = np.cumsum([e - a for a, e in sl])
        idx = np.concatenate([np.arange(a, e, dtype=np.int32) for a, e in sl]) if self.steps else np.zeros(0, np.int32)
        self._epoch_idx = torch.as_tensor(idx, device=dev)
        self._neg_epoch = tuple(torch.empty(max(1, int(self._loc_off[-1]) * neg_per_pos), **i32) for _ in range(3))
        self._neg_epoch_of = -1   # epoch whose negatives the buffers hold
        self._counts_last = torch.zeros(G, **i32)
        self.keep_stats = False  # tests switch this on (costs one tiny copy per step)
        self._cuda = self.device.type == "cuda" and self.lookahead > 0   # side-stream machinery only when pipelining
        if self._cuda:
            self._plan_stream = torch.cuda.Stream(device=self.device)
            self._plan_done = [torch.cuda.Event() for _ in range(nslot)]
            self._main_done = [torch.cuda.Event() for _ in range(nslot)]
            self._main_done_valid = [False] * nslot
        self._plan_group = dist.new_group() if (dist.is_initialized() and G > 1 and self.lookahead > 0) else None
        self._planned = -1    # plans of global steps <= this index have been enqueued
        self._stepped = -1    # main steps <= this index have been enqueued
        self.score_events = None  # set to a list to collect (start, end, triples) HIP events of the score kernel

    def _calibrate_capacity(self, c_bound: int, max_local: int, probe_steps: int = 3, slack: float = 1.15) -> int:
        """Fixed-capacity buffers are what every rank moves per step, so size them from the data instead of the
        all-distinct worst case: build the row set of the first few steps (setup time, synchronising), take the
        largest per-owner count over all ranks, add slack.  An overflow later is detected on device and raised."""
        G, dev, be = self.world, self.device, self.backend
        i32 = dict(dtype=torch.int32, device=dev)
        flags, id_map = torch.zeros(self.n_ent, **i32), torch.zeros(self.n_ent, **i32)
        worst = 0
        for s in range(min(probe_steps, self.steps)):
            a, e = self.my_slice(s)
            n_pos = e - a
            if n_pos == 0:
                continue
            b = self.bat
            pos = (b.pos_h[a:e], b.pos_r[a:e], b.pos_t[a:e])
            neg = tuple(torch.empty(n_pos * self.N, **i32) for _ in range(3))
            if self.N:
                be.sample(pos, a, b.pos_kg[a:e], b.side1, b.side2, self.N, b.rng_seed, b.rng_stream, neg)
            counts, overflow = torch.zeros(G, **i32), torch.zeros(1, **i32)
            req = torch.full((G * c_bound,), -1, **i32)
            streams = [pos[0], pos[2], neg[0], neg[2]]
            be.rowset_build(streams, flags, counts, req, id_map, overflow, G, c_bound)
            be.rowset_remap(streams, [torch.empty_like(x) for x in streams], id_map, flags)
            worst = max(worst, int(counts.max()))
        t = torch.tensor([worst], dtype=torch.int64, device=dev)
        if dist.is_initialized() and G > 1:
            self.comm.all_reduce(t, op=dist.ReduceOp.MAX)
        worst = int(t)
        return int(min(max_local, c_bound, math.ceil(worst * slack) + 64)) if worst else c_bound

    # ------------------------------------------------------------------------------------------------
    def my_slice(self, s: int):
        """[a, b) epoch positions of this rank's share of global step s (contiguous, ceil split)."""
        lo, hi = int(self.bat.off[s]), int(self.bat.off[s + 1])
        per = int(math.ceil((hi - lo) / self.world)) if hi > lo else 0
        a = min(hi, lo + self.rank * per)
        return a, min(hi, a + per)

    def _step_neg(self, s: int):
        lo, hi = int(self._loc_off[s]) * self.N, int(self._loc_off[s + 1]) * self.N
        return tuple(x[lo:hi] for x in self._neg_epoch)

    def global_scored(self, i: int) -> int:
        s = i % self.steps
        return int(self.bat.off[s + 1] - self.bat.off[s]) * (1 + self.N)

    def _plan(self, i: int):
        """Table-independent half of step i: negatives, row set, id exchange, compact indices -> slot i % nslot."""
        s = i % self.steps
        slot = i % self._nslot
        b, N, G, C, be = self.bat, self.N, self.world, self.C, self.backend
        a, e = self.my_slice(s)
        n_pos = e - a
        pos = (b.pos_h[a:e], b.pos_r[a:e], b.pos_t[a:e])
        if N and self._neg_epoch_of != i // self.steps:       # first plan of an epoch: sample the rank's whole share
            il = self._epoch_idx.long()
            if il.numel():
                be.sample_at((b.pos_h[il], b.pos_r[il], b.pos_t[il]), self._epoch_idx, b.pos_kg[il], b.side1, b.side2, N,
                             b.rng_seed, b.rng_stream, self._neg_epoch)
            self._neg_epoch_of = i // self.steps
        neg = self._step_neg(s)
        streams = [pos[0], pos[2], neg[0], neg[2]]
        be.rowset_build(streams, self._flags, self._counts, self._req, self._id_map, self._overflow, G, C)
        self.comm.all_to_all_single(self._want2[slot], self._req, group=self._plan_group)
        torch.maximum(self._max_rows, self._counts.max().reshape(1), out=self._max_rows)
        if self.keep_stats:
            self._counts_last.copy_(self._counts)
        cidx = [self._cidx2[slot][k][:streams[k].numel()] for k in range(4)]
        # remap also re-initialises req / counts for the next build (they are consumed: the id exchange is enqueued) and
        # inverts the requests this owner received, for its reduce-and-update launch
        be.rowset_remap(streams, cidx, self._id_map, self._flags, self._req, self._counts, self._want2[slot],
                        self._slot_of[slot], G, C)

    def _enqueue_plan(self, i: int):
        """Plans are issued strictly in step order; the epoch shuffle happens right before the first plan of the next
        epoch.  The shuffle rewrites the epoch buffers in place, so a plan of the NEXT epoch may only be enqueued once
        every main step of the current epoch has been enqueued (`step` guarantees it)."""
        if i % self.steps == 0 and i > 0:
            self.bat.shuffle()                                # random.shuffle of both lists at the epoch boundary
        if not self._cuda:
            self._plan(i)
        else:
            slot = i % self._nslot
            ps = self._plan_stream
            ps.wait_stream(torch.cuda.current_stream())      # epoch shuffle / setup on the main stream is visible
            if self._main_done_valid[slot]:
                ps.wait_event(self._main_done[slot])         # the slot's previous user has finished with it
            with torch.cuda.stream(ps):
                old = _lib.pin_stream(ps.cuda_stream)
                try:
                    self._plan(i)
                finally:
                    _lib.pin_stream(old)
                self._plan_done[slot].record(ps)
        self._planned = i

    def step(self, i: int):
        """Global step i (steps must be issued in order).  Plans run `lookahead` steps ahead, except across an epoch
        boundary: the next epoch's first plans wait until this epoch's last main step is enqueued, because the shuffle
        rewrites the epoch buffers in place."""
        s = i % self.steps
        epoch_end = (i // self.steps + 1) * self.steps        # first step of the next epoch
        while self._planned < min(i + self.lookahead, epoch_end - 1) or self._planned < i:
            self._enqueue_plan(self._planned + 1)
        slot = i % self._nslot
        b, N, G, C, be = self.bat, self.N, self.world, self.C, self.backend
        a, e = self.my_slice(s)
        n_pos = e - a
        pos_r = b.pos_r[a:e]
        neg_r = self._step_neg(s)[1]
        want = self._want2[slot]
        cidx = [self._cidx2[slot][k][:(n_pos if k < 2 else n_pos * N)] for k in range(4)]
        cur = torch.cuda.current_stream() if self._cuda else None
        old = _lib.pin_stream(cur.cuda_stream) if self._cuda else None
        try:
            if self._cuda:
                cur.wait_event(self._plan_done[slot])
            # ---- requested rows: owner gathers raw rows, equal-split all-to-all back ---------------------------------
            be.gather_padded(self.ent, want, self._rows_out, self._cgrad)       # also clears the compact grad scratch
            self.comm.all_to_all_single(self._rows_in, self._rows_out)
            # ---- local fused step on the compact row set --------------------------------------------------------
            self.tag += 1
            tag = self.tag
            ev = self.score_events
            if ev is not None:                                # bench instrumentation: HIP events around the score kernel
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            be.score(self._rows_in_all, True, self.rel, True, self.dim, (cidx[0], pos_r, cidx[1]), (cidx[2], neg_r, cidx[3]), N,
                     self._cgrad_all, self.rel_grad, self._ctouched, self.rel_touched, tag, self.loss_ring[s])
            if ev is not None:
                e1.record()
                ev.append((e0, e1, n_pos * (1 + N)))
            # ---- gradient rows home; the owner reduces and updates each row once ---------------------------------
            self.comm.all_to_all_single(self._ggot, self._cgrad)
            # ---- replicated relation table: all-reduce the (tiny) dense gradient -----------------------------------
            self.comm.all_reduce(self.rel_grad)
            # ---- one launch: identical relation update on every rank (touched=None: all rows) + this shard's rows, each
            #      summing what the ranks sent for it (no scatter pass, no dense gradient scratch on the owner)
            be.reduce_update((self.rel, self.rel_acc, self.rel_grad, None, True),
                             dict(table=self.ent, acc=self.ent_acc, normalize=True, src_rows=self._ggot,
                                  slot_of=self._slot_of[slot], n_ranks=G, capacity=C), tag, self.dim, self.lr)
            if self._cuda:
                self._main_done[slot].record(cur)
                self._main_done_valid[slot] = True
        finally:
            if self._cuda:
                _lib.pin_stream(old)
        self._stepped = i
        # ---- look ahead: this step is enqueued, so the next epoch's plans may start if we are at the boundary --------
        nxt_limit = i + 1 + self.lookahead if i + 1 < epoch_end else i + 1
        while self._planned < min(nxt_limit, (epoch_end if i + 1 < epoch_end else epoch_end + self.steps) - 1):
            self._enqueue_plan(self._planned + 1)

    def pending_slots(self) -> int:
        """Synchronising invariant check: request slots the owner should have consumed and has not (0 between steps).
        Plans enqueued ahead of the last executed step legitimately hold theirs."""
        ahead = {p % self._nslot for p in range(self._stepped + 1, self._planned + 1)}
        bad = 0
        for k, so in enumerate(self._slot_of):
            expect = int((self._want2[k] >= 0).sum()) if k in ahead else 0
            bad += abs(int((so >= 0).sum()) - expect)
        return bad

    def stats(self) -> dict:
        """Synchronising debug view of the last step's row set."""
        c = self._counts_last.tolist()
        return {"unique_rows": int(sum(c)), "remote_rows": int(sum(c) - c[self.rank]), "capacity": self.C,
                "overflow": int(self._overflow.item())}

    # ------------------------------------------------------------------------------------------------
    def gather_entity_table(self) -> torch.Tensor:
        """Reassemble the full [n_ent, dim] raw table on every rank (tests / checkpoint)."""
        pad = int(math.ceil(self.n_ent / self.world))
        mine = torch.zeros(pad, self.stride, dtype=self.ent.dtype, device=self.device)
        mine[:self.n_local] = self.ent
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.comm.all_gather(parts, mine)
        full = torch.zeros(self.n_ent, self.dim, dtype=self.ent.dtype, device=self.device)
        for r in range(self.world):
            n = len(range(r, self.n_ent, self.world))
            full[r::self.world] = parts[r][:n, :self.dim]
        return full

    def check(self) -> dict:
        """Synchronising: raise if ANY rank's row set overflowed the exchange capacity at any time (the steps
        concerned dropped the overflowing rows' contributions: their results are invalid); returns the capacity and the
        largest per-owner request count any rank has seen.  Call it before results are used (bench.py: after the timed
        region; training: `epoch_loss` calls it every epoch)."""
        if self._cuda:   # look-ahead plans already in flight on the plan stream write both words: read them behind it
            torch.cuda.current_stream().wait_stream(self._plan_stream)
        t = torch.stack([self._overflow[0].to(torch.int64), self._max_rows[0].to(torch.int64)])
        if dist.is_initialized() and self.world > 1:
            self.comm.all_reduce(t, op=dist.ReduceOp.MAX)
        over, worst = int(t[0]), int(t[1])
        # the flag is STICKY (never cleared): an overflow invalidates the trainer's state for good, and clearing it here could
        # erase one raised by a plan that was enqueued between this read and the clear
        if over:
            raise _lib.MultiKEHipError(f"row-set capacity {self.C} per owner exceeded (largest request seen: > {self.C}): the "
                                       f"steps concerned dropped rows; rebuild the trainer with a larger capacity")
        return {"capacity_rows_per_owner": self.C, "max_rows_per_owner_seen": worst}

    def epoch_loss(self) -> float:
        self.check()
        t = self.loss_ring.sum()
        self.comm.all_reduce(t)
        self.loss_ring.zero_()
        return float(t)
