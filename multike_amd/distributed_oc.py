"""Entity-row sharded relation-view training, "owner computes" form (RCCL over xGMI on MI355X).

The reference has no multi-device code; this is new design.  One process per GPU.
  * entity table + its Adagrad slot + gradient scratch are row-sharded by  id % world  (local row = id // world);
  * the relation table is replicated;
  * a global step is `world` x batch_size positives in the reference's epoch order; rank g is HOME of the g-th
    contiguous slice and samples its negatives (Philox stream indexed by the GLOBAL epoch position: the negatives of a
    positive do not depend on the world size);
  * rows never leave their owner.  A negative differs from its positive (h, r, t) in one entity c and its score needs only
    c's row and one of two vectors of the positive — HR_p = h^ + r^ (corrupted tail: d = HR_p - c^) or RT_p = r^ - t^
    (corrupted head: d = c^ + RT_p) — so the NEGATIVES go to the rows.  And the reference's sampler tosses ONE coin per round
    (code/base/batch.py:97-105): a positive's negatives almost always corrupt the same side, so only ONE of the two vectors
    travels for it (both for the few positives whose re-draw rounds fell on the other side); the positive's own term
    d = HR_p - t^ (or h^ + RT_p) is scored like a negative by the owner of t (of h).  Per global step:
        (once per epoch, prefetched on a side stream: every rank draws 1 / world of the epoch's negatives, packs them as
         (entity, side) codes with the group's need flags, one all-gather of the codes)
        owner of h_p builds HR_p, owner of t_p builds RT_p — the needed ones                                [mke_oc_bases]
        ALL-GATHER of the blocks (~1 vector per positive)
        reference counts of the own rows over the whole global step (needs only the codes)                 [mke_oc_count]
        every rank scores, for ALL world x batch positives, the negatives whose corrupt entity it owns: corrupt-row
        gradient applied locally (in place when referenced once, else scattered), partial dL/dHR_p, dL/dRT_p written into
        the slot the vector came from; the positive's own term by the owner of its other entity            [mke_oc_score]
        REDUCE-SCATTER of the gradient vectors (same layout): the owner of h_p / t_p receives the sum
        head / tail rows' and relation rows' gradient from it                                              [mke_oc_apply]
        ALL-REDUCE of the relation gradient;  one update of every touched shard row and relation row  [mke_rows_update_multi]
    i.e. every row is updated once per step from the sum of all its contributions (dense-Adagrad-equivalent, SURVEY.md
    §8e "semantics note").  Slots are assigned per epoch from the (replicated) epoch order, so capacity is known exactly
    before the epoch starts: nothing can overflow mid-epoch.
Bytes per rank and step over the links: (G-1)/G * 2 * ~1 * P stride 4 against (G-1)/G * 2 * P (N + 2) stride 4 of a
row exchange — 27x less at N = 25 / dim 75, 66x less at N = 64 / dim 256 (DESIGN.md §5 has the latency model).

`chunks` > 1 splits the global step's positives into that many parts whose all-gather / reduce-scatter run on the
communicator's own stream while the previous / next part is scored (split-batch pipelining; the single update at the end
sees every part's gradients, so the result is the same function).

The compute steps go through a backend object: `OcHipBackend` (the product, HIP kernels) — tests inject a CPU backend
built on the oracle to exercise this logic under `gloo`.
"""
from __future__ import annotations

import math
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .sampling import KGSide, KnownTripleSet, RelationBatcher, side_array
from .tables import ADAGRAD_INIT_ACC, PLACEMENT_LOG, placed_rows


@dataclass
class OcStep:
    """One part of a global step as the backends see it (tensors on the trainer's device)."""
    pos_h: torch.Tensor
    pos_r: torch.Tensor
    pos_t: torch.Tensor
    per: int
    slot_h: torch.Tensor
    slot_t: torch.Tensor
    own_h: torch.Tensor
    own_t: torch.Tensor
    tag: int
    codes: torch.Tensor = None      # the epoch's negative codes of every rank, [world][codes_per_rank]
    code_off: tuple = ()            # per home rank: offset of its codes of this part inside `codes`
    native: object = None           # backend-private cache (the ctypes mke_oc_step of the HIP backend)
    pos_w: torch.Tensor = None      # per-positive weights of the part (weighted cross-KG loops), or None


class OcHipBackend:
    """Product backend: every compute step is a HIP kernel of libmultike_hip.so (mke_oc.hip, mke_update.hip)."""

    device_type = "cuda"

    def make_known(self, h, r, t):
        return KnownTripleSet(h, r, t)

    def sample_at(self, pos, pos_index, pos_kg, side1, side2, neg_per_pos, seed, stream_id, out):
        _lib.neg_sample_at(pos, pos_index, pos_kg, side_array(side1, side2), neg_per_pos, 10, seed, stream_id, out)

    def block_elems(self, capacity, stride):
        return _lib.oc_block_floats(capacity, stride)

    def pack_codes(self, pos_h, neg_h, neg_t, neg_per_pos, codes):
        _lib.oc_pack_codes(pos_h, neg_h, neg_t, neg_per_pos, codes)

    def plan(self, pos_h, pos_t, codes, neg_per_pos, part_lo, n_parts, n_ranks, rank, slot_h, slot_t, own_h, own_t, counts):
        """slots, owned lists and per-(part, owner) counts of the whole epoch in ONE launch (mke_oc_plan)."""
        _lib.oc_plan(pos_h, pos_t, codes, neg_per_pos, part_lo, n_parts, n_ranks, rank, slot_h, slot_t, own_h, own_t, counts)

    def em_plan(self, tr, ph, pr, pt, codes, slot, bufs):
        """mke_oc_em_plan: the epoch's references to this rank's rows sorted by (step, row, positive, kind), the touched rows of
        every global step and their CSR offsets (entity-major second pass) — one native call, nothing synchronises."""
        i32, i64 = torch.int32, torch.int64
        a = _lib.OcEmPlanArgs()
        a.pos_h, a.pos_r, a.pos_t = _lib.ptr(ph, i32, "pos"), _lib.ptr(pr, i32, "pos"), _lib.ptr(pt, i32, "pos")
        a.codes, a.neg_per_pos = _lib.ptr(codes, i32, "codes"), tr.N
        a.slot_h, a.slot_t = _lib.ptr(slot[0], i32, "slot"), _lib.ptr(slot[1], i32, "slot")
        a.step_lo, a.n_steps, a.chunks = _lib.ptr(tr._step_lo, i64, "step_lo"), tr.steps, tr.chunks
        a.n_all, a.max_step = tr._n_all, tr._max_step
        a.n_ranks, a.rank, a.n_local, a.n_rel = tr.world, tr.rank, max(1, tr.n_local), tr.rel.shape[0]
        a.keys, a.keys_alt, a.capacity = _lib.ptr(bufs["keys"], i64, "keys"), _lib.ptr(bufs["keys_alt"], i64, "keys"), bufs["capacity"]
        a.vals_alt, a.wave_scratch = _lib.ptr(bufs["vals_alt"], i32, "vals_alt"), _lib.ptr(bufs["waves"], i32, "waves")
        a.scratch8 = _lib.ptr(bufs["scratch8"], i64, "scratch8")
        a.refs, a.rows, a.off = _lib.ptr(bufs["refs"], i32, "refs"), _lib.ptr(bufs["rows"], i32, "rows"), _lib.ptr(bufs["off"], i32, "off")
        a.flags, a.scan = _lib.ptr(bufs["flags"], i32, "flags"), _lib.ptr(bufs["scan"], i32, "scan")
        a.step_row0, a.n_refs = _lib.ptr(bufs["row0"], i64, "row0"), _lib.ptr(bufs["n_refs"], i64, "n_refs")
        a.item_row, a.item_off, a.item_part = (_lib.ptr(bufs[k], i32, k) for k in ("item_row", "item_off", "item_part"))
        a.long_row, a.long_part0 = _lib.ptr(bufs["long_row"], i32, "long_row"), _lib.ptr(bufs["long_part0"], i32, "long_part0")
        a.step_item0, a.step_long0, a.step_part0 = (_lib.ptr(bufs["steps3"][k], i64, "steps3") for k in range(3))
        a.temp, a.temp_bytes = _lib.ptr(bufs["temp"], torch.uint8, "temp"), bufs["temp"].numel()
        _lib.oc_em_plan(a)

    def em_temp_bytes(self, capacity):
        return _lib.oc_em_plan_temp_bytes(capacity)

    def _struct(self, tr: "OwnerComputesTrainer", st: OcStep):
        f32, i32 = torch.float32, torch.int32
        s = _lib.OcStepStruct()
        s.ent, s.ent_acc = _lib.ptr(tr.ent, f32, "ent"), _lib.ptr(tr.ent_acc, f32, "acc")
        s.ent_grad = _lib.ptr(tr.ent_grad, f32, "grad") if tr.ent_grad is not None else None      # entity-major: no entity scratch
        s.ent_touched = _lib.ptr(tr.ent_touched, i32, "touched") if tr.ent_touched is not None else None
        s.ref_count = _lib.ptr(tr.ref_count, i32, "ref_count") if tr.ref_count is not None else None
        s.n_local = tr.n_local
        s.rel, s.rel_grad = _lib.ptr(tr.rel, f32, "rel"), _lib.ptr(tr.rel_grad, f32, "rel_grad")
        s.rel_grad_copies = 1 if tr.rel_grad.dim() == 2 else tr.rel_grad.shape[0]     # privatised relation gradient (all-reduced whole)
        s.rel_acc = _lib.ptr(tr.rel_acc, f32, "rel_acc")
        s.rel_touched, s.n_rel = _lib.ptr(tr.rel_touched, i32, "rel_touched"), tr.rel.shape[0]
        s.stride, s.dim, s.rank, s.n_ranks = tr.stride, tr.dim, tr.rank, tr.world
        s.pos_h, s.pos_r, s.pos_t = (_lib.ptr(x, i32, "pos") for x in (st.pos_h, st.pos_r, st.pos_t))
        s.n_pos, s.per = st.pos_h.numel(), st.per
        s.slot_h, s.slot_t = _lib.ptr(st.slot_h, i32, "slot"), _lib.ptr(st.slot_t, i32, "slot")
        s.own_h, s.n_own_h = _lib.ptr(st.own_h, i32, "own"), st.own_h.numel()
        s.own_t, s.n_own_t = _lib.ptr(st.own_t, i32, "own"), st.own_t.numel()
        s.neg_per_pos, s.capacity = tr.N, tr.C
        s.codes = _lib.ptr(st.codes, i32, "codes")
        for g, o in enumerate(st.code_off):
            s.code_off[g] = int(o)
        s.optimizer, s.lr, s.scale, s.tag = tr.OPTIMIZER, tr.lr, tr.scale, st.tag
        s.pos_w = _lib.ptr(st.pos_w, f32, "pos_w") if st.pos_w is not None else None
        if tr.hot_slot is not None:           # hub rows of the shard: private gradient copies behind the shard's own rows
            s.hot.slot, s.hot.n_hot = _lib.ptr(tr.hot_slot, i32, "hot_slot"), tr.n_hot
            s.hot.copies, s.hot.row0 = tr.HOT_COPIES, tr.ent_grad_rows
        s.tuning = _lib.tuning_ptr(tr.tuning)
        s.n_peers = 0
        if tr.peer_direct and tr.world > 1:   # peer-mapped blocks (chunk 0: peer-direct runs unchunked)
            gb = 2 * tr.C * tr.stride * 4
            s.n_peers = tr.world
            for g in range(tr.world):
                s.peer_v[g] = tr._peer_send[g].data_ptr()
                s.peer_g[g] = tr._peer_inbox[g].data_ptr() + tr.rank * gb
        return s

    def _cached(self, tr, st):
        """The part's mke_oc_step is built once per epoch (st.native) and only re-tagged per step."""
        s = st.native
        if s is None:
            s = st.native = self._struct(tr, st)
        s.tag = st.tag
        return s

    def prepare_epoch(self, tr):
        """One mke_oc_step per part of the epoch, from raw device addresses (the epoch buffers are persistent: positives,
        slots, owned lists keep their addresses; only the owned-list offsets change from epoch to epoch) — no tensor
        slicing and no struct building on the step path."""
        i32 = torch.int32
        b = tr.bat
        if not tr._parts:
            self._steps = []
            return
        oh, ot = _lib.ptr(tr._own[0], i32, "own"), _lib.ptr(tr._own[1], i32, "own")
        em = tr._em if tr.em else None
        key = (tr.C, b.pos_h.data_ptr(), tr._slot[0].data_ptr(), tr._slot[1].data_ptr(), oh, ot, tr._codes.data_ptr(), len(tr._parts),
               tr._peer_send[0].data_ptr() if tr.peer_direct and tr.world > 1 else 0,
               (em["refs"].data_ptr(), em["item_row"].data_ptr(), em["item_off"].data_ptr(), tr._em_coef.data_ptr(),
                tr._em_partials.data_ptr()) if em else 0)
        cache = self.__dict__.setdefault("_tables", {})
        if key in cache:                                  # the two epoch buffer sets alternate: one table each
            self._steps = cache[key]
        else:                                             # first use of this buffer set, or a buffer was re-allocated
            base = self._struct(tr, tr._build_part_step(0, 0))
            ph, pr, pt = (_lib.ptr(x, i32, "pos") for x in (b.pos_h, b.pos_r, b.pos_t))
            sh, stt = _lib.ptr(tr._slot[0], i32, "slot"), _lib.ptr(tr._slot[1], i32, "slot")
            pw = getattr(b, "pos_w", None)
            pw = _lib.ptr(pw, torch.float32, "pos_w") if pw is not None else None
            arr = (_lib.OcStepStruct * len(tr._parts))()     # contiguous: mke_oc_steps walks it (the list below holds views)
            out = []
            for k, (_, lo, hi) in enumerate(tr._parts):
                s = arr[k]
                C.memmove(C.byref(s), C.byref(base), C.sizeof(s))
                s.pos_h, s.pos_r, s.pos_t = ph + 4 * lo, pr + 4 * lo, pt + 4 * lo
                s.slot_h, s.slot_t = sh + 4 * lo, stt + 4 * lo
                s.pos_w = (pw + 4 * lo) if pw is not None else None
                s.n_pos = hi - lo
                s.per = max(1, -(-(hi - lo) // tr.world))
                for g in range(tr.world):
                    s.code_off[g] = (lo + g * int(s.per)) * tr.N      # codes are laid out by epoch position
                if em:      # entity-major: the step's coefficient buffer, this part's first positive in it, the chunks' vector blocks
                    step = tr._parts[k][0]
                    s.em_coef, s.em_pos0 = tr._em_coef.data_ptr(), lo - int(b.off[step])
                    s.em_refs = em["refs"].data_ptr()
                    s.em_chunks, s.em_block_floats = len(tr._parts_of[step]), tr.block
                    for c in range(int(s.em_chunks)):
                        s.em_v[c], s.em_gv[c] = tr._addr[c][1], tr._addr[c][3]
                out.append(s)
            if len(cache) > 4:
                cache.clear()
            self._steps = cache[key] = out
            out_arr = self.__dict__.setdefault("_arrays", {})
            if len(out_arr) > 4:
                out_arr.clear()
            out_arr[key] = arr
            self._ring = tr.loss_ring.data_ptr()
            self._ring_stride = tr.loss_ring.shape[1] * 8
        self._parts_arr = self._arrays[key]
        cnth, cntt = (x.tolist() for x in tr._own_cnt)
        for k, (s, (_, lo, _hi)) in enumerate(zip(self._steps, tr._parts)):     # a part's owned list starts at the part's own offset
            s.own_h, s.n_own_h = oh + 4 * lo, cnth[k]
            s.own_t, s.n_own_t = ot + 4 * lo, cntt[k]
        if em:              # the work items / long rows of each global step: positions change from epoch to epoch
            i0, l0, p0 = (em[k].tolist() for k in ("item0_host", "long0_host", "part0_host"))
            rp, op, pp = em["item_row"].data_ptr(), em["item_off"].data_ptr(), em["item_part"].data_ptr()
            lr, lp = em["long_row"].data_ptr(), em["long_part0"].data_ptr()
            for s, (step, _, _) in zip(self._steps, tr._parts):
                s.em_rows, s.em_off, s.em_n_rows = rp + 4 * i0[step], op + 4 * i0[step], i0[step + 1] - i0[step]
                s.em_part = pp + 4 * i0[step]
                s.em_long_rows, s.em_long_part0, s.em_n_long = lr + 4 * l0[step], lp + 4 * l0[step], l0[step + 1] - l0[step]
                s.em_part0, s.em_partials = p0[step], tr._em_partials.data_ptr()
        # the whole epoch's schedule for mke_oc_steps (one native call per run of steps)
        lp = _lib.OcLoopStruct()
        lp.parts, lp.n_steps, lp.chunks = C.addressof(self._parts_arr), tr.steps, tr.chunks
        first = (C.c_int32 * (tr.steps + 1))()
        k = 0
        for st in range(tr.steps):
            first[st] = k
            k += len(tr._parts_of.get(st, ()))
        first[tr.steps] = k
        self._step_part0 = first
        lp.step_part0 = C.addressof(first)
        for c in range(tr.chunks):
            lp.send[c], lp.v_all[c], lp.g_all[c], lp.gv[c] = tr._addr[c]
        lp.block_floats = tr.block
        lp.loss_ring, lp.loss_stride = self._ring, tr.loss_ring.shape[1]
        self._loop = lp

    def run_steps(self, tr, s0, s1, tag_base, comm_struct, comm_stream, overlap_rs=False):
        """Global steps [s0, s1) of the current epoch in ONE native call (mke_oc_steps): kernels, collectives and — with several
        parts per step — the two-stream pipeline are enqueued from C++."""
        lp = self._loop
        lp.tag_base = tag_base
        lp.comm = C.addressof(comm_struct) if comm_struct is not None else None
        lp.comm_stream = comm_stream
        lp.overlap_rs = int(bool(overlap_rs))
        _lib.oc_steps(lp, s0, s1)

    def bases(self, tr, st, send):
        _lib.oc_bases(self._cached(tr, st), send)

    def count(self, tr, st):
        _lib.oc_count(self._cached(tr, st))

    def score(self, tr, st, v_all, g_all, loss_partials):
        _lib.oc_score(self._cached(tr, st), v_all, tr.block, g_all, loss_partials)

    def apply(self, tr, st, gv):
        _lib.oc_apply(self._cached(tr, st), gv)

    def update(self, tr, tag):
        # relation table: EVERY row (touched = None) — after the all-reduce a row may carry a gradient no local triple touched
        hot = None
        if tr.hot_slot is not None:
            hot = _lib.HotRowsStruct(_lib.ptr(tr.hot_slot, torch.int32, "hot_slot"), tr.n_hot, tr.HOT_COPIES, tr.ent_grad_rows)
        _lib.rows_update_multi([(tr.rel, tr.rel_acc, tr.rel_grad, None, True),
                                (tr.ent, tr.ent_acc, tr.ent_grad, tr.ent_touched, True, tr.ref_count, hot)],
                               tag, tr.stride, tr.dim, tr.OPTIMIZER, tr.lr)

    def run(self, tr, k, tag, phases, c, loss_slot):
        """The phases of `phases` (OC_* bit mask) of part k (chunk buffers c) in ONE native call; buffers by raw address
        (validated when they were allocated)."""
        a = tr._addr[c]
        s = self._steps[k]
        s.tag = tag
        _lib.oc_run(s, phases, a[0], a[1], tr.block, a[2], a[3], self._ring + loss_slot * self._ring_stride)


BASES, COUNT, SCORE, APPLY, UPDATE, PASS2 = _lib.OC_BASES, _lib.OC_COUNT, _lib.OC_SCORE, _lib.OC_APPLY, _lib.OC_UPDATE, _lib.OC_PASS2


class OcComm:
    """The three collectives of the step on torch.distributed (RCCL over xGMI on MI355X; gloo in the CPU tests).
    `async_op` returns a work handle whose wait() orders the CURRENT stream after the collective."""

    def __init__(self, group=None):
        self.group = group

    def all_gather(self, out, mine, async_op=False):
        return dist.all_gather_into_tensor(out, mine, group=self.group, async_op=async_op)

    def reduce_scatter(self, out, inp, async_op=False):
        return dist.reduce_scatter_tensor(out, inp, group=self.group, async_op=async_op)

    def all_reduce(self, t, op=None):
        dist.all_reduce(t, group=self.group) if op is None else dist.all_reduce(t, op=op, group=self.group)

    def all_gather_list(self, parts, mine):
        dist.all_gather(parts, mine, group=self.group)

    def barrier(self, token):
        """Stream-ordered cross-rank barrier (peer-direct mode): a one-element all-reduce — every rank's stream passes it only
        after every rank's stream has reached it; the host is not blocked."""
        dist.all_reduce(token, group=self.group)

    def all_gather_object(self, obj):
        out = [None] * dist.get_world_size(self.group)
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def for_plan(self):
        """A communicator of its own for the once-per-epoch all-gather of the negative codes: that collective is issued from
        the plan's side stream while the step collectives run on the main stream — on one communicator the steps would queue
        behind it (a process group's collectives execute in issue order)."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return self
        return type(self)(dist.new_group(ranks=list(range(dist.get_world_size(self.group)))) if self.group is None else
                          dist.new_group(ranks=dist.get_process_group_ranks(self.group)))


class _EventWork:
    """Handle of an asynchronous collective: wait() orders the CURRENT stream after it (no host wait)."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class OcRcclComm(OcComm):
    """The three collectives through RCCL directly (multike_amd/rccl.py), enqueued ON THE CALLER'S STREAM: stream order is the
    dependency, no event and no second stream per collective (torch.distributed's two stream hops per collective cost ~26 us
    of device time and ~30 us of host time each on this part: EXPERIMENTS R5.3).  async_op=True (the chunk-pipelined schedule)
    goes to this communicator's own stream with two pooled events.  The default of the HIP trainers on an "nccl" process
    group; MKE_OC_COMM=torch selects OcComm."""

    def __init__(self, group=None):
        super().__init__(group)
        from .rccl import Communicator
        self.c = Communicator(group)
        self.c.self_check()                 # all three collectives give the right sums, or raise before any training step
        self._side, self._events, self._k = None, None, 0

    def _async(self, fn):
        if self._side is None:
            self._side = torch.cuda.Stream()
            self._events = [torch.cuda.Event() for _ in range(64)]
        cur = torch.cuda.current_stream()
        e_in, e_out = self._events[self._k % 64], self._events[(self._k + 1) % 64]
        self._k += 2
        e_in.record(cur)
        self._side.wait_event(e_in)
        fn(self._side)
        e_out.record(self._side)
        return _EventWork(e_out)

    def all_gather(self, out, mine, async_op=False):
        o, m = out.view(-1), mine.reshape(-1)
        if async_op:
            return self._async(lambda st: self.c.all_gather(o, m, st))
        self.c.all_gather(o, m)

    def reduce_scatter(self, out, inp, async_op=False):
        o, i = out.view(-1), inp.view(-1)
        if async_op:
            return self._async(lambda st: self.c.reduce_scatter(o, i, st))
        self.c.reduce_scatter(o, i)

    def all_reduce(self, t, op=None):
        if op is not None:
            return super().all_reduce(t, op)
        self.c.all_reduce(t.view(-1))

    def barrier(self, token):
        self.c.all_reduce(token.view(-1))

    def native(self, tr=None):
        """mke_oc_comm over this communicator: RCCL's own entry points, called from the native step loop (mke_oc_steps)."""
        if getattr(self, "_native", None) is None:
            from . import rccl
            L = rccl.lib()
            cs = _lib.OcCommStruct()
            cs.kind, cs.ctx = _lib.OC_COMM_NCCL, self.c._comm.value
            cs.all_gather = C.cast(L.ncclAllGather, C.c_void_p).value
            cs.reduce_scatter = C.cast(L.ncclReduceScatter, C.c_void_p).value
            cs.all_reduce = C.cast(L.ncclAllReduce, C.c_void_p).value
            cs.world, cs.rank = self.c.world, self.c.rank
            self._native = cs
        return self._native

    def for_plan(self):
        """A second RCCL communicator (concurrent with the step collectives), one per step communicator."""
        if getattr(self, "_plan", None) is None:
            self._plan = OcRcclComm(self.group)
        return self._plan


_DEFAULT_RCCL = None        # the process's step communicator over the world: every trainer of a model shares it


def default_comm(device, world, force=False):
    """The communicator a HIP trainer uses when none is given."""
    import os
    global _DEFAULT_RCCL
    if device.type == "cuda" and dist.is_initialized() and dist.get_backend() == "nccl" and (world > 1 or force) \
            and os.environ.get("MKE_OC_COMM", "rccl") != "torch":
        if _DEFAULT_RCCL is None or (_DEFAULT_RCCL is not False and _DEFAULT_RCCL.c.world != dist.get_world_size()):
            # every rank tries; the ranks then agree (one torch.distributed all-reduce) on whether ALL of them succeeded — a
            # communicator that came up on some ranks only must not be used by any
            try:
                cand, err = OcRcclComm(), None
            except Exception as e:      # noqa: BLE001 — reported below, the torch.distributed communicator takes over
                cand, err = None, e
            ok = torch.tensor([1 if cand is not None else 0], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok) == 1:
                _DEFAULT_RCCL = cand
            else:
                import warnings
                warnings.warn(f"multike_amd: RCCL through ctypes did not come up on every rank ({err!r} on this one): using torch.distributed for the collectives")
                _DEFAULT_RCCL = False
        if _DEFAULT_RCCL is not False:
            return _DEFAULT_RCCL
    return OcComm() if (device.type == "cuda" or not dist.is_initialized()) else OcGlooComm()


class OcGlooComm(OcComm):
    """gloo has no reduce-scatter: all-reduce the whole buffer and keep this rank's block (CPU tests only)."""

    def all_gather(self, out, mine, async_op=False):
        w = dist.get_world_size(self.group)
        dist.all_gather(list(out.view(w, -1).unbind(0)), mine.reshape(-1), group=self.group)

    def reduce_scatter(self, out, inp, async_op=False):
        w, r = dist.get_world_size(self.group), dist.get_rank(self.group)
        tmp = inp.clone()
        dist.all_reduce(tmp, group=self.group)
        out.copy_(tmp.view(w, -1)[r].view_as(out))


class OcHostStagedComm(OcGlooComm):
    """Test vehicle: the same collectives on DEVICE tensors through gloo, staged over the host.  Lets two ranks that SHARE
    one GPU run the device kernels with world_size 2 (RCCL refuses two ranks on one device)."""

    def all_gather(self, out, mine, async_op=False):
        o = torch.empty(out.shape, dtype=out.dtype)
        super().all_gather(o, mine.cpu())
        out.copy_(o)

    def reduce_scatter(self, out, inp, async_op=False):
        o = torch.empty(out.shape, dtype=out.dtype)
        super().reduce_scatter(o, inp.cpu())
        out.copy_(o)

    def all_reduce(self, t, op=None):
        c = t.cpu()
        super().all_reduce(c, op)
        t.copy_(c)

    def all_gather_list(self, parts, mine):
        cp = [torch.empty(p.shape, dtype=p.dtype) for p in parts]
        dist.all_gather(cp, mine.cpu(), group=self.group)
        for p, c in zip(parts, cp):
            p.copy_(c)

    def barrier(self, token):
        torch.cuda.synchronize()           # gloo orders hosts, not streams
        dist.barrier(group=self.group)

    def native(self, tr):
        """mke_oc_comm of kind CALLBACK: the native step loop calls back into these staged collectives (the buffers are found by
        their device address among the trainer's exchange buffers; the callback works on the stream the loop hands it)."""
        def find(addr, count):
            for t in tr._exchange_tensors():
                if t.data_ptr() == addr:
                    return t.view(-1)[:count]
            raise _lib.MultiKEHipError("native callback: unknown exchange buffer")

        def on(stream):
            # the stream the loop enqueues on, as a torch stream: handle 0 is torch's default stream (torch.cuda.ExternalStream(0)
            # is NOT — it makes a stream of its own, and the staged copies then raced the kernels: caught at world 8)
            return torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()

        def move(fn, scale_in, scale_out):
            def cb(ctx, send, recv, count, stream):
                try:
                    with torch.cuda.stream(on(stream)):
                        fn(find(recv, count * scale_out), find(send, count * scale_in))
                    return 0
                except Exception:      # noqa: BLE001 — an exception must not unwind through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1
            return _lib.OC_CB_MOVE(cb)

        def reduce(ctx, buf, count, stream):
            try:
                with torch.cuda.stream(on(stream)):
                    self.all_reduce(find(buf, count))
                return 0
            except Exception:          # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        G = dist.get_world_size(self.group)
        keep = (move(self.all_gather, 1, G), move(self.reduce_scatter, G, 1), _lib.OC_CB_REDUCE(reduce))
        cs = _lib.OcCommStruct()
        cs.kind = _lib.OC_COMM_CALLBACK
        cs.all_gather, cs.reduce_scatter, cs.all_reduce = (C.cast(f, C.c_void_p).value for f in keep)
        cs.world, cs.rank = G, dist.get_rank(self.group)
        cs._keep = keep                # the thunks live as long as the struct
        return cs


class TripleListBatcher:
    """Epoch source of the owner-computes trainer for the cross-KG inference loops (code/MultiKE_model.py:349-369, 393-414):
    `steps = ceil(len / B)` steps per epoch, each `random.sample(triples, B)` (B = len when there is one step) — distinct
    inside a step, steps independent.  Positives only (no negative sampler: `neg_per_pos` must be 0).  The draws are a
    function of (seed, epoch) alone, so every rank lays out the same epoch without exchanging a byte: on the GPU through
    `mke_sample_distinct` (the sampler of the single-GPU loops, multike_amd/MultiKE_model.py `_positives_epoch`), on the CPU
    (gloo tests) through NumPy's generator."""

    def __init__(self, triples, batch_size: int, device="cuda", seed: int = 0):
        arr = np.ascontiguousarray(np.asarray([t[:3] for t in triples], dtype=np.int32).reshape(-1, 3))
        self.device, self.seed, self.n = torch.device(device), int(seed), int(arr.shape[0])
        self.cols = tuple(torch.as_tensor(np.ascontiguousarray(arr[:, k]), device=self.device) for k in range(3))
        # 4-tuples (h, r, t, w): the weighted loops (code/MultiKE_model.py:393-414); drawn with their triples
        self.w_all = (torch.as_tensor(np.asarray([t[3] for t in triples], dtype=np.float32), device=self.device)
                      if self.n and len(triples[0]) > 3 else None)
        self.steps = int(math.ceil(self.n / batch_size)) if self.n else 0
        self.bs = int(batch_size) if self.steps > 1 else self.n
        self.off = np.arange(self.steps + 1, dtype=np.int64) * self.bs
        self.epoch, self._alt = 0, None
        self.pos_kg = self.side1 = self.side2 = None           # no negatives: nothing the sampler would need
        (self.pos_h, self.pos_r, self.pos_t), self.pos_w = self._draw(0)

    def _draw(self, epoch: int):
        """((h, r, t), w or None) of every step of `epoch`, in step order."""
        total = self.steps * self.bs
        if total == 0:
            z = torch.zeros(1, dtype=torch.int32, device=self.device)
            return (z, z.clone(), z.clone()), None
        if self.device.type == "cuda":
            idx = _lib.sample_distinct(self.n, self.bs, self.steps, (self.seed & 0xFFFFFFFF, 0x434B47), epoch + 1,
                                       device=self.device).reshape(-1).long()
        else:
            rng = np.random.default_rng([self.seed, epoch])
            idx = torch.as_tensor(np.concatenate([rng.choice(self.n, self.bs, replace=False) for _ in range(self.steps)]))
        return tuple(c[idx].contiguous() for c in self.cols), (self.w_all[idx].contiguous() if self.w_all is not None else None)

    def shuffle(self):
        """The next epoch's draws, into the SAME buffers (native step descriptors point into them)."""
        self.epoch += 1
        cols, w = self._draw(self.epoch)
        for dst, src in zip((self.pos_h, self.pos_r, self.pos_t), cols):
            dst.copy_(src)
        if w is not None:
            self.pos_w.copy_(w)

    def stage_next_epoch(self):
        new, self._w_next = self._draw(self.epoch + 1)
        if self._alt is None:
            self._alt = new
        else:
            for dst, src in zip(self._alt, new):
                dst.copy_(src)
        return self._alt

    def commit_staged(self):
        cur = (self.pos_h, self.pos_r, self.pos_t)
        self.pos_h, self.pos_r, self.pos_t = self._alt
        self._alt = cur
        if self.pos_w is not None:
            self.pos_w.copy_(self._w_next)   # one weight buffer: the staged epoch's weights arrive with the swap
        self.epoch += 1

    @property
    def rng_seed(self):
        return (self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)

    @property
    def rng_stream(self):
        return (self.epoch * 2) & 0xFFFFFFFF


def hub_rows_of_shard(triples, n_ent: int, global_batch: int, rank: int, world: int, hot_min: float = 20.0, hot_max: int = 1024):
    """LOCAL rows (id // world) of this rank's entities that are head or tail of >= hot_min positives of an average global
    step, from the two KGs' relation triples (static degrees: an epoch is the same triples in a new order) — what a holder
    of an EmbeddingTable shard passes to `set_hot_rows` before it builds the trainers on it."""
    ids = np.concatenate([np.asarray(t, dtype=np.int64).reshape(-1, 3)[:, [0, 2]].reshape(-1) for t in triples])
    n = sum(len(t) for t in triples)
    steps = max(1, -(-n // max(1, int(global_batch))))
    deg = np.bincount(ids, minlength=n_ent) / steps
    hot = np.nonzero((deg >= hot_min) & (np.arange(n_ent) % world == rank))[0]
    if len(hot) > hot_max:
        hot = hot[np.argsort(-deg[hot])[:hot_max]]
    return np.sort(hot // world)


class OwnerComputesTrainer:
    # hub rows of the shard (mke_oc_step.hot): entities that are head or tail of >= HOT_MIN positives of an average GLOBAL step get
    # HOT_COPIES private copies of their gradient row for mke_oc_apply and the positives' own terms (the fused runner's rule,
    # multike_amd/runner.py; measured as rank 0 of 8 on Zipf(1.0) triples: EXPERIMENTS R5.12)
    HOT_MIN, HOT_MAX, HOT_COPIES = 20.0, 1024, 8
    REL_COPIES = 4
    SAMPLE_RUN = 1 << 26          # ids per column of the epoch sampler's scratch (the plan samples its share in runs of SAMPLE_RUN / N positions)

    def __init__(self, kgs, ent0: np.ndarray, rel0: np.ndarray, batch_size: int, neg_per_pos: int, rank: int, world: int,
                 seed: int = 0, lr: float = 0.001, backend=None, device=None, dtype=torch.float32, comm=None,
                 exclusive_rows: bool = True, chunks: int = 1, peer_direct: bool = False, prefetch: bool = True,
                 batcher=None, scale: float = 1.0, tables_of: "OwnerComputesTrainer" = None, ent_table=None, rel_table=None,
                 opt_name: str = "relation", n_ent: int = None, tag_base: int = None, global_batch: int = None,
                 entity_major: bool = None, tuning: dict = None):
        """batcher: an epoch source other than the two KGs' shuffled triples (`TripleListBatcher`: the cross-KG inference
        loops — positives only, `neg_per_pos` 0, `kgs` unused and `batch_size` the GLOBAL step size the batcher was built
        with); scale: the loss factor (2 for code/MultiKE_model.py:349-369); tables_of: another trainer of the same
        (rank, world, device, dtype) whose entity shard and relation table this one trains too — the graphs of one view
        share their variables and have one optimizer each (code/MultiKE_model.py:17-31): shared tables and (zero-invariant)
        gradient / flag scratch, own Adagrad accumulators, own tag range; `ent0` / `rel0` are then unused."""
        self.scale = float(scale)
        self._em_request = entity_major
        self.tuning = _lib.tuning(**tuning) if tuning else None     # this trainer's knobs (mke_oc_step.tuning), e.g. {"oc_score_quarter": 1}
        # ent_table / rel_table (multike_amd.tables.EmbeddingTable: this rank's shard of `n_ent` global rows, and the
        # replicated relation table): train THOSE — the trainer then shares them with whatever else holds them (other
        # trainers, the common-space step of multike_amd.distributed_views) and takes its Adagrad slot by `opt_name`.
        if ent_table is not None:
            if rel_table is None or n_ent is None:
                raise _lib.MultiKEHipError("ent_table needs rel_table and n_ent (the global row count)")
            ent0 = np.empty((int(n_ent), ent_table.dim), dtype=np.float32)          # shapes only
            rel0 = np.empty((rel_table.n_rows, rel_table.dim), dtype=np.float32)
        if tables_of is not None:
            ent0 = np.empty((tables_of.n_ent, tables_of.dim), dtype=np.float32)    # shapes only
            rel0 = np.empty((tables_of.rel.shape[0], tables_of.dim), dtype=np.float32)
        self.backend = backend or OcHipBackend()
        self.device = torch.device(device or ("cuda" if self.backend.device_type == "cuda" else "cpu"))
        import os as _os
        self.force_collectives = _os.environ.get("MKE_OC_FORCE_COLLECTIVES", "0") == "1"
        if comm is None:
            comm = default_comm(self.device, world, self.force_collectives)
        self.comm = comm
        # the epoch plan's collective (the ranks' shares of the epoch's negative codes) on a communicator of its own
        # (round 5 put the epoch plan's collective on a communicator of its own, issued from the side stream: nothing ordered
        # it against the step collectives across ranks — it now goes through `comm` at a fixed point of the step sequence)
        self.rank, self.world, self.lr = rank, world, float(lr)
        self.dim = ent0.shape[1]
        self.stride = _lib.stride_for(self.dim)
        self.N = int(neg_per_pos)
        if not 0 <= self.N <= 64:   # 0: positives only (the shape of the cross-KG inference loops, code/MultiKE_model.py:349-369)
            raise _lib.MultiKEHipError("the sharded relation view takes 0..64 negatives per positive")
        self.n_ent = ent0.shape[0]
        if self.n_ent >= 1 << 29:   # a code is (entity << 1) | side with the group's two need flags above it
            raise _lib.MultiKEHipError("the sharded relation view packs entity ids into 29 bits")
        self.batch_size = int(batch_size)
        self.chunks = max(1, int(chunks))
        # peer-direct (opt-in): no all-gather / reduce-scatter — every rank maps the other ranks' send blocks and gradient
        # inboxes (IPC handles exchanged once) and mke_oc_score reads / writes them straight over xGMI; two stream-ordered
        # barriers per step.  Correct by construction and tested with two ranks on one GPU; not measured on several.
        self.prefetch = bool(prefetch)
        # MKE_OC_FORCE_COLLECTIVES=1: a one-rank group takes the G > 1 step path — its three collectives issued for real on the
        # one-rank communicator — so that the host cost of that path (Python + torch.distributed per step) can be measured on
        # one GPU (bench.py --force-sharded reports `host_us_per_step`)
        self.peer_direct = bool(peer_direct) and world > 1
        if self.peer_direct:
            self.chunks = 1
        # ENTITY-MAJOR second pass (round 6): the score launch stores one coefficient per (positive, owned negative),
        # mke_oc_pass2 finishes every touched owned row in place from the row's reference list of the epoch plan — no gradient
        # scratch, flags, reference counts, hub-row copies or atomics on entity rows, results bit-reproducible run to run.  The
        # default of the HIP backend (MKE_OC_EM=0 / entity_major=False: the atomics form of rounds 2-5); not with peer-direct.
        em = self._em_request
        if em is None:
            em = _os.environ.get("MKE_OC_EM", "1") != "0"
        self.em = bool(em) and hasattr(self.backend, "em_plan") and not self.peer_direct and self.chunks <= _lib.OC_EM_MAX_CHUNKS \
            and dtype == torch.float32
        if self.em:
            exclusive_rows = False
        dev, st = self.device, self.stride
        i32 = dict(dtype=torch.int32, device=dev)
        # --- row-sharded entity state ---------------------------------------------------------------
        mine = np.arange(rank, self.n_ent, world)
        self.n_local = len(mine)
        if ent_table is not None:
            if ent_table.n_rows != max(1, self.n_local) or ent_table.stride != st or rel_table.stride != st:
                raise _lib.MultiKEHipError("ent_table: the shard must hold ceil-share rows of n_ent with the trainer's stride")
            self.ent, self.ent_grad, self.ent_touched = ent_table.data, ent_table.grad, ent_table.touched
            self.rel, self.rel_grad, self.rel_touched = rel_table.data, rel_table.grad, rel_table.touched
            self.ref_count = ent_table.refcount if exclusive_rows else None
        elif tables_of is not None:
            o = tables_of
            if (o.rank, o.world, o.device, o.ent.dtype) != (rank, world, dev, dtype):
                raise _lib.MultiKEHipError("tables_of: the two trainers must agree on rank / world / device / dtype")
            self.ent, self.ent_grad, self.ent_touched = o.ent, o.ent_grad, o.ent_touched
            self.rel, self.rel_grad, self.rel_touched = o.rel, o.rel_grad, o.rel_touched
            if exclusive_rows and o.ref_count is None:
                o.ref_count = torch.zeros(max(1, self.n_local), **i32)
            self.ref_count = o.ref_count if exclusive_rows else None
        else:
            place = dtype == torch.float32          # big float32 shards: on the fastest of a few candidate allocations (tables.placed_rows)
            mk = (lambda fill, with_=(): placed_rows(max(1, self.n_local), st, dev, fill, PLACEMENT_LOG, with_)) if place else \
                (lambda fill, with_=(): torch.full((max(1, self.n_local), st), fill, dtype=dtype, device=dev))
            self.ent = mk(0.0)
            self.ent[:self.n_local, :self.dim] = torch.as_tensor(ent0[mine], dtype=dtype, device=dev)
            self.ent_grad = None                    # allocated by _declare_hot_rows (with the hub rows' copies behind the shard's rows)
            self._mk_rows = mk
            self.ent_touched = None if self.em else torch.zeros(max(1, self.n_local), **i32)
            self.ref_count = torch.zeros(max(1, self.n_local), **i32) if exclusive_rows else None
            # --- replicated relation state ----------------------------------------------------------
            self.rel = torch.zeros(rel0.shape[0], st, dtype=dtype, device=dev)
            self.rel[:, :self.dim] = torch.as_tensor(rel0, dtype=dtype, device=dev)
            # relation gradient: mke_oc_apply adds one vector per owned slot to the relation's row, and relation frequencies are
            # heavy-tailed (relation ids ~ Zipf(1.0): apply 7.7 -> 23.2 us as rank 0 of 8, Zipf(1.5): 68 us).  Privatised REL_COPIES
            # ways (slot k adds to copy k % copies; the all-reduce carries the copies, the update sums them) while that stays
            # under 1 MB on the wire — beyond (2K relations x 256 floats) one copy: the all-reduce would cost more than it saves
            copies = 1
            if self.backend.device_type == "cuda" and dtype == torch.float32 and not self.em:    # entity-major: one writer per relation row
                copies = max(1, min(self.REL_COPIES, (1 << 20) // max(1, self.rel.numel() * 4)))
            self.rel_grad = torch.zeros_like(self.rel) if copies == 1 else torch.zeros((copies,) + tuple(self.rel.shape), dtype=dtype, device=dev)
            self.rel_touched = torch.zeros(rel0.shape[0], **i32)
        if ent_table is not None:
            self.ent_acc, self.rel_acc = ent_table.slot(opt_name), rel_table.slot(opt_name)
        else:
            self.ent_acc = placed_rows(self.ent.shape[0], st, dev, ADAGRAD_INIT_ACC, PLACEMENT_LOG, [self.ent]) if self.ent.dtype == torch.float32 \
                else torch.full_like(self.ent, ADAGRAD_INIT_ACC)           # per-optimizer slots (code/MultiKE_model.py:17)
            self.rel_acc = torch.full_like(self.rel, ADAGRAD_INIT_ACC)
        # --- global epoch order (identical on every rank: same seed) ----------------------------------
        if batcher is not None:
            if self.N != 0:
                raise _lib.MultiKEHipError("an explicit epoch source carries positives only: neg_per_pos must be 0")
            self.bat = batcher
        else:
            sides = []
            for k in (0, 1):
                # the sampler's membership filter: `kgs.known[k]` when the caller has one (the reference filters against
                # local_relation_triples_set, which also holds the swapped training-link triples: SURVEY 3.1), else the KG's own
                kt = getattr(kgs, "known", None)
                t = torch.as_tensor(np.asarray(kt[k] if kt is not None else kgs.triples[k], dtype=np.int32).reshape(-1, 3), device=dev)
                known = self.backend.make_known(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())
                sides.append(KGSide(kgs.entities(k), known, device=dev))
            # a global step = `global_batch` positives when given (the reference's batch_size whatever the world size: every
            # rank is home of ceil(global / world) of them, the last of fewer), else batch_size per rank (weak scaling)
            self.bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1],
                                       int(global_batch) if global_batch else batch_size * world, neg_per_pos,
                                       device=dev, seed=seed)
        self.steps = self.bat.steps
        self._declare_hot_rows(ent_table, tables_of)
        # trainers sharing the touched-flag arrays keep apart in tag space (a flag is `touched[row] == tag`)
        self._n_sharing = 0
        if tables_of is not None:
            tables_of._n_sharing += 1
        self.tag = 0 if tables_of is None else (tables_of._n_sharing << 26)
        if tag_base is not None:
            self.tag = int(tag_base)
        self.loss_ring = torch.zeros(max(1, self.steps) * self.chunks, _lib.LOSS_PARTIALS, dtype=torch.float64, device=dev)
        self.score_events = None  # set to a list to collect (start, end, triples) HIP events of the score kernel
        self.C = 0
        self._em_capacity = {}
        self._dtype = dtype
        self._planned_epoch = -1
        self._stepped = -1
        self._parts = None
        self._persistent = {}
        self._plan_epoch()

    def _declare_hot_rows(self, ent_table, tables_of):
        """hot_slot (int32 [n_local]: index among this rank's hub rows or -1), n_hot, and the gradient scratch with the copies
        behind the shard's own rows.  Performance only: any row may or may not be declared (the arithmetic is the same sum)."""
        self.hot_slot, self.n_hot, self.ent_grad_rows = None, 0, max(1, self.n_local)
        if self.em:                                         # entity-major: no entity scratch at all, nothing to privatise
            if ent_table is None and tables_of is None:
                self.ent_grad_full = self.ent_grad = None
            return
        if tables_of is not None:                           # shared scratch: the declaration comes with it
            self.hot_slot, self.n_hot, self.HOT_COPIES = tables_of.hot_slot, tables_of.n_hot, tables_of.HOT_COPIES
            return
        if ent_table is not None:                           # an EmbeddingTable shard: whatever its holder declared
            if ent_table.n_hot:
                self.hot_slot, self.n_hot, self.HOT_COPIES = ent_table.hot_slot, ent_table.n_hot, ent_table.hot_copies
                self.ent_grad = ent_table.grad
            return
        b, G = self.bat, self.world
        n_all = int(b.off[-1]) if self.steps else 0
        if self.device.type == "cuda" and isinstance(self.backend, OcHipBackend) and n_all and self._dtype_is_f32():
            ids = torch.cat([b.pos_h[:n_all], b.pos_t[:n_all]]).long()
            deg = torch.bincount(ids, minlength=self.n_ent).float() / max(1, self.steps)      # references per global step
            hot = torch.nonzero(deg >= self.HOT_MIN).reshape(-1)
            hot = hot[hot % G == self.rank]
            if hot.numel() > self.HOT_MAX:
                hot = hot[torch.topk(deg[hot], self.HOT_MAX).indices]
            if hot.numel():
                slot = torch.full((max(1, self.n_local),), -1, dtype=torch.int32, device=self.device)
                slot[(hot // G)] = torch.arange(hot.numel(), dtype=torch.int32, device=self.device)
                self.hot_slot, self.n_hot = slot, int(hot.numel())
        rows = self.ent_grad_rows + self.HOT_COPIES * self.n_hot
        full = self._mk_rows(0.0, [self.ent, self.ent_acc]) if rows == self.ent.shape[0] else torch.zeros(rows, self.stride, dtype=self.ent.dtype, device=self.device)
        self.ent_grad_full, self.ent_grad = full, full[:self.ent.shape[0]]

    def _dtype_is_f32(self):
        return self.ent.dtype == torch.float32

    # ------------------------------------------------------------------------------------------------
    def parts_of_step(self, s: int):
        """[(lo, hi)] epoch-position ranges of the `chunks` parts of global step s (contiguous, ceil split; empty parts
        are dropped)."""
        lo, hi = int(self.bat.off[s]), int(self.bat.off[s + 1])
        if hi <= lo:
            return []
        size = int(math.ceil((hi - lo) / self.chunks))
        return [(a, min(hi, a + size)) for a in range(lo, hi, size)]

    def my_slice(self, lo: int, hi: int):
        """(per, a, e): positives per home rank in [lo, hi) and this rank's contiguous share [a, e)."""
        per = max(1, int(math.ceil((hi - lo) / self.world)))
        a = min(hi, lo + self.rank * per)
        return per, a, min(hi, a + per)

    def _layout(self):
        """What depends only on the sizes of the epoch (fixed across epochs: a shuffle permutes contents, not the step /
        part / slice boundaries): parts, every rank's slice of every part, this rank's epoch positions, part ids."""
        b, G, dev = self.bat, self.world, self.device
        parts = [(s, lo, hi) for s in range(self.steps) for (lo, hi) in self.parts_of_step(s)]
        self._parts = parts
        self._n_all = int(b.off[-1]) if self.steps else 0
        lo = np.array([p[1] for p in parts], dtype=np.int64)
        hi = np.array([p[2] for p in parts], dtype=np.int64)
        per = np.maximum(1, -(-(hi - lo) // G))
        self._all_idx = torch.arange(self._n_all, dtype=torch.int32, device=dev)
        self._part_id = torch.repeat_interleave(torch.arange(len(parts), device=dev), torch.as_tensor(hi - lo, device=dev)) \
            if len(parts) else torch.zeros(0, dtype=torch.int64, device=dev)
        self._lo_of = torch.as_tensor(lo, device=dev)
        self._part_lo = torch.as_tensor(np.concatenate([lo, [self._n_all]]).astype(np.int64), device=dev)   # [parts + 1]
        self._parts_of = {}
        for k, (ps, _, _) in enumerate(parts):
            self._parts_of.setdefault(ps, []).append(k)
        off = np.asarray(b.off[:self.steps + 1], dtype=np.int64) if self.steps else np.zeros(1, dtype=np.int64)
        self._step_lo = torch.as_tensor(off, device=dev)                       # [steps + 1] epoch positions of the global steps
        self._max_step = int((off[1:] - off[:-1]).max()) if self.steps else 0

    # ---- per-epoch plan: table-independent, so the NEXT epoch's is computed on a side stream while this epoch trains -----
    def _compute_plan(self, pos, rng_stream, bs):
        """The whole plan of an epoch order in line on the current stream: this rank's share of the negatives as codes, the
        all-gather of the codes ON THE STEP COMMUNICATOR, the slots and reference lists."""
        plan = self._plan_sample(pos, rng_stream, bs)
        self._plan_gather(plan)
        return self._plan_rest(plan)

    def _plan_gather(self, plan):
        """The ONE collective of an epoch plan: every rank's 1 / G of the epoch's negative codes, all-gathered on the step
        communicator, on the stream the steps run on, at a fixed point of the step sequence (`_gather_at`) — so the ranks issue
        every collective of the job in one order (round 5 issued it from the side stream on a communicator of its own:
        concurrent collectives on two communicators, with nothing ordering them across ranks)."""
        if "mine" in plan and (self.world > 1 or self.force_collectives):
            if "sampled" in plan:
                torch.cuda.current_stream().wait_event(plan["sampled"])
            self.comm.all_gather(plan["codes_all"], plan["mine"])
            if self.device.type == "cuda":
                plan["gathered"] = torch.cuda.Event()
                plan["gathered"].record()
        plan["stage"] = "gathered"

    def _plan_sample(self, pos, rng_stream, bs):
        """Device work only (no host synchronisation), into buffer set `bs`: the negatives of EVERY positive of the epoch
        (one sampler launch — every rank draws all of them itself: the Philox stream is a function of the epoch position,
        so the ranks agree without exchanging a byte) packed as codes; the slot of every positive's HR / RT vector in its
        owner's block; the owned positives per part in slot order; per (part, owner) counts, copied to pinned host memory
        asynchronously."""
        b, G, dev, N = self.bat, self.world, self.device, self.N
        i32 = dict(dtype=torch.int32, device=dev)
        ph, pr, pt = pos
        n_all, parts, part_id = self._n_all, self._parts, self._part_id
        plan = {"bs": bs}
        # Every rank draws the negatives of ITS contiguous 1 / G of the epoch positions (the Philox stream is a function of
        # the epoch position, so who draws a positive's negatives does not matter), packs them as codes, and ONE all-gather
        # per epoch (`_plan_gather`: on the step communicator, at a fixed point of the step sequence) gives every rank the whole
        # epoch's codes in position order.  (Rounds 2-4: every rank drew all of them — 63 us per step of rank compute at the
        # C5 shape with 8 ranks.)
        n_per = -(-n_all // G) if n_all else 0            # positions per rank (the last rank's share may be shorter)
        codes = self._persist(("codes", bs), torch.zeros(0, **i32), max(1, G * n_per * N))
        if n_all and N:
            lo_r, hi_r = min(n_all, self.rank * n_per), min(n_all, (self.rank + 1) * n_per)
            mine = codes[self.rank * n_per * N:(self.rank + 1) * n_per * N] if G == 1 else \
                self._persist(("codes_mine", bs), torch.zeros(0, **i32), n_per * N)[:n_per * N]
            if hi_r > lo_r:
                # scratch of the sampler's (h, r, t) output: kept across epochs (a fresh allocation per epoch was tens of ms of
                # hipMalloc inside the plan) and BOUNDED — the share is sampled in runs of at most 2^26 / N positions (256 MB per
                # column; the whole share at once was 2.4 GB per column at the C5 shape on one rank: round-4 advice); one plan at a
                # time writes it (plans are computed in epoch order on one stream)
                run = max(1, self.SAMPLE_RUN // N)
                neg = tuple(self._persist(("neg", k_), torch.zeros(0, **i32), min(n_per, run) * N) for k_ in range(3))
                for a in range(lo_r, hi_r, run):
                    e = min(hi_r, a + run)
                    out = tuple(x[:(e - a) * N] for x in neg)
                    self.backend.sample_at((ph[a:e], pr[a:e], pt[a:e]), self._all_idx[a:e], b.pos_kg[a:e],
                                           b.side1, b.side2, N, b.rng_seed, rng_stream, out)
                    self.backend.pack_codes(ph[a:e], out[0], out[2], N, mine[(a - lo_r) * N:(e - lo_r) * N])
            if G > 1 or self.force_collectives:      # (forced at one rank: an in-place all-gather of the whole array)
                plan["codes_all"], plan["mine"] = codes[:G * n_per * N], mine[:n_per * N]
        plan["codes"], plan["pos"], plan["stage"] = codes, pos, "sampled"
        if dev.type == "cuda":
            plan["sampled"] = torch.cuda.Event()
            plan["sampled"].record()
        return plan

    def _plan_rest(self, plan):
        """What follows the codes' all-gather (device work only): slots, owned lists, counts, the entity-major reference lists."""
        b, G, dev, N = self.bat, self.world, self.device, self.N
        i32 = dict(dtype=torch.int32, device=dev)
        ph, pr, pt = plan["pos"]
        bs, codes = plan["bs"], plan["codes"]
        n_all, parts, part_id = self._n_all, self._parts, self._part_id
        if "gathered" in plan:
            torch.cuda.current_stream().wait_event(plan["gathered"])
        # slot of every positive's HR / RT vector in its owner's block (rank among the positives of its part that NEED that
        # vector — the group flags in the first code of every positive — and have the same owner, epoch order; -1 when not
        # needed), this rank's owned positives per part in slot order (part k's list starts at own[lo_k]), and the
        # per-(part, owner) counts.  HIP backend: one launch (mke_oc_plan); other backends (the CPU tests): torch.
        slot = [self._persist(("slot", x, bs), torch.zeros(0, **i32), max(1, n_all)) for x in range(2)]
        own = [self._persist(("own", x, bs), torch.zeros(0, **i32), max(1, n_all)) for x in range(2)]
        plan["slot"], plan["own"] = slot, own
        if n_all:
            cnt = self._persist(("cnt", bs), torch.zeros(0, **i32), 2 * len(parts) * G)
            if hasattr(self.backend, "plan"):
                self.backend.plan(ph, pt, codes, N, self._part_lo, len(parts), G, self.rank, slot[0], slot[1], own[0], own[1], cnt)
            else:
                if N:
                    first = codes[:n_all * N].view(n_all, N)[:, 0].long() & 0xFFFFFFFF
                    needs = ((first & _lib.OC_NEED_HR) != 0, (first & _lib.OC_NEED_RT) != 0)
                else:
                    needs = (torch.ones(n_all, dtype=torch.bool, device=dev), torch.zeros(n_all, dtype=torch.bool, device=dev))
                for x, ids in enumerate((ph, pt)):
                    owner = torch.where(needs[x], ids[:n_all].long() % G, G)        # bucket G: the vector does not travel
                    key = part_id * (G + 1) + owner
                    order = torch.argsort(key, stable=True)
                    ks = key[order]
                    counts = torch.bincount(ks, minlength=len(parts) * (G + 1))
                    start = torch.cumsum(counts, 0) - counts
                    sl = torch.empty(n_all, dtype=torch.int64, device=dev)
                    sl[order] = torch.arange(n_all, device=dev) - start[ks]
                    slot[x][:n_all].copy_(torch.where(needs[x], sl, -1).to(torch.int32))
                    mine = torch.nonzero(owner == self.rank).reshape(-1)
                    lo_m = self._lo_of[part_id[mine]]
                    own[x][(lo_m + sl[mine])] = (mine - lo_m).to(torch.int32)
                    cnt[x * len(parts) * G:(x + 1) * len(parts) * G].copy_(counts.view(len(parts), G + 1)[:, :G].reshape(-1).to(torch.int32))
            c = cnt[:2 * len(parts) * G].view(2, len(parts), G)
            host = self._persistent.get(("cnt_host", bs))
            if host is None or host.shape != c.shape:
                host = torch.empty(c.shape, dtype=torch.int32, pin_memory=dev.type == "cuda")
                self._persistent[("cnt_host", bs)] = host
            host.copy_(c, non_blocking=True)
            plan["cnt_host"] = host
        if self.em:
            plan["em"] = self._compute_em_plan(ph, pr, pt, codes, slot, bs)
        if dev.type == "cuda":
            plan["event"] = torch.cuda.Event()
            plan["event"].record()
        plan["stage"] = "done"
        return plan

    def _em_buffers(self, bs, capacity):
        """Scratch (shared by the two buffer sets: plans are computed one at a time on one stream) and outputs (per buffer set)
        of the entity-major plan at `capacity` references."""
        dev = self.device
        i32, i64 = dict(dtype=torch.int32, device=dev), dict(dtype=torch.int64, device=dev)
        z32, z64 = torch.zeros(0, **i32), torch.zeros(0, **i64)
        out = {"capacity": capacity}
        out["keys"] = self._persist(("em_keys",), z64, capacity + 1)
        out["keys_alt"] = self._persist(("em_keys_alt",), z64, capacity + 1)
        out["flags"] = self._persist(("em_flags",), z32, capacity + 1)
        out["scan"] = self._persist(("em_scan",), z32, capacity + 1)
        out["vals_alt"] = self._persist(("em_vals_alt",), z32, capacity + 1)
        out["scratch8"] = self._persist(("em_scratch8",), z64, capacity + 1)
        out["waves"] = self._persist(("em_waves",), z32, 2 * (_lib.OC_EM_WAVES + 1))
        out["temp"] = self._persist(("em_temp",), torch.zeros(0, dtype=torch.uint8, device=dev), self.backend.em_temp_bytes(capacity))
        out["refs"] = self._persist(("em_refs", bs), z32, 2 * capacity)
        out["rows"] = self._persist(("em_rows", bs), z32, capacity)
        out["off"] = self._persist(("em_off", bs), z32, capacity + 1)
        out["row0"] = self._persist(("em_row0", bs), z64, self.steps + 1)
        # the work items of the second pass (rows, or 32-reference segments of long rows), the long rows and their partial slots
        for k in ("item_row", "item_off", "item_part"):
            out[k] = self._persist(("em_" + k, bs), z32, capacity + 1)
        for k in ("long_row", "long_part0"):
            out[k] = self._persist(("em_" + k, bs), z32, capacity // 32 + 2)
        out["steps3"] = self._persist(("em_steps3", bs), z64, 3 * (self.steps + 1)).view(-1)[:3 * (self.steps + 1)].view(3, self.steps + 1)
        out["n_refs"] = self._persist(("em_n_refs", bs), z64, 1)
        return out

    def _compute_em_plan(self, ph, pr, pt, codes, slot, bs, capacity=None):
        """The entity-major reference lists of the epoch in buffer set `bs` (device work only; the touched-row offsets of the
        steps and the reference count go to pinned host memory asynchronously, read by `_finish_plan`)."""
        G, N = self.world, self.N
        upper = max(1, self._n_all * (N + 5))                          # every element of every positive
        if capacity is None:
            capacity = self._em_capacity.get(bs, 0)
            if not capacity:
                # 1 / G of the epoch's references + 6 % + 4,096: the sort, the gather and the row walks run over the CAPACITY (the unused
                # tail is sentinels), so slack is paid every epoch — uniform corruptions put a rank within 0.1 % of its share, hub
                # entities move only the five non-negative elements of a position; a rank that owns more re-plans once at the exact size
                capacity = upper if G == 1 else min(upper, int(1.06 * self._n_all * (N + 3) / G) + 4096)
        self._em_capacity[bs] = capacity
        bufs = self._em_buffers(bs, capacity)
        self.backend.em_plan(self, ph, pr, pt, codes, slot, bufs)
        S1 = self.steps + 1
        host = self._persistent.get(("em_host", bs))
        if host is None or host.numel() != 4 * S1 + 1:
            host = self._persistent[("em_host", bs)] = torch.empty(4 * S1 + 1, dtype=torch.int64, pin_memory=self.device.type == "cuda")
        host[:S1].copy_(bufs["row0"][:S1], non_blocking=True)
        host[S1:4 * S1].copy_(bufs["steps3"].reshape(-1), non_blocking=True)
        host[4 * S1:].copy_(bufs["n_refs"][:1], non_blocking=True)
        bufs["host"] = host
        return bufs

    def _finish_plan(self, plan):
        """Make a computed plan the current one: wait for its counts (the only host synchronisation of an epoch), size the
        exchange blocks exactly, rebuild the native step descriptors."""
        G, dev = self.world, self.device
        if "event" in plan:
            plan["event"].synchronize()
        parts = self._parts
        self._codes, self._slot, self._own = plan["codes"], plan["slot"], plan["own"]
        self._own_cnt = []                      # per part: how many HR / RT vectors of it this rank owns
        worst = 0
        self.vectors_planned = 0                # HR + RT vectors that travel in this epoch (all owners): ~1 per positive
        for x in range(2):
            mine = np.zeros(len(parts), dtype=np.int64)
            if self._n_all:
                cnt = plan["cnt_host"][x].numpy().reshape(len(parts), G)
                worst = max(worst, int(cnt.max()))
                self.vectors_planned += int(cnt.sum())
                mine = cnt[:, self.rank].astype(np.int64)
            self._own_cnt.append(mine)
        # -- capacity: exact for this epoch, buffers only ever grow ----------------------------------------
        need = max(worst, 1)
        if need > self.C:
            self.C = int(need * 1.05) + 16
            self.block = int(self.backend.block_elems(self.C, self.stride))
            gb = 2 * self.C * self.stride
            mk = lambda n: torch.zeros(n, dtype=self._dtype, device=dev)
            self._send = [mk(self.block) for _ in range(self.chunks)]
            self._v_all = [self._send[c] if G == 1 else mk(G * self.block) for c in range(self.chunks)]
            self._g_all = [mk(G * gb) for _ in range(self.chunks)]
            self._gv = [self._g_all[c] if G == 1 else mk(gb) for c in range(self.chunks)]
            if self.peer_direct:
                self._map_peers(gb)
            self._addr = [tuple(t.data_ptr() for t in (self._send[c], self._v_all[c], self._g_all[c], self._gv[c]))
                          for c in range(self.chunks)]
        if self.em:
            em = plan["em"]
            S1 = self.steps + 1
            n_refs = int(em["host"][4 * S1])
            if n_refs > em["capacity"]:        # more references than the 1 / G estimate allowed for (skewed ownership): re-plan in line, exactly
                b = self.bat
                pos = (b.pos_h, b.pos_r, b.pos_t)
                em = self._compute_em_plan(pos[0], pos[1], pos[2], plan["codes"], plan["slot"], plan["bs"], capacity=int(n_refs * 1.1) + 4096)
                if dev.type == "cuda":
                    torch.cuda.current_stream().synchronize()
                n_refs = int(em["host"][4 * S1])
            em["row0_host"] = em["host"][:S1].clone()
            em["item0_host"], em["long0_host"], em["part0_host"] = (em["host"][(k + 1) * S1:(k + 2) * S1].clone() for k in range(3))
            em["n_refs_host"] = n_refs
            # the long rows' partial sums of ONE step (stride + 16 floats per slot: the gradient vector and the coefficient sum)
            need_parts = int((em["part0_host"][1:] - em["part0_host"][:-1]).max()) if self.steps else 0
            need_parts = max(1, need_parts) * (self.stride + 16)
            if getattr(self, "_em_partials", None) is None or self._em_partials.numel() < need_parts:
                self._em_partials = torch.zeros(need_parts, dtype=torch.float32, device=dev)
            self._em = em
            need_coef = max(1, self._max_step * (self.N + 1))
            if getattr(self, "_em_coef", None) is None or self._em_coef.numel() < need_coef:
                self._em_coef = torch.zeros(need_coef, dtype=torch.float32, device=dev)
        self._planned_epoch = self.bat.epoch
        self._plan_bs = plan["bs"]
        self._st_cache = {}
        if hasattr(self.backend, "prepare_epoch"):
            self.backend.prepare_epoch(self)

    def _plan_epoch(self):
        """Plan of the CURRENT epoch order, in line (construction, or an epoch boundary without a prefetched plan)."""
        if getattr(self, "_parts", None) is None:
            self._layout()
        b = self.bat
        self._finish_plan(self._compute_plan((b.pos_h, b.pos_r, b.pos_t), b.rng_stream, 0))
        self._next_plan = None

    def _prefetch_next_epoch(self):
        """Draw the next epoch's permutation into the batcher's alternate buffers and compute its plan on a side stream (the
        sampler, two sorts): by the time the epoch ends it is waiting in the other buffer set."""
        b = self.bat
        nxt = ((b.epoch + 1) * 2) & 0xFFFFFFFF
        bs = 1 - getattr(self, "_plan_bs", 0)
        split = self.world > 1 or self.force_collectives
        first = self._plan_sample if split else self._compute_plan               # G > 1: the collective waits for `_gather_at`
        if self.device.type != "cuda":
            self._next_plan = first(b.stage_next_epoch(), nxt, bs)
            return
        # the permutation too goes to the side stream (two device sorts: ~190 us per epoch at the C2 shape — 8 us per global step
        # of an 8-rank epoch when it sat on the steps' stream)
        self._on_side(lambda: setattr(self, "_next_plan", first(b.stage_next_epoch(), nxt, bs)))

    def _on_side(self, fn):
        """Run fn with the plan's side stream current (and pinned for the native calls), ordered after the current stream."""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            old = _lib.pin_stream(self._side.cuda_stream)
            try:
                fn()
            finally:
                _lib.pin_stream(old)

    @property
    def _gather_at(self):
        """The step of an epoch before which the NEXT epoch's codes are all-gathered: an eighth into the epoch — the sampling of a
        rank's share is short (0.2 ms at the C2 shape with 8 ranks, 1.5 ms at C5) and the lists that follow the gather (1 - 12 ms)
        want the rest of the epoch, or the host waits for them at the boundary with the GPU idle (measured at C2 with 8 ranks and
        the gather mid-epoch: 117 us per global step for 65 us of kernels)."""
        return min(self.steps - 1, max(1, self.steps // 8)) if self.steps > 1 else 0

    def _plan_midpoint(self):
        """At step `_gather_at` of every epoch, on every rank: the prefetched plan's collective (main stream, step communicator),
        then the rest of the plan back on the side stream."""
        plan = getattr(self, "_next_plan", None)
        if plan is None or plan.get("stage") != "sampled":
            return
        self._plan_gather(plan)
        if self.device.type != "cuda":
            self._plan_rest(plan)
        else:
            self._on_side(lambda: self._plan_rest(plan))

    def _advance_epoch(self):
        """Epoch boundary: random.shuffle of both positive lists (code/MultiKE_model.py:314-315) + the new epoch's plan."""
        if getattr(self, "_next_plan", None) is not None:
            self._plan_midpoint()                            # (an epoch of one step: its midpoint is the boundary itself)
            plan, self._next_plan = self._next_plan, None
            if "event" in plan:
                torch.cuda.current_stream().wait_event(plan["event"])
            self.bat.commit_staged()
            self._finish_plan(plan)
        else:
            self.bat.shuffle()
            self._plan_epoch()

    def set_neighbours(self, tables):
        """Truncated negative sampling (code/MultiKE_CSL.py:89-102): install ((cand_table, cand_valid), (...)) for the two KGs
        (None = back to uniform) — identical on every rank.  Takes effect with the NEXT epoch, as in the reference (the refresh
        sits at the end of an epoch): a plan of that epoch prefetched with the old candidates is dropped."""
        for side, nb in ((self.bat.side1, tables[0]), (self.bat.side2, tables[1])):
            side.set_neighbours(*(nb if nb is not None else (None, None)))
        if getattr(self, "_next_plan", None) is not None:
            if getattr(self, "_side", None) is not None:
                torch.cuda.current_stream().wait_stream(self._side)    # the dropped plan may still be writing its buffer set
            self._next_plan = None

    def _map_peers(self, gb):
        """Exchange IPC handles of this rank's send block and gradient inbox ([world][2 C][stride]: one slice per writer) and
        map every peer's (torch's CUDA-IPC tensor reductions: hipIpcGetMemHandle / hipIpcOpenMemHandle underneath)."""
        from torch.multiprocessing.reductions import reduce_tensor
        G = self.world
        self._inbox = torch.zeros(G * gb, dtype=self._dtype, device=self.device)
        self._gv = [self._inbox]
        self._v_all = [self._send[0]]                      # unused in peer mode (kept non-null for the address table)
        self._g_all = [self._inbox]
        mine = (reduce_tensor(self._send[0]), reduce_tensor(self._inbox))
        every = self.comm.all_gather_object(mine)
        self._peer_send, self._peer_inbox = [], []
        for g, ((f1, a1), (f2, a2)) in enumerate(every):
            self._peer_send.append(self._send[0] if g == self.rank else f1(*a1))
            self._peer_inbox.append(self._inbox if g == self.rank else f2(*a2))
        self._bar = torch.zeros(1, dtype=torch.float32, device=self.device)

    def _persist(self, key, value, capacity):
        """Epoch buffers keep their device addresses across epochs (the native step descriptors point into them)."""
        buf = self._persistent.get(key)
        if buf is None or buf.numel() < max(capacity, value.numel(), 1):
            buf = self._persistent[key] = torch.zeros(max(capacity, value.numel(), 1), dtype=value.dtype, device=value.device)
        buf[:value.numel()].copy_(value)
        return buf

    def global_scored(self, i: int) -> int:
        s = i % self.steps
        return int(self.bat.off[s + 1] - self.bat.off[s]) * (1 + self.N)

    def _part_step(self, k: int, tag: int) -> OcStep:
        st = self._st_cache.get(k)
        if st is not None:
            st.tag = tag
            return st
        st = self._st_cache[k] = self._build_part_step(k, tag)
        return st

    def _build_part_step(self, k: int, tag: int) -> OcStep:
        _, lo, hi = self._parts[k]
        b = self.bat
        per, _, _ = self.my_slice(lo, hi)
        nh, nt = int(self._own_cnt[0][k]), int(self._own_cnt[1][k])
        code_off = tuple((lo + g * per) * self.N for g in range(self.world))   # codes are laid out by epoch position
        pw = getattr(b, "pos_w", None)
        return OcStep(b.pos_h[lo:hi], b.pos_r[lo:hi], b.pos_t[lo:hi], per, self._slot[0][lo:hi], self._slot[1][lo:hi],
                      self._own[0][lo:lo + nh], self._own[1][lo:lo + nt], tag, self._codes, code_off,
                      pos_w=(pw[lo:hi] if pw is not None else None))

    def _exchange_tensors(self):
        return [*self._send, *self._v_all, *self._g_all, *self._gv, self.rel_grad]

    def _native_loop(self):
        """(usable, mke_oc_comm or None): the step loop can go through mke_oc_steps — HIP backend, a communicator with a native
        form (RCCL through ctypes, the tests' host-staged ranks, the tools' loop-back) or a single rank, no peer-direct, no
        per-launch event collection.  MKE_OC_NATIVE=0 keeps the Python loop."""
        import os
        if not hasattr(self.backend, "run_steps") or self.peer_direct or self.score_events is not None or self.chunks > _lib.OC_EM_MAX_CHUNKS \
                or os.environ.get("MKE_OC_NATIVE", "1") == "0":
            return False, None
        if self.world == 1 and not self.force_collectives:
            return True, None
        if not hasattr(self.comm, "native"):
            return False, None
        key = (self.C, self._send[0].data_ptr())
        if getattr(self, "_comm_native_key", None) != key:      # callbacks look the exchange buffers up by address
            self._comm_native, self._comm_native_key = self.comm.native(self), key
        return True, self._comm_native

    OVERLAP_RS_MIN_BYTES = 16 << 20
    # the update rule of the step descriptors: the reference's relation view trains with Adagrad (code/MultiKE_model.py:17-31) and the
    # multi-GPU drivers accept nothing else (distributed_run.py); plain SGD (code/MultiKE_model.py:24) is the kernels' other rule,
    # reachable through the C-ABI — a subclass sets it for the tests
    OPTIMIZER = _lib.OPT_ADAGRAD

    def _overlap_rs(self) -> bool:
        """Entity-major, one part per step: put the reduce-scatter on the communication stream and run, under it, the second pass's
        work items that do not need its result (at 8 ranks ~95 % of the rows: the corrupt entities) — two stream hops per step
        (~26 us), so only when the reduce-scatter is long: >= 16 MB received per rank (the C5 shape at 8 ranks: 40 MB = 122 us in the
        link model; C2: 12 MB = 48 us, not worth the hops).  MKE_OC_OVERLAP_RS=0 / 1 forces it."""
        import os
        if not self.em or self.chunks != 1 or self.world < 2 and not self.force_collectives:
            return False
        env = os.environ.get("MKE_OC_OVERLAP_RS")
        if env is not None:
            return env == "1"
        return (self.world - 1) * 2 * self.C * self.stride * 4 >= self.OVERLAP_RS_MIN_BYTES

    def run(self, i0: int, n: int):
        """Global steps i0 .. i0 + n - 1 (in order): one native call per run of steps inside an epoch (mke_oc_steps) when the
        communicator has a native form, else the Python step loop."""
        i, end = i0, i0 + n
        while i < end:
            s = i % self.steps
            m = min(end - i, self.steps - s)
            ok, cs = self._native_loop()
            if not ok:
                for k in range(m):
                    self.step(i + k)
                i += m
                continue
            if s == 0 and i > 0:
                self._advance_epoch()
            if s == 0 and self.prefetch:
                self._prefetch_next_epoch()                  # the next epoch's plan overlaps this epoch's steps
            k = self._gather_at
            if s == k:
                self._plan_midpoint()
            elif s < k < s + m and getattr(self, "_next_plan", None) is not None and self._next_plan.get("stage") == "sampled":
                m = k - s                                    # stop at the epoch's gather point: the collective goes between two steps
            ok, cs = self._native_loop()                     # the exchange buffers may have grown with the new epoch's plan
            comm_stream = None
            overlap = self._overlap_rs() if cs is not None else False
            if cs is not None and (self.chunks > 1 or overlap) and self.device.type == "cuda":
                if getattr(self, "_comm_stream", None) is None:
                    self._comm_stream = torch.cuda.Stream(device=self.device)
                comm_stream = self._comm_stream.cuda_stream
            self.backend.run_steps(self, s, s + m, self.tag, cs, comm_stream, overlap)
            self.tag += m
            i += m
            self._stepped = i - 1

    def step(self, i: int):
        """Global step i (steps must be issued in order)."""
        s = i % self.steps
        if s == 0 and i > 0:
            self._advance_epoch()
        if s == 0 and self.prefetch:
            self._prefetch_next_epoch()                      # the next epoch's plan overlaps this epoch's steps
        if s == self._gather_at:
            self._plan_midpoint()
        be, G, cm = self.backend, self.world, self.comm
        ks = self._parts_of.get(s, [])
        self.tag += 1
        tag = self.tag
        ev = self.score_events
        slot0 = s * self.chunks
        if self.em:
            self._step_em(s, ks, tag, slot0)
            self._stepped = i
            return
        if G == 1 and not self.force_collectives:  # every row is local: no collective between the phases
            last = len(ks) - 1
            if last == 0 and ev is None:
                be.run(self, ks[0], tag, BASES | COUNT | SCORE | APPLY | UPDATE, 0, slot0)
            else:
                for c, k in enumerate(ks):
                    be.run(self, k, tag, BASES | COUNT, c, slot0 + c)       # counts of ALL parts before any is scored
                for c, k in enumerate(ks):
                    if ev is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    be.run(self, k, tag, SCORE, c, slot0 + c)
                    if ev is not None:
                        e1.record()
                        ev.append((e0, e1, (self._parts[k][2] - self._parts[k][1]) * (1 + self.N)))
                for c, k in enumerate(ks):
                    be.run(self, k, tag, APPLY | (UPDATE if c == last else 0), c, slot0 + c)
            self._stepped = i
            return
        if self.peer_direct:
            for k in ks:
                be.run(self, k, tag, BASES | COUNT, 0, slot0)
                cm.barrier(self._bar)                        # every rank's vectors are in its send block
                be.run(self, k, tag, SCORE, 0, slot0)        # reads peers' blocks, writes its slice of peers' inboxes
                cm.barrier(self._bar)                        # every writer's slice of every inbox is complete
                be.run(self, k, tag, APPLY, 0, slot0)
            cm.all_reduce(self.rel_grad)
            if ks:
                be.run(self, ks[-1], tag, UPDATE, 0, slot0)
            self._stepped = i
            return
        pipelined = len(ks) > 1 and self.device.type == "cuda"
        works = {}
        # ---- HR / RT vectors of every part, all-gathered (asynchronously when pipelining) -------------------------
        # The reference counts over the WHOLE global step (all parts) are complete before any part is scored; they need
        # only the epoch's codes and ride on blocks of the bases launch (one kernel boundary less per part, and nothing
        # of theirs left behind the all-gather)
        for c, k in enumerate(ks):
            be.run(self, k, tag, BASES | (COUNT if self.ref_count is not None else 0), c, slot0 + c)
            works[("ag", c)] = cm.all_gather(self._v_all[c], self._send[c], async_op=pipelined)
        # ---- score part c while part c+1's all-gather / part c-1's reduce-scatter are on the wire ------------------
        for c, k in enumerate(ks):
            if works.get(("ag", c)) is not None:
                works[("ag", c)].wait()
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            be.run(self, k, tag, SCORE, c, slot0 + c)
            if ev is not None:
                e1.record()
                ev.append((e0, e1, (self._parts[k][2] - self._parts[k][1]) * (1 + self.N) // G))
            works[("rs", c)] = cm.reduce_scatter(self._gv[c], self._g_all[c], async_op=pipelined)
        for c, k in enumerate(ks):
            if works.get(("rs", c)) is not None:
                works[("rs", c)].wait()
            be.run(self, k, tag, APPLY, c, slot0 + c)
        # ---- replicated relation table: all-reduce the (small) dense gradient; one update of everything ---------------
        cm.all_reduce(self.rel_grad)
        if ks:
            be.run(self, ks[-1], tag, UPDATE, 0, slot0)
        self._stepped = i

    def _step_em(self, s, ks, tag, slot0):
        """Global step s in the entity-major form: per part  bases -> ALL-GATHER -> score (one coefficient per owned negative,
        the partial gradient vectors) -> REDUCE-SCATTER;  then ONE second pass over the touched owned rows of the whole step
        (mke_oc_pass2: finishes the entity rows in place, stores this rank's partial relation gradient), the relation gradient's
        ALL-REDUCE and the relation table's update."""
        be, G, cm, ev = self.backend, self.world, self.comm, self.score_events
        if not ks:
            return
        last = len(ks) - 1

        def score(c, k, share):
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            be.run(self, k, tag, SCORE, c, slot0 + c)
            if ev is not None:
                e1.record()
                ev.append((e0, e1, (self._parts[k][2] - self._parts[k][1]) * (1 + self.N) // share))

        if G == 1 and not self.force_collectives:        # every row is local: no collective between the phases
            if last == 0 and ev is None:
                be.run(self, ks[0], tag, BASES | SCORE | PASS2 | UPDATE, 0, slot0)
                return
            for c, k in enumerate(ks):
                be.run(self, k, tag, BASES, c, slot0 + c)
                score(c, k, 1)
            be.run(self, ks[-1], tag, PASS2 | UPDATE, 0, slot0)
            return
        pipelined = len(ks) > 1 and self.device.type == "cuda"
        works = {}
        for c, k in enumerate(ks):
            be.run(self, k, tag, BASES, c, slot0 + c)
            works[("ag", c)] = cm.all_gather(self._v_all[c], self._send[c], async_op=pipelined)
        for c, k in enumerate(ks):
            if works.get(("ag", c)) is not None:
                works[("ag", c)].wait()
            score(c, k, G)
            works[("rs", c)] = cm.reduce_scatter(self._gv[c], self._g_all[c], async_op=pipelined)
        for c in range(len(ks)):
            if works.get(("rs", c)) is not None:
                works[("rs", c)].wait()
        be.run(self, ks[-1], tag, PASS2, 0, slot0)              # every part's vectors and gradient vectors are in place
        cm.all_reduce(self.rel_grad)                            # this rank's partial relation gradient, stored by the second pass
        be.run(self, ks[-1], tag, UPDATE, 0, slot0)

    # ------------------------------------------------------------------------------------------------
    def check(self) -> dict:
        """Capacity is fixed per epoch from the data before the epoch runs (`_plan_epoch`): nothing to flag."""
        out = {"capacity_vectors_per_owner": self.C, "block_bytes": self.block * 4, "chunks": self.chunks,
               "vectors_per_positive": self.vectors_planned / max(1, self._n_all),
               "entity_major": bool(self.em), "native_step_loop": bool(self._native_loop()[0]),
               "reduce_scatter_under_second_pass": bool(self._overlap_rs()), "communicator": type(self.comm).__name__}
        if self.em:
            out["references_per_global_step"] = self._em["n_refs_host"] / max(1, self.steps)
            out["long_rows_per_global_step"] = int(self._em["long0_host"][-1]) / max(1, self.steps)
        return out

    def scratch_clean(self) -> bool:
        """The zero invariants between steps: gradient scratch all zero, reference counts all zero (the entity-major form has
        no entity scratch and no counts: only the relation gradient)."""
        ok = float(self.rel_grad.abs().max()) == 0.0
        g = getattr(self, "ent_grad_full", None)
        g = g if g is not None else self.ent_grad
        if g is not None:
            ok = ok and float(g.abs().max()) == 0.0
        if self.ref_count is not None:
            ok = ok and int(self.ref_count.abs().sum()) == 0
        return ok

    def gather_entity_table(self) -> torch.Tensor:
        """Reassemble the full [n_ent, dim] raw table on every rank (tests / checkpoint)."""
        pad = int(math.ceil(self.n_ent / self.world))
        mine = torch.zeros(pad, self.stride, dtype=self.ent.dtype, device=self.device)
        mine[:self.n_local] = self.ent[:self.n_local]
        if self.world == 1:
            return mine[:self.n_local, :self.dim].clone()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.comm.all_gather_list(parts, mine)
        full = torch.zeros(self.n_ent, self.dim, dtype=self.ent.dtype, device=self.device)
        for r in range(self.world):
            n = len(range(r, self.n_ent, self.world))
            full[r::self.world] = parts[r][:n, :self.dim]
        return full

    def epoch_loss(self) -> float:
        t = self.loss_ring.sum()
        if self.world > 1:
            self.comm.all_reduce(t)
        self.loss_ring.zero_()
        return float(t)
