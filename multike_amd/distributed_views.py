"""The other views of MultiKE on several GPUs (new design, the reference is single-device).

Every entity table (`rv_ent_embeds`, `av_ent_embeds`, `ent_embeds`, the constant `name_embeds`) is row-sharded the same
way, id % world (local row = id // world); `attr_embeds`, the literal table and the CNN / mapping parameters are small and
replicated.  Two consequences:

* **Common-space learning** (code/MultiKE_model.py:225-239, 458-473): every term is between rows of the SAME entity in
  different tables, so each rank trains the sampled entities it owns and nothing crosses the links but the reported loss
  (`ShardedCommonSpace`).
* **Attribute view** (code/MultiKE_model.py:134-151, 319-345; `conv` :34-63): a triple (h, a, v, w) is trained by the
  owner of h — its `av_ent_embeds` row is local, the attribute / literal rows and the CNN are replicated — i.e. plain data
  parallelism over triples.  What couples the ranks is the batch-wide `tf.nn.l2_normalize` of the CNN output (:60,
  "important!!"): sum z^2 over the WHOLE batch in the forward and sum g.z in the backward — one scalar all-reduce each —
  and the gradients of the replicated parameters (CNN pack ~91 KB, attribute table) — one all-reduce each before the
  identical update on every rank (`ShardedAttributeView`; native phases: `mke_attr_step_phases`).

* **Space mapping** (SSL driver; code/losses.py:53-63, code/MultiKE_model.py:241-261, 439-454): the sampled entities' rows of
  the shared table and of the three view tables are local to the entity's owner; the three d x d mapping matrices are
  replicated.  Per view the mapped batch is normalised as a WHOLE (`tf.nn.l2_normalize` without an axis, losses.py:55):
  sum P_k^2 in the forward and sum G_k . out_k in its backward — one all-reduce of three scalars each — and the data part of
  the matrix gradients — one all-reduce of 3 d^2 floats; the orthogonality / norm terms depend on the replicated matrices only
  and are added after that reduction, identically on every rank (`ShardedSpaceMapping`; native phases:
  `mke_mapping_step_phases`).

* **Literal auto-encoder** (code/literal_encoder.py:41-112): plain data parallelism over the rows of a batch (rank r takes
  rows r, r + world, ...), parameters replicated.  The code matrix of a batch is normalised as a WHOLE (:65-66): sum code^2
  in the forward and sum dcn . code in its backward — one scalar all-reduce each — then the all-reduce of the packed
  gradient buffer (4.3 M floats at 1500-1024-512-75) and the identical update on every rank (`ShardedAutoEncoder`; native
  phases: `mke_ae_step_phases`).

Compute goes through a backend object (HIP kernels in production; tests inject a NumPy backend built on the oracle to run
this logic under gloo with world_size 2).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .tables import EmbeddingTable, StepEngine


class ViewComm:
    def all_reduce(self, t):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(t)


def _bcast(self, t, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src)


ViewComm.broadcast = _bcast


class NoComm(ViewComm):
    """world == 1: a view's collectives are no-ops whatever process group the process is part of (a one-rank view may live
    inside a multi-rank job: the self-test's reference run, a replicated small model)."""

    def all_reduce(self, t):
        pass

    def broadcast(self, t, src=0):
        pass


def _comm_for(world, comm, default=None):
    return NoComm() if world == 1 else (comm or default or ViewComm())


class HostStagedViewComm(ViewComm):
    """Test vehicle: two ranks sharing one GPU, collectives staged through gloo over the host."""

    def all_reduce(self, t):
        c = t.cpu()
        dist.all_reduce(c)
        t.copy_(c)

    def broadcast(self, t, src=0):
        c = t.cpu()
        dist.broadcast(c, src)
        t.copy_(c)


def _gather_device():
    """Where the shards meet in gather(): the GPU under RCCL ("nccl" moves device tensors only), the host under gloo."""
    return torch.device("cuda") if (dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")


def _owned(ids: np.ndarray, rank: int, world: int):
    """(positions of the batch this rank trains, their local rows)."""
    ids = np.asarray(ids, dtype=np.int64)
    pos = np.nonzero(ids % world == rank)[0]
    return pos, ids[pos] // world


# ======================================================================================================================
# Attribute view
# ======================================================================================================================
class HipAttrBackend:
    device_type = "cuda"

    def __init__(self, view: "ShardedAttributeView", ent0, attr0, lit, cnn_params, tables_of=None, tables=None):
        from .attr_cnn import AttrCNN
        d = view.dim
        if tables is not None:      # (av_ent shard, attr, literal) EmbeddingTables held by the caller (multike_amd.distributed_model)
            self.ent, self.attr, self.lit = tables
        elif tables_of is not None:   # another graph of the same view: the same tables (their Adagrad slots are per optimizer name)
            self.ent, self.attr, self.lit = tables_of.ent, tables_of.attr, tables_of.lit
        else:
            self.ent = EmbeddingTable(max(1, len(ent0)), d, "av_ent_embeds", normalize=True, values=ent0 if len(ent0) else np.zeros((1, d)))
            self.attr = EmbeddingTable(attr0.shape[0], d, "attr_embeds", normalize=False, values=attr0)
            self.lit = EmbeddingTable(lit.shape[0], d, "literal_embeds", normalize=False, trainable=False, values=lit)
        self.cnn = AttrCNN(d, params=cnn_params)
        self.eng = StepEngine()
        self.loss = torch.zeros((), dtype=torch.float64, device="cuda")

    def _i32(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device="cuda")

    def stage(self, lh, ia, iv, w):
        """Host arrays of this rank's triples (one step or a whole epoch) -> device (one copy for the three id columns)."""
        wt = None if w is None else torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32), device="cuda")
        ids = self._i32(np.stack([np.asarray(lh), np.asarray(ia), np.asarray(iv)]).reshape(3, -1))
        return ids[0], ids[1], ids[2], wt

    def forward(self, view, lh, ia, iv, w, scale, staged=None):
        """Conv stack + dense layer on this rank's triples; returns the device scalar sum z^2 of its part.  staged: the
        triples as device tensors (slices of an epoch staged once) instead of the host arrays."""
        self._keep = staged if staged is not None else self.stage(lh, ia, iv, w)
        self.args, self.part = self.cnn._args(self.eng, self.ent, self.attr, self.lit, *self._keep, int(self._keep[0].numel()), scale,
                                              view.opt_name, view.lr, "Adagrad", True, 1)
        _lib.attr_step_phases(self.args, _lib.ATTR_FWD)
        return self._scalar(1)

    def step_alone(self, view, lh, ia, iv, w, scale, staged=None):
        """world == 1: nothing separates the phases — the whole step as ONE native call (`mke_attr_step`)."""
        h, a, v, wt = staged if staged is not None else self.stage(lh, ia, iv, w)
        if h.numel() == 0:
            return
        self.loss += self.cnn.step(self.eng, self.ent, self.attr, self.lit, h, a, v, wt, scale=scale, opt_name=view.opt_name,
                                   lr=view.lr).sum()

    def _scalar(self, k):
        LP = _lib.LOSS_PARTIALS
        return self.part[k * LP:(k + 1) * LP].sum().reshape(1)

    def _set_scalar(self, k, v):
        LP = _lib.LOSS_PARTIALS
        seg = self.part[k * LP:(k + 1) * LP]
        seg.zero_()
        seg[0:1] = v

    def tail(self, view, S):
        self._set_scalar(1, S)
        _lib.attr_step_phases(self.args, _lib.ATTR_TAIL)
        self.loss += self.part[:_lib.LOSS_PARTIALS].sum()
        return self._scalar(2)

    def backward(self, view, T):
        self._set_scalar(2, T)
        _lib.attr_step_phases(self.args, _lib.ATTR_BWD)
        return [self.cnn.grads, self.attr.grad]

    def update(self, view):
        self.args.attr_touched = None          # after the all-reduce a row may carry a gradient no local triple touched
        _lib.attr_step_phases(self.args, _lib.ATTR_UPD)

    def scalar_like(self):
        return torch.zeros(1, dtype=torch.float64, device="cuda")

    # --- replicated-compute step (ShardedAttributeView mode="replicated") -------------------------------------------------
    def gather_heads(self, view, pos, lh, n):
        """[n, dim] float32: the RAW rows of the batch's head entities at the positions this rank owns, zero elsewhere (the
        all-reduce that follows assembles the whole batch's heads on every rank: adding zeros is exact)."""
        out = torch.zeros(n, view.dim, dtype=torch.float32, device="cuda")
        if len(pos):
            out[torch.as_tensor(pos, device="cuda")] = self.ent.raw()[torch.as_tensor(np.asarray(lh, dtype=np.int64), device="cuda")]
        return out

    def replicated_step(self, view, H, ia, iv, w, scale):
        """The WHOLE batch's forward + backward on this rank, on a compact table of the gathered head rows (read through
        l2_normalize like the shard).  Leaves the heads' gradient rows in the compact table's scratch and returns the packed
        [CNN parameter gradients | attribute-table gradient] buffer (a view into it is what `apply_replicated` consumes)."""
        n, d = H.shape
        if getattr(self, "_cent", None) is None or self._cent.n_rows < n:
            self._cent = EmbeddingTable(max(n, 1), d, "batch_heads", normalize=True)
        ct = self._cent
        ct.data[:n, :d] = H
        idx = torch.arange(n, dtype=torch.int32, device="cuda")
        ia_, iv_ = self._i32(ia), self._i32(iv)
        wt = None if w is None else torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32), device="cuda")
        self._keep = (idx, ia_, iv_, wt)
        self.args, self.part = self.cnn._args(self.eng, ct, self.attr, self.lit, idx, ia_, iv_, wt, n, scale, view.opt_name, view.lr,
                                              "Adagrad", True, 1)
        self.args.update = 0
        _lib.attr_step_phases(self.args, _lib.ATTR_FWD | _lib.ATTR_TAIL | _lib.ATTR_BWD)
        if view.rank == 0:                       # every rank computed the whole batch's loss: count it once
            self.loss += self.part[:_lib.LOSS_PARTIALS].sum()
        self._n_rep = n
        return torch.cat([self.cnn.grads, self.attr.grad.reshape(-1)])

    def apply_replicated(self, view, pos, lh, canon):
        """canon: rank 0's [parameter | attribute] gradients (identical updates of the replicated state on every rank); the
        heads this rank owns take their gradient rows from its own (bit-wise equally good) copy of the batch's."""
        ct, n = self._cent, self._n_rep
        npar = self.cnn.grads.numel()
        self.cnn.grads.copy_(canon[:npar])
        self.attr.grad.copy_(canon[npar:].view_as(self.attr.grad))
        a = self.args
        if len(pos):
            rows = ct.grad[torch.as_tensor(pos, device="cuda")].contiguous()
            _lib.rows_scatter_add(self._i32(lh), rows, view.dim, self.ent.grad, self.ent.touched, a.tag)
        ct.grad[:n].zero_()
        f32, i32 = torch.float32, torch.int32
        a.ent_table, a.n_ent = _lib.ptr(self.ent.data, f32, "ent"), self.ent.n_rows
        a.ent_acc = _lib.ptr(self.ent.slot(view.opt_name), f32, "acc")
        a.ent_grad, a.ent_touched = _lib.ptr(self.ent.grad, f32, "grad"), _lib.ptr(self.ent.touched, i32, "touched")
        a.attr_touched = None
        a.update = 1
        _lib.attr_step_phases(a, _lib.ATTR_UPD)

    def tables(self):
        return (self.ent.raw().cpu().numpy(), self.attr.raw().cpu().numpy(), self.cnn.numpy_params())

    def take_loss(self):
        v = self.loss.reshape(1).clone()
        self.loss.zero_()
        return v


class ShardedAttributeView:
    def __init__(self, ent0: np.ndarray, attr0: np.ndarray, lit: np.ndarray, cnn_params: dict, rank: int, world: int,
                 lr: float = 0.001, opt_name: str = "attribute", backend_cls=None, comm=None, tables_of: "ShardedAttributeView" = None,
                 tables=None, n_ent: int = None, mode: str = None):
        """tables_of: another attribute graph of the same run (code/MultiKE_model.py:134-151, 153-190: the attribute view and
        the two cross-KG attribute-inference graphs share `av_ent_embeds` / `attr_embeds` and have a CNN parameter set and an
        optimizer each): this one trains ITS tables — pass a different `opt_name`; `ent0` / `attr0` / `lit` are then unused."""
        self.rank, self.world, self.lr, self.opt_name = rank, world, float(lr), opt_name
        # mode "parallel" (default): each rank trains the triples whose head it owns — 2 scalar all-reduces + 2 gradient
        #   all-reduces per step.  mode "replicated": the batch's head rows are assembled on every rank (1 all-reduce of
        #   [B, dim]), EVERY rank computes the whole step (at 5000 triples the step is a latency chain: a rank's 1 / world share
        #   costs what the whole batch costs), rank 0's parameter / attribute-table gradients are broadcast
        #   (replicas stay bit-identical: each rank's own sums differ in the order of their fp32 atomics) and every rank
        #   updates the head rows it owns: 2 collectives per step instead of 4.  MKE_ATTR_MODE overrides.
        import os
        self.mode = mode or os.environ.get("MKE_ATTR_MODE", "parallel")
        if self.mode not in ("parallel", "replicated"):
            raise ValueError(f"ShardedAttributeView: mode {self.mode!r}")
        if tables is not None:      # EmbeddingTables of the caller: (this rank's av_ent shard of n_ent rows, attr, literal)
            self.dim, self.n_ent = tables[0].dim, int(n_ent)
            self.comm = _comm_for(world, comm)
            self.backend = (backend_cls or HipAttrBackend)(self, None, None, None, cnn_params, tables=tables)
            return
        if tables_of is not None:
            if (tables_of.rank, tables_of.world) != (rank, world) or tables_of.opt_name == opt_name:
                raise ValueError("tables_of: same rank / world and a different optimizer name")
            self.dim, self.n_ent = tables_of.dim, tables_of.n_ent
            self.comm = _comm_for(world, comm, tables_of.comm)
            self.backend = (backend_cls or type(tables_of.backend))(self, None, None, None, cnn_params, tables_of=tables_of.backend)
            return
        self.dim = ent0.shape[1]
        self.n_ent = ent0.shape[0]
        self.comm = _comm_for(world, comm)
        self.backend = (backend_cls or HipAttrBackend)(self, ent0[rank::world], attr0, lit, cnn_params)

    def step(self, ih, ia, iv, w=None, scale: float = 1.0):
        """One `session.run([loss, optimizer])` of an attribute-view graph on the GLOBAL batch (ih, ia, iv, w: host arrays,
        identical on every rank): this rank trains the triples whose head it owns."""
        be, cm = self.backend, self.comm
        pos, lh = _owned(ih, self.rank, self.world)
        ia_all, iv_all, w_all = ia, iv, w
        ia, iv = np.asarray(ia)[pos], np.asarray(iv)[pos]
        w = None if w is None else np.asarray(w)[pos]
        if self.world == 1 and hasattr(be, "step_alone"):
            be.step_alone(self, lh, ia, iv, w, scale)
            return
        if self.mode == "replicated" and self.world > 1:
            self._replicated_step(np.asarray(ih), pos, lh, np.asarray(ia_all), np.asarray(iv_all), w_all, scale)
            return
        S = be.forward(self, lh, ia, iv, w, scale)
        cm.all_reduce(S)                                   # sum z^2 over the whole batch (code/MultiKE_model.py:60)
        T = be.tail(self, S)
        cm.all_reduce(T)                                   # sum g.z over the whole batch (its backward)
        for g in be.backward(self, T):
            cm.all_reduce(g)                               # replicated parameters: CNN pack, attribute table
        be.update(self)

    def _replicated_step(self, ih, pos, lh, ia, iv, w, scale):
        be, cm = self.backend, self.comm
        H = be.gather_heads(self, pos, lh, len(ih))
        cm.all_reduce(H)                                   # every rank: the raw rows of all the batch's heads
        canon = be.replicated_step(self, H, ia, iv, w, scale)
        cm.broadcast(canon, 0)                             # rank 0's sums are everybody's: replicas stay bit-identical
        be.apply_replicated(self, pos, lh, canon)

    def steps(self, ih, ia, iv, w, step_off, scale: float = 1.0):
        """Consecutive steps over epoch-ordered host arrays (step s = positions [step_off[s], step_off[s + 1])): what `step`
        does per step, with this rank's triples of ALL steps filtered and copied to the device once."""
        be = self.backend
        if not hasattr(be, "stage") or (self.mode == "replicated" and self.world > 1):
            if isinstance(ih, torch.Tensor):
                ih, ia, iv = (t.cpu().numpy() for t in (ih, ia, iv))
                w = None if w is None else w.cpu().numpy()
            for s in range(len(step_off) - 1):
                lo, hi = int(step_off[s]), int(step_off[s + 1])
                self.step(ih[lo:hi], ia[lo:hi], iv[lo:hi], None if w is None else w[lo:hi], scale)
            return
        if isinstance(ih, torch.Tensor):     # the epoch already on the device (shuffled there): filter by ownership there too
            mine = torch.nonzero(ih % self.world == self.rank).reshape(-1)
            off = torch.searchsorted(mine, torch.as_tensor(np.asarray(step_off, dtype=np.int64), device=ih.device)).cpu().numpy()
            i32 = lambda t: t.to(torch.int32).contiguous()
            staged = (i32(ih[mine] // self.world), i32(ia[mine]), i32(iv[mine]), None if w is None else w[mine].to(torch.float32).contiguous())
        else:
            ih = np.asarray(ih, dtype=np.int64)
            mine = np.nonzero(ih % self.world == self.rank)[0]
            off = np.searchsorted(mine, np.asarray(step_off, dtype=np.int64))      # this rank's slice of every step
            staged = be.stage(ih[mine] // self.world, np.asarray(ia)[mine], np.asarray(iv)[mine], None if w is None else np.asarray(w)[mine])
        cm = self.comm
        for s in range(len(step_off) - 1):
            part = tuple(None if t is None else t[off[s]:off[s + 1]] for t in staged)
            if self.world == 1 and hasattr(be, "step_alone"):
                be.step_alone(self, None, None, None, None, scale, staged=part)
                continue
            S = be.forward(self, None, None, None, None, scale, staged=part)
            cm.all_reduce(S)
            T = be.tail(self, S)
            cm.all_reduce(T)
            for g in be.backward(self, T):
                cm.all_reduce(g)
            be.update(self)

    def epoch_loss(self) -> float:
        t = self.backend.take_loss()
        self.comm.all_reduce(t)
        return float(t)

    def gather(self):
        """(full av_ent table [n_ent, dim], attr table, CNN parameter dict) — tests / checkpoint."""
        ent, attr, params = self.backend.tables()
        pad = -(-self.n_ent // self.world)
        mine = torch.zeros(pad, self.dim, dtype=torch.float64, device=_gather_device())
        n_local = len(range(self.rank, self.n_ent, self.world))
        mine[:n_local] = torch.as_tensor(ent[:n_local], dtype=torch.float64)
        if self.world == 1 or not dist.is_initialized():
            return mine[:n_local].cpu().numpy(), attr, params
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        full = np.zeros((self.n_ent, self.dim))
        for r in range(self.world):
            n = len(range(r, self.n_ent, self.world))
            full[r::self.world] = parts[r][:n].cpu().numpy()
        return full, attr, params


# ======================================================================================================================
# Common-space learning (ITC)
# ======================================================================================================================
class HipCommonSpaceBackend:
    device_type = "cuda"

    def __init__(self, view, shards: dict, tables: dict = None):
        d = view.dim
        self.eng = StepEngine()
        self.loss = torch.zeros((), dtype=torch.float64, device="cuda")
        if tables is not None:      # {"ent", "name", "rv", "av"}: EmbeddingTables held by the caller
            self.ent, self.name, self.rv, self.av = (tables[k] for k in ("ent", "name", "rv", "av"))
            return
        mk = lambda name, trainable, norm: EmbeddingTable(max(1, len(shards[name])), d, name, normalize=norm, trainable=trainable,
                                                          values=shards[name] if len(shards[name]) else np.zeros((1, d)))
        # name_embeds is a constant read as-is (code/MultiKE_model.py:88); the other three are normalise-on-read variables
        self.ent, self.name = mk("ent", True, True), mk("name", False, False)
        self.rv, self.av = mk("rv", True, True), mk("av", True, True)

    def step(self, view, rows):
        if len(rows) == 0:
            return
        idx = rows if isinstance(rows, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(rows, dtype=np.int32), device="cuda")
        cvw = view.cv_weight
        terms = [(self.ent, idx, self.name, idx, cvw * view.cv_name_weight), (self.ent, idx, self.rv, idx, cvw),
                 (self.ent, idx, self.av, idx, cvw)]
        self.loss += self.eng.alignment_step(terms, "cross_name", view.lr)

    def take_loss(self):
        v = self.loss.reshape(1).clone()
        self.loss.zero_()
        return v

    def tables(self):
        return {k: getattr(self, k).raw().cpu().numpy() for k in ("ent", "rv", "av")}


class ShardedCommonSpace:
    def __init__(self, ent0, name0, rv0, av0, rank: int, world: int, lr: float = 0.004, cv_name_weight: float = 1.0,
                 cv_weight: float = 1.0, backend_cls=None, comm=None, tables: dict = None, n_ent: int = None):
        self.rank, self.world, self.lr = rank, world, float(lr)
        self.cv_name_weight, self.cv_weight = float(cv_name_weight), float(cv_weight)
        self.comm = _comm_for(world, comm)
        if tables is not None:      # the caller's EmbeddingTable shards {"ent", "name", "rv", "av"} of n_ent global rows
            self.dim, self.n_ent = tables["ent"].dim, int(n_ent)
            self.backend = (backend_cls or HipCommonSpaceBackend)(self, None, tables=tables)
            return
        self.dim, self.n_ent = ent0.shape[1], ent0.shape[0]
        shards = {k: v[rank::world] for k, v in (("ent", ent0), ("name", name0), ("rv", rv0), ("av", av0))}
        self.backend = (backend_cls or HipCommonSpaceBackend)(self, shards)

    def step(self, entities):
        """One common-space step on the GLOBAL sample of entity ids (distinct; identical on every rank)."""
        _, rows = _owned(entities, self.rank, self.world)
        self.backend.step(self, rows)

    def steps(self, entities: torch.Tensor, step_off):
        """Consecutive steps over a device tensor of GLOBAL entity ids in step order (step s = [step_off[s], step_off[s + 1])):
        ownership filtered on the device, one host read of the slice boundaries per call."""
        mine = torch.nonzero(entities % self.world == self.rank).reshape(-1)
        off = torch.searchsorted(mine, torch.as_tensor(np.asarray(step_off, dtype=np.int64), device=entities.device)).cpu().numpy()
        rows = (entities[mine] // self.world).to(torch.int32).contiguous()
        for s in range(len(step_off) - 1):
            self.backend.step(self, rows[off[s]:off[s + 1]])

    def epoch_loss(self) -> float:
        t = self.backend.take_loss()
        self.comm.all_reduce(t)
        return float(t)

    def gather(self) -> dict:
        out = {}
        for k, v in self.backend.tables().items():
            pad = -(-self.n_ent // self.world)
            n_local = len(range(self.rank, self.n_ent, self.world))
            mine = torch.zeros(pad, self.dim, dtype=torch.float64, device=_gather_device())
            mine[:n_local] = torch.as_tensor(v[:n_local], dtype=torch.float64)
            if self.world == 1 or not dist.is_initialized():
                out[k] = mine[:n_local].cpu().numpy()
                continue
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine)
            full = np.zeros((self.n_ent, self.dim))
            for r in range(self.world):
                full[r::self.world] = parts[r][:len(range(r, self.n_ent, self.world))].cpu().numpy()
            out[k] = full
        return out


# ======================================================================================================================
# Space mapping (SSL driver)
# ======================================================================================================================
class HipSpaceMappingBackend:
    device_type = "cuda"

    def __init__(self, view: "ShardedSpaceMapping", ent0, views0, matrices, tables=None):
        from .runner import SpaceMappingState
        d = view.dim
        mk = lambda name, vals, trainable: EmbeddingTable(max(1, len(vals)), d, name, normalize=True, trainable=trainable,
                                                          values=vals if len(vals) else np.zeros((1, d)))
        if tables is not None:      # (shared-table shard, [view shards]): EmbeddingTables held by the caller
            self.ent, self.views = tables[0], list(tables[1])
        else:
            self.ent = mk("ent_embeds", ent0, True)
            self.views = [mk(f"view{k}", v, False) for k, v in enumerate(views0)]
        self.state = SpaceMappingState([torch.as_tensor(np.asarray(m)) for m in matrices], "cuda")
        self.eng = StepEngine()
        LP = _lib.LOSS_PARTIALS
        self.lossp = torch.zeros((_lib.MAPPING_MAX_VIEWS + 1) * LP, dtype=torch.float64, device="cuda")
        self.loss = torch.zeros(2, dtype=torch.float64, device="cuda")      # [sum of the ranks' map losses, orthogonality + norm terms]
        self.args = None

    def _args(self, view, rows):
        f32, i32 = torch.float32, torch.int32
        n = len(rows)
        self._idx = torch.as_tensor(np.ascontiguousarray(rows, dtype=np.int32), device="cuda")
        a = _lib.MappingStepArgs()
        e = self.ent
        a.ent_table, a.n_ent, a.ent_normalize = _lib.ptr(e.data, f32, "ent"), e.n_rows, 1
        a.ent_acc = _lib.ptr(e.slot("mapping"), f32, "acc")
        a.ent_grad, a.ent_touched = _lib.ptr(e.grad, f32, "grad"), _lib.ptr(e.touched, i32, "touched")
        a.n_views = len(self.views)
        for k, t in enumerate(self.views):
            a.views[k].table, a.views[k].normalize = _lib.ptr(t.data, f32, "view"), 1
        a.stride, a.dim = e.stride, e.dim
        a.idx, a.n = (_lib.ptr(self._idx, i32, "idx") if n else None), n
        st = self.state
        a.M, a.gM, a.accM = _lib.ptr(st.M, f32, "M"), _lib.ptr(st.gM, f32, "gM"), _lib.ptr(st.accM, f32, "accM")
        a.orthogonal_weight, a.norm_w = view.orthogonal_weight, view.norm_w
        a.scratch = _lib.ptr(st.scratch(max(n, 1)), f32, "scratch")
        a.partials = _lib.ptr(st.partials, torch.float64, "partials")
        tag, _ = self.eng._next()
        a.optimizer, a.lr, a.tag, a.update = _lib.OPT_ADAGRAD, view.lr, tag, 1
        return a

    def _scalars(self, which):
        LP, V = _lib.LOSS_PARTIALS, len(self.views)
        return self.state.partials.view(_lib.MAPPING_MAX_VIEWS, 2, LP)[:V, which].sum(dim=1)

    def _set_scalars(self, which, v):
        LP, V = _lib.LOSS_PARTIALS, len(self.views)
        blk = self.state.partials.view(_lib.MAPPING_MAX_VIEWS, 2, LP)[:V, which]
        blk.zero_()
        blk[:, 0] = v

    def forward(self, view, rows):
        self.args = self._args(view, rows)
        _lib.mapping_step_phases(self.args, self.lossp, _lib.MAP_FWD)
        return self._scalars(0)

    def tail(self, view, S):
        self._set_scalars(0, S)
        _lib.mapping_step_phases(self.args, self.lossp, _lib.MAP_TAIL)
        return self._scalars(1)

    def backward(self, view, T):
        self._set_scalars(1, T)
        _lib.mapping_step_phases(self.args, self.lossp, _lib.MAP_BWD)
        return self.state.gM

    def update(self, view):
        _lib.mapping_step_phases(self.args, self.lossp, _lib.MAP_UPD)
        LP, MV = _lib.LOSS_PARTIALS, _lib.MAPPING_MAX_VIEWS
        self.loss[0] += self.lossp[:MV * LP].sum()
        self.loss[1] += self.lossp[MV * LP:].sum()

    def take_loss(self):
        v = self.loss.clone()
        self.loss.zero_()
        return v

    def tables(self):
        return self.ent.raw().cpu().numpy(), self.state.M.double().cpu().numpy()


class ShardedSpaceMapping:
    def __init__(self, ent0, views0, matrices, rank: int, world: int, lr: float = 0.01, orthogonal_weight: float = 2.0,
                 norm_w: float = 0.0001, backend_cls=None, comm=None, tables=None, n_ent: int = None):
        """ent0: the shared table [n_ent, dim]; views0: the (constant) view tables mapped onto it; matrices: [dim, dim] each.
        tables: (shared-table shard, [view shards]) EmbeddingTables of the caller instead of ent0 / views0 (n_ent global rows)."""
        self.rank, self.world, self.lr = rank, world, float(lr)
        self.orthogonal_weight, self.norm_w = float(orthogonal_weight), float(norm_w)
        self.comm = _comm_for(world, comm)
        if tables is not None:
            self.dim, self.n_ent = tables[0].dim, int(n_ent)
            self.backend = (backend_cls or HipSpaceMappingBackend)(self, None, None, matrices, tables=tables)
            return
        self.dim, self.n_ent = ent0.shape[1], ent0.shape[0]
        self.backend = (backend_cls or HipSpaceMappingBackend)(self, ent0[rank::world], [v[rank::world] for v in views0], matrices)

    def step(self, entities):
        """One `session.run([shared_comb_loss, shared_comb_optimizer])` on the GLOBAL sample of entity ids (distinct;
        identical on every rank): this rank maps the entities it owns."""
        be, cm = self.backend, self.comm
        _, rows = _owned(entities, self.rank, self.world)
        S = be.forward(self, rows)
        cm.all_reduce(S)                                   # per view: sum P^2 over the whole batch (code/losses.py:55)
        T = be.tail(self, S)
        cm.all_reduce(T)                                   # per view: sum G . out over the whole batch (its backward)
        cm.all_reduce(be.backward(self, T))                # data part of the three matrix gradients
        be.update(self)

    def epoch_loss(self) -> float:
        t = self.backend.take_loss()
        data = t[0:1].clone()
        self.comm.all_reduce(data)
        return float(data) + float(t[1])                   # the orthogonality / norm terms are the same on every rank

    def gather(self):
        """(full shared table [n_ent, dim], matrices [n_views, dim, dim]) — tests / checkpoint."""
        ent, M = self.backend.tables()
        pad = -(-self.n_ent // self.world)
        n_local = len(range(self.rank, self.n_ent, self.world))
        mine = torch.zeros(pad, self.dim, dtype=torch.float64, device=_gather_device())
        mine[:n_local] = torch.as_tensor(ent[:n_local], dtype=torch.float64)
        if self.world == 1 or not dist.is_initialized():
            return mine[:n_local].cpu().numpy(), M
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        full = np.zeros((self.n_ent, self.dim))
        for r in range(self.world):
            full[r::self.world] = parts[r][:len(range(r, self.n_ent, self.world))].cpu().numpy()
        return full, M


# ======================================================================================================================
# Literal auto-encoder
# ======================================================================================================================
class HipAutoEncoderBackend:
    device_type = "cuda"

    def __init__(self, view: "ShardedAutoEncoder", params: dict):
        from types import SimpleNamespace
        from .literal_encoder import AutoEncoderModel
        args = SimpleNamespace(dim=view.dims[-1], encoder_normalize=view.normalize, encoder_active=view.active,
                               optimizer="Adagrad", learning_rate=view.lr)
        self.model = AutoEncoderModel(np.zeros((0, view.dims[0]), dtype=np.float32), args, input_dimension=view.dims[0],
                                      hidden_dimensions=list(view.dims[1:]), seed=0)
        self.model.set_params(params)
        self.loss = torch.zeros(1, dtype=torch.float64, device="cuda")
        self._one = torch.zeros(1, dtype=torch.float64, device="cuda")
        self._x, self._mg = None, 0

    def _block(self, k):
        LP = _lib.LOSS_PARTIALS
        return self.model._partials[k * LP:(k + 1) * LP]

    def _run(self, phases):
        m, p = self.model, self.model._plan
        p.optimizer, p.update, p.lr = _lib.OPT_ADAGRAD, 1, float(m.args.learning_rate)
        rows = 0 if self._x is None else self._x.shape[0]
        _lib.ae_step_phases(p, self._x if rows else None, rows, (self._x.stride(0) if rows else m.input_dimension), self._mg,
                            phases, self._one)

    def encode(self, view, x_rows, m_global):
        self._x = torch.as_tensor(np.ascontiguousarray(x_rows, dtype=np.float32), device="cuda") if len(x_rows) else None
        self._mg = int(m_global)
        self.model._ensure_scratch(max(1, len(x_rows)))
        self._run(_lib.AE_ENC)
        return self._block(0).sum().reshape(1)

    def decode(self, view, S):
        b = self._block(0)
        b.zero_()
        b[0] = S[0]
        self._run(_lib.AE_DEC)
        self.loss += self._one
        return self._block(1).sum().reshape(1)

    def backward(self, view, T):
        b = self._block(1)
        b.zero_()
        b[0] = T[0]
        self._run(_lib.AE_BWD)
        return self.model.grads

    def update(self, view):
        self._run(_lib.AE_UPD)

    def take_loss(self):
        v = self.loss.clone()
        self.loss.zero_()
        return v

    def params(self) -> dict:
        return {k: v.astype(np.float64) for k, v in self.model.numpy_params().items()}


class ShardedAutoEncoder:
    def __init__(self, params: dict, dims, rank: int, world: int, lr: float = 0.01, active: str = "thah", normalize: bool = True,
                 backend_cls=None, comm=None):
        """params: {encoder_h0, encoder_b0, ..., decoder_h0, ...} (replicated; code/literal_encoder.py:41-61); dims: input width,
        hidden widths, code width."""
        self.rank, self.world, self.lr = rank, world, float(lr)
        self.dims, self.active, self.normalize = [int(v) for v in dims], active, bool(normalize)
        self.comm = _comm_for(world, comm)
        self.backend = (backend_cls or HipAutoEncoderBackend)(self, params)

    def step(self, x):
        """One `session.run([loss, optimizer])` (code/literal_encoder.py:63-69, 98-107) on the GLOBAL batch x [M, dims[0]]
        (identical on every rank): this rank trains rows rank, rank + world, ..."""
        be, cm = self.backend, self.comm
        x = np.asarray(x)
        S = be.encode(self, x[self.rank::self.world], x.shape[0])
        cm.all_reduce(S)                                   # sum code^2 over the whole batch (:65-66)
        T = be.decode(self, S)
        cm.all_reduce(T)                                   # sum dcn . code over the whole batch (its backward)
        cm.all_reduce(be.backward(self, T))                # the packed gradient of the replicated parameters
        be.update(self)

    def epoch_loss(self) -> float:
        """Sum over the steps of mean((decoded - x)^2)."""
        t = self.backend.take_loss()
        self.comm.all_reduce(t)
        return float(t)

    def params(self) -> dict:
        return self.backend.params()
