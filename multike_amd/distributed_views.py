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


class HostStagedViewComm(ViewComm):
    """Test vehicle: two ranks sharing one GPU, collectives staged through gloo over the host."""

    def all_reduce(self, t):
        c = t.cpu()
        dist.all_reduce(c)
        t.copy_(c)


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

    def __init__(self, view: "ShardedAttributeView", ent0, attr0, lit, cnn_params):
        from .attr_cnn import AttrCNN
        d = view.dim
        self.ent = EmbeddingTable(max(1, len(ent0)), d, "av_ent_embeds", normalize=True, values=ent0 if len(ent0) else np.zeros((1, d)))
        self.attr = EmbeddingTable(attr0.shape[0], d, "attr_embeds", normalize=False, values=attr0)
        self.lit = EmbeddingTable(lit.shape[0], d, "literal_embeds", normalize=False, trainable=False, values=lit)
        self.cnn = AttrCNN(d, params=cnn_params)
        self.eng = StepEngine()
        self.loss = torch.zeros((), dtype=torch.float64, device="cuda")

    def _i32(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device="cuda")

    def forward(self, view, lh, ia, iv, w, scale):
        """Conv stack + dense layer on this rank's triples; returns the device scalar sum z^2 of its part."""
        wt = None if w is None else torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32), device="cuda")
        self._keep = (self._i32(lh), self._i32(ia), self._i32(iv), wt)
        self.args, self.part = self.cnn._args(self.eng, self.ent, self.attr, self.lit, *self._keep, len(lh), scale, view.opt_name,
                                              view.lr, "Adagrad", True, 1)
        _lib.attr_step_phases(self.args, _lib.ATTR_FWD)
        return self._scalar(1)

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

    def tables(self):
        return (self.ent.raw().cpu().numpy(), self.attr.raw().cpu().numpy(), self.cnn.numpy_params())

    def take_loss(self):
        v = self.loss.reshape(1).clone()
        self.loss.zero_()
        return v


class ShardedAttributeView:
    def __init__(self, ent0: np.ndarray, attr0: np.ndarray, lit: np.ndarray, cnn_params: dict, rank: int, world: int,
                 lr: float = 0.001, opt_name: str = "attribute", backend_cls=None, comm=None):
        self.rank, self.world, self.lr, self.opt_name = rank, world, float(lr), opt_name
        self.dim = ent0.shape[1]
        self.n_ent = ent0.shape[0]
        self.comm = comm or ViewComm()
        self.backend = (backend_cls or HipAttrBackend)(self, ent0[rank::world], attr0, lit, cnn_params)

    def step(self, ih, ia, iv, w=None, scale: float = 1.0):
        """One `session.run([loss, optimizer])` of an attribute-view graph on the GLOBAL batch (ih, ia, iv, w: host arrays,
        identical on every rank): this rank trains the triples whose head it owns."""
        be, cm = self.backend, self.comm
        pos, lh = _owned(ih, self.rank, self.world)
        ia, iv = np.asarray(ia)[pos], np.asarray(iv)[pos]
        w = None if w is None else np.asarray(w)[pos]
        S = be.forward(self, lh, ia, iv, w, scale)
        cm.all_reduce(S)                                   # sum z^2 over the whole batch (code/MultiKE_model.py:60)
        T = be.tail(self, S)
        cm.all_reduce(T)                                   # sum g.z over the whole batch (its backward)
        for g in be.backward(self, T):
            cm.all_reduce(g)                               # replicated parameters: CNN pack, attribute table
        be.update(self)

    def epoch_loss(self) -> float:
        t = self.backend.take_loss()
        self.comm.all_reduce(t)
        return float(t)

    def gather(self):
        """(full av_ent table [n_ent, dim], attr table, CNN parameter dict) — tests / checkpoint."""
        ent, attr, params = self.backend.tables()
        pad = -(-self.n_ent // self.world)
        mine = torch.zeros(pad, self.dim, dtype=torch.float64)
        n_local = len(range(self.rank, self.n_ent, self.world))
        mine[:n_local] = torch.as_tensor(ent[:n_local], dtype=torch.float64)
        if self.world == 1 or not dist.is_initialized():
            return mine[:n_local].numpy(), attr, params
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        full = np.zeros((self.n_ent, self.dim))
        for r in range(self.world):
            n = len(range(r, self.n_ent, self.world))
            full[r::self.world] = parts[r][:n].numpy()
        return full, attr, params


# ======================================================================================================================
# Common-space learning (ITC)
# ======================================================================================================================
class HipCommonSpaceBackend:
    device_type = "cuda"

    def __init__(self, view, shards: dict):
        d = view.dim
        mk = lambda name, trainable, norm: EmbeddingTable(max(1, len(shards[name])), d, name, normalize=norm, trainable=trainable,
                                                          values=shards[name] if len(shards[name]) else np.zeros((1, d)))
        # name_embeds is a constant read as-is (code/MultiKE_model.py:88); the other three are normalise-on-read variables
        self.ent, self.name = mk("ent", True, True), mk("name", False, False)
        self.rv, self.av = mk("rv", True, True), mk("av", True, True)
        self.eng = StepEngine()
        self.loss = torch.zeros((), dtype=torch.float64, device="cuda")

    def step(self, view, rows):
        if len(rows) == 0:
            return
        idx = torch.as_tensor(np.ascontiguousarray(rows, dtype=np.int32), device="cuda")
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
                 cv_weight: float = 1.0, backend_cls=None, comm=None):
        self.rank, self.world, self.lr = rank, world, float(lr)
        self.cv_name_weight, self.cv_weight = float(cv_name_weight), float(cv_weight)
        self.dim, self.n_ent = ent0.shape[1], ent0.shape[0]
        self.comm = comm or ViewComm()
        shards = {k: v[rank::world] for k, v in (("ent", ent0), ("name", name0), ("rv", rv0), ("av", av0))}
        self.backend = (backend_cls or HipCommonSpaceBackend)(self, shards)

    def step(self, entities):
        """One common-space step on the GLOBAL sample of entity ids (distinct; identical on every rank)."""
        _, rows = _owned(entities, self.rank, self.world)
        self.backend.step(self, rows)

    def epoch_loss(self) -> float:
        t = self.backend.take_loss()
        self.comm.all_reduce(t)
        return float(t)

    def gather(self) -> dict:
        out = {}
        for k, v in self.backend.tables().items():
            pad = -(-self.n_ent // self.world)
            n_local = len(range(self.rank, self.n_ent, self.world))
            mine = torch.zeros(pad, self.dim, dtype=torch.float64)
            mine[:n_local] = torch.as_tensor(v[:n_local], dtype=torch.float64)
            if self.world == 1 or not dist.is_initialized():
                out[k] = mine[:n_local].numpy()
                continue
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine)
            full = np.zeros((self.n_ent, self.dim))
            for r in range(self.world):
                full[r::self.world] = parts[r][:len(range(r, self.n_ent, self.world))].numpy()
            out[k] = full
        return out
