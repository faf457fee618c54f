"""The training phases of an ITC epoch (code/MultiKE_CSL.py:57-79) on row-sharded tables, one process per GPU.

New design (the reference is single-device): every entity table is sharded id % world, the small tables and the CNN sets are
replicated, and each phase of the epoch runs in its sharded form on the SAME tables —

    relation view            OwnerComputesTrainer (negatives, optimizer "relation")
    ckge relation            OwnerComputesTrainer on a TripleListBatcher (positives only, x 2)
    ckgp relation            the same, weighted 4-tuples
    attribute view           ShardedAttributeView (CNN set 0)
    ckge attribute           ShardedAttributeView (CNN set 1, x 2)
    ckga attribute           ShardedAttributeView (CNN set 2, weighted)
    common-space learning    ShardedCommonSpace

(multike_amd/distributed_oc.py, multike_amd/distributed_views.py; byte / latency model in DESIGN.md §5).  The host-side
batch draws (attribute batches, cross-KG samples, entity samples) are made ON THE DEVICE from (seed, epoch, phase) — a seeded
`torch.randperm` for the shuffled attribute view, `mke_sample_distinct` for the `random.sample` loops: identical on every rank,
no exchange, no per-step host-to-device copy.  Not here: the soft predicate alignment refresh, truncated-sampling k-NN refresh,
validation (host-side or evaluator work that the single-GPU drivers do between epochs).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from .distributed_oc import OwnerComputesTrainer, TripleListBatcher
from .distributed_views import ShardedAttributeView, ShardedCommonSpace, ShardedSpaceMapping
from .tables import EmbeddingTable


class ShardedITC:
    def __init__(self, kgs, tables: dict, cnn_sets, lists: dict, rank: int, world: int, batch_size: int = 5000,
                 attribute_batch_size: int = 5000, entity_batch_size: int = 5000, neg_triple_num: int = 10,
                 learning_rate: float = 0.001, itc_learning_rate: float = 0.004, cv_name_weight: float = 1.0, cv_weight: float = 1.0,
                 seed: int = 0, comm_oc=None, comm_views=None, mapping_matrices=None, mapping_learning_rate: float = 0.01,
                 orthogonal_weight: float = 2.0, start_predicate_soft_alignment: int = 0, attr_steps: int = None):
        """kgs: the two KGs' relation triples (multike_amd.synthetic.SyntheticKGs / base.kgs.KGs interface: `triples`,
        `entities(k)`, `ent_range`); tables: full float32 arrays {"rv_ent", "av_ent", "ent", "name", "rel", "attr", "lit"}
        (every rank passes the same; each keeps its shard); cnn_sets: three CNN parameter dicts (attribute view, ckge, ckga);
        lists: {"attr": [(h, a, v, w)], "ckge_rel": [(h, r, t)], "ckgp_rel": [(h, r, t, w)], "ckge_attr": [(h, a, v)],
        "ckga_attr": [(h, a, v, w)], "entities": [ids]}; batch sizes are GLOBAL (a step trains that many across the ranks).
        start_predicate_soft_alignment: the two predicate-alignment phases (ckgp_rel, ckga_attr) run in epoch i only when
        i > this (code/MultiKE_CSL.py:62-70, code/args.json "start_predicate_soft_alignment": 10); their lists are replaced
        between epochs with `set_lists` (the reference rebuilds them every 10 epochs, code/MultiKE_CSL.py:80-87)."""
        self.rank, self.world, self.seed = rank, world, int(seed)
        self.start_soft = int(start_predicate_soft_alignment)
        d = tables["ent"].shape[1]
        self.n_ent = n_ent = tables["ent"].shape[0]
        shard = lambda k, norm=True, train=True: EmbeddingTable(max(1, len(range(rank, n_ent, world))), d, k, normalize=norm,
                                                                trainable=train, values=_pad1(tables[k][rank::world], d))
        # the relation / attribute tables' gradient scratch is privatised as in the one-GPU model (heavy-tailed relation / attribute
        # frequencies: EXPERIMENTS R5.16-17); it is all-reduced whole, so only while it stays under 1 MB on the wire
        def full(k, norm, train=True, copies=1):
            c = max(1, min(copies, (1 << 20) // max(1, tables[k].shape[0] * _lib.stride_for(d) * 4)))
            return EmbeddingTable(tables[k].shape[0], d, k, normalize=norm, trainable=train, values=tables[k], grad_copies=c)
        self.rv_ent, self.av_ent, self.ent = shard("rv_ent"), shard("av_ent"), shard("ent")
        self.name = shard("name", norm=False, train=False)
        self.rel, self.attr = full("rel", True, copies=OwnerComputesTrainer.REL_COPIES), full("attr", False, copies=4)
        self.lit = full("lit", False, train=False)
        self.sizes = (int(batch_size), int(attribute_batch_size), int(entity_batch_size))
        # hub rows of the relation view's shard (performance only; before any trainer takes the table's gradient scratch):
        # mke_oc_apply and the positives' own terms add to private copies of the rows many positives of every step share
        from .distributed_oc import hub_rows_of_shard
        hubs = hub_rows_of_shard(kgs.triples, n_ent, batch_size, rank, world)
        if len(hubs):
            self.rv_ent.set_hot_rows(hubs, OwnerComputesTrainer.HOT_COPIES)
        # one tag range per component (they share the tables' touched-flag arrays: a flag is `touched[row] == tag`)
        base = iter(k << 26 for k in range(1, 16))
        oc = dict(rank=rank, world=world, seed=seed, lr=learning_rate, comm=comm_oc, ent_table=self.rv_ent, rel_table=self.rel, n_ent=n_ent)
        self.relation = OwnerComputesTrainer(kgs, None, None, max(1, -(-batch_size // world)), neg_triple_num, opt_name="relation",
                                             tag_base=next(base), global_batch=batch_size, **oc)
        self._oc_args, self._list_tags, self._list_gen = oc, {}, {}

        def mk_list(key, opt):
            self._list_tags.setdefault(key, next(base))
            return self._make_list_trainer(key, lists.get(key, ()))
        self.ckge_rel, self.ckgp_rel = mk_list("ckge_rel", "ckge_rel"), mk_list("ckgp_rel", "ckgp_rel")
        av = dict(rank=rank, world=world, lr=learning_rate, comm=comm_views, tables=(self.av_ent, self.attr, self.lit), n_ent=n_ent)
        self.attr_views = [ShardedAttributeView(None, None, None, cnn_sets[k], opt_name=name, **av)
                           for k, name in enumerate(("attribute", "ckge_attr", "ckga_attr"))]
        for v in self.attr_views:
            v.backend.eng.tag = next(base)
        self.common = ShardedCommonSpace(None, None, None, None, rank, world, lr=itc_learning_rate, cv_name_weight=cv_name_weight,
                                         cv_weight=cv_weight, comm=comm_views, n_ent=n_ent,
                                         tables={"ent": self.ent, "name": self.name, "rv": self.rv_ent, "av": self.av_ent})
        self.common.backend.eng.tag = next(base)
        # SSL schedule (code/MultiKE_Late.py:201-280): the shared table is learned by mapping the three views onto it
        # (code/MultiKE_model.py:439-454) instead of the common-space step
        self.mapping = None
        if mapping_matrices is not None:
            self.mapping = ShardedSpaceMapping(None, None, mapping_matrices, rank, world, lr=mapping_learning_rate,
                                               orthogonal_weight=orthogonal_weight, comm=comm_views, n_ent=n_ent,
                                               tables=(self.ent, [self.name, self.rv_ent, self.av_ent]))
            self.mapping.backend.eng.tag = next(base)
        self.lists = {k: lists.get(k, []) for k in ("attr", "ckge_attr", "ckga_attr", "entities")}
        dev = lambda a, dt: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
        # lists["attr"] = (KG1's weighted attribute triples, KG2's): the attribute view then batches as the reference does —
        # both lists shuffled after every epoch, step s = [KG1 slice s | KG2 slice s] in the proportion of the list sizes
        # (code/attr_batch.py:28-41, code/MultiKE_model.py:319-345); a single list: one shuffled list cut into steps
        self._attr_pair, self.attr_steps = None, attr_steps
        if isinstance(self.lists["attr"], tuple):
            self._attr_pair = tuple(tuple(dev(c, torch.float32 if j == 3 else torch.int64) for j, c in enumerate(_columns(l)))
                                    for l in self.lists["attr"])
            self.lists["attr"] = []
        self._cols = {k: tuple(dev(c, torch.float32 if j == 3 else torch.int64) for j, c in enumerate(_columns(v)))
                      for k, v in self.lists.items() if k != "entities"}
        self._entities = dev(np.asarray(self.lists["entities"], dtype=np.int64), torch.int64)
        self._gen = torch.Generator(device="cuda")
        self._oc_steps = {id(t): 0 for t in (self.relation, self.ckge_rel, self.ckgp_rel) if t is not None}

    # ------------------------------------------------------------------------------------------------
    def _make_list_trainer(self, key, triples):
        """Owner-computes trainer of a cross-KG relation loop over `triples` (positives only, x 2).  A replacement trainer
        (set_lists) keeps the loop's optimizer — the Adagrad slots live in the tables under the loop's name — and continues
        its tag range; its draws are seeded by (seed, generation) so that a new list does not replay the old one's."""
        if not len(triples):
            return None
        gen = self._list_gen.get(key, 0)
        self._list_gen[key] = gen + 1
        bs = self.sizes[0]
        tr = OwnerComputesTrainer(None, None, None, bs, 0, opt_name=key, scale=2.0, tag_base=self._list_tags[key],
                                  batcher=TripleListBatcher(triples, bs, device="cuda", seed=self.seed + 7919 * gen),
                                  **self._oc_args)
        return tr

    def set_lists(self, **lists):
        """Replace supervision lists between epochs: `ckgp_rel=[(h, r, t, w)]`, `ckga_attr=[(h, a, v, w)]` after a soft
        predicate-alignment update (code/MultiKE_CSL.py:80-87), or any of the other keys.  Every rank must pass the same
        lists.  Optimizer state is untouched (the slots belong to the tables / CNN sets)."""
        dev = lambda a, dt: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
        for key, lst in lists.items():
            if key in ("ckge_rel", "ckgp_rel"):
                old = getattr(self, key)
                if old is not None:
                    self._list_tags[key] = old.tag            # continue the tag range: flags of earlier steps stay stale
                    self._oc_steps.pop(id(old), None)
                new = self._make_list_trainer(key, lst)
                setattr(self, key, new)
                if new is not None:
                    self._oc_steps[id(new)] = 0
            elif key in ("attr", "ckge_attr", "ckga_attr"):
                self.lists[key] = lst
                self._cols[key] = tuple(dev(c, torch.float32 if j == 3 else torch.int64) for j, c in enumerate(_columns(lst)))
            elif key == "entities":
                self.lists[key] = lst
                self._entities = dev(np.asarray(lst, dtype=np.int64), torch.int64)
            else:
                raise _lib.MultiKEHipError(f"set_lists: unknown list {key!r}")

    def _oc_epoch(self, tr):
        if tr is None:
            return 0.0
        i0 = self._oc_steps[id(tr)]
        tr.run(i0, tr.steps)              # one native call per run of steps inside an epoch (mke_oc_steps); the Python step loop otherwise
        self._oc_steps[id(tr)] = i0 + tr.steps
        return tr.epoch_loss()

    def _draw(self, n: int, B: int, epoch: int, phase: int, sampled: bool):
        """(positions of an epoch in step order [device int64], step_off): `steps` x random.sample(range(n), B) when sampled
        (code/MultiKE_model.py:358, 378, 446), else random.shuffle + consecutive slices (:336-343)."""
        steps = int(math.ceil(n / B))
        if sampled:
            bs = B if steps > 1 else n
            idx = _lib.sample_distinct(n, bs, steps, (self.seed & 0xFFFFFFFF, 0x495443 + phase), epoch, device="cuda").reshape(-1).long()
            return idx, np.arange(steps + 1, dtype=np.int64) * bs
        self._gen.manual_seed((self.seed * 1000003 + epoch * 101 + phase) & 0x7FFFFFFFFFFFFFFF)
        idx = torch.randperm(n, generator=self._gen, device="cuda")
        return idx, np.minimum(np.arange(steps + 1, dtype=np.int64) * B, n)

    def _attr_epoch(self, view, key, epoch, phase, scale, sampled):
        cols = self._cols[key]
        n = int(cols[0].numel())
        if n == 0:
            return 0.0
        order, off = self._draw(n, self.sizes[1], epoch, phase, sampled)
        view.steps(cols[0][order], cols[1][order], cols[2][order], cols[3][order] if cols[3] is not None else None, off, scale=scale)
        return view.epoch_loss()

    def _attr_pair_epoch(self, view, epoch, phase):
        """The attribute view on two per-KG lists: the reference's proportional split (code/attr_batch.py:28-41 through
        code/base/batch.py:36-37 arithmetic) and per-epoch shuffles, laid out on the device from (seed, epoch)."""
        c1, c2 = self._attr_pair
        n1, n2 = int(c1[0].numel()), int(c2[0].numel())
        if n1 + n2 == 0:
            return 0.0
        B = self.sizes[1]
        b1 = int(n1 / (n1 + n2) * B)
        b2 = B - b1
        steps = self.attr_steps if self.attr_steps is not None else int(math.ceil((n1 + n2) / B))
        key = (n1, n2, b1, b2, steps)
        if getattr(self, "_attr_layout_key", None) != key:
            k1 = np.clip(n1 - np.arange(steps) * b1, 0, b1) if b1 > 0 else np.zeros(steps, np.int64)
            k2 = np.clip(n2 - np.arange(steps) * b2, 0, b2) if b2 > 0 else np.zeros(steps, np.int64)
            off = np.zeros(steps + 1, dtype=np.int64)
            off[1:] = np.cumsum(k1 + k2)
            p1, p2 = np.arange(int(k1.sum())), np.arange(int(k2.sum()))
            d1 = off[p1 // max(b1, 1)] + p1 % max(b1, 1) if len(p1) else p1
            d2 = off[p2 // max(b2, 1)] + k1[p2 // max(b2, 1)] + p2 % max(b2, 1) if len(p2) else p2
            tod = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64), device="cuda")
            self._attr_layout, self._attr_layout_key = (off, tod(d1), tod(d2)), key
        off, d1, d2 = self._attr_layout
        total = int(off[-1])
        if total == 0:
            return 0.0
        cols = [torch.empty(total, dtype=torch.int64, device="cuda") for _ in range(3)] + [torch.empty(total, dtype=torch.float32, device="cuda")]
        for kg, (c, dest, n) in enumerate(((c1, d1, n1), (c2, d2, n2))):
            m = int(dest.numel())
            if m == 0:
                continue
            if epoch <= 1:
                src = torch.arange(m, device="cuda")                       # the first epoch runs in list order
            else:
                self._gen.manual_seed((self.seed * 1000003 + epoch * 101 + phase * 7 + kg) & 0x7FFFFFFFFFFFFFFF)
                src = torch.randperm(n, generator=self._gen, device="cuda")[:m]
            for k in range(4):
                cols[k][dest] = c[k][src] if c[k] is not None else torch.ones(m, device="cuda")
        view.steps(cols[0], cols[1], cols[2], cols[3], off, scale=1.0)
        return view.epoch_loss()

    def _common_epoch(self, epoch, phase):
        n = int(self._entities.numel())
        if n == 0:
            return 0.0
        order, off = self._draw(n, self.sizes[2], epoch, phase, True)      # random.sample(entity_list, B) (code/MultiKE_model.py:446)
        self.common.steps(self._entities[order], off)
        return self.common.epoch_loss()

    def _mapping_epoch(self, epoch, phase):
        n = int(self._entities.numel())
        if n == 0 or self.mapping is None:
            return 0.0
        order, off = self._draw(n, self.sizes[2], epoch, phase, True)
        ids = self._entities[order].cpu().numpy()
        for s in range(len(off) - 1):
            self.mapping.step(ids[off[s]:off[s + 1]])
        return self.mapping.epoch_loss()

    def _view_phases(self, i: int) -> dict:
        """The six view / cross-KG phases of epoch i in the reference's order; the two predicate-alignment phases only when
        i > start_predicate_soft_alignment (code/MultiKE_CSL.py:62-70, code/MultiKE_Late.py:218-228) — a skipped phase is
        absent from the result."""
        soft = i > self.start_soft
        out = {"relation": self._oc_epoch(self.relation), "ckge_rel": self._oc_epoch(self.ckge_rel)}
        if soft:
            out["ckgp_rel"] = self._oc_epoch(self.ckgp_rel)
        out["attribute"] = (self._attr_pair_epoch(self.attr_views[0], i, 0) if self._attr_pair is not None else
                            self._attr_epoch(self.attr_views[0], "attr", i, 0, 1.0, sampled=False))
        out["ckge_attr"] = self._attr_epoch(self.attr_views[1], "ckge_attr", i, 1, 2.0, sampled=True)
        if soft:
            out["ckga_attr"] = self._attr_epoch(self.attr_views[2], "ckga_attr", i, 2, 1.0, sampled=True)
        return out

    # the drivers' building blocks (multike_amd/distributed_run.py): one training phase of epoch i -> (summed loss, positives)
    def run_phase(self, name: str, i: int):
        if name in ("relation", "ckge_rel", "ckgp_rel"):
            tr = getattr(self, name)
            return (self._oc_epoch(tr), int(tr.bat.off[-1])) if tr is not None else (0.0, 0)
        if name == "attribute":
            if self._attr_pair is not None:
                loss = self._attr_pair_epoch(self.attr_views[0], i, 0)
                return loss, int(self._attr_layout[0][-1]) if getattr(self, "_attr_layout", None) else 0
            return self._attr_epoch(self.attr_views[0], "attr", i, 0, 1.0, sampled=False), int(self._cols["attr"][0].numel())
        if name in ("ckge_attr", "ckga_attr"):
            k = 1 if name == "ckge_attr" else 2
            n = int(self._cols[name][0].numel())
            B = self.sizes[1]
            steps = int(math.ceil(n / B)) if n else 0
            return self._attr_epoch(self.attr_views[k], name, i, k, 2.0 if k == 1 else 1.0, sampled=True), steps * (B if steps > 1 else n)
        n = int(self._entities.numel())
        B = self.sizes[2]
        steps = int(math.ceil(n / B)) if n else 0
        if name == "common":
            return self._common_epoch(i, 3), steps * (B if steps > 1 else n)
        if name == "mapping":
            return self._mapping_epoch(i, 4), steps * (B if steps > 1 else n)
        raise _lib.MultiKEHipError(f"run_phase: unknown phase {name!r}")

    def views_epoch(self, i: int) -> dict:
        return self._view_phases(i)

    def common_epoch(self, i: int) -> float:
        return self._common_epoch(i, 3)

    def mapping_epoch(self, i: int) -> float:
        return self._mapping_epoch(i, 4)

    def epoch_ssl(self, i: int) -> dict:
        """The SSL schedule's training phases (code/MultiKE_Late.py:216-243): the six view phases, then the space mapping."""
        out = self._view_phases(i)
        out["mapping"] = self._mapping_epoch(i, 4)
        return out

    def epoch(self, i: int) -> dict:
        """The seven training phases of epoch i in the reference's order (code/MultiKE_CSL.py:62-79); returns their summed losses."""
        out = self._view_phases(i)
        out["common"] = self._common_epoch(i, 3)
        return out

    def gather(self) -> dict:
        """Full tables (tests / checkpoint): the three entity tables gathered from their shards, the replicated ones as they are."""
        out = self.common.gather()              # ent, rv, av
        out["rel"] = self.rel.raw().cpu().numpy()
        out["attr"] = self.attr.raw().cpu().numpy()
        out["cnn"] = [v.backend.cnn.numpy_params() for v in self.attr_views]
        if self.mapping is not None:
            out["matrices"] = self.mapping.backend.state.M.double().cpu().numpy()
        return out


def _pad1(a, d):
    return a if len(a) else np.zeros((1, d), dtype=np.float32)


def _columns(lst):
    """[(h, a, v[, w])] -> (h, a, v, w or None) int64 / float arrays."""
    if len(lst) == 0:
        z = np.zeros(0, dtype=np.int64)
        return z, z, z, None
    arr = np.asarray([t[:3] for t in lst], dtype=np.int64)
    w = np.asarray([t[3] for t in lst], dtype=np.float64) if len(lst[0]) > 3 else None
    return arr[:, 0], arr[:, 1], arr[:, 2], w
