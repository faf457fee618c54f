"""CPU ORACLE (test infrastructure, not product code) — the WHOLE training schedule in float64.

`OracleMultiKE` holds every variable of the reference's model (code/MultiKE_model.py:86-107) as a float64 NumPy array and
one Adagrad accumulator per (optimizer, variable) exactly where the reference creates optimizers (:131, 150, 169, 184, 200,
220, 237, 258), and replays the epoch loops (:291-473) on index streams that are GIVEN to it — the batches
some other implementation drew — with the reference's dense-table semantics (whole-table l2_normalize, dense Jacobian, TF1
ApplyAdagrad with acc0 = 0.1 and no epsilon):

    relation   code/MultiKE_model.py:114-132, 291-317   relation_logistic_loss over positives + sampled negatives
    ckge_rel   :158-169, 349-369                        2 x relation_logistic_loss_wo_negs
    ckgp_rel   :187-201, 393-414                        2 x logistic_loss_wo_negs (weighted)
    attribute  :134-151, 319-345                        sum w log(1 + exp(-conv))            CNN set 0
    ckge_attr  :171-185, 371-391                        2 x sum log(1 + exp(-conv))          CNN set 1
    ckga_attr  :203-221, 416-437                        sum w log(1 + exp(-conv))            CNN set 2
    common     :225-239, 458-473                        cv_name_weight align(ent, name) + align(ent, rv) + align(ent, av)
    mapping    :241-261, 439-454                        3 x space_mapping_loss

Every `*_epoch` returns the figure the reference prints for that loop (sum of batch losses / trained positives).  Built from
the step functions of `multike_oracle.py` / `attr_cnn_oracle.py` (pinned there: losses.py executed, torch autograd,
torch.optim.Adagrad); **parity unpinned at the TF boundary** like them.  The COMPOSITION of every graph — tables, views, factors,
learning rates, the optimizers' variable lists — is pinned by the reference's `_define_variables` / `_define_*_graph` methods EXECUTED
(tests/golden/make_golden.py `graphs_fixture` -> tests/golden/graphs_golden.npz; tests/test_graphs_golden.py: one step of each graph).
"""
from __future__ import annotations

import numpy as np

from . import attr_cnn_oracle as ao
from . import multike_oracle as mo

ACC0 = 0.1

# (optimizer name, variables it updates) — SURVEY.md §9.3 item 4
SLOTS = {
    "relation": ("rv_ent", "rel"), "ckge_rel": ("rv_ent", "rel"), "ckgp_rel": ("rv_ent", "rel"),
    "attribute": ("av_ent", "attr"), "ckge_attr": ("av_ent", "attr"), "ckga_attr": ("av_ent", "attr"),
    "cross_name": ("ent", "rv_ent", "av_ent"), "shared_comb": ("ent",),
}


class OracleMultiKE:
    def __init__(self, tables: dict, cnn_sets, mappings=None, learning_rate=0.001, itc_learning_rate=0.004,
                 cv_name_weight=1.0, cv_weight=1.0, orthogonal_weight=2.0, dtype=np.float64):
        """tables: raw values {"rv_ent", "av_ent", "ent", "rel", "attr"} (trainable) + {"name", "lit"} (constants);
        cnn_sets: three parameter dicts (attribute view, ckge_attr, ckga_attr); mappings: [nv, rv, av] d x d or None.
        dtype: float64 = the truth; float32 = the same schedule in the reference's WORKING precision with NumPy's summation
        order — what any other correct fp32 implementation of the same arithmetic looks like next to the truth (the yardstick
        tests/test_schedule_trace_gpu.py holds the HIP tables against)."""
        f = lambda a: np.array(a, dtype=dtype)
        self.t = {k: f(v) for k, v in tables.items()}
        self.acc = {(opt, var): np.full_like(self.t[var], ACC0) for opt, vs in SLOTS.items() for var in vs}
        self.cnn = [{k: f(v) for k, v in p.items()} for p in cnn_sets]
        self.cnn_acc = [{k: np.full_like(v, ACC0) for k, v in p.items()} for p in self.cnn]
        self.M = None if mappings is None else [f(m) for m in mappings]
        self.M_acc = None if mappings is None else [np.full_like(m, ACC0) for m in self.M]
        self.lr, self.itc_lr = float(learning_rate), float(itc_learning_rate)
        self.cv_name_weight, self.cv_weight, self.orthogonal_weight = float(cv_name_weight), float(cv_weight), float(orthogonal_weight)

    # --- relation-type loops ------------------------------------------------------------------------------
    def relation_epoch(self, pos, neg, off, neg_per_pos):
        """pos / neg: (h, r, t) arrays of the whole epoch in step order (negatives grouped neg_per_pos per positive)."""
        t, N = self.t, int(neg_per_pos)
        total = 0.0
        for s in range(len(off) - 1):
            lo, hi = int(off[s]), int(off[s + 1])
            p = tuple(a[lo:hi] for a in pos)
            n = tuple(a[lo * N:hi * N] for a in neg) if N else None
            L, _, _ = mo.relation_view_step_dense(t["rv_ent"], t["rel"], self.acc[("relation", "rv_ent")],
                                                  self.acc[("relation", "rel")], p, n, self.lr)
            total += L
        return total / max(int(off[-1]), 1)

    def relation_positives_epoch(self, opt, cols, w, off, scale=2.0):
        """ckge_rel (w None) / ckgp_rel (weighted): positives only, loss x scale."""
        t = self.t
        total = 0.0
        for s in range(len(off) - 1):
            lo, hi = int(off[s]), int(off[s + 1])
            L, _, _ = mo.relation_view_step_dense(t["rv_ent"], t["rel"], self.acc[(opt, "rv_ent")], self.acc[(opt, "rel")],
                                                  tuple(c[lo:hi] for c in cols), None, self.lr,
                                                  pos_w=None if w is None else np.asarray(w[lo:hi], dtype=t["rel"].dtype), scale=scale)
            total += L
        return total / max(int(off[-1]), 1)

    # --- attribute-type loops -----------------------------------------------------------------------------
    def attribute_epoch(self, opt, cols, w, off, scale=1.0):
        k = ("attribute", "ckge_attr", "ckga_attr").index(opt)
        t = self.t
        total = 0.0
        for s in range(len(off) - 1):
            lo, hi = int(off[s]), int(off[s + 1])
            if hi <= lo:
                continue
            L, _ = ao.attribute_step_dense(self.cnn[k], self.cnn_acc[k], t["av_ent"], t["attr"], t["lit"],
                                           self.acc[(opt, "av_ent")], self.acc[(opt, "attr")],
                                           np.asarray(cols[0][lo:hi], dtype=np.int64), np.asarray(cols[1][lo:hi], dtype=np.int64),
                                           np.asarray(cols[2][lo:hi], dtype=np.int64),
                                           None if w is None else np.asarray(w[lo:hi], dtype=t["attr"].dtype), scale, self.lr)
            total += L
        return total / max(int(off[-1]), 1)

    # --- combination ----------------------------------------------------------------------------------------
    def common_space_epoch(self, idx, off):
        """The optimizer minimises cv_weight * loss with ITC_learning_rate; the printed figure is the unscaled loss."""
        t = self.t
        total = 0.0
        for s in range(len(off) - 1):
            ids = np.asarray(idx[int(off[s]):int(off[s + 1])], dtype=np.int64)
            total += mo.common_space_step_dense(t["ent"], t["name"], t["rv_ent"], t["av_ent"], self.acc[("cross_name", "ent")],
                                                self.acc[("cross_name", "rv_ent")], self.acc[("cross_name", "av_ent")], ids,
                                                self.itc_lr, self.cv_name_weight, self.cv_weight)
        return total / self.cv_weight / max(int(off[-1]), 1) if self.cv_weight != 0 else 0.0

    def space_mapping_epoch(self, idx, off):
        """Only the variables named 'shared*' move (code/MultiKE_model.py:257-261): ent_embeds and the three matrices."""
        t = self.t
        views = [(t["name"], False), (t["rv_ent"], True), (t["av_ent"], True)]
        total = 0.0
        for s in range(len(off) - 1):
            ids = np.asarray(idx[int(off[s]):int(off[s + 1])], dtype=np.int64)
            total += mo.space_mapping_step_dense(t["ent"], self.acc[("shared_comb", "ent")], views, self.M, self.M_acc, ids,
                                                 self.lr, self.orthogonal_weight)
        return total / max(int(off[-1]), 1)

    # --- replay of a recorded phase (multike_amd.MultiKE_model.MultiKE._recorder payloads as NumPy arrays) ---------
    def replay(self, phase, rec):
        if phase == "relation":
            return self.relation_epoch(rec["pos"], rec["neg"], rec["off"], rec["neg_per_pos"])
        if phase in ("ckge_rel", "ckgp_rel"):
            return self.relation_positives_epoch(phase, rec["cols"], rec.get("w"), rec["off"], 2.0)
        if phase in ("attribute", "ckge_attr", "ckga_attr"):
            return self.attribute_epoch(phase, rec["cols"], rec.get("w"), rec["off"], 2.0 if phase == "ckge_attr" else 1.0)
        if phase == "common":
            return self.common_space_epoch(rec["idx"], rec["off"])
        if phase == "mapping":
            return self.space_mapping_epoch(rec["idx"], rec["off"])
        raise ValueError(phase)

    def view(self, name):
        """What `model.<name>.eval()` returns: the normalised view of a table (raw for attr / constants)."""
        return mo.l2_normalize_rows(self.t[name]) if name in ("rv_ent", "av_ent", "ent", "rel") else self.t[name]
