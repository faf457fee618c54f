"""Whole-schedule parity (SURVEY.md §8 f1 "epoch-loss traces from the CPU oracle"; north_star "per-batch loss within 1e-4,
reproduces Hits@1/10"): the product's ITC and SSL drivers run their FULL schedules — every training phase of every epoch, the
soft-alignment gate, validation, the truncated-sampling refresh, and for SSL the shared-space mapping epochs — while a recorder
captures the exact index streams each phase consumed (positives, the device sampler's negatives, weights, random.sample draws).
A float64 oracle of the whole model (oracle/model_oracle.py: dense-table semantics, one Adagrad accumulator per optimizer and
variable as in the reference) then replays the same schedule on the same batches.  Asserted:

  * every phase's printed epoch loss, epoch by epoch, to 1e-4 relative (the north_star tolerance);
  * every trainable table, the three CNN parameter sets and the mapping matrices after the last epoch, to fp32 tolerance;
  * Hits@1 / Hits@10 / MRR of the product's evaluator (k_align_rank) on the HIP tables == the float64 evaluator oracle on the
    oracle's tables, for every view the drivers test.

DBP-WD itself is absent (/root/reference/.MISSING_LARGE_BLOBS); this is the stand-in SURVEY §7 names for its acceptance row."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import eval_oracle as eo
from oracle.model_oracle import OracleMultiKE

pytestmark = pytest.mark.gpu

PHASE_OF = {
    "train_relation_view_1epo": "relation", "train_cross_kg_entity_inference_relation_view_1epo": "ckge_rel",
    "train_cross_kg_relation_inference_1epo": "ckgp_rel", "train_attribute_view_1epo": "attribute",
    "train_cross_kg_entity_inference_attribute_view_1epo": "ckge_attr", "train_cross_kg_attribute_inference_1epo": "ckga_attr",
    "train_common_space_learning_1epo": "common", "train_shared_space_mapping_1epo": "mapping",
}
DIM = 24


def _data(n_ent=1800, n_rel=24, n_attr=20, n_values=400, dim=DIM, seed=21):
    """Two KGs that share 80 % of their relation / attribute structure and whose counterpart entities have noisy copies of one
    name vector: every view has something to learn, none becomes perfect in a few epochs — the Hits figures sit in the
    sensitive middle of their range."""
    from multike_amd.synthetic import SyntheticData
    data = SyntheticData(n_ent=n_ent, n_rel=n_rel, n_attr=n_attr, n_values=n_values, dim=dim, seed=seed, shared_structure=0.8)
    n1 = data.kgs.entities_num // 2
    rng = np.random.default_rng(4)
    base = rng.standard_normal((n1, dim)).astype(np.float32)
    nm = np.concatenate([base, base + 0.9 * rng.standard_normal((n1, dim)).astype(np.float32)])
    data.local_name_vectors = nm / np.linalg.norm(nm, axis=1, keepdims=True)
    return data


def _np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy().copy()
    if isinstance(v, (tuple, list)):
        return tuple(_np(x) for x in v)
    return v


def _run(method, data_kw=None, args_kw=None, snapshots=None):
    from multike_amd.MultiKE_CSL import MultiKE_CV
    from multike_amd.MultiKE_Late import MultiKE_Late
    from multike_amd.synthetic import synthetic_args
    data = _data(**(data_kw or {}))
    args = synthetic_args(**(args_kw or {})) if args_kw else \
        synthetic_args(dim=DIM, batch_size=900, attribute_batch_size=700, entity_batch_size=500, neg_triple_num=6,
                          learning_rate=0.03, ITC_learning_rate=0.05, cv_name_weight=0.8, cv_weight=1.5, orthogonal_weight=2,
                          max_epoch=10, shared_learning_max_epoch=8, start_valid=2, eval_freq=2, start_predicate_soft_alignment=2,
                          truncated_freq=3, truncated_epsilon=0.9, neg_sampling="truncated", seed=5)
    model = (MultiKE_CV if method == "ITC" else MultiKE_Late)(data, args, data.predicate_align_model)
    model.overlap_views = False               # one stream: the phases then run (and are recorded) in the reference's order
    raw = lambda t: t.raw().cpu().numpy()
    tables = {"rv_ent": raw(model.rv_ent_embeds), "av_ent": raw(model.av_ent_embeds), "ent": raw(model.ent_embeds),
              "rel": raw(model.rel_embeds), "attr": raw(model.attr_embeds), "name": data.local_name_vectors, "lit": data.value_vectors}
    cnn = [c.numpy_params() for c in (model._attr_cnn, model._ckge_attr_cnn, model._ckga_attr_cnn)]
    maps = [m.detach().cpu().numpy() for m in (model.nv_mapping, model.rv_mapping, model.av_mapping)] if method == "SSL" else None
    oracle = OracleMultiKE(tables, cnn, maps, learning_rate=args.learning_rate, itc_learning_rate=args.ITC_learning_rate,
                           cv_name_weight=args.cv_name_weight, cv_weight=args.cv_weight, orthogonal_weight=args.orthogonal_weight)
    oracle.M0 = None if maps is None else [np.array(m, dtype=np.float64) for m in maps]
    recs, losses = [], []
    def record(phase, **kw):
        recs.append((phase, {k: _np(v) for k, v in kw.items()}))
        if snapshots is not None:          # the trainable tables right after the phase (tools/parity_noise.py)
            torch.cuda.synchronize()
            snap = {k: raw(t) for k, t in (("rv_ent", model.rv_ent_embeds), ("av_ent", model.av_ent_embeds),
                                           ("ent", model.ent_embeds), ("rel", model.rel_embeds), ("attr", model.attr_embeds))}
            for k, t in (("rv_ent", model.rv_ent_embeds), ("av_ent", model.av_ent_embeds), ("ent", model.ent_embeds)):
                if "cross_name" in t.slots:
                    snap["acc_cross_name_" + k] = t.slots["cross_name"][:, :t.dim].cpu().numpy()
            snapshots.append(snap)
    model._recorder = record
    for name, phase in PHASE_OF.items():
        orig = getattr(model, name)
        setattr(model, name, (lambda f, ph: lambda *a, **k: losses.append((ph, a[0], f(*a, **k))) or losses[-1][2])(orig, phase))
    with contextlib.redirect_stdout(io.StringIO()):
        results = model.run()
    torch.cuda.synchronize()
    return model, oracle, recs, losses, results, data, args


@pytest.mark.timeout(900)
@pytest.mark.parametrize("method", ["ITC", "SSL"])
def test_whole_schedule_tracks_the_float64_oracle(method):
    model, oracle, recs, losses, results, data, args = _run(method)
    # the schedule that ran: every phase of every epoch, gated phases from epoch 3 (i > 2), SSL's mapping epochs at the end
    n_ep = args.max_epoch
    expected = []
    for i in range(1, n_ep + 1):
        expected += [("relation", i), ("ckge_rel", i)] + ([("ckgp_rel", i)] if i > 2 else [])
        expected += [("attribute", i), ("ckge_attr", i)] + ([("ckga_attr", i)] if i > 2 else [])
        expected += [("common", i)] if method == "ITC" else []
    expected += [("mapping", i) for i in range(1, args.shared_learning_max_epoch + 1)] if method == "SSL" else []
    assert [(p, i) for p, i, _ in losses] == expected
    assert [p for p, _ in recs] == [p for p, _ in expected]
    # the truncated-sampling refresh happened (after epoch 3) and later epochs drew their negatives from the k-NN lists
    assert model._neighbors[0] is not None
    # ---- replay on the float64 oracle, phase by phase -------------------------------------------------------------
    # ... and on the SAME oracle in float32 (NumPy's summation order): what another correct fp32 implementation of this
    # schedule looks like next to the float64 truth — the yardstick of the final-state check below
    o32 = OracleMultiKE(dict(oracle.t), oracle.cnn, oracle.M0, learning_rate=args.learning_rate, itc_learning_rate=args.ITC_learning_rate,
                        cv_name_weight=args.cv_name_weight, cv_weight=args.cv_weight, orthogonal_weight=args.orthogonal_weight,
                        dtype=np.float32)
    worst = 0.0
    for (phase, rec), (p2, epoch, got) in zip(recs, losses):
        exp = oracle.replay(phase, rec)
        o32.replay(phase, rec)
        assert np.isfinite(got) and got > 0.0, (phase, epoch, got)
        err = abs(got - exp) / abs(exp)
        worst = max(worst, err)
        assert err <= 1e-4, f"{method}: epoch {epoch} phase {phase}: product {got!r} vs oracle {exp!r} (rel {err:.2e})"
    # ---- final state -------------------------------------------------------------------------------------------------
    pairs = {"rv_ent": model.rv_ent_embeds, "av_ent": model.av_ent_embeds, "ent": model.ent_embeds, "rel": model.rel_embeds,
             "attr": model.attr_embeds}
    # Round 3 widened this check (rtol 2e-3, 5e-4 of the elements outside) and blamed near-zero-norm rows.  Round 4 looked
    # (tools/parity_noise.py, profiles/r04_parity_noise.log): the offending rows have ordinary norms (0.25 .. 0.45).  What
    # happens is that early Adagrad steps of the common-space phase EXPAND perturbations: for a row read through l2_normalize
    # d w_new / d w_old = 1 - lr W / ||w||^2 * acc / (acc + g^2)^1.5 per element, and with this schedule's ITC_learning_rate
    # 0.05, summed loss weight W = 8.4, ||w||^2 = 0.07 and a young accumulator (0.1) that is up to -18 where an element's
    # gradient is near zero.  float64 finite differences (a 1e-9 nudge of one element, the phase replayed) put the
    # amplification of ONE common-space phase at x139 on the worst (row, element) and at x1.00 in the median (90th percentile
    # x1.7).  ANY fp32 implementation inherits it: the float32 run of the oracle itself is off by 3e-5 on exactly those rows
    # (2e-7 elsewhere), and the HIP tables' error there is a constant ~16x the float32 oracle's on every such row — the ratio
    # of the two implementations' rounding noise (hardware exp / rcp / rsqrt, atomic order) carried through the same
    # linearised dynamics.  So: (1) every element is inside the ORIGINAL band (rtol 1e-3, atol 2e-5) widened per row by 40 x
    # what NumPy's float32 makes of that row; (2) rows with an element outside the plain original band are < 1 % of the rows
    # and every one of them is a row where float32 itself leaves its noise floor (> 5 x the table's median row); (3) the absolute cap is 2e-3
    # (5e-3 before).
    for k, tab in pairs.items():
        got = tab.raw().cpu().numpy().astype(np.float64)
        ref = oracle.t[k]
        err = np.abs(got - ref)
        row_noise = np.abs(o32.t[k].astype(np.float64) - ref).max(axis=1, keepdims=True)
        plain = 2e-5 + 1e-3 * np.abs(ref)
        bad = err > plain + 40.0 * row_noise
        assert not bad.any(), (k, int(bad.sum()), float(err.max()), np.argwhere(bad)[:5].tolist(),
                               float((err / np.maximum(row_noise, 1e-12))[bad].max()))
        out_rows = (err > plain).any(axis=1)
        assert out_rows.mean() < 0.01, (k, int(out_rows.sum()))
        floor = max(2e-7, float(np.median(row_noise)))          # what float32 makes of an ordinary row of this table
        assert (row_noise[out_rows, 0] > 5.0 * floor).all(), (k, floor, np.flatnonzero(out_rows)[:8].tolist(), row_noise[out_rows, 0][:8].tolist())
        assert float(err.max()) < 2e-3, (k, float(err.max()))
    for c, P in zip((model._attr_cnn, model._ckge_attr_cnn, model._ckga_attr_cnn), oracle.cnn):
        for name, got in c.numpy_params().items():
            np.testing.assert_allclose(got, P[name], rtol=5e-3, atol=5e-4, err_msg=name)
    if method == "SSL":
        for got, ref in zip((model.nv_mapping, model.rv_mapping, model.av_mapping), oracle.M):
            np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=1e-3, atol=2e-5)
        assert float(np.abs(oracle.M[1] - oracle.M0[1]).max()) > 1e-3                         # the matrices did move
    # ---- Hits / MRR: product evaluator on HIP tables == float64 evaluator on oracle tables ---------------------------
    from multike_amd.base.alignment import alignment_counts, tie_aware_metrics
    kgs = data.kgs
    e1, e2 = kgs.test_entities1, kgs.test_entities2
    views = {"nv": ("name", model.name_embeds), "rv": ("rv_ent", model.rv_ent_embeds), "av": ("av_ent", model.av_ent_embeds),
             "final": ("ent", model.ent_embeds)}
    top_k = [1, 5, 10, 50]
    summary = {}
    for choice, (oname, tab) in views.items():
        hv = tab.eval()
        greater, ties, _ = alignment_counts(hv[e1], hv[e2])
        hits, mr, mrr = tie_aware_metrics(greater, ties, top_k)
        hits = np.round(np.array(hits) / len(e1) * 100, 3)
        ov = oracle.view(oname)
        orank, _ = eo.ranks(ov[e1], ov[e2])
        ohits, omr, omrr = eo.metrics(orank, top_k)
        summary[choice] = (hits, ohits)
        # ranks are integers: a handful of near-ties may flip between fp32 tables / fp32 similarities and float64
        n_diff = int(np.sum(greater.cpu().numpy() != orank))
        assert n_diff <= max(2, len(e1) // 50), (choice, n_diff)
        assert np.all(np.abs(hits - ohits) <= 0.5), (choice, hits, ohits)       # north_star: Hits within +-0.5
        assert abs(mrr - omrr) <= 2e-3, (choice, mrr, omrr)
    # the views are neither trivially perfect nor at chance: the comparison is sensitive
    assert 20.0 < summary["nv"][1][0] < 99.0
    if method == "ITC":       # the shared table of the SSL schedule only starts to move in its few mapping epochs
        assert summary["final"][1][2] > 5.0
    # the drivers' own closing tests report the same figures (results[...] is the MRR `test` returns)
    assert abs(results["final"] - eo.metrics(eo.ranks(oracle.view("ent")[e1], oracle.view("ent")[e2])[0], top_k)[2]) <= 2e-3
    print(f"{method}: worst phase-loss error {worst:.2e}; Hits@[1,5,10,50] product / oracle: "
          + "; ".join(f"{k} {v[0].tolist()} / {v[1].tolist()}" for k, v in summary.items()))
