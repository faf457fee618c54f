"""The MultiKE model surface (multike_amd/MultiKE_model.py) driven the way code/MultiKE_CSL.py drives the reference:
define variables + all graphs, then the per-epoch train loops.  Checks the loops' bookkeeping (step counts, loss
normalisation, x2 factors, per-graph optimizer slots) against the oracle on the very batches the model drew."""
import math

import numpy as np
import pytest

from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu


def _model(**over):
    from multike_amd.MultiKE_model import MultiKE
    from multike_amd.synthetic import SyntheticData, synthetic_args
    data = SyntheticData(dim=20)
    args = synthetic_args(dim=20, batch_size=700, attribute_batch_size=600, entity_batch_size=800, neg_triple_num=5,
                          learning_rate=0.01, **over)
    m = MultiKE(data, args, data.predicate_align_model)
    m._define_variables()
    m._define_name_view_graph()
    m._define_relation_view_graph()
    m._define_attribute_view_graph()
    m._define_cross_kg_entity_reference_relation_view_graph()
    m._define_cross_kg_entity_reference_attribute_view_graph()
    m._define_cross_kg_attribute_reference_graph()
    m._define_cross_kg_relation_reference_graph()
    m._define_common_space_learning_graph()
    m._define_space_mapping_graph()
    return m, data, args


def test_itc_style_epochs_run_and_learn():
    m, data, args = _model()
    kgs = data.kgs
    rel_steps = int(math.ceil((kgs.kg1.local_relation_triples_num + kgs.kg2.local_relation_triples_num) / args.batch_size))
    attr_steps = int(math.ceil((kgs.kg1.local_attribute_triples_num + kgs.kg2.local_attribute_triples_num) / args.batch_size))
    ents = kgs.kg1.entities_list + kgs.kg2.entities_list
    pam = data.predicate_align_model
    hist = []
    for i in range(1, 4):
        r = m.train_relation_view_1epo(i, rel_steps, None, None, None, None)
        c1 = m.train_cross_kg_entity_inference_relation_view_1epo(i, kgs.kg1.sup_relation_triples_list + kgs.kg2.sup_relation_triples_list)
        c2 = m.train_cross_kg_relation_inference_1epo(i, pam.sup_relation_alignment_triples1 + pam.sup_relation_alignment_triples2)
        a = m.train_attribute_view_1epo(i, attr_steps, None, None, None, None)
        c3 = m.train_cross_kg_entity_inference_attribute_view_1epo(i, kgs.kg1.sup_attribute_triples_list + kgs.kg2.sup_attribute_triples_list)
        c4 = m.train_cross_kg_attribute_inference_1epo(i, pam.sup_attribute_alignment_triples1 + pam.sup_attribute_alignment_triples2)
        cs = m.train_common_space_learning_1epo(i, ents)
        hist.append((r, c1, c2, a, c3, c4, cs))
    sm = m.train_shared_space_mapping_1epo(1, ents)
    h = np.array(hist)
    assert np.all(np.isfinite(h)) and np.isfinite(sm)
    assert np.all(h[-1] < h[0])            # every loss went down over three epochs
    # relation view: avg loss per POSITIVE with N=5 starts near (1+N)*log(2) = 4.16 (unit rows, random init)
    assert 2.5 < h[0, 0] < 6 * math.log(2) + 1.5
    # per-graph optimizer slots exist exactly where the reference creates them
    assert set(m.rv_ent_embeds.slots) == {"relation", "ckge_rel", "ckgp_rel", "cross_name"}
    assert set(m.rel_embeds.slots) == {"relation", "ckge_rel", "ckgp_rel"}
    assert set(m.av_ent_embeds.slots) == {"attribute", "ckge_attr", "ckga_attr", "cross_name"}
    assert set(m.attr_embeds.slots) == {"attribute", "ckge_attr", "ckga_attr"}
    assert set(m.ent_embeds.slots) == {"cross_name", "shared_comb"}
    # every gradient scratch consumed; constants untouched
    for t in (m.rv_ent_embeds, m.rel_embeds, m.av_ent_embeds, m.attr_embeds, m.ent_embeds):
        assert float(t.grad.abs().max()) == 0.0 and float(t.data[:, t.dim:].abs().max()) == 0.0
    nm = m.name_embeds.eval()
    np.testing.assert_allclose(nm, data.local_name_vectors, rtol=0, atol=0)
    # .eval(session=...) returns the NORMALISED view
    np.testing.assert_allclose(np.linalg.norm(m.rv_ent_embeds.eval(session=m.session), axis=1), 1.0, rtol=1e-5)
    m.save()


def test_cross_kg_and_common_space_loops_match_oracle():
    """Replays the batch sampler to recover the batches the loops drew, then checks losses and tables against the
    float64 oracle: ckge_rel (x2), ckgp_rel (weighted, x2), common space (three alignment terms, ITC lr)."""
    m, data, args = _model()
    kgs, pam = data.kgs, data.predicate_align_model
    d = args.dim
    sup = kgs.kg1.sup_relation_triples_list + kgs.kg2.sup_relation_triples_list
    supw = pam.sup_relation_alignment_triples1 + pam.sup_relation_alignment_triples2
    ents = kgs.kg1.entities_list + kgs.kg2.entities_list
    raw = lambda t: t.raw().cpu().numpy().astype(np.float64)
    E, R, ENT, AV = raw(m.rv_ent_embeds), raw(m.rel_embeds), raw(m.ent_embeds), raw(m.av_ent_embeds)
    NM = data.local_name_vectors.astype(np.float64)
    acc = lambda x: np.full_like(x, 0.1)
    state = None

    def replay(n, bs, steps):
        """The batches the loop just drew: its (seed, stream) through the sampler oracle (mke_sample_distinct is bit-exact
        against it, tests/test_sampler_gpu.py)."""
        from oracle.sampler_oracle import distinct_sample
        seed, stream, n_, bs_, steps_ = m._last_sample
        assert (n_, bs_, steps_) == (n, bs, steps)
        return list(distinct_sample(n, bs, steps, seed, stream)), None

    # --- ckge_rel --------------------------------------------------------------------------------------
    got = m.train_cross_kg_entity_inference_relation_view_1epo(1, sup)
    steps = int(math.ceil(len(sup) / args.batch_size)); bs = args.batch_size if steps > 1 else len(sup)
    idxs, state = replay(len(sup), bs, steps)
    arr = np.asarray(sup)
    aE, aR = acc(E), acc(R)
    tot = 0.0
    for ix in idxs:
        b = arr[ix]
        L, _, _ = mo.relation_view_step_dense(E, R, aE, aR, (b[:, 0], b[:, 1], b[:, 2]), None, args.learning_rate, scale=2.0)
        tot += L
    np.testing.assert_allclose(got, tot / (steps * bs), rtol=1e-5)
    np.testing.assert_allclose(raw(m.rv_ent_embeds), E, rtol=2e-4, atol=2e-6)
    # --- ckgp_rel (weights) ----------------------------------------------------------------------------
    got = m.train_cross_kg_relation_inference_1epo(1, supw)
    steps = int(math.ceil(len(supw) / args.batch_size)); bs = args.batch_size if steps > 1 else len(supw)
    idxs, state = replay(len(supw), bs, steps)
    arrw = np.asarray([t[:3] for t in supw]); ww = np.asarray([t[3] for t in supw])
    aE2, aR2 = acc(E), acc(R)      # a different optimizer => fresh accumulators
    tot = 0.0
    for ix in idxs:
        b = arrw[ix]
        L, _, _ = mo.relation_view_step_dense(E, R, aE2, aR2, (b[:, 0], b[:, 1], b[:, 2]), None, args.learning_rate,
                                              pos_w=ww[ix], scale=2.0)
        tot += L
    np.testing.assert_allclose(got, tot / (steps * bs), rtol=1e-5)
    np.testing.assert_allclose(raw(m.rv_ent_embeds), E, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(raw(m.rel_embeds), R, rtol=2e-4, atol=2e-6)
    # --- common space ------------------------------------------------------------------------------------
    got = m.train_common_space_learning_1epo(1, ents)
    steps = int(math.ceil(len(ents) / args.entity_batch_size)); bs = args.entity_batch_size if steps > 1 else len(ents)
    idxs, state = replay(len(ents), bs, steps)
    ea = np.asarray(ents)
    a_ent, a_rv, a_av = acc(ENT), acc(E), acc(AV)
    lr = args.ITC_learning_rate
    tot = 0.0
    for ix in idxs:
        ids = ea[ix]
        # the three terms share one optimizer step: accumulate gradients, then update each table once
        l1, g1a, _ = mo.alignment_step_dense(ENT, NM, None, None, ids, ids, lr, weight=1.0, b_norm=False, update=False)
        l2, g2a, g2b = mo.alignment_step_dense(ENT, E, None, None, ids, ids, lr, update=False)
        l3, g3a, g3b = mo.alignment_step_dense(ENT, AV, None, None, ids, ids, lr, update=False)
        mo.adagrad_dense(ENT, a_ent, mo.l2_normalize_rows_backward(ENT, g1a + g2a + g3a), lr)
        mo.adagrad_dense(E, a_rv, mo.l2_normalize_rows_backward(E, g2b), lr)
        mo.adagrad_dense(AV, a_av, mo.l2_normalize_rows_backward(AV, g3b), lr)
        tot += l1 + l2 + l3
    np.testing.assert_allclose(got, tot / (steps * bs), rtol=1e-5)
    np.testing.assert_allclose(raw(m.ent_embeds), ENT, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(raw(m.rv_ent_embeds), E, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(raw(m.av_ent_embeds), AV, rtol=2e-4, atol=2e-6)


def test_truncated_sampling_neighbours_are_used():
    m, data, args = _model()
    kgs = data.kgs
    rng = np.random.default_rng(0)
    nb1 = {e: [int(x) for x in rng.choice(kgs.kg1.entities_list, 40, replace=False)] for e in kgs.kg1.entities_list}
    nb2 = {e: [int(x) for x in rng.choice(kgs.kg2.entities_list, 40, replace=False)] for e in kgs.kg2.entities_list}
    m._set_neighbours(nb1, nb2)
    pos, neg = m._rel_batcher.batch(0)
    ph, pt = pos[0].cpu().numpy(), pos[2].cpu().numpy()
    nh, nt = neg[0].cpu().numpy().reshape(len(ph), -1), neg[2].cpu().numpy().reshape(len(ph), -1)
    nb = {**nb1, **nb2}
    for i in range(len(ph)):
        for a, b in zip(nh[i], nt[i]):
            assert (a == ph[i] and b in nb[pt[i]]) or (b == pt[i] and a in nb[ph[i]])


@pytest.mark.parametrize("mode", ["ITC", "SSL"])
def test_drivers_run_end_to_end(mode):
    """run_ITC.py / run_SSL.py shape: Model(data, args, predicate_align_model).run() — a few epochs with every gate
    exercised (soft-alignment phases from epoch 2, validation every 2 epochs, truncated-sampling refresh every 2)."""
    from multike_amd.MultiKE_CSL import MultiKE_CV
    from multike_amd.MultiKE_Late import MultiKE_Late
    from multike_amd.synthetic import SyntheticData, synthetic_args
    data = SyntheticData(dim=20)
    # name vectors that carry the alignment signal: counterpart entities share a (noisy) name vector
    n1 = data.kgs.entities_num // 2
    rng = np.random.default_rng(1)
    base = rng.standard_normal((n1, 20)).astype(np.float32)
    nm = np.concatenate([base, base + 0.3 * rng.standard_normal((n1, 20)).astype(np.float32)])
    data.local_name_vectors = nm / np.linalg.norm(nm, axis=1, keepdims=True)
    args = synthetic_args(dim=20, batch_size=700, attribute_batch_size=600, entity_batch_size=800, neg_triple_num=5,
                          learning_rate=0.01, max_epoch=4, shared_learning_max_epoch=2, start_valid=2, eval_freq=2,
                          start_predicate_soft_alignment=1, truncated_freq=2, truncated_epsilon=0.9)
    cls = MultiKE_CV if mode == "ITC" else MultiKE_Late
    model = cls(data, args, data.predicate_align_model)
    res = model.run()
    assert all(np.isfinite(v) for v in res.values())
    assert res["nv"] > 0.5                       # the name view alone aligns the synthetic pairs
    assert model._neighbors[0] is not None and model._rel_batcher.side1.cand_table is not None   # truncated mode active


def test_two_stream_epochs_equal_single_stream_epochs():
    """The drivers enqueue the relation group and the attribute group of an epoch on two streams (disjoint state).  Same
    seeds => same batches => the tables after a few epochs must agree with the single-stream schedule up to fp32
    atomic-order noise.  One model also alternates between the two schedules, the case in which a buffer allocated on one
    stream is released while the other stream still reads it (caught once as a memory fault at C2 scale)."""
    import contextlib
    import io
    from multike_amd.MultiKE_CSL import MultiKE_CV
    from multike_amd.synthetic import SyntheticData, synthetic_args

    def run(schedule):
        data = SyntheticData(n_ent=40_000, n_rel=60, n_attr=40, n_values=5000, dim=32, seed=3)
        args = synthetic_args(dim=32, batch_size=4000, attribute_batch_size=4000, entity_batch_size=4000, neg_triple_num=8,
                              learning_rate=0.01, start_predicate_soft_alignment=0, neg_sampling="uniform")
        m = MultiKE_CV(data, args, data.predicate_align_model)
        m._prepare()
        losses = []
        with contextlib.redirect_stdout(io.StringIO()):
            for i, ov in enumerate(schedule, 1):
                m.overlap_views = ov
                m._train_views(i)
                losses.append(m.train_common_space_learning_1epo(i, m._entity_list))
        torch.cuda.synchronize()
        return m, losses

    import torch
    a, la = run([False] * 6)
    a2, _ = run([False] * 6)          # same schedule twice: the run-to-run noise of fp32 atomics through six Adagrad epochs
    b, lb = run([True] * 6)
    c, lc = run([False, True, True, False, True, True])
    names = ("rv_ent_embeds", "av_ent_embeds", "rel_embeds", "attr_embeds", "ent_embeds")
    tab = lambda m, n: getattr(m, n).raw().cpu().numpy()
    for other, lo in ((b, lb), (c, lc)):
        np.testing.assert_allclose(lo, la, rtol=1e-4)
        for name in names:
            noise = np.abs(tab(a2, name) - tab(a, name))
            diff = np.abs(tab(other, name) - tab(a, name))
            # same statistics as two runs of one schedule (a stale or recycled buffer would move whole rows by O(0.1))
            assert float(diff.mean()) <= 3.0 * float(noise.mean()) + 1e-8, (name, float(diff.mean()), float(noise.mean()))
            assert float(diff.max()) <= 5e-3, (name, float(diff.max()), float(noise.max()))
