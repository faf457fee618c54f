"""GPU parity of the fused relation-view step (mke_triple_score_fwd_bwd + mke_rows_update) against the
oracle and the golden vectors.  Tolerances: per-batch loss relative 1e-4 (north_star; we hold 2e-6),
rows after the update rtol 2e-5 / atol 1e-7 against float64 (fp32 arithmetic + atomic order)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu

LOSS_RTOL = 2e-6
ROW_RTOL, ROW_ATOL = 2e-5, 2e-7


def _case(g, ci):
    pre = f"c{ci}_"
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize("copies", [1, 8])
@pytest.mark.parametrize("ci", range(5))
def test_golden_three_steps(losses_golden, ci, copies):
    from gpu_util import dev_i32, make_tables
    from multike_amd.tables import StepEngine
    c = _case(losses_golden, ci)
    N = int(c["meta"][5])
    E, R = make_tables(c["ent"], c["rel"], rel_grad_copies=copies)
    eng = StepEngine()
    pos = tuple(dev_i32(c[k]) for k in ("ph", "pr", "pt"))
    neg = tuple(dev_i32(c[k]) for k in ("nh", "nr", "nt"))
    losses = []
    for step in range(3):
        lp = eng.relation_step(E, R, "relation", pos, neg, neg_per_pos=N, lr=0.001)
        losses.append(float(lp.sum()))
        if step in (0, 2):
            np.testing.assert_allclose(E.raw().cpu().numpy(), c[f"a1_ent_after{step + 1}"], rtol=ROW_RTOL, atol=ROW_ATOL)
            np.testing.assert_allclose(R.raw().cpu().numpy(), c[f"a1_rel_after{step + 1}"], rtol=ROW_RTOL, atol=ROW_ATOL)
    np.testing.assert_allclose(losses, c["a1_step_losses_f64"], rtol=LOSS_RTOL)
    # invariants: gradient scratch consumed, pad columns still zero
    assert float(E.grad.abs().max()) == 0.0 and float(R.grad.abs().max()) == 0.0
    assert float(E.data[:, E.dim:].abs().max()) == 0.0
    assert int(E.refcount.abs().sum()) == 0          # exclusive-row bookkeeping restored


@pytest.mark.parametrize("d,P,N,grouped", [(75, 300, 25, True), (75, 300, 25, False), (4, 50, 3, True),
                                           (256, 120, 64, True), (100, 64, 10, True), (32, 77, 1, True),
                                           (75, 513, 0, True)])
def test_scatter_gradients_and_update_vs_oracle(d, P, N, grouped):
    from gpu_util import dev_i32, grouped_batch, make_tables
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(d * 1000 + P + N)
    n_ent, n_rel = 2000, 37
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    rel = mo.xavier_truncated_normal((n_rel, d), rng)
    pos, neg = grouped_batch(rng, n_ent, n_rel, P, max(N, 1))
    if N == 0:
        neg = None
    E, R = make_tables(ent, rel)
    eng = StepEngine()
    dpos = tuple(dev_i32(a) for a in pos)
    dneg = None if neg is None else tuple(dev_i32(a) for a in neg)
    # forward + scatter only: inspect the normalised-space gradient buffers
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    L, ghat_e, ghat_r = mo.relation_view_step_dense(e64, r64, None, None, pos, neg, 0.0, update=False)
    from multike_amd import _lib
    tag, lp = eng._next()
    _lib.triple_score_fwd_bwd(E.data, True, R.data, True, d, dpos, None, dneg, None, (N if grouped else 0), 1.0,
                              E.grad, R.grad, E.touched, R.touched, tag, lp)
    np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
    np.testing.assert_allclose(E.grad[:, :d].cpu().numpy(), ghat_e, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(R.grad[:, :d].cpu().numpy(), ghat_r, rtol=1e-4, atol=2e-5)
    if E.stride > d:
        assert float(E.grad[:, d:].abs().max()) == 0.0
    touched = np.zeros(n_ent, bool)
    touched[pos[0]] = touched[pos[2]] = True
    if neg is not None:
        touched[neg[0]] = touched[neg[2]] = True
    assert np.array_equal((E.touched == tag).cpu().numpy(), touched)
    # now the row update
    before = E.data.clone()
    eng._apply(E, "relation", "Adagrad", 0.001, tag)
    eng._apply(R, "relation", "Adagrad", 0.001, tag)
    acc_e, acc_r = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    mo.relation_view_step_dense(e64, r64, acc_e, acc_r, pos, neg, 0.001)
    np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=ROW_RTOL, atol=ROW_ATOL)
    np.testing.assert_allclose(R.raw().cpu().numpy(), r64, rtol=ROW_RTOL, atol=ROW_ATOL)
    np.testing.assert_allclose(E.slot("relation")[:, :d].cpu().numpy(), acc_e, rtol=1e-4, atol=1e-7)
    # untouched rows are bit-identical (dense TF semantics == touched-rows-only)
    ut = torch.as_tensor(~touched, device="cuda")
    assert torch.equal(E.data[ut], before[ut])
    assert float(E.grad.abs().max()) == 0.0


@pytest.mark.parametrize("variant", ["a2_x2", "a3_weighted_x2", "a4_both_weighted", "sgd", "rel_unnormalised"])
def test_variants(variant):
    """a2: positives only, caller's x2 (MultiKE_model.py:168); a3: weighted positives x2 (:198);
    a4: weights on both signs; SGD optimizer; a table read without normalisation (attr_embeds, :97)."""
    from gpu_util import dev_f32, dev_i32, grouped_batch, make_tables
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(5)
    d, n_ent, n_rel, P, N = 75, 900, 11, 200, 4
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    rel = mo.xavier_truncated_normal((n_rel, d), rng)
    pos, neg = grouped_batch(rng, n_ent, n_rel, P, N)
    pw = rng.uniform(0.2, 1, P).astype(np.float32)
    nw = rng.uniform(0.2, 1, P * N).astype(np.float32)
    kw = dict(scale=1.0, pos_w=None, neg_w=None)
    use_neg, rel_norm, opt = True, True, "Adagrad"
    if variant == "a2_x2":
        use_neg, kw["scale"] = False, 2.0
    elif variant == "a3_weighted_x2":
        use_neg, kw["scale"], kw["pos_w"] = False, 2.0, pw
    elif variant == "a4_both_weighted":
        kw["pos_w"], kw["neg_w"] = pw, nw
    elif variant == "sgd":
        opt = "SGD"
    elif variant == "rel_unnormalised":
        rel_norm = False
    E, R = make_tables(ent, rel, rel_norm=rel_norm)
    eng = StepEngine()
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    acc_e, acc_r = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    lp = eng.relation_step(E, R, "x", tuple(dev_i32(a) for a in pos),
                           tuple(dev_i32(a) for a in neg) if use_neg else None, neg_per_pos=N if use_neg else 0,
                           lr=0.01, pos_w=None if kw["pos_w"] is None else dev_f32(kw["pos_w"]),
                           neg_w=None if kw["neg_w"] is None else dev_f32(kw["neg_w"]), scale=kw["scale"],
                           optimizer=opt)
    if opt == "SGD":
        L, ge, gr = mo.relation_view_step_dense(e64, r64, None, None, pos, neg, 0.0, update=False)
        e64 -= 0.01 * mo.l2_normalize_rows_backward(e64, ge)
        r64 -= 0.01 * mo.l2_normalize_rows_backward(r64, gr)
    else:
        L, _, _ = mo.relation_view_step_dense(e64, r64, acc_e, acc_r, pos, neg if use_neg else None, 0.01,
                                              pos_w=kw["pos_w"], neg_w=kw["neg_w"], scale=kw["scale"],
                                              rel_norm=rel_norm)
    np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
    np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=ROW_RTOL, atol=ROW_ATOL)
    np.testing.assert_allclose(R.raw().cpu().numpy(), r64, rtol=ROW_RTOL, atol=ROW_ATOL)


@pytest.mark.parametrize("P,N,half", [(333, 10, 12), (333, 10, 0), (1, 3, 12), (64, 12, 12), (77, 1, 12),
                                      # long groups (more entries than a half-wave has lanes: several id blocks per group)
                                      (333, 25, 64), (101, 40, 64), (51, 64, 64), (333, 25, -1)])
def test_two_groups_per_wavefront_equals_one(P, N, half):
    """Short groups are scored two per wavefront (each half owns a group; `score_half_groups`): same losses and tables as
    the one-group-per-wavefront form and as the oracle, odd group counts (the last wavefront's second half idle) included."""
    from gpu_util import dev_i32, grouped_batch, make_tables
    from multike_amd import _lib
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(P * 100 + N)
    d, n_ent, n_rel = 75, 4000, 13
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    rel = mo.xavier_truncated_normal((n_rel, d), rng)
    pos, neg = grouped_batch(rng, n_ent, n_rel, P, N)
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    old_s = _lib.set_option("score_splits", 1)
    old_h = _lib.set_option("score_half_groups", half)
    try:
        E, R = make_tables(ent, rel)
        eng = StepEngine()
        for step in range(2):
            L, _, _ = mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01)
            lp = eng.relation_step(E, R, "relation", tuple(dev_i32(a) for a in pos), tuple(dev_i32(a) for a in neg),
                                   neg_per_pos=N, lr=0.01)
            np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
        np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=ROW_RTOL, atol=ROW_ATOL)
        np.testing.assert_allclose(R.raw().cpu().numpy(), r64, rtol=1e-4, atol=2e-6)
        assert float(E.grad.abs().max()) == 0.0 and float(R.grad.abs().max()) == 0.0 and int(E.refcount.abs().sum()) == 0
    finally:
        _lib.set_option("score_splits", old_s)
        _lib.set_option("score_half_groups", old_h)


@pytest.mark.parametrize("n_ent,P,N,excl", [(300, 400, 25, True), (300, 400, 25, False), (5000, 2000, 10, True), (200, 300, 0, True)])
def test_deterministic_mode_hub_rows(n_ent, P, N, excl):
    """`mke_set_option("deterministic", 1)`: fixed-order gradient sums.  On the heavy-collision (Zipf) batches whose hub
    rows sum hundreds of cancelling fp32 terms: (1) two runs are BIT-identical (tables, accumulators, losses), (2) rows agree
    with the float64 oracle to 1e-4 and the accumulator to 1e-3 on EVERY run (the atomic path's accumulator check is 1e-2
    because single elements land a few 1e-3 off in about one run out of 25; what is left here — 2 elements of 375,000 at
    5e-4 — is the fp32 Jacobian (g - w^(w^.g)) / |w| of rows whose gradient is nearly parallel to the row, not the sum)."""
    from gpu_util import dev_i32, make_tables
    from multike_amd import _lib
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(n_ent + P)
    d, n_rel = 75, 7
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    rel = mo.xavier_truncated_normal((n_rel, d), rng)
    pr_ = 1.0 / np.arange(1, n_ent + 1)
    pr_ /= pr_.sum()
    ph, pt = rng.choice(n_ent, P, p=pr_), rng.choice(n_ent, P, p=pr_)
    prl = rng.integers(0, n_rel, P)
    M = max(N, 1)
    nh, nt, nr = np.repeat(ph, M), np.repeat(pt, M), np.repeat(prl, M)
    side = rng.integers(0, 2, P * M).astype(bool)
    c = rng.integers(0, n_ent, P * M)
    nh, nt = np.where(side, c, nh), np.where(side, nt, c)
    nh[3], nt[3] = rng.integers(0, n_ent, 2)                       # an irregular negative (independent-triple path)
    pos = tuple(a.astype(np.int32) for a in (ph, prl, pt))
    neg = tuple(a.astype(np.int32) for a in (nh, nr, nt)) if N else None
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    losses64 = [mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01)[0] for _ in range(2)]
    old = _lib.set_option("deterministic", 1)
    try:
        runs = []
        for rep in range(2):
            E, R = make_tables(ent, rel)
            eng = StepEngine()
            ls = []
            for step in range(2):
                lp = eng.relation_step(E, R, "relation", tuple(dev_i32(a) for a in pos),
                                       None if neg is None else tuple(dev_i32(a) for a in neg), neg_per_pos=N, lr=0.01,
                                       exclusive_rows=excl)
                ls.append(lp.clone())
            runs.append((E.data.clone(), R.data.clone(), E.slot("relation").clone(), R.slot("relation").clone(), torch.stack(ls)))
            assert float(E.grad.abs().max()) == 0.0 and float(R.grad.abs().max()) == 0.0
            assert E._refcount is None or int(E.refcount.abs().sum()) == 0
        for x, y in zip(*runs):
            assert torch.equal(x, y)                               # bit-identical from run to run
        E_data, R_data, E_acc, R_acc, ls = runs[0]
        np.testing.assert_allclose(ls.sum(1).cpu().numpy(), losses64, rtol=LOSS_RTOL)
        np.testing.assert_allclose(E_data[:, :d].cpu().numpy(), e64, rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(R_data[:, :d].cpu().numpy(), r64, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(E_acc[:, :d].cpu().numpy(), a64, rtol=1e-3, atol=1e-6)
        assert np.mean(np.abs(E_acc[:, :d].cpu().numpy() - a64) > 1e-4 * np.abs(a64) + 1e-6) < 1e-5
    finally:
        _lib.set_option("deterministic", old)


def test_edge_cases():
    from gpu_util import dev_i32, make_tables
    from multike_amd import _lib
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(9)
    ent = mo.xavier_truncated_normal((64, 75), rng)
    rel = mo.xavier_truncated_normal((5, 75), rng)
    E, R = make_tables(ent, rel)
    eng = StepEngine()
    z = dev_i32(np.zeros(0))
    # empty batch (the reference's last slice can be empty, code/base/batch.py:45-54): loss 0, nothing moves
    before = E.data.clone()
    lp = eng.relation_step(E, R, "o", (z, z, z), (z, z, z), neg_per_pos=3)
    assert float(lp.sum()) == 0.0 and torch.equal(E.data, before)
    # one positive whose head == tail, negatives that all equal the positive
    one = (dev_i32([7]), dev_i32([2]), dev_i32([7]))
    neg = (dev_i32([7, 7]), dev_i32([2, 2]), dev_i32([7, 7]))
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    a, b = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    L, _, _ = mo.relation_view_step_dense(e64, r64, a, b, ([7], [2], [7]), ([7, 7], [2, 2], [7, 7]), 0.001)
    lp = eng.relation_step(E, R, "o", one, neg, neg_per_pos=2)
    np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
    np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=ROW_RTOL, atol=ROW_ATOL)
    # a row of zeros: ssq <= eps branch of the normalisation and its gradient
    E.data[3].zero_()
    e64 = E.raw().cpu().numpy().astype(np.float64)
    r64 = R.raw().cpu().numpy().astype(np.float64)
    a, b = E.slot("o")[:, :75].cpu().numpy().astype(np.float64), R.slot("o")[:, :75].cpu().numpy().astype(np.float64)
    L, _, _ = mo.relation_view_step_dense(e64, r64, a, b, ([3], [1], [9]), None, 0.001)
    lp = eng.relation_step(E, R, "o", (dev_i32([3]), dev_i32([1]), dev_i32([9])), None)
    np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
    np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=1e-4, atol=1e-6)
    # argument errors come back as exceptions with the library's message
    with pytest.raises(_lib.MultiKEHipError, match="n_neg|grouped negatives"):
        eng.relation_step(E, R, "o", one, neg, neg_per_pos=3)
    with pytest.raises(_lib.MultiKEHipError, match="CUDA"):
        eng.relation_step(E, R, "o", (torch.zeros(1, dtype=torch.int32),) * 3, None)


def test_full_size_c2_batch_vs_c_oracle():
    """BASELINE configs[1] shape: |E|=200K |R|=550 d=75 N=25 P=5000 (T=130K).  Checked against the C oracle in
    float64, plus size-independent properties (scratch consumed, untouched rows bit-identical, loss additivity)."""
    from gpu_util import dev_i32, make_tables
    from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import StepEngine
    kgs = SyntheticKGs()
    rng = np.random.default_rng(3)
    d = 75
    ent = mo.xavier_truncated_normal((kgs.entities_num, d), rng)
    rel = mo.xavier_truncated_normal((kgs.relations_num, d), rng)
    E, R = make_tables(ent, rel, rel_grad_copies=8)
    eng = StepEngine()
    sides = []
    for k in (0, 1):
        t = dev_i32(kgs.triples[k])
        sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, 25, seed=11)
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    orc = co.RelationStepOracle(kgs.entities_num, kgs.relations_num, d, np.float64)
    for step in (0, 1, bat.steps - 1):
        pos, neg = bat.batch(step)
        before = E.data.clone()
        lp = eng.relation_step(E, R, "relation", pos, neg, neg_per_pos=25, lr=0.001)
        hp = tuple(x.cpu().numpy() for x in pos)
        hn = tuple(x.cpu().numpy() for x in neg)
        L = orc.step(e64, r64, a64, b64, hp, hn, 0.001)
        np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
        np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=1e-4, atol=5e-7)
        np.testing.assert_allclose(R.raw().cpu().numpy(), r64, rtol=1e-4, atol=5e-7)
        touched = torch.zeros(kgs.entities_num, dtype=torch.bool, device="cuda")
        for x in (pos[0], pos[2], neg[0], neg[2]):
            touched[x.long()] = True
        assert torch.equal(E.data[~touched], before[~touched])
        assert float(E.grad.abs().max()) == 0.0 and float(R.grad.abs().max()) == 0.0
        assert int(E.refcount.abs().sum()) == 0
    # additivity: loss(batch) == loss(first half) + loss(second half), forward only
    pos, neg = bat.batch(2)
    h = pos[0].numel() // 2
    full = float(eng.relation_step(E, R, "relation", pos, neg, neg_per_pos=25, update=False).sum())
    p1 = tuple(x[:h] for x in pos); n1 = tuple(x[:h * 25] for x in neg)
    p2 = tuple(x[h:] for x in pos); n2 = tuple(x[h * 25:] for x in neg)
    parts = float(eng.relation_step(E, R, "relation", p1, n1, neg_per_pos=25, update=False).sum()) + \
        float(eng.relation_step(E, R, "relation", p2, n2, neg_per_pos=25, update=False).sum())
    np.testing.assert_allclose(full, parts, rtol=1e-6)  # fp32 per-lane partial sums regroup with the split


def test_wide_rows_many_negatives_vs_c_oracle():
    """configs[4]-like rows (dim 256, 64 negatives, batch 5000) on a 500K-entity table (the full 2M x 256 shape is
    exercised by tools/kbench.py; here the float64 oracle has to fit the host): one step against the C oracle."""
    from gpu_util import dev_i32, make_tables
    from multike_amd.sampling import KGSide, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import StepEngine
    kgs = SyntheticKGs(n_ent=500_000, n_rel=2000, triples_per_entity=1.0, seed=4)
    rng = np.random.default_rng(4)
    d, N = 256, 64
    ent = mo.xavier_truncated_normal((kgs.entities_num, d), rng)
    rel = mo.xavier_truncated_normal((kgs.relations_num, d), rng)
    E, R = make_tables(ent, rel)
    eng = StepEngine()
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None), KGSide(kgs.entities(1), None), 5000, N,
                          seed=1)
    pos, neg = bat.batch(3)
    lp = eng.relation_step(E, R, "relation", pos, neg, neg_per_pos=N, lr=0.001)
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    del ent, rel
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    orc = co.RelationStepOracle(kgs.entities_num, kgs.relations_num, d, np.float64)
    L = orc.step(e64, r64, a64, b64, tuple(x.cpu().numpy() for x in pos), tuple(x.cpu().numpy() for x in neg), 0.001)
    np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
    np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(R.raw().cpu().numpy(), r64, rtol=1e-4, atol=5e-7)


def test_full_size_c5_step_vs_compact_f64_oracle():
    """BASELINE configs[4] per-GPU shape at FULL size: |E|=2M, |R|=2000, dim=256, N=64, P=5000 (T=325K scored triples, a
    2 GB table: the HBM-resident shape).  The float64 C oracle runs on the COMPACTED problem — the ~300K rows the two
    steps touch, renumbered — which is the same function: untouched rows take no part in a step and stay bit-identical
    (asserted on the device table).  Plus the size-independent properties: gradient scratch consumed, reference counts
    back to zero, loss additivity over a split of the batch."""
    from gpu_util import make_tables
    from multike_amd.sampling import KGSide, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import EmbeddingTable, StepEngine
    n_ent, n_rel, d, N, P = 2_000_000, 2000, 256, 64, 5000
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, triples_per_entity=1.0, seed=5)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    sigma = float(np.sqrt(2.6 / (n_ent + d)))
    E = EmbeddingTable(n_ent, d, "ent", trainable=False)
    E.trainable = True
    E.data[:, :d] = torch.randn(n_ent, d, device="cuda", generator=g).clamp_(-2, 2) * sigma
    R = EmbeddingTable(n_rel, d, "rel", values=mo.xavier_truncated_normal((n_rel, d), np.random.default_rng(6)))
    eng = StepEngine()
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None), KGSide(kgs.entities(1), None), P, N,
                          seed=2)
    steps = [bat.batch(s) for s in (0, 7)]
    used = torch.unique(torch.cat([x.long() for pos, neg in steps for x in (pos[0], pos[2], neg[0], neg[2])]))
    remap = torch.full((n_ent,), -1, dtype=torch.int64, device="cuda")
    remap[used] = torch.arange(used.numel(), device="cuda")
    e64 = E.raw()[used].double().cpu().numpy()
    r64 = R.raw().double().cpu().numpy()
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    orc = co.RelationStepOracle(len(e64), n_rel, d, np.float64)
    before = E.data.clone()
    for pos, neg in steps:
        lp = eng.relation_step(E, R, "relation", pos, neg, neg_per_pos=N, lr=0.001)
        cp = (remap[pos[0].long()].cpu().numpy(), pos[1].cpu().numpy(), remap[pos[2].long()].cpu().numpy())
        cn = (remap[neg[0].long()].cpu().numpy(), neg[1].cpu().numpy(), remap[neg[2].long()].cpu().numpy())
        L = orc.step(e64, r64, a64, b64, cp, cn, 0.001)
        np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
    np.testing.assert_allclose(E.raw()[used].cpu().numpy(), e64, rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(R.raw().cpu().numpy(), r64, rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(E.slot("relation")[used][:, :d].cpu().numpy(), a64, rtol=1e-3, atol=1e-7)
    mask = torch.ones(n_ent, dtype=torch.bool, device="cuda")
    mask[used] = False
    assert torch.equal(E.data[mask], before[mask])                      # 1.7M untouched rows: bit-identical
    assert float(E.slot("relation")[mask].min()) == float(E.slot("relation")[mask].max()) == np.float32(0.1)
    assert float(E.grad.abs().max()) == 0.0 and float(R.grad.abs().max()) == 0.0
    assert int(E.refcount.abs().sum()) == 0
    del before
    pos, neg = bat.batch(3)
    h = pos[0].numel() // 2
    full = float(eng.relation_step(E, R, "relation", pos, neg, neg_per_pos=N, update=False).sum())
    parts = float(eng.relation_step(E, R, "relation", tuple(x[:h] for x in pos), tuple(x[:h * N] for x in neg), neg_per_pos=N,
                                    update=False).sum()) + \
        float(eng.relation_step(E, R, "relation", tuple(x[h:] for x in pos), tuple(x[h * N:] for x in neg), neg_per_pos=N,
                                update=False).sum())
    np.testing.assert_allclose(full, parts, rtol=1e-6)


@pytest.mark.parametrize("n_ent,P,N", [(300, 400, 25), (5000, 2000, 10)])
def test_heavy_collisions_exclusive_row_path(n_ent, P, N):
    """Few entities, many references: almost every row is referenced many times, some exactly once, the same corrupt
    entity shows up twice inside one group — exercises every branch of the exclusive-row bookkeeping against the oracle."""
    from gpu_util import dev_i32, make_tables
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(n_ent + P)
    d, n_rel = 75, 7
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    rel = mo.xavier_truncated_normal((n_rel, d), rng)
    # Zipf-like heads/tails, uniform corrupt entities
    pr = 1.0 / np.arange(1, n_ent + 1)
    pr /= pr.sum()
    ph, pt = rng.choice(n_ent, P, p=pr), rng.choice(n_ent, P, p=pr)
    prl = rng.integers(0, n_rel, P)
    nh, nt, nr = np.repeat(ph, N), np.repeat(pt, N), np.repeat(prl, N)
    side = rng.integers(0, 2, P * N).astype(bool)
    c = rng.integers(0, n_ent, P * N)
    nh = np.where(side, c, nh)
    nt = np.where(side, nt, c)
    nh[1], nt[1] = nh[0], nt[0]                    # the same negative twice in one group
    pos = tuple(a.astype(np.int32) for a in (ph, prl, pt))
    neg = tuple(a.astype(np.int32) for a in (nh, nr, nt))
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    E, R = make_tables(ent, rel)
    eng = StepEngine()
    for step in range(2):
        L, _, _ = mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01)
        lp = eng.relation_step(E, R, "relation", tuple(dev_i32(a) for a in pos), tuple(dev_i32(a) for a in neg),
                               neg_per_pos=N, lr=0.01)
        np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
    # hub rows sum hundreds of fp32 contributions (atomic order free): absolute tolerance ~1e-4 of a typical weight
    np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(R.raw().cpu().numpy(), r64, rtol=1e-4, atol=2e-5)
    # accumulator = sum of g^2 where a hub row's g is itself a cancelling sum of hundreds of fp32 terms in atomic order:
    # single elements land a few 1e-3 off in about one run out of 25
    acc = E.slot("relation")[:, :d].cpu().numpy()
    np.testing.assert_allclose(acc, a64, rtol=1e-2, atol=1e-6)
    # ... and that is the tail: all but a handful of elements agree to 2e-4 (the deterministic mode, `test_deterministic_*`,
    # holds 1e-3 on every element)
    assert np.mean(np.isclose(acc, a64, rtol=2e-4, atol=1e-6)) > 0.9995, float(np.mean(np.isclose(acc, a64, rtol=2e-4, atol=1e-6)))
    assert int(E.refcount.abs().sum()) == 0 and float(E.grad.abs().max()) == 0.0
    # and the two paths agree with each other
    E2, R2 = make_tables(ent, rel)
    eng2 = StepEngine()
    for step in range(2):
        eng2.relation_step(E2, R2, "relation", tuple(dev_i32(a) for a in pos), tuple(dev_i32(a) for a in neg), neg_per_pos=N,
                           lr=0.01, exclusive_rows=False)
    np.testing.assert_allclose(E.raw().cpu().numpy(), E2.raw().cpu().numpy(), rtol=1e-4, atol=5e-6)


@pytest.mark.parametrize("d,P,N", [(75, 100, 70), (75, 3000, 64), (32, 40, 100), (75, 9, 25), (256, 60, 64), (75, 700, 12)])
def test_lane_ids_path_blocks_and_slices(d, P, N):
    """The training instantiation that fetches a group's ids and reference counts once, one negative per lane (`score_lane_ids`):
    more than 64 negatives per positive (two id blocks), small batches (a group sliced over several wavefronts), two groups
    per wavefront (N <= 12), weights, an irregular negative — two steps against the float64 dense oracle, and against the
    per-round id fetch (`score_lane_ids = 0`)."""
    from gpu_util import dev_i32, make_tables
    from multike_amd import _lib
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(d + P * 7 + N)
    n_ent, n_rel = 4000, 11
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    rel = mo.xavier_truncated_normal((n_rel, d), rng)
    ph, pt, prl = rng.integers(0, n_ent, P), rng.integers(0, n_ent, P), rng.integers(0, n_rel, P)
    nh, nt, nr = np.repeat(ph, N), np.repeat(pt, N), np.repeat(prl, N)
    side = rng.integers(0, 2, P * N).astype(bool)
    c = rng.integers(0, n_ent, P * N)
    nh, nt = np.where(side, c, nh), np.where(side, nt, c)
    nh[N + 1], nt[N + 1] = rng.integers(0, n_ent, 2)                # an irregular negative in the second group
    nh[2 * N - 1] = ph[1]; nt[2 * N - 1] = pt[1]                    # a negative equal to its positive
    pos = tuple(a.astype(np.int32) for a in (ph, prl, pt))
    neg = tuple(a.astype(np.int32) for a in (nh, nr, nt))
    pw, nw = rng.uniform(0.5, 1.5, P).astype(np.float32), rng.uniform(0.5, 1.5, P * N).astype(np.float32)
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    losses64 = [mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01, pos_w=pw.astype(np.float64),
                                            neg_w=nw.astype(np.float64))[0] for _ in range(2)]
    outs = []
    for lane_ids in (1, 0):
        old = _lib.set_option("score_lane_ids", lane_ids)
        try:
            E, R = make_tables(ent, rel)
            eng = StepEngine()
            ls = [float(eng.relation_step(E, R, "relation", tuple(dev_i32(a) for a in pos), tuple(dev_i32(a) for a in neg),
                                          neg_per_pos=N, lr=0.01, pos_w=torch.as_tensor(pw, device="cuda"),
                                          neg_w=torch.as_tensor(nw, device="cuda")).sum()) for _ in range(2)]
            assert float(E.grad.abs().max()) == 0.0 and int(E.refcount.abs().sum()) == 0
            outs.append((ls, E.raw().cpu().numpy(), R.raw().cpu().numpy(), E.slot("relation")[:, :d].cpu().numpy()))
        finally:
            _lib.set_option("score_lane_ids", old)
    for ls, e, r, acc in outs:
        np.testing.assert_allclose(ls, losses64, rtol=LOSS_RTOL)
        np.testing.assert_allclose(e, e64, rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(r, r64, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(acc, a64, rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("d", [256, 128, 64, 180, 75])
def test_row_update_one_flag_per_lane(d):
    """`update_chunk = 64` (what tables of more than 500K rows get): one touched flag per lane; on strides that are a multiple of 64
    floats (d = 256, 128, 64, 180 -> 192) the whole wavefront then visits a row, 16 bytes per lane; d = 75 (stride 80) keeps the
    quarter-wave shape.  Two training steps against the float64 dense oracle, exclusive rows on and off."""
    from gpu_util import dev_i32, grouped_batch, make_tables
    from multike_amd import _lib
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(d)
    n_ent, n_rel, P, N = 20000, 23, 400, 10
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    rel = mo.xavier_truncated_normal((n_rel, d), rng)
    pos, neg = grouped_batch(rng, n_ent, n_rel, P, N)
    old = _lib.set_option("update_chunk", 64)
    try:
        for excl in (True, False):
            e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
            a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
            E, R = make_tables(ent, rel)
            eng = StepEngine()
            for step in range(2):
                L = mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01)[0]
                lp = eng.relation_step(E, R, "relation", tuple(dev_i32(a) for a in pos), tuple(dev_i32(a) for a in neg),
                                       neg_per_pos=N, lr=0.01, exclusive_rows=excl)
                np.testing.assert_allclose(float(lp.sum()), L, rtol=LOSS_RTOL)
            np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=ROW_RTOL, atol=ROW_ATOL)
            np.testing.assert_allclose(R.raw().cpu().numpy(), r64, rtol=ROW_RTOL, atol=ROW_ATOL)
            np.testing.assert_allclose(E.slot("relation")[:, :d].cpu().numpy(), a64, rtol=1e-4, atol=1e-7)
            assert float(E.grad.abs().max()) == 0.0 and float(R.grad.abs().max()) == 0.0
    finally:
        _lib.set_option("update_chunk", old)

