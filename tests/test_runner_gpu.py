"""The native step runner (mke_relation_steps) against (a) the Python-driven per-step path and (b) the CPU
oracle over a whole synthetic epoch: sampler chunking must not change the negatives, per-step losses and the
final tables must agree."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu


def _setup(seed=3, n_ent=4000, n_rel=30, d=75, B=500, N=10, zipf=0.0):
    from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import EmbeddingTable
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, seed=seed, zipf=zipf)
    rng = np.random.default_rng(seed)
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    rel = mo.xavier_truncated_normal((n_rel, d), rng)
    sides = []
    for k in (0, 1):
        t = torch.as_tensor(kgs.triples[k], device="cuda")
        sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))

    def fresh():
        E = EmbeddingTable(n_ent, d, "e", values=ent)
        R = EmbeddingTable(n_rel, d, "r", values=rel, grad_copies=2)
        bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], B, N, seed=42)
        return E, R, bat
    return kgs, ent, rel, fresh


@pytest.mark.parametrize("chunk,overlap", [(1, False), (7, False), (None, False), (1, True), (3, True), (None, True)])
def test_runner_equals_python_steps_and_oracle(chunk, overlap):
    from multike_amd.runner import RelationViewRunner
    from multike_amd.tables import StepEngine
    kgs, ent, rel, fresh = _setup()
    d, N = 75, 10
    # (a) native runner, one call for the whole epoch
    E1, R1, bat1 = fresh()
    run = RelationViewRunner(E1, R1, bat1, "relation", lr=0.01, sample_chunk=chunk, overlap=overlap)
    run.run()
    l_native = run.step_losses().cpu().numpy()
    # (b) python-driven steps
    E2, R2, bat2 = fresh()
    eng = StepEngine()
    l_py, negs = [], []
    for s in range(bat2.steps):
        pos, neg = bat2.batch(s)
        negs.append([x.cpu().numpy() for x in neg])
        l_py.append(float(eng.relation_step(E2, R2, "relation", pos, neg, neg_per_pos=N, lr=0.01).sum()))
    np.testing.assert_allclose(l_native, l_py, rtol=2e-6)
    np.testing.assert_allclose(E1.raw().cpu().numpy(), E2.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(R1.raw().cpu().numpy(), R2.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)
    # (c) oracle: same epoch with the oracle's sampler (bit-exact negatives) and float64 step
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    orc = co.RelationStepOracle(len(e64), len(r64), d, np.float64)
    sets = [co.TripleSet(t[:, 0], t[:, 1], t[:, 2]) for t in kgs.triples]
    ph, pr, pt = (x.cpu().numpy() for x in (bat2.pos_h, bat2.pos_r, bat2.pos_t))
    l_or = []
    for s in range(bat2.steps):
        lo, hi = int(bat2.off[s]), int(bat2.off[s + 1])
        mid = lo + int(bat2.cnt1[s])
        parts = []
        for k, (a, b) in enumerate(((lo, mid), (mid, hi))):
            elo, ehi = kgs.ent_range[k]
            parts.append(co.neg_sample(ph[a:b], pr[a:b], pt[a:b], N, ehi - elo, ent_lo=elo, known=sets[k], seed=(42, 0),
                                       stream_id=k, pos_offset=a))
        neg = [np.concatenate([parts[0][i], parts[1][i]]) for i in range(3)]
        for i in range(3):
            assert np.array_equal(neg[i], negs[s][i]), f"negatives differ at step {s}"
        l_or.append(orc.step(e64, r64, a64, b64, (ph[lo:hi], pr[lo:hi], pt[lo:hi]), neg, 0.01))
    np.testing.assert_allclose(l_native, l_or, rtol=1e-5)
    np.testing.assert_allclose(E1.raw().cpu().numpy(), e64, rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(R1.raw().cpu().numpy(), r64, rtol=5e-4, atol=5e-6)
    # the epoch metric the reference prints: sum(batch_loss)/#positives (code/MultiKE_model.py:313)
    np.testing.assert_allclose(float(run.epoch_loss_sum()) / len(ph), np.sum(l_or) / len(ph), rtol=1e-5)


def test_runner_partial_ranges_and_second_epoch():
    from multike_amd.runner import RelationViewRunner
    kgs, ent, rel, fresh = _setup(seed=5)
    E1, R1, bat1 = fresh()
    E2, R2, bat2 = fresh()
    r1 = RelationViewRunner(E1, R1, bat1, lr=0.01, sample_chunk=4, overlap=True)
    r2 = RelationViewRunner(E2, R2, bat2, lr=0.01, overlap=False)
    for ep in range(2):
        r1.run(0, 5); r1.run(5, 6); r1.run(6, None)          # same epoch in three calls
        r2.run()
        np.testing.assert_allclose(r1.step_losses().cpu().numpy(), r2.step_losses().cpu().numpy(), rtol=2e-6)
        bat1.shuffle(); bat2.shuffle()
    np.testing.assert_allclose(E1.raw().cpu().numpy(), E2.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert int(r1.refcount.abs().sum()) == 0 and int(r2.refcount.abs().sum()) == 0     # zero-invariant restored
    assert not torch.equal(bat1.pos_h, _setup(seed=5)[3]()[2].pos_h)  # the shuffle really permuted the epoch


def test_next_step_counting_in_the_score_or_in_the_update_launch():
    """`count_in_score` (default 1): the next step's reference counts are taken by rider blocks of the score launch; 0: of the
    update launch.  Same steps either way, and the same as counting in a launch of its own (sample_chunk = 1 never counts
    ahead)."""
    from multike_amd import _lib
    from multike_amd.runner import RelationViewRunner
    kgs, ent, rel, fresh = _setup(seed=9)
    outs = []
    for opt, chunk in ((1, None), (0, None), (1, 1)):
        old = _lib.set_option("count_in_score", opt)
        try:
            E, R, bat = fresh()
            r = RelationViewRunner(E, R, bat, lr=0.01, sample_chunk=chunk)
            r.run()
            outs.append((r.step_losses().cpu().numpy(), E.raw().cpu().numpy(), R.raw().cpu().numpy()))
            assert int(r.refcount.abs().sum()) == 0
        finally:
            _lib.set_option("count_in_score", old)
    for l, e, rr in outs[1:]:
        np.testing.assert_allclose(l, outs[0][0], rtol=2e-6)
        np.testing.assert_allclose(e, outs[0][1], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(rr, outs[0][2], rtol=1e-4, atol=1e-6)


def test_run_epochs_prefetch_equals_plain_epochs():
    """run_epochs (next epoch permuted + sampled on a side stream) == the plain loop run(); shuffle(); run(); ..."""
    from multike_amd.runner import RelationViewRunner
    kgs, ent, rel, fresh = _setup(seed=8)
    E1, R1, bat1 = fresh()
    E2, R2, bat2 = fresh()
    r1 = RelationViewRunner(E1, R1, bat1, lr=0.01)
    r2 = RelationViewRunner(E2, R2, bat2, lr=0.01)
    per_epoch = []
    r1.run_epochs(3, on_epoch_end=lambda e, r: per_epoch.append(r.step_losses().cpu().numpy().copy()), prefetch=True)
    for e in range(3):
        if e > 0:
            bat2.shuffle()
        r2.run()
        np.testing.assert_allclose(per_epoch[e], r2.step_losses().cpu().numpy(), rtol=2e-6)
    assert torch.equal(bat1.pos_h, bat2.pos_h) and bat1.epoch == bat2.epoch == 2
    np.testing.assert_allclose(E1.raw().cpu().numpy(), E2.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(R1.raw().cpu().numpy(), R2.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert int(r1.refcount.abs().sum()) == 0


@pytest.mark.parametrize("d,N,n_ent,wide", [(75, 10, 4000, False), (256, 40, 3000, False), (32, 3, 2000, False), (64, 25, 20000, True)])
def test_hub_rows_private_copies_are_the_same_function(d, N, n_ent, wide):
    """Heavy-tailed KGs (SURVEY 8d's Zipf(1.0) variant; code/base/batch.py:45-54 feeds real, hub-heavy triples): the runner
    declares the entities that several positives of every step share as hub rows and sends the groups' flushes of their
    gradient to private copies (include/multike_hip.h mke_hot_rows).  Same function: losses and tables equal the run without
    hub rows and the Python-driven steps; the copy rows are all zero again after every epoch.  wide: the whole-wavefront row
    form of the update kernel (forced through "update_chunk")."""
    from multike_amd import _lib
    from multike_amd.runner import RelationViewRunner
    from multike_amd.tables import StepEngine
    kgs, ent, rel, fresh = _setup(seed=11, n_ent=n_ent, d=d, N=N, zipf=1.0)
    old = _lib.set_option("update_chunk", 64 if wide else 0)
    old_min = RelationViewRunner.HOT_MIN
    RelationViewRunner.HOT_MIN = 6.0          # these small KGs: 15-20 hub rows instead of the 3-4 above the product's threshold
    try:
        E1, R1, bat1 = fresh()
        r1 = RelationViewRunner(E1, R1, bat1, lr=0.01)
        assert E1.n_hot >= 12 and E1.hot_copies == 8 and E1._grad_full.shape[0] == n_ent + 8 * E1.n_hot
        E2, R2, bat2 = fresh()
        r2 = RelationViewRunner(E2, R2, bat2, lr=0.01, hot_rows=False)
        assert E2.n_hot == 0
        for ep in range(2):
            r1.run(); r2.run()
            np.testing.assert_allclose(r1.step_losses().cpu().numpy(), r2.step_losses().cpu().numpy(), rtol=2e-6)
            assert float(E1._grad_full.abs().max()) == 0.0            # the table's own rows AND the copies consumed
            bat1.shuffle(); bat2.shuffle()
        # hub rows: a gradient row is a float32 sum of tens of terms in two different orders (and the atomics' order changes from
        # run to run: one element of 1.28M was seen 3.7e-6 apart at |x| = 7e-3)
        np.testing.assert_allclose(E1.raw().cpu().numpy(), E2.raw().cpu().numpy(), rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(R1.raw().cpu().numpy(), R2.raw().cpu().numpy(), rtol=2e-4, atol=1e-5)
        # ... and the Python-driven steps on a hub-declared table (no copies used there) still agree
        E3, R3, bat3 = fresh()
        eng = StepEngine()
        for s in range(bat3.steps):
            pos, neg = bat3.batch(s)
            eng.relation_step(E3, R3, "relation", pos, neg, neg_per_pos=N, lr=0.01)
        E4, R4, bat4 = fresh()
        r4 = RelationViewRunner(E4, R4, bat4, lr=0.01)
        r4.run()
        np.testing.assert_allclose(E4.raw().cpu().numpy(), E3.raw().cpu().numpy(), rtol=2e-4, atol=1e-5)   # same sums, other orders: one element of 1.28M was seen 4.6e-6 apart (1 run in 10)
    finally:
        _lib.set_option("update_chunk", old)
        RelationViewRunner.HOT_MIN = old_min
