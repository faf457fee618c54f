"""Per-plan tuning (mke_tuning, ABI 105; round-5 review item 7): the performance knobs travel with the plan / the call, so two
trainers in ONE process hold different settings at the same time; mke_set_option only sets the defaults."""
import threading

import numpy as np
import pytest
import torch

from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu


def test_two_step_engines_with_different_settings_concurrently():
    """Two StepEngines, two host threads, two streams, opposite knob settings (score_splits 1 / 4, half groups off / on, 32- /
    64-bit offsets, lane ids on / off, update chunk 16 / 64) stepping their own tables at the same time: both agree with the
    float64 oracle, and the process defaults are what they were."""
    from gpu_util import dev_i32, grouped_batch, make_tables
    from multike_amd import _lib
    from multike_amd.tables import StepEngine
    before = {k: _lib.get_option(k) for k in ("score_splits", "score_half_groups", "score_offsets32", "score_lane_ids", "update_chunk")}
    settings = [dict(score_splits=1, score_half_groups=0, score_offsets32=1, score_lane_ids=1, update_chunk=16),
                dict(score_splits=4, score_half_groups=64, score_offsets32=0, score_lane_ids=0, update_chunk=64)]
    out, err = [None, None], []

    def work(k):
        try:
            rng = np.random.default_rng(77 + k)
            d, n_ent, n_rel, P, N = 75, 20_000, 13, 900, 10
            ent = mo.xavier_truncated_normal((n_ent, d), rng)
            rel = mo.xavier_truncated_normal((n_rel, d), rng)
            pos, neg = grouped_batch(rng, n_ent, n_rel, P, N)
            e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
            a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                E, R = make_tables(ent, rel)
                eng = StepEngine(tuning=settings[k])
                losses = []
                for step in range(6):
                    L, _, _ = mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01)
                    lp = eng.relation_step(E, R, "relation", tuple(dev_i32(a) for a in pos), tuple(dev_i32(a) for a in neg),
                                           neg_per_pos=N, lr=0.01)
                    losses.append((float(lp.sum()), L))
                st.synchronize()
                out[k] = (losses, E.raw().cpu().numpy(), e64, R.raw().cpu().numpy(), r64)
        except Exception as ex:      # noqa: BLE001
            err.append(ex)

    ts = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for losses, e, e64, r, r64 in out:
        for got, exp in losses:
            np.testing.assert_allclose(got, exp, rtol=2e-6)
        np.testing.assert_allclose(e, e64, rtol=2e-4, atol=5e-6)      # six float32 steps against float64
        np.testing.assert_allclose(r, r64, rtol=2e-4, atol=5e-6)
    assert before == {k: _lib.get_option(k) for k in before}


def test_a_plans_tuning_is_read_and_defaults_are_untouched():
    """A knob in a plan's mke_tuning decides the launch whatever the process default says: with the default of `score_half_groups`
    at 0 (never two groups per wavefront) and the plan's at 64, and the other way round, the runner's epoch is the same function
    (losses to 2e-6 against each other and a third runner on plain defaults)."""
    from multike_amd import _lib
    from multike_amd.runner import RelationViewRunner
    from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import EmbeddingTable
    kgs = SyntheticKGs(n_ent=4000, n_rel=20, seed=3)
    rng = np.random.default_rng(3)
    ent0, rel0 = mo.xavier_truncated_normal((4000, 75), rng), mo.xavier_truncated_normal((20, 75), rng)

    def epoch(tuning, default_half):
        old = _lib.set_option("score_half_groups", default_half)
        try:
            sides = []
            for k in (0, 1):
                t = torch.as_tensor(kgs.triples[k], device="cuda")
                sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
            bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 500, 10, seed=9)
            E, R = EmbeddingTable(4000, 75, "e", values=ent0), EmbeddingTable(20, 75, "r", values=rel0)
            run = RelationViewRunner(E, R, bat, lr=0.01, tuning=tuning)
            run.run()
            torch.cuda.synchronize()
            return run.loss.sum(1).cpu().numpy(), E.raw().cpu().numpy()
        finally:
            _lib.set_option("score_half_groups", old)

    base_l, base_e = epoch(None, -1)
    for tuning, dflt in ((dict(score_half_groups=64, count_in_score=0), 0), (dict(score_half_groups=0, sampler_fast=0), 64)):
        l, e = epoch(tuning, dflt)
        np.testing.assert_allclose(l, base_l, rtol=2e-6)
        np.testing.assert_allclose(e, base_e, rtol=1e-4, atol=2e-6)
    t = _lib.tuning(score_splits=3)
    assert t.score_splits == 3 and t.update_chunk == _lib.TUNE_DEFAULT and t.reserved[0] == _lib.TUNE_DEFAULT
    with pytest.raises(_lib.MultiKEHipError):
        _lib.tuning(no_such_knob=1)
