"""The seven training phases of an ITC epoch on row-sharded tables (multike_amd/distributed_model.py): two ranks SHARING the
one GPU of the test box (collectives staged through gloo) against ONE rank on the same global batches — sharding must not change
the result beyond fp32 atomic-order noise — and the one-rank run's phases against the dense float64 oracle where one exists for
the exact batches (the relation view's first epoch)."""
import os

import numpy as np
import pytest
import torch

from oracle import attr_cnn_oracle as ao
from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu

N_ENT, N_REL, N_ATTR, N_LIT, DIM, SEED = 1200, 12, 30, 200, 32, 9
B, AB, EB, NEG, EPOCHS = 400, 300, 250, 4, 2


def _setup():
    from multike_amd.synthetic import SyntheticKGs
    kgs = SyntheticKGs(n_ent=N_ENT, n_rel=N_REL, seed=SEED)
    rng = np.random.default_rng(SEED)
    t = lambda n: mo.xavier_truncated_normal((n, DIM), rng).astype(np.float32)
    unit = lambda n: (lambda x: (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32))(rng.standard_normal((n, DIM)))
    tables = {"rv_ent": t(N_ENT), "av_ent": t(N_ENT), "ent": t(N_ENT), "name": unit(N_ENT), "rel": t(N_REL), "attr": t(N_ATTR),
              "lit": unit(N_LIT)}
    cnn = []
    for k in range(3):
        P = ao.init_params(DIM, rng)
        P["bias"] = 0.05 * rng.standard_normal(DIM)
        cnn.append(P)
    ri = lambda hi, n: rng.integers(0, hi, n)
    lists = {
        "attr": [(int(h), int(a), int(v), float(w)) for h, a, v, w in zip(ri(N_ENT, 700), ri(N_ATTR, 700), ri(N_LIT, 700), rng.uniform(0.3, 1, 700))],
        "ckge_rel": [(int(h), int(r), int(t_)) for h, r, t_ in zip(ri(N_ENT, 500), ri(N_REL, 500), ri(N_ENT, 500))],
        "ckgp_rel": [(int(h), int(r), int(t_), float(w)) for h, r, t_, w in zip(ri(N_ENT, 300), ri(N_REL, 300), ri(N_ENT, 300), rng.uniform(0.3, 1, 300))],
        "ckge_attr": [(int(h), int(a), int(v)) for h, a, v in zip(ri(N_ENT, 450), ri(N_ATTR, 450), ri(N_LIT, 450))],
        "ckga_attr": [(int(h), int(a), int(v), float(w)) for h, a, v, w in zip(ri(N_ENT, 200), ri(N_ATTR, 200), ri(N_LIT, 200), rng.uniform(0.3, 1, 200))],
        "entities": [int(x) for x in rng.choice(N_ENT, 600, replace=False)],
    }
    return kgs, tables, cnn, lists


def _matrices():
    rng = np.random.default_rng(SEED + 1)
    return [np.linalg.qr(rng.standard_normal((DIM, DIM)))[0].astype(np.float32) + 0.01 * rng.standard_normal((DIM, DIM)).astype(np.float32)
            for _ in range(3)]


def _train(rank, world, comm_oc=None, comm_views=None, ssl=False):
    from multike_amd.distributed_model import ShardedITC
    kgs, tables, cnn, lists = _setup()
    m = ShardedITC(kgs, tables, cnn, lists, rank, world, batch_size=B, attribute_batch_size=AB, entity_batch_size=EB,
                   neg_triple_num=NEG, learning_rate=0.01, itc_learning_rate=0.02, cv_name_weight=0.7, cv_weight=1.3, seed=SEED,
                   comm_oc=comm_oc, comm_views=comm_views, mapping_matrices=_matrices() if ssl else None)
    losses = [(m.epoch_ssl(i) if ssl else m.epoch(i)) for i in range(1, EPOCHS + 1)]
    return m, losses


def _worker(rank, world, port, ret, ssl=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_oc import OcHostStagedComm
        from multike_amd.distributed_views import HostStagedViewComm
        torch.cuda.set_device(0)
        m, losses = _train(rank, world, OcHostStagedComm(), HostStagedViewComm(), ssl)
        out = m.gather()
        if rank == 0:
            ret.put((out, losses))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("ssl", [False, True])
def test_two_ranks_equal_one_rank_over_two_epochs(ssl):
    """ssl: the SSL schedule's phases — space mapping of the three views onto the shared table instead of the common-space step."""
    import tempfile
    import torch.multiprocessing as mp
    m1, l1 = _train(0, 1, ssl=ssl)
    ref = m1.gather()
    port = tempfile.mktemp(prefix="mke_rdv_")
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, ssl)) for r in range(2)]
    for p in procs:
        p.start()
    got, l2 = ret.get(timeout=800)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for ep in range(EPOCHS):
        for k in l1[ep]:
            assert np.isfinite(l1[ep][k]) and l1[ep][k] > 0.0, (ep, k)
            np.testing.assert_allclose(l2[ep][k], l1[ep][k], rtol=2e-5, err_msg=f"epoch {ep} {k}")
    for k in ("ent", "rv", "av", "rel", "attr"):
        np.testing.assert_allclose(got[k], ref[k], rtol=2e-4, atol=2e-6, err_msg=k)
    if ssl:
        np.testing.assert_allclose(got["matrices"], ref["matrices"], rtol=2e-4, atol=2e-6)
        assert np.abs(ref["matrices"] - np.stack(_matrices())).max() > 1e-4
    for a, b in zip(got["cnn"], ref["cnn"]):
        for k in a:
            np.testing.assert_allclose(a[k], b[k], rtol=2e-3, atol=1e-4, err_msg=k)
    # every table moved (each phase trained something) and the scratch is consumed
    _, tables, _, _ = _setup()
    assert np.abs(ref["ent"] - tables["ent"]).max() > 1e-4 and np.abs(ref["rv"] - tables["rv_ent"]).max() > 1e-4
    assert np.abs(ref["av"] - tables["av_ent"]).max() > 1e-4 and np.abs(ref["attr"] - tables["attr"]).max() > 1e-5
    for t in (m1.rv_ent, m1.av_ent, m1.ent, m1.rel, m1.attr):
        assert float(t.grad.abs().max()) == 0.0


def test_soft_alignment_gate_and_list_refresh_one_rank():
    """code/MultiKE_CSL.py:62-70, 80-87: the two predicate-alignment phases run only when i > start_predicate_soft_alignment,
    and their lists are rebuilt between epochs (`set_lists`) without resetting the loops' optimizers."""
    from multike_amd.distributed_model import ShardedITC
    kgs, tables, cnn, lists = _setup()
    m = ShardedITC(kgs, tables, cnn, lists, 0, 1, batch_size=B, attribute_batch_size=AB, entity_batch_size=EB, neg_triple_num=NEG,
                   learning_rate=0.01, itc_learning_rate=0.02, seed=SEED, start_predicate_soft_alignment=1)
    first = m.epoch(1)
    assert "ckgp_rel" not in first and "ckga_attr" not in first and first["relation"] > 0 and first["ckge_rel"] > 0
    # nothing has trained under the gated loops' optimizers yet: their Adagrad slots are still at the initial 0.1
    assert "ckgp_rel" not in m.rv_ent.slots or float((m.rv_ent.slot("ckgp_rel")[:, :DIM] - 0.1).abs().max()) == 0.0
    second = m.epoch(2)
    assert second["ckgp_rel"] > 0 and second["ckga_attr"] > 0
    acc_before = m.rel.slot("ckgp_rel").clone()
    assert float((acc_before[:, :DIM] - 0.1).abs().max()) > 0.0
    steps_before = m.ckgp_rel.steps
    rng = np.random.default_rng(3)
    new_rel = [(int(h), int(r), int(t_), 0.5) for h, r, t_ in zip(rng.integers(0, N_ENT, 900), rng.integers(0, N_REL, 900),
                                                                   rng.integers(0, N_ENT, 900))]
    new_attr = [(int(h), int(a), int(v), 0.9) for h, a, v in zip(rng.integers(0, N_ENT, 50), rng.integers(0, N_ATTR, 50),
                                                                 rng.integers(0, N_LIT, 50))]
    m.set_lists(ckgp_rel=new_rel, ckga_attr=new_attr)
    assert m.ckgp_rel.steps == 3 and steps_before == 1           # ceil(900 / 400) against ceil(300 / 400)
    third = m.epoch(3)
    assert third["ckgp_rel"] > 0 and third["ckga_attr"] > 0
    acc_after = m.rel.slot("ckgp_rel")
    assert bool((acc_after >= acc_before).all()) and float((acc_after - acc_before).abs().max()) > 0.0   # same slot, grown
    m.set_lists(ckgp_rel=[])                                      # an empty list switches the phase off
    assert m.epoch(4)["ckgp_rel"] == 0.0
    for t in (m.rv_ent, m.rel, m.av_ent, m.attr):
        assert float(t.grad.abs().max()) == 0.0
