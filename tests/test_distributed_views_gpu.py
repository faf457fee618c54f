"""The sharded attribute view / common-space step with the HIP kernels (mke_attr_step_phases, mke_align_*): two ranks
SHARING the one GPU of the test box (collectives staged through gloo) against the float64 dense oracle on the same global
batches; the exchange logic itself is covered under gloo in tests/test_distributed_views_cpu.py."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import attr_cnn_oracle as ao
from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu

N_ENT, N_ATTR, N_LIT, DIM, B, STEPS, SEED = 900, 40, 300, 75, 700, 3, 5


def _attr_data():
    rng = np.random.default_rng(SEED)
    ent = mo.xavier_truncated_normal((N_ENT, DIM), rng)
    attr = mo.xavier_truncated_normal((N_ATTR, DIM), rng)
    lit = rng.standard_normal((N_LIT, DIM)).astype(np.float32)
    lit /= np.linalg.norm(lit, axis=1, keepdims=True)
    P = ao.init_params(DIM, rng)
    P["bias"] = 0.05 * rng.standard_normal(DIM)
    batches = []
    for s in range(STEPS):
        ih = rng.integers(0, N_ENT, B)
        if s == 1:
            ih = 2 * rng.integers(0, N_ENT // 2, 64)           # rank 1 of 2 owns none of this step's triples
        n = len(ih)
        batches.append((ih, rng.integers(0, N_ATTR, n), rng.integers(0, N_LIT, n), rng.uniform(0.2, 1.0, n)))
    return ent, attr, lit, P, batches


def _attr_worker(rank, world, port, ret, mode="parallel"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed_views import HostStagedViewComm, ShardedAttributeView
        torch.cuda.set_device(0)
        ent, attr, lit, P, batches = _attr_data()
        v = ShardedAttributeView(ent, attr, lit, P, rank, world, lr=0.01, comm=HostStagedViewComm(), mode=mode)
        for (ih, ia, iv, w) in batches:
            v.step(ih, ia, iv, w, scale=2.0)
        loss = v.epoch_loss()
        full, a, p = v.gather()
        # replicated state must be BIT-identical on the two ranks (rank 0's gradients are everybody's in "replicated" mode;
        # all-reduced sums in "parallel" mode)
        import torch.distributed as dist2
        mine = torch.cat([v.backend.cnn.params.detach().reshape(-1), v.backend.attr.data.reshape(-1)]).cpu()
        other = mine.clone()
        dist2.broadcast(other, 0)
        same = torch.tensor([1 if torch.equal(mine, other) else 0])
        dist2.all_reduce(same, op=dist2.ReduceOp.MIN)
        assert int(same) == 1, "replicated parameters differ between the ranks"
        ok = float(v.backend.cnn.grads.abs().max()) == 0.0 and float(v.backend.attr.grad.abs().max()) == 0.0 and \
            float(v.backend.ent.grad.abs().max()) == 0.0
        if rank == 0:
            ret.put((full, a, p, loss, ok))
    finally:
        dist.destroy_process_group()


def _run(worker, world=2, extra=()):
    import torch.multiprocessing as mp
    import tempfile
    port = tempfile.mktemp(prefix="mke_rdv_")   # rendezvous file (init_method="file://..."): no TCP port to collide on
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, ret) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    out = ret.get(timeout=480)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return out


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["parallel", "replicated"])
def test_two_ranks_attribute_view_equals_dense_oracle(mode):
    """mode "replicated": every rank computes the whole step on the gathered head rows (1 all-reduce + 1 broadcast per step
    instead of 4 all-reduces), applies the heads it owns; a rank that owns none of a step's heads still updates the replicated
    state identically."""
    full, a, p, loss, ok = _run(_attr_worker, extra=(mode,))
    ent, attr, lit, P, batches = _attr_data()
    p64 = {k: v.astype(np.float64) for k, v in P.items()}
    acc = {k: np.full_like(v, 0.1) for k, v in p64.items()}
    e64, a64, l64 = ent.astype(np.float64), attr.astype(np.float64), lit.astype(np.float64)
    ae, aa = np.full_like(e64, 0.1), np.full_like(a64, 0.1)
    tot = 0.0
    for (ih, ia, iv, w) in batches:
        L, _ = ao.attribute_step_dense(p64, acc, e64, a64, l64, ae, aa, ih, ia, iv, w.astype(np.float32).astype(np.float64), 2.0, 0.01)
        tot += L
    assert ok
    np.testing.assert_allclose(loss, tot, rtol=1e-5)
    np.testing.assert_allclose(full, e64, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(a, a64, rtol=2e-3, atol=2e-5)
    for k in ao.PARAM_NAMES:
        np.testing.assert_allclose(p[k], p64[k], rtol=2e-3, atol=1e-4, err_msg=k)


def _cs_data():
    rng = np.random.default_rng(SEED + 1)
    mk = lambda: mo.xavier_truncated_normal((N_ENT, DIM), rng)
    ent, rv, av = mk(), mk(), mk()
    name = rng.standard_normal((N_ENT, DIM)).astype(np.float32)
    name /= np.linalg.norm(name, axis=1, keepdims=True)
    batches = [rng.choice(N_ENT, 400, replace=False) for _ in range(STEPS)]
    return ent, name, rv, av, batches


def _cs_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed_views import HostStagedViewComm, ShardedCommonSpace
        torch.cuda.set_device(0)
        ent, name, rv, av, batches = _cs_data()
        v = ShardedCommonSpace(ent, name, rv, av, rank, world, lr=0.02, cv_name_weight=0.7, cv_weight=1.3, comm=HostStagedViewComm())
        for ids in batches:
            v.step(ids)
        loss = v.epoch_loss()
        out = v.gather()
        if rank == 0:
            ret.put((out, loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_common_space_equals_dense_oracle():
    out, loss = _run(_cs_worker)
    ent, name, rv, av, batches = (x.astype(np.float64) if isinstance(x, np.ndarray) else x for x in _cs_data())
    accs = [np.full_like(ent, 0.1) for _ in range(3)]
    tot = sum(mo.common_space_step_dense(ent, name, rv, av, accs[0], accs[1], accs[2], ids, 0.02, 0.7, 1.3) for ids in batches)
    np.testing.assert_allclose(loss, tot, rtol=2e-6)
    for k, ref in (("ent", ent), ("rv", rv), ("av", av)):
        np.testing.assert_allclose(out[k], ref, rtol=2e-4, atol=2e-6, err_msg=k)


def _sm_data():
    rng = np.random.default_rng(SEED + 2)
    mk = lambda: mo.xavier_truncated_normal((N_ENT, DIM), rng)
    ent, views = mk(), [mk(), mk(), mk()]
    mats = [(np.eye(DIM) + 0.05 * rng.standard_normal((DIM, DIM))).astype(np.float32) for _ in range(3)]
    batches = [rng.choice(N_ENT, 500, replace=False) for _ in range(STEPS)]
    batches[1] = 2 * rng.choice(N_ENT // 2, 60, replace=False)   # rank 1 of 2 owns none of this step's entities
    return ent, views, mats, batches


def _sm_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed_views import HostStagedViewComm, ShardedSpaceMapping
        torch.cuda.set_device(0)
        ent, views, mats, batches = _sm_data()
        v = ShardedSpaceMapping(ent, views, mats, rank, world, lr=0.01, orthogonal_weight=2.0, comm=HostStagedViewComm())
        for ids in batches:
            v.step(ids)
        loss = v.epoch_loss()
        full, M = v.gather()
        ok = float(v.backend.state.gM.abs().max()) == 0.0 and float(v.backend.ent.grad.abs().max()) == 0.0
        if rank == 0:
            ret.put((full, M, loss, ok))
    finally:
        dist.destroy_process_group()


def _sm_reference():
    ent, views, mats, batches = _sm_data()
    e64 = ent.astype(np.float64)
    v64 = [(v.astype(np.float64), True) for v in views]
    m64 = [m.astype(np.float64) for m in mats]
    acc_e, acc_m = np.full_like(e64, 0.1), [np.full_like(m, 0.1) for m in m64]
    tot = sum(mo.space_mapping_step_dense(e64, acc_e, v64, m64, acc_m, ids, 0.01, 2.0) for ids in batches)
    return e64, np.stack(m64), tot


@pytest.mark.timeout(600)
def test_two_ranks_space_mapping_equals_dense_oracle():
    """mke_mapping_step_phases on two ranks sharing the GPU: per view the batch-wide sums and the matrix gradients travel
    through the (host-staged) all-reduces; a step in which rank 1 owns nothing."""
    full, M, loss, ok = _run(_sm_worker)
    e64, m64, tot = _sm_reference()
    assert ok
    np.testing.assert_allclose(loss, tot, rtol=2e-5)
    np.testing.assert_allclose(full, e64, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(M, m64, rtol=2e-4, atol=2e-6)


def test_one_rank_space_mapping_phases_equal_the_single_call():
    """The phased step at world 1 (no process group) against the float64 oracle — and therefore against mke_mapping_step,
    which tests/test_mapping_gpu.py pins to the same oracle."""
    from multike_amd.distributed_views import ShardedSpaceMapping
    ent, views, mats, batches = _sm_data()
    v = ShardedSpaceMapping(ent, views, mats, 0, 1, lr=0.01, orthogonal_weight=2.0)
    for ids in batches:
        v.step(ids)
    loss = v.epoch_loss()
    full, M = v.gather()
    e64, m64, tot = _sm_reference()
    np.testing.assert_allclose(loss, tot, rtol=2e-5)
    np.testing.assert_allclose(full, e64, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(M, m64, rtol=2e-4, atol=2e-6)


# ----------------------------------------------------------------------------------------------------------------------
# literal auto-encoder (mke_ae_step_phases)
# ----------------------------------------------------------------------------------------------------------------------
AE_DIMS, AE_LR = [150, 96, 40, 20], 0.01


def _ae_data():
    from oracle import literal_oracle as lo
    rng = np.random.default_rng(23)
    p = {k: 0.2 * v for k, v in lo.init_params(AE_DIMS, rng).items()}
    batches = [rng.standard_normal((m, AE_DIMS[0])).astype(np.float32) for m in (300, 1, 129)]   # the 1-row batch: rank 1 has none
    return p, batches


def _ae_reference(active):
    from oracle import literal_oracle as lo
    p, batches = _ae_data()
    acc = {k: np.full_like(v, 0.1) for k, v in p.items()}
    tot = 0.0
    for x in batches:
        L, g = lo.loss_and_grads(p, x.astype(np.float64), len(AE_DIMS) - 1, active, True)
        lo.adagrad_step(p, acc, g, AE_LR)
        tot += L
    return p, tot


def _ae_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_views import HostStagedViewComm, ShardedAutoEncoder
        torch.cuda.set_device(0)
        p, batches = _ae_data()
        v = ShardedAutoEncoder(p, AE_DIMS, rank, world, lr=AE_LR, active="tanh", normalize=True, comm=HostStagedViewComm())
        for x in batches:
            v.step(x)
        loss = v.epoch_loss()
        m = v.backend.model
        ok = float(m.grads.abs().max()) == 0.0 and float(m._partials.abs().max()) == 0.0
        if rank == 0:
            ret.put((v.params(), loss, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_auto_encoder_equals_dense_oracle():
    """mke_ae_step_phases on two ranks sharing the GPU (rows r, r + 2, ... of every batch each): the two batch-wide sums of the
    whole-matrix normalisation of the code and the packed gradient travel through the (host-staged) all-reduces; a batch in
    which rank 1 has no row."""
    got, loss, ok = _run(_ae_worker)
    p64, tot = _ae_reference("tanh")
    assert ok
    np.testing.assert_allclose(loss, tot, rtol=2e-5)
    for k in p64:
        np.testing.assert_allclose(got[k], p64[k], rtol=2e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("active", ["thah", "tanh"])
def test_one_rank_auto_encoder_phases_equal_the_single_call(active):
    """The phased step at world 1 against mke_ae_train_steps on the same batches and against the float64 oracle."""
    from types import SimpleNamespace
    from multike_amd.distributed_views import ShardedAutoEncoder
    from multike_amd.literal_encoder import AutoEncoderModel
    p, batches = _ae_data()
    v = ShardedAutoEncoder(p, AE_DIMS, 0, 1, lr=AE_LR, active=active, normalize=True)
    for x in batches:
        v.step(x)
    loss = v.epoch_loss()
    args = SimpleNamespace(dim=AE_DIMS[-1], encoder_normalize=True, encoder_active=active, optimizer="Adagrad", learning_rate=AE_LR)
    m = AutoEncoderModel(np.zeros((0, AE_DIMS[0]), dtype=np.float32), args, input_dimension=AE_DIMS[0], hidden_dimensions=AE_DIMS[1:], seed=0)
    m.set_params(p)
    tot1 = sum(float(m.train_step(torch.as_tensor(x, device="cuda"))) for x in batches)
    got, ref1 = v.params(), m.numpy_params()
    for k in got:   # same kernels in the same order; the split-K products and the bias column sums add atomically
        np.testing.assert_allclose(got[k], ref1[k], rtol=1e-4, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(loss, tot1, rtol=1e-6)
    p64, tot = _ae_reference(active)
    np.testing.assert_allclose(loss, tot, rtol=2e-5)
    for k in p64:
        np.testing.assert_allclose(got[k], p64[k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_two_attribute_graphs_on_shared_tables_one_rank():
    """The attribute view and a cross-KG attribute-inference graph on the SAME tables (HIP kernels, one rank): a CNN parameter
    set and an Adagrad slot each (EmbeddingTable slots are per optimizer name), alternating steps, against the dense oracle."""
    from multike_amd.distributed_views import ShardedAttributeView
    ent, attr, lit, P, batches = _attr_data()
    P2 = {k: (0.5 * np.asarray(v)).astype(np.asarray(v).dtype) for k, v in P.items()}
    v1 = ShardedAttributeView(ent, attr, lit, P, 0, 1, lr=0.01)
    v2 = ShardedAttributeView(None, None, None, P2, 0, 1, lr=0.01, opt_name="ckge_attr", tables_of=v1)
    assert v2.backend.ent is v1.backend.ent and v2.backend.cnn is not v1.backend.cnn
    for (ih, ia, iv, w) in batches:
        v1.step(ih, ia, iv, w, scale=1.0)
        v2.step(ih[::-1], ia, iv, None, scale=2.0)
    l1, l2 = v1.epoch_loss(), v2.epoch_loss()
    full, a, p1 = v1.gather()
    _, _, p2 = v2.gather()
    f64 = lambda d: {k: np.asarray(v).astype(np.float64) for k, v in d.items()}
    q1, q2 = f64(P), f64(P2)
    acc1 = {k: np.full_like(v, 0.1) for k, v in q1.items()}
    acc2 = {k: np.full_like(v, 0.1) for k, v in q2.items()}
    e64, a64, l64 = ent.astype(np.float64), attr.astype(np.float64), lit.astype(np.float64)
    ae1, aa1, ae2, aa2 = (np.full_like(x, 0.1) for x in (e64, a64, e64, a64))
    t1 = t2 = 0.0
    for (ih, ia, iv, w) in batches:
        t1 += ao.attribute_step_dense(q1, acc1, e64, a64, l64, ae1, aa1, ih, ia, iv, w.astype(np.float32).astype(np.float64), 1.0, 0.01)[0]
        t2 += ao.attribute_step_dense(q2, acc2, e64, a64, l64, ae2, aa2, ih[::-1], ia, iv, None, 2.0, 0.01)[0]
    np.testing.assert_allclose(l1, t1, rtol=1e-5)
    np.testing.assert_allclose(l2, t2, rtol=1e-5)
    np.testing.assert_allclose(full, e64, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(a, a64, rtol=2e-3, atol=2e-5)
    for k in ao.PARAM_NAMES:
        np.testing.assert_allclose(p1[k], q1[k], rtol=2e-3, atol=1e-4, err_msg=k)
        np.testing.assert_allclose(p2[k], q2[k], rtol=2e-3, atol=1e-4, err_msg=k)
