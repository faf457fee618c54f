"""world_size-2 gloo runs of the sharded attribute view and the sharded common-space step
(multike_amd/distributed_views.py) with NumPy backends built on the oracle: ownership by head entity, the two scalar
all-reduces of the batch-wide l2_normalize (code/MultiKE_model.py:60), the all-reduce of the replicated parameters'
gradients, a rank that owns none of a step's triples — against a single-process dense oracle on the same global batches."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import attr_cnn_oracle as ao
from oracle import multike_oracle as mo

N_ENT, N_ATTR, N_LIT, DIM, B, STEPS, SEED = 61, 9, 40, 12, 50, 4, 3


def _free_port():
    """A fresh rendezvous file for init_method="file://..." (a TCP port picked by bind-and-close can be taken again before the
    workers listen on it: one EADDRINUSE in ~200 runs on the GPU boxes)."""
    import tempfile
    return tempfile.mktemp(prefix="mke_rdv_")


def _attr_data():
    rng = np.random.default_rng(SEED)
    ent = mo.xavier_truncated_normal((N_ENT, DIM), rng).astype(np.float64)
    attr = mo.xavier_truncated_normal((N_ATTR, DIM), rng).astype(np.float64)
    lit = rng.standard_normal((N_LIT, DIM))
    lit /= np.linalg.norm(lit, axis=1, keepdims=True)
    P = ao.init_params(DIM, rng)
    P["bias"] = 0.05 * rng.standard_normal(DIM)
    batches = []
    for s in range(STEPS):
        ih = rng.integers(0, N_ENT, B)
        if s == 2:
            ih = 2 * rng.integers(0, N_ENT // 2, 7)            # only even heads: rank 1 of 2 owns nothing in this step
        n = len(ih)
        batches.append((ih, rng.integers(0, N_ATTR, n), rng.integers(0, N_LIT, n), rng.uniform(0.2, 1.0, n)))
    return ent, attr, lit, P, batches


def _attr_worker(rank, world, port, ret, mode="parallel"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed_views import ShardedAttributeView
        from oracle_backend import OracleAttrBackend
        ent, attr, lit, P, batches = _attr_data()
        v = ShardedAttributeView(ent, attr, lit, P, rank, world, lr=0.05, backend_cls=OracleAttrBackend, mode=mode)
        for (ih, ia, iv, w) in batches:
            v.step(ih, ia, iv, w, scale=2.0)
        loss = v.epoch_loss()
        full, a, p = v.gather()
        if rank == 0:
            ret.put((full, a, p, loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,mode", [(2, "parallel"), (3, "parallel"), (2, "replicated"), (3, "replicated")])
def test_sharded_attribute_view_equals_single_process_oracle(world, mode):
    """mode "replicated": the batch's head rows assembled on every rank, the whole step computed by every rank, rank 0's
    replicated-state gradients broadcast, every rank updating the heads it owns (2 collectives per step instead of 4)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_attr_worker, args=(r, world, port, ret, mode)) for r in range(world)]
    for p in procs:
        p.start()
    full, a, p, loss = ret.get(timeout=240)
    for q in procs:
        q.join(60)
        assert q.exitcode == 0
    ent, attr, lit, P, batches = _attr_data()
    acc = {k: np.full_like(x, 0.1) for k, x in P.items()}
    ae, aa = np.full_like(ent, 0.1), np.full_like(attr, 0.1)
    tot = 0.0
    for (ih, ia, iv, w) in batches:
        L, _ = ao.attribute_step_dense(P, acc, ent, attr, lit, ae, aa, ih, ia, iv, w, 2.0, 0.05)
        tot += L
    np.testing.assert_allclose(loss, tot, rtol=1e-11)
    np.testing.assert_allclose(full, ent, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a, attr, rtol=1e-9, atol=1e-12)
    for k in ao.PARAM_NAMES:
        np.testing.assert_allclose(p[k], P[k], rtol=1e-8, atol=1e-12, err_msg=k)


def _cs_data():
    rng = np.random.default_rng(SEED + 1)
    mk = lambda: mo.xavier_truncated_normal((N_ENT, DIM), rng).astype(np.float64)
    ent, rv, av = mk(), mk(), mk()
    name = rng.standard_normal((N_ENT, DIM))
    name /= np.linalg.norm(name, axis=1, keepdims=True)
    name[5] = 0.0                                             # an entity without a name vector
    batches = [rng.choice(N_ENT, 23, replace=False) for _ in range(STEPS)]
    return ent, name, rv, av, batches


def _cs_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed_views import ShardedCommonSpace
        from oracle_backend import OracleCommonSpaceBackend
        ent, name, rv, av, batches = _cs_data()
        v = ShardedCommonSpace(ent, name, rv, av, rank, world, lr=0.05, cv_name_weight=0.7, cv_weight=1.3,
                               backend_cls=OracleCommonSpaceBackend)
        for ids in batches:
            v.step(ids)
        loss = v.epoch_loss()
        out = v.gather()
        if rank == 0:
            ret.put((out, loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_common_space_equals_single_process_oracle():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cs_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    out, loss = ret.get(timeout=240)
    for q in procs:
        q.join(60)
        assert q.exitcode == 0
    ent, name, rv, av, batches = _cs_data()
    accs = [np.full_like(ent, 0.1) for _ in range(3)]
    tot = sum(mo.common_space_step_dense(ent, name, rv, av, accs[0], accs[1], accs[2], ids, 0.05, 0.7, 1.3) for ids in batches)
    np.testing.assert_allclose(loss, tot, rtol=1e-12)
    for k, ref in (("ent", ent), ("rv", rv), ("av", av)):
        np.testing.assert_allclose(out[k], ref, rtol=1e-10, atol=1e-13, err_msg=k)


def _sm_data():
    rng = np.random.default_rng(SEED + 2)
    mk = lambda: mo.xavier_truncated_normal((N_ENT, DIM), rng).astype(np.float64)
    ent, views = mk(), [mk(), mk(), mk()]
    mats = [np.eye(DIM) + 0.1 * rng.standard_normal((DIM, DIM)) for _ in range(3)]
    batches = [rng.choice(N_ENT, 19, replace=False) for _ in range(STEPS)]
    batches[1] = 3 * rng.choice(N_ENT // 3, 6, replace=False)   # multiples of 3: ranks 1, 2 of 3 own nothing in this step
    return ent, views, mats, batches


def _sm_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed_views import ShardedSpaceMapping
        from oracle_backend import OracleSpaceMappingBackend
        ent, views, mats, batches = _sm_data()
        v = ShardedSpaceMapping(ent, views, mats, rank, world, lr=0.05, orthogonal_weight=2.0, backend_cls=OracleSpaceMappingBackend)
        for ids in batches:
            v.step(ids)
        loss = v.epoch_loss()
        full, M = v.gather()
        if rank == 0:
            ret.put((full, M, loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_space_mapping_equals_single_process_oracle(world):
    """SSL space-mapping step (code/losses.py:53-63): per view the two batch-wide sums of the axis-less l2_normalize and the
    matrix gradients are all-reduced; the orthogonality terms are added once, after the reduction."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sm_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    full, M, loss = ret.get(timeout=240)
    for q in procs:
        q.join(60)
        assert q.exitcode == 0
    ent, views, mats, batches = _sm_data()
    acc_e = np.full_like(ent, 0.1)
    acc_m = [np.full_like(m, 0.1) for m in mats]
    tot = sum(mo.space_mapping_step_dense(ent, acc_e, [(v, True) for v in views], mats, acc_m, ids, 0.05, 2.0) for ids in batches)
    np.testing.assert_allclose(loss, tot, rtol=1e-11)
    np.testing.assert_allclose(full, ent, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(M, np.stack(mats), rtol=1e-9, atol=1e-12)


# ----------------------------------------------------------------------------------------------------------------------
# literal auto-encoder
# ----------------------------------------------------------------------------------------------------------------------
AE_DIMS, AE_STEPS = [24, 16, 8, 5], 3


def _ae_data(active):
    from oracle import literal_oracle as lo
    rng = np.random.default_rng(17)
    p = lo.init_params(AE_DIMS, rng)
    p = {k: 0.3 * v for k, v in p.items()}
    # batches: ragged, and one smaller than the world (a rank without rows still takes part in the reductions)
    batches = [rng.standard_normal((m, AE_DIMS[0])) for m in (37, 2, 20)][:AE_STEPS]
    return p, batches


def _ae_worker(rank, world, port, active, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_views import ShardedAutoEncoder
        from oracle_backend import OracleAutoEncoderBackend
        p, batches = _ae_data(active)
        v = ShardedAutoEncoder(p, AE_DIMS, rank, world, lr=0.05, active=active, normalize=True, backend_cls=OracleAutoEncoderBackend)
        for x in batches:
            v.step(x)
        loss = v.epoch_loss()
        if rank == 0:
            ret.put((v.params(), loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,active", [(2, "thah"), (3, "tanh")])
def test_sharded_auto_encoder_equals_single_process_oracle(world, active):
    """Literal auto-encoder, data parallel over the rows of a batch (code/literal_encoder.py:63-69): the two batch-wide sums of
    the whole-matrix l2_normalize of the code (:65-66) and the packed parameter gradient are all-reduced."""
    from oracle import literal_oracle as lo
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ae_worker, args=(r, world, port, active, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got, loss = ret.get(timeout=240)
    for q in procs:
        q.join(60)
        assert q.exitcode == 0
    p, batches = _ae_data(active)
    acc = {k: np.full_like(v, 0.1) for k, v in p.items()}
    tot = 0.0
    for x in batches:
        L, g = lo.loss_and_grads(p, x, len(AE_DIMS) - 1, active, True)
        lo.adagrad_step(p, acc, g, 0.05)
        tot += L
    np.testing.assert_allclose(loss, tot, rtol=1e-11)
    for k in p:
        np.testing.assert_allclose(got[k], p[k], rtol=1e-9, atol=1e-12, err_msg=k)


# ----------------------------------------------------------------------------------------------------------------------
# the attribute group: two attribute graphs (own CNN set, own optimizer) on shared sharded tables
# ----------------------------------------------------------------------------------------------------------------------
def _attr_group_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_views import ShardedAttributeView
        from oracle_backend import OracleAttrBackend
        ent, attr, lit, P, batches = _attr_data()
        P2 = {k: 0.5 * v for k, v in P.items()}
        v1 = ShardedAttributeView(ent, attr, lit, P, rank, world, lr=0.05, backend_cls=OracleAttrBackend)
        v2 = ShardedAttributeView(None, None, None, P2, rank, world, lr=0.05, opt_name="ckge_attr", tables_of=v1)
        for (ih, ia, iv, w) in batches:
            v1.step(ih, ia, iv, w, scale=1.0)
            v2.step(ih[::-1], ia, iv, None, scale=2.0)
        l1, l2 = v1.epoch_loss(), v2.epoch_loss()
        full, a, p1 = v1.gather()
        _, _, p2 = v2.gather()
        if rank == 0:
            ret.put((full, a, p1, p2, l1, l2))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_attribute_graphs_on_shared_sharded_tables():
    """The attribute view and the cross-KG entity-inference graph of the attribute view (code/MultiKE_model.py:134-151,
    371-391): same `av_ent_embeds` / `attr_embeds`, a CNN parameter set and an optimizer each; alternating steps."""
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_attr_group_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    full, a, p1, p2, l1, l2 = ret.get(timeout=240)
    for q in procs:
        q.join(60)
        assert q.exitcode == 0
    ent, attr, lit, P, batches = _attr_data()
    P2 = {k: 0.5 * v for k, v in P.items()}
    acc1 = {k: np.full_like(x, 0.1) for k, x in P.items()}
    acc2 = {k: np.full_like(x, 0.1) for k, x in P2.items()}
    ae1, aa1 = np.full_like(ent, 0.1), np.full_like(attr, 0.1)
    ae2, aa2 = np.full_like(ent, 0.1), np.full_like(attr, 0.1)
    t1 = t2 = 0.0
    for (ih, ia, iv, w) in batches:
        t1 += ao.attribute_step_dense(P, acc1, ent, attr, lit, ae1, aa1, ih, ia, iv, w, 1.0, 0.05)[0]
        t2 += ao.attribute_step_dense(P2, acc2, ent, attr, lit, ae2, aa2, ih[::-1], ia, iv, None, 2.0, 0.05)[0]
    np.testing.assert_allclose(l1, t1, rtol=1e-11)
    np.testing.assert_allclose(l2, t2, rtol=1e-11)
    np.testing.assert_allclose(full, ent, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a, attr, rtol=1e-9, atol=1e-12)
    for k in ao.PARAM_NAMES:
        np.testing.assert_allclose(p1[k], P[k], rtol=1e-8, atol=1e-12, err_msg=k)
        np.testing.assert_allclose(p2[k], P2[k], rtol=1e-8, atol=1e-12, err_msg=k)
