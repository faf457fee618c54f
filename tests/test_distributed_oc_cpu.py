"""world_size 2 / 3 gloo runs of the owner-computes sharded trainer (multike_amd/distributed_oc.py) with the oracle as the
compute backend: slots, per-epoch code exchange, ownership filtering, the all-gather / reduce-scatter / all-reduce
sequence, split-batch chunks and the epoch boundary.  The sharded result must equal a single-process dense float64
oracle run on the same global batches (every row updated once per step from the sum of all its contributions)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle as co
from oracle import multike_oracle as mo


def _free_port():
    """A fresh rendezvous file for init_method="file://..." (a TCP port picked by bind-and-close can be taken again before the
    workers listen on it: one EADDRINUSE in ~200 runs on the GPU boxes)."""
    import tempfile
    return tempfile.mktemp(prefix="mke_rdv_")


N_REL, DIM, B, NEG, SEED = 12, 20, 64, 5, 7


def _worker(rank, world, port, ret, n_ent, steps, chunks, excl):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed_oc import OwnerComputesTrainer
        from multike_amd.synthetic import SyntheticKGs
        from oracle_backend import OcOracleBackend
        if n_ent < 100:
            OwnerComputesTrainer.SAMPLE_RUN = 23 * NEG        # the rank's share of an epoch sampled in several runs (bounded scratch)
        kgs = SyntheticKGs(n_ent=n_ent, n_rel=N_REL, seed=SEED)
        rng = np.random.default_rng(SEED)
        ent0 = mo.xavier_truncated_normal((n_ent, DIM), rng).astype(np.float64)
        rel0 = mo.xavier_truncated_normal((N_REL, DIM), rng).astype(np.float64)
        be = OcOracleBackend()
        drawn, orig_sample_at = [], be.sample_at
        be.sample_at = lambda pos, *a, **k: (drawn.append(int(pos[0].numel())), orig_sample_at(pos, *a, **k))[1]
        tr = OwnerComputesTrainer(kgs, ent0, rel0, B, NEG, rank, world, seed=SEED, lr=0.05, backend=be,
                                  device="cpu", dtype=torch.float64, chunks=chunks, exclusive_rows=excl)
        for i in range(steps):
            tr.step(i)
        # no rank draws more than its 1 / world share of an epoch's negatives (one sampler call per planned epoch)
        assert drawn and max(drawn) <= -(-tr._n_all // world), (drawn, tr._n_all)
        full = tr.gather_entity_table().numpy()
        assert float(tr.ent_grad.abs().max()) == 0.0 and float(tr.rel_grad.abs().max()) == 0.0   # scratch consumed
        assert tr.ref_count is None or int(tr.ref_count.abs().sum()) == 0
        loss = tr.epoch_loss()
        if rank == 0:
            ret.put((full, tr.rel[:, :DIM].numpy().copy(), loss, tr.steps, tr.check()))
    finally:
        dist.destroy_process_group()


def _reference(world, n_ent, steps):
    """Single process, dense float64 oracle, same global batches (world * B positives per step), crossing epochs."""
    from multike_amd.sampling import KGSide, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=N_REL, seed=SEED)
    rng = np.random.default_rng(SEED)
    e = mo.xavier_truncated_normal((n_ent, DIM), rng).astype(np.float64)
    r = mo.xavier_truncated_normal((N_REL, DIM), rng).astype(np.float64)
    ae, ar = np.full_like(e, 0.1), np.full_like(r, 0.1)
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None, device="cpu"),
                          KGSide(kgs.entities(1), None, device="cpu"), B * world, NEG, device="cpu", seed=SEED)
    sets = [co.TripleSet(t[:, 0], t[:, 1], t[:, 2]) for t in kgs.triples]
    losses = []
    for i in range(steps):
        s = i % bat.steps
        if s == 0 and i > 0:
            bat.shuffle()
        ph, pr, pt = (x.numpy() for x in (bat.pos_h, bat.pos_r, bat.pos_t))
        lo, hi = int(bat.off[s]), int(bat.off[s + 1])
        mid = lo + int(bat.cnt1[s])
        parts = []
        for k, (a, b) in enumerate(((lo, mid), (mid, hi))):
            elo, ehi = kgs.ent_range[k]
            parts.append(co.neg_sample(ph[a:b], pr[a:b], pt[a:b], NEG, ehi - elo, ent_lo=elo, known=sets[k], seed=bat.rng_seed,
                                       stream_id=bat.rng_stream + k, pos_offset=a))
        neg = [np.concatenate([parts[0][j], parts[1][j]]) for j in range(3)]
        L, _, _ = mo.relation_view_step_dense(e, r, ae, ar, (ph[lo:hi], pr[lo:hi], pt[lo:hi]), neg, 0.05)
        losses.append(L)
    return e, r, losses, bat.steps


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n_ent,chunks,excl", [(2, 600, 1, True), (3, 602, 1, True), (2, 600, 3, True), (2, 600, 2, False),
                                                     (3, 80, 2, True)])   # 80 entities: a sixth of the positives need BOTH vectors
def test_owner_computes_equals_single_process_oracle(world, n_ent, chunks, excl):
    """world 3 with 602 entities: shards of unequal size, KG id ranges that do not fall on shard boundaries, ragged slices;
    chunks 2 / 3: split-batch parts; steps run past the epoch boundary (shuffle + re-plan)."""
    ref_e, ref_r, ref_losses, steps_per_epoch = _reference(world, n_ent, 1)   # steps per epoch only
    steps = steps_per_epoch + 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret, n_ent, steps, chunks, excl)) for r in range(world)]
    for p in procs:
        p.start()
    full, rel, loss, tsteps, info = ret.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    e, r, losses, _ = _reference(world, n_ent, steps)
    assert tsteps == steps_per_epoch
    np.testing.assert_allclose(full, e, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rel, r, rtol=1e-9, atol=1e-12)
    # the loss ring holds one slot per step of an epoch: steps 0 and 1 were overwritten by the second epoch's first two
    np.testing.assert_allclose(loss, sum(losses[2:]), rtol=1e-11)
    assert info["chunks"] == chunks and info["capacity_vectors_per_owner"] >= B * world // world // 4
    # one coin per round (code/base/batch.py:97-105): a positive's negatives corrupt one side unless a re-draw round fell on the
    # other, so ~1 vector per positive travels — and never fewer than one (the positive's own term needs it)
    assert 1.0 <= info["vectors_per_positive"] < 1.25, info
    assert n_ent > 100 or info["vectors_per_positive"] > 1.05


def test_parts_and_slices_partition_every_global_step():
    """Parts are disjoint, ordered and cover each global step; rank slices partition each part — ragged last steps too."""
    from multike_amd.distributed_oc import OwnerComputesTrainer

    class Fake:
        pass
    for world in (2, 3, 8):
        for chunks in (1, 2, 3):
            t = Fake()
            t.bat = Fake()
            t.bat.off = np.array([0, 40, 80, 97, 97])
            t.world, t.chunks = world, chunks
            for s in range(4):
                parts = OwnerComputesTrainer.parts_of_step(t, s)
                cur = t.bat.off[s]
                for lo, hi in parts:
                    assert lo == cur and hi > lo
                    c2 = lo
                    for rank in range(world):
                        t.rank = rank
                        per, a, e = OwnerComputesTrainer.my_slice(t, lo, hi)
                        assert a == min(c2, hi) and a <= e <= hi and e - a <= per
                        c2 = e
                    assert c2 == hi
                    cur = hi
                assert cur == t.bat.off[s + 1]


# ----------------------------------------------------------------------------------------------------------------------
# cross-KG inference loops: positives only, `random.sample` batches of a triple list, loss x 2
# ----------------------------------------------------------------------------------------------------------------------
CK_N_ENT, CK_TRIPLES, CK_B = 400, 333, 100      # 4 steps per epoch (the last draws like the others: B of 333)


def _ck_setup():
    rng = np.random.default_rng(SEED + 1)
    triples = np.stack([rng.integers(0, CK_N_ENT, CK_TRIPLES), rng.integers(0, N_REL, CK_TRIPLES),
                        rng.integers(0, CK_N_ENT, CK_TRIPLES)], 1).astype(np.int32)
    ent0 = mo.xavier_truncated_normal((CK_N_ENT, DIM), rng).astype(np.float64)
    rel0 = mo.xavier_truncated_normal((N_REL, DIM), rng).astype(np.float64)
    return triples, ent0, rel0


def _ck_list(triples, weighted):
    """The model's list form: (h, r, t) or weighted 4-tuples (h, r, t, w) (code/MultiKE_model.py:393-414)."""
    if not weighted:
        return triples
    w = np.random.default_rng(SEED + 2).uniform(0.2, 1.0, len(triples))
    return [(int(h), int(r), int(t), float(x)) for (h, r, t), x in zip(triples, w)]


def _ck_worker(rank, world, port, ret, steps, chunks, weighted=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_oc import OwnerComputesTrainer, TripleListBatcher
        from oracle_backend import OcOracleBackend
        triples, ent0, rel0 = _ck_setup()
        bat = TripleListBatcher(_ck_list(triples, weighted), CK_B, device="cpu", seed=SEED)
        tr = OwnerComputesTrainer(None, ent0, rel0, CK_B, 0, rank, world, seed=SEED, lr=0.05, backend=OcOracleBackend(), device="cpu",
                                  dtype=torch.float64, chunks=chunks, batcher=bat, scale=2.0)
        for i in range(steps):
            tr.step(i)
        full = tr.gather_entity_table().numpy()
        assert float(tr.ent_grad.abs().max()) == 0.0 and float(tr.rel_grad.abs().max()) == 0.0
        loss = tr.epoch_loss()
        if rank == 0:
            ret.put((full, tr.rel[:, :DIM].numpy().copy(), loss, tr.steps))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,chunks,weighted", [(2, 1, False), (3, 2, False), (2, 1, True)])
def test_cross_kg_positive_steps_equal_single_process_oracle(world, chunks, weighted):
    """The owner-computes exchange on the cross-KG entity-inference loop (code/MultiKE_model.py:349-369): positives only,
    every step a `random.sample` of the triple list (drawn identically on every rank), 2 x the loss; past the epoch boundary."""
    from multike_amd.distributed_oc import TripleListBatcher
    steps = 6
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ck_worker, args=(r, world, port, ret, steps, chunks, weighted)) for r in range(world)]
    for p in procs:
        p.start()
    full, rel, loss, spe = ret.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    triples, e, r = _ck_setup()
    ae, ar = np.full_like(e, 0.1), np.full_like(r, 0.1)
    bat = TripleListBatcher(_ck_list(triples, weighted), CK_B, device="cpu", seed=SEED)
    assert spe == bat.steps == 4
    losses = []
    for i in range(steps):
        s = i % bat.steps
        if s == 0 and i > 0:
            bat.shuffle()
        lo, hi = int(bat.off[s]), int(bat.off[s + 1])
        pos = tuple(x.numpy()[lo:hi] for x in (bat.pos_h, bat.pos_r, bat.pos_t))
        assert len(set(map(tuple, np.stack(pos, 1)))) <= hi - lo
        pw = bat.pos_w.numpy()[lo:hi].astype(np.float64) if weighted else None
        L, _, _ = mo.relation_view_step_dense(e, r, ae, ar, pos, None, 0.05, pos_w=pw, scale=2.0)
        losses.append(L)
    np.testing.assert_allclose(full, e, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rel, r, rtol=1e-9, atol=1e-12)
    # the loss ring holds one slot per step of an epoch: steps 0 and 1 were overwritten by the second epoch's first two
    np.testing.assert_allclose(loss, sum(losses[2:]), rtol=1e-11)


# ----------------------------------------------------------------------------------------------------------------------
# the relation group of an epoch on shared sharded tables: relation view, then a cross-KG loop, one optimizer each
# ----------------------------------------------------------------------------------------------------------------------
def _group_setup():
    from multike_amd.synthetic import SyntheticKGs
    kgs = SyntheticKGs(n_ent=CK_N_ENT, n_rel=N_REL, seed=SEED)
    rng = np.random.default_rng(SEED + 5)
    triples, _, _ = _ck_setup()
    ent0 = mo.xavier_truncated_normal((CK_N_ENT, DIM), rng).astype(np.float64)
    rel0 = mo.xavier_truncated_normal((N_REL, DIM), rng).astype(np.float64)
    return kgs, _ck_list(triples, True), ent0, rel0


def _group_worker(rank, world, port, ret, epochs):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_oc import OwnerComputesTrainer, TripleListBatcher
        from oracle_backend import OcOracleBackend
        kgs, lst, ent0, rel0 = _group_setup()
        a = OwnerComputesTrainer(kgs, ent0, rel0, B, NEG, rank, world, seed=SEED, lr=0.05, backend=OcOracleBackend(), device="cpu",
                                 dtype=torch.float64)
        b = OwnerComputesTrainer(None, None, None, CK_B, 0, rank, world, seed=SEED, lr=0.05, backend=OcOracleBackend(), device="cpu",
                                 dtype=torch.float64, batcher=TripleListBatcher(lst, CK_B, device="cpu", seed=SEED), scale=2.0,
                                 tables_of=a)
        assert b.ent is a.ent and b.rel is a.rel and b.ent_acc is not a.ent_acc
        losses = []
        for ep in range(epochs):
            for s in range(a.steps):
                a.step(ep * a.steps + s)
            la = a.epoch_loss()
            for s in range(b.steps):
                b.step(ep * b.steps + s)
            losses.append((la, b.epoch_loss()))
        full = a.gather_entity_table().numpy()
        if rank == 0:
            ret.put((full, a.rel[:, :DIM].numpy().copy(), losses, a.steps, b.steps))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_relation_group_on_shared_sharded_tables_equals_single_process_oracle():
    """Two trainers on ONE pair of sharded tables — the relation view (negatives, Adagrad slot 1) and the weighted cross-KG
    relation-inference loop (positives only, x 2, Adagrad slot 2), alternating for two epochs: what the relation group of an
    ITC epoch does (code/MultiKE_model.py:291-317, 393-414), against the dense oracle with one accumulator pair per optimizer."""
    from multike_amd.distributed_oc import TripleListBatcher
    from multike_amd.sampling import KGSide, RelationBatcher
    world, epochs = 2, 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, ret, epochs)) for r in range(world)]
    for p in procs:
        p.start()
    full, rel, losses, sa, sb = ret.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    kgs, lst, e, r = _group_setup()
    acc_a = (np.full_like(e, 0.1), np.full_like(r, 0.1))
    acc_b = (np.full_like(e, 0.1), np.full_like(r, 0.1))
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None, device="cpu"),
                          KGSide(kgs.entities(1), None, device="cpu"), B * world, NEG, device="cpu", seed=SEED)
    sets = [co.TripleSet(t[:, 0], t[:, 1], t[:, 2]) for t in kgs.triples]
    lb = TripleListBatcher(lst, CK_B, device="cpu", seed=SEED)
    assert (sa, sb) == (bat.steps, lb.steps)
    for ep in range(epochs):
        if ep > 0:
            bat.shuffle()
            lb.shuffle()
        la = 0.0
        ph, pr, pt = (x.numpy() for x in (bat.pos_h, bat.pos_r, bat.pos_t))
        for s in range(bat.steps):
            lo, hi = int(bat.off[s]), int(bat.off[s + 1])
            mid = lo + int(bat.cnt1[s])
            parts = []
            for k, (a, c) in enumerate(((lo, mid), (mid, hi))):
                elo, ehi = kgs.ent_range[k]
                parts.append(co.neg_sample(ph[a:c], pr[a:c], pt[a:c], NEG, ehi - elo, ent_lo=elo, known=sets[k], seed=bat.rng_seed,
                                           stream_id=bat.rng_stream + k, pos_offset=a))
            neg = [np.concatenate([parts[0][j], parts[1][j]]) for j in range(3)]
            L, _, _ = mo.relation_view_step_dense(e, r, acc_a[0], acc_a[1], (ph[lo:hi], pr[lo:hi], pt[lo:hi]), neg, 0.05)
            la += L
        lbs = 0.0
        for s in range(lb.steps):
            lo, hi = int(lb.off[s]), int(lb.off[s + 1])
            pos = tuple(x.numpy()[lo:hi] for x in (lb.pos_h, lb.pos_r, lb.pos_t))
            L, _, _ = mo.relation_view_step_dense(e, r, acc_b[0], acc_b[1], pos, None, 0.05, pos_w=lb.pos_w.numpy()[lo:hi].astype(np.float64),
                                                  scale=2.0)
            lbs += L
        np.testing.assert_allclose(losses[ep][0], la, rtol=1e-11)
        np.testing.assert_allclose(losses[ep][1], lbs, rtol=1e-11)
    np.testing.assert_allclose(full, e, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rel, r, rtol=1e-9, atol=1e-12)
