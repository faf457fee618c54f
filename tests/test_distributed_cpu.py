"""world_size-2 gloo test of the entity-row sharded trainer's exchange logic (ids out / rows back / gradients
home / owner-side single update / relation all-reduce) with the oracle as the compute backend.  The sharded
result must equal a single-process dense oracle run on the same global batches."""
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


N_REL, DIM, B, NEG, STEPS, SEED = 12, 20, 64, 5, 4, 7


def _worker(rank, world, port, ret, N_ENT):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed import ShardedRelationTrainer
        from multike_amd.synthetic import SyntheticKGs
        from oracle_backend import OracleBackend
        kgs = SyntheticKGs(n_ent=N_ENT, n_rel=N_REL, seed=SEED)
        rng = np.random.default_rng(SEED)
        ent0 = mo.xavier_truncated_normal((N_ENT, DIM), rng).astype(np.float64)
        rel0 = mo.xavier_truncated_normal((N_REL, DIM), rng).astype(np.float64)
        tr = ShardedRelationTrainer(kgs, ent0, rel0, B, NEG, rank, world, seed=SEED, lr=0.05, backend=OracleBackend(),
                                    device="cpu", dtype=torch.float64)
        stats = []
        tr.keep_stats = True
        for i in range(STEPS):
            tr.step(i)
            stats.append(tr.stats())
        full = tr.gather_entity_table().numpy()
        loss = tr.epoch_loss()
        if rank == 0:
            ret.put((full, tr.rel[:, :DIM].numpy().copy(), loss, stats, float(tr.pending_slots())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,N_ENT", [(2, 600), (3, 602)])
def test_sharded_equals_single_process_oracle(world, N_ENT):
    """world 3 with 602 entities: shards of unequal size (201 / 201 / 200 rows), KG id ranges that do not fall on shard
    boundaries."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret, N_ENT)) for r in range(world)]
    for p in procs:
        p.start()
    full, rel, loss, stats, gmax = ret.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: same global batches (world*B positives per step), dense float64 oracle
    from multike_amd.sampling import KGSide, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    kgs = SyntheticKGs(n_ent=N_ENT, n_rel=N_REL, seed=SEED)
    rng = np.random.default_rng(SEED)
    e = mo.xavier_truncated_normal((N_ENT, DIM), rng).astype(np.float64)
    r = mo.xavier_truncated_normal((N_REL, DIM), rng).astype(np.float64)
    ae, ar = np.full_like(e, 0.1), np.full_like(r, 0.1)
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None, device="cpu"),
                          KGSide(kgs.entities(1), None, device="cpu"), B * world, NEG, device="cpu", seed=SEED)
    sets = [co.TripleSet(t[:, 0], t[:, 1], t[:, 2]) for t in kgs.triples]
    ph, pr, pt = (x.numpy() for x in (bat.pos_h, bat.pos_r, bat.pos_t))
    tot = 0.0
    for s in range(STEPS):
        lo, hi = int(bat.off[s]), int(bat.off[s + 1])
        mid = lo + int(bat.cnt1[s])
        parts = []
        for k, (a, b) in enumerate(((lo, mid), (mid, hi))):
            elo, ehi = kgs.ent_range[k]
            parts.append(co.neg_sample(ph[a:b], pr[a:b], pt[a:b], NEG, ehi - elo, ent_lo=elo, known=sets[k], seed=(SEED, 0),
                                       stream_id=k, pos_offset=a))
        neg = [np.concatenate([parts[0][i], parts[1][i]]) for i in range(3)]
        L, _, _ = mo.relation_view_step_dense(e, r, ae, ar, (ph[lo:hi], pr[lo:hi], pt[lo:hi]), neg, 0.05)
        tot += L
    np.testing.assert_allclose(full, e, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(rel, r, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(loss, tot, rtol=1e-12)
    assert gmax == 0.0                                   # owner consumed every request slot
    assert all(st["remote_rows"] > 0 and st["overflow"] == 0 and st["unique_rows"] <= world * st["capacity"] for st in stats)


def test_slices_partition_every_global_step():
    """Rank slices are disjoint, ordered and cover each global step — for ragged last steps too."""
    from multike_amd.distributed import ShardedRelationTrainer

    class Fake:
        pass
    for world in (2, 3, 8):
        off = np.array([0, 40, 80, 97, 97])
        got = []
        for rank in range(world):
            t = Fake()
            t.bat = Fake()
            t.bat.off = off
            t.world, t.rank = world, rank
            got.append([ShardedRelationTrainer.my_slice(t, s) for s in range(4)])
        for s in range(4):
            cur = off[s]
            for rank in range(world):
                a, b = got[rank][s]
                assert a == min(cur, off[s + 1]) or a == cur
                assert a <= b <= off[s + 1]
                cur = b
            assert cur == off[s + 1]
