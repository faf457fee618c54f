"""The sharded trainer with the HIP backend on one GPU (1-rank RCCL group): same global steps as the
single-table StepEngine path => same losses and tables.  (The N>1 exchange logic is covered under gloo in
tests/test_distributed_cpu.py; 8-GPU runs are the driver's.)"""
import os

import numpy as np
import pytest
import torch

from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lookahead", [0, 2])
def test_one_rank_sharded_equals_single_table_path(lookahead):
    import torch.distributed as dist
    from multike_amd.distributed import ShardedRelationTrainer
    from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import EmbeddingTable, StepEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import tempfile
    dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="mke_rdv_"), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n_ent, n_rel, d, B, N = 6000, 40, 75, 700, 10
        kgs = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, seed=2)
        rng = np.random.default_rng(2)
        ent0 = mo.xavier_truncated_normal((n_ent, d), rng)
        rel0 = mo.xavier_truncated_normal((n_rel, d), rng)
        tr = ShardedRelationTrainer(kgs, ent0, rel0, B, N, 0, 1, seed=9, lr=0.01, lookahead=lookahead)
        E = EmbeddingTable(n_ent, d, "e", values=ent0)
        R = EmbeddingTable(n_rel, d, "r", values=rel0)
        sides = []
        for k in (0, 1):
            t = torch.as_tensor(kgs.triples[k], device="cuda")
            sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
        bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], B, N, seed=9)
        eng = StepEngine()
        tot = 0.0
        nsteps = bat.steps + 3          # crosses an epoch boundary (shuffle) on both paths
        for s in range(nsteps):
            tr.step(s)
            if s > 0 and s % bat.steps == 0:
                bat.shuffle()
            pos, neg = bat.batch(s % bat.steps)
            l = float(eng.relation_step(E, R, "relation", pos, neg, neg_per_pos=N, lr=0.01).sum())
            if s >= 3:                  # the trainer's loss ring holds the last `steps` global steps
                tot += l
        np.testing.assert_allclose(tr.epoch_loss(), tot, rtol=2e-6)
        np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), E.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(tr.rel[:, :d].cpu().numpy(), R.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)
        assert tr.pending_slots() == 0
    finally:
        dist.destroy_process_group()


# ---- two ranks sharing the one GPU: world_size-2 run of the DEVICE kernels (HipBackend) ------------------------------
N_ENT2, N_REL2, DIM2, B2, NEG2, STEPS2, SEED2 = 3000, 20, 75, 300, 8, 7, 11


def _two_rank_worker(rank, world, port, ret, lookahead=0):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed import HostStagedComm, ShardedRelationTrainer
        from multike_amd.synthetic import SyntheticKGs
        torch.cuda.set_device(0)
        kgs = SyntheticKGs(n_ent=N_ENT2, n_rel=N_REL2, seed=SEED2)
        rng = np.random.default_rng(SEED2)
        ent0 = mo.xavier_truncated_normal((N_ENT2, DIM2), rng)
        rel0 = mo.xavier_truncated_normal((N_REL2, DIM2), rng)
        tr = ShardedRelationTrainer(kgs, ent0, rel0, B2, NEG2, rank, world, seed=SEED2, lr=0.02, comm=HostStagedComm(), lookahead=lookahead)
        tr.keep_stats = True
        stats = []
        for i in range(STEPS2):
            tr.step(i)
            stats.append(tr.stats())
        full = tr.gather_entity_table().cpu().numpy()
        loss = tr.epoch_loss()
        gmax = float(tr.pending_slots())
        if rank == 0:
            ret.put((full, tr.rel[:, :DIM2].cpu().numpy().copy(), loss, stats, gmax))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("lookahead", [0, 2])
def test_two_ranks_on_one_gpu_equal_single_process_oracle(lookahead):
    """world_size 2 with the HIP backend: both ranks run their kernels on cuda:0, the collectives are staged through gloo
    (`HostStagedComm`) because RCCL refuses two ranks on one device.  Owner = id % 2, remote rows, gradient rows coming
    home from another rank, per-epoch `mke_neg_sample_at`, relation all-reduce -- against the float64 dense oracle on the
    same global batches.  lookahead = 2: the plan half of each step (row set, id exchange, remap) runs two steps ahead on
    its own stream and process group."""
    import socket
    import torch.multiprocessing as mp
    from oracle import c_oracle as co
    from multike_amd.sampling import KGSide, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    import tempfile
    port = tempfile.mktemp(prefix="mke_rdv_")   # rendezvous file (init_method="file://..."): no TCP port to collide on
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, ret, lookahead)) for r in range(world)]
    for p in procs:
        p.start()
    full, rel, loss, stats, gmax = ret.get(timeout=480)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    kgs = SyntheticKGs(n_ent=N_ENT2, n_rel=N_REL2, seed=SEED2)
    rng = np.random.default_rng(SEED2)
    e = mo.xavier_truncated_normal((N_ENT2, DIM2), rng).astype(np.float32).astype(np.float64)
    r = mo.xavier_truncated_normal((N_REL2, DIM2), rng).astype(np.float32).astype(np.float64)
    ae, ar = np.full_like(e, 0.1), np.full_like(r, 0.1)
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None, device="cpu"),
                          KGSide(kgs.entities(1), None, device="cpu"), B2 * world, NEG2, device="cpu", seed=SEED2)
    sets = [co.TripleSet(t[:, 0], t[:, 1], t[:, 2]) for t in kgs.triples]
    ph, pr, pt = (x.numpy() for x in (bat.pos_h, bat.pos_r, bat.pos_t))
    tot = 0.0
    assert STEPS2 <= bat.steps
    for st in range(STEPS2):
        lo, hi = int(bat.off[st]), int(bat.off[st + 1])
        mid = lo + int(bat.cnt1[st])
        parts = []
        for k, (a, b) in enumerate(((lo, mid), (mid, hi))):
            elo, ehi = kgs.ent_range[k]
            parts.append(co.neg_sample(ph[a:b], pr[a:b], pt[a:b], NEG2, ehi - elo, ent_lo=elo, known=sets[k], seed=(SEED2, 0),
                                       stream_id=k, pos_offset=a))
        neg = [np.concatenate([parts[0][i], parts[1][i]]) for i in range(3)]
        L, _, _ = mo.relation_view_step_dense(e, r, ae, ar, (ph[lo:hi], pr[lo:hi], pt[lo:hi]), neg, 0.02)
        tot += L
    np.testing.assert_allclose(loss, tot, rtol=2e-6)
    np.testing.assert_allclose(full, e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(rel, r, rtol=2e-4, atol=2e-6)
    assert gmax == 0.0
    assert all(x["remote_rows"] > 0 and x["overflow"] == 0 for x in stats)
