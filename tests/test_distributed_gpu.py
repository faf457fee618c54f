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
    os.environ.setdefault("MASTER_PORT", "29641")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
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
        assert float(tr.ent_grad.abs().max()) == 0.0
    finally:
        dist.destroy_process_group()
