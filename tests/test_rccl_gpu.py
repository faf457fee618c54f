"""multike_amd/rccl.py (RCCL through ctypes, collectives enqueued on the compute stream) on a ONE-rank communicator — what a
single-GPU box can run: initialisation through torch.distributed's rendezvous, the three collectives against their definitions,
the self-check, and the owner-computes trainer taking the G > 1 step path over it (MKE_OC_FORCE_COLLECTIVES=1) with the same
result as the torch.distributed communicator and as the path without collectives.  Several ranks need several GPUs: the
multi-rank logic of the trainer is covered under gloo (tests/test_distributed_oc_cpu.py) and with host-staged ranks sharing
the GPU (tests/test_distributed_oc_gpu.py)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="mke_rdv_"), rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


def test_collectives_on_the_compute_stream(one_rank_group):
    from multike_amd.rccl import Communicator
    c = Communicator()
    assert c.world == 1 and c.rank == 0
    c.self_check()
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    out = torch.zeros(1000, device="cuda")
    c.all_gather(out, x)
    red = torch.zeros(1000, device="cuda")
    c.reduce_scatter(red, x * 2)
    y = x.clone()
    c.all_reduce(y)
    codes = torch.arange(77, dtype=torch.int32, device="cuda")
    cout = torch.zeros(77, dtype=torch.int32, device="cuda")
    c.all_gather(cout, codes)
    side = torch.cuda.Stream()
    z = torch.zeros(1000, device="cuda")
    side.wait_stream(torch.cuda.current_stream())
    c.all_gather(z, x, stream=side)              # an explicit stream
    torch.cuda.synchronize()
    assert torch.equal(out, x) and torch.equal(red, x * 2) and torch.equal(y, x) and torch.equal(cout, codes) and torch.equal(z, x)
    with pytest.raises(Exception):
        c.all_gather(torch.zeros(10, device="cuda"), x)      # sizes are checked before the call
    c.destroy()


@pytest.mark.parametrize("chunks", [1, 2])
def test_trainer_over_rccl_equals_the_other_paths(one_rank_group, chunks, monkeypatch):
    from multike_amd.distributed_oc import OcComm, OcRcclComm, OwnerComputesTrainer
    from multike_amd.synthetic import SyntheticKGs
    from oracle import multike_oracle as mo
    kgs = SyntheticKGs(n_ent=6000, n_rel=20, seed=4)
    rng = np.random.default_rng(4)
    ent0 = mo.xavier_truncated_normal((6000, 75), rng)
    rel0 = mo.xavier_truncated_normal((20, 75), rng)

    def run(force, comm):
        monkeypatch.setenv("MKE_OC_FORCE_COLLECTIVES", "1" if force else "0")
        tr = OwnerComputesTrainer(kgs, ent0, rel0, 800, 10, 0, 1, seed=3, lr=0.01, chunks=chunks, comm=comm)
        assert tr.force_collectives == force
        for i in range(tr.steps + 3):           # crosses an epoch boundary (the prefetched plan's code exchange)
            tr.step(i)
        torch.cuda.synchronize()
        return tr.ent[:, :75].cpu().numpy(), tr.rel[:, :75].cpu().numpy(), tr
    e0, r0, _ = run(False, None)                 # no collectives (every row local)
    e1, r1, t1 = run(True, None)                 # the G > 1 path over RCCL on the compute stream (the default communicator)
    assert isinstance(t1.comm, OcRcclComm)
    e2, r2, _ = run(True, OcComm())              # the same over torch.distributed
    for e, r in ((e1, r1), (e2, r2)):
        np.testing.assert_allclose(e, e0, rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(r, r0, rtol=2e-4, atol=2e-6)


@pytest.mark.timeout(400)
def test_unique_id_reaches_a_second_rank():
    """tools/rccl_two_ranks_one_gpu.py: two processes on this GPU hand the ncclUniqueId through torch.distributed as the
    communicator's constructor does.  RCCL refuses the second rank of a device — AFTER its bootstrap, which needs the id's
    socket address intact on rank 1: both ranks must get that refusal (or a communicator) promptly.  With the id cut at its
    first NUL byte (rounds 5-6 until this test) both ranks sat in ncclCommInitRank for 63 s and got a network error."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_two_ranks_one_gpu.py")], capture_output=True, text=True,
                         timeout=380, cwd=root)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    import json
    ranks = [json.loads(l) for l in out.stdout.splitlines() if l.startswith('{"rank"') and "outcome" in l]
    assert sorted(r["rank"] for r in ranks) == [0, 1]
    for r in ranks:
        assert r["outcome"] == "communicator" or "invalid usage" in r["error"], r
        assert r["seconds"] < 45, r          # the bootstrap's own connect time-out is ~60 s
