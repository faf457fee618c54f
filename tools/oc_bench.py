"""Owner-computes sharded step on ONE GPU (G = 1: no collectives): per-step wall time, the per-epoch plan cost, and how the
step splits between host and device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd.distributed_oc import OwnerComputesTrainer
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import xavier_truncated_normal
cfg = dict(n_ent=200_000, n_rel=550, dim=75, neg=25) if os.environ.get("OC_CFG", "c2") == "c2" else dict(n_ent=2_000_000, n_rel=2000, dim=256, neg=64)
kgs = SyntheticKGs(n_ent=cfg["n_ent"], n_rel=cfg["n_rel"], seed=1234)
ent0 = xavier_truncated_normal(cfg["n_ent"], cfg["dim"], "cpu", seed=1).numpy()
rel0 = xavier_truncated_normal(cfg["n_rel"], cfg["dim"], "cpu", seed=2).numpy()
tr = OwnerComputesTrainer(kgs, ent0, rel0, 5000, cfg["neg"], 0, 1, seed=1, chunks=int(os.environ.get("OC_CHUNKS", "1")))
for i in range(10):
    tr.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); tr._plan_epoch(); torch.cuda.synchronize(); print(f"plan epoch: {(time.perf_counter() - t0) * 1e3:.2f} ms")
n = min(150, tr.steps - 12)
t0 = time.perf_counter()
for i in range(10, 10 + n):
    tr.step(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{n} steps: host enqueue {t_host / n * 1e6:.1f} us/step, wall {t_all / n * 1e6:.1f} us/step -> {5000 * (1 + cfg['neg']) / (t_all / n) / 1e9:.3f} G triples/s")
