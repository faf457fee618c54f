import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import numpy as np, torch
import oc_rank_compute as T
from multike_amd.distributed_oc import OwnerComputesTrainer
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import xavier_truncated_normal
G, B = 8, 5000
cfg = dict(n_ent=200_000, n_rel=550, dim=75, neg=25)
kgs = SyntheticKGs(n_ent=cfg["n_ent"], n_rel=cfg["n_rel"], seed=1234)
g = torch.Generator(device="cpu"); g.manual_seed(1)
ent0 = (torch.randn(cfg["n_ent"], cfg["dim"], generator=g) * 0.01).numpy()
rel0 = xavier_truncated_normal(cfg["n_rel"], cfg["dim"], "cpu", seed=2).numpy()
comm = T.LoopbackComm(G, 0, 0.0, 0.0)
tr = OwnerComputesTrainer(kgs, ent0, rel0, B, cfg["neg"], 0, G, seed=1, comm=comm, prefetch=True)
acc = {}
def wrap(name):
    f = getattr(tr, name)
    def g_(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(tr, name, g_)
for n in ("_advance_epoch", "_prefetch_next_epoch", "_plan_midpoint", "_finish_plan", "_plan_sample", "_plan_rest", "_plan_gather", "_compute_em_plan"):
    wrap(n)
be = tr.backend
for n in ("prepare_epoch", "run_steps", "em_plan", "plan"):
    f = getattr(be, n)
    def mk(f, n):
        def g_(*a, **k):
            t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault("be." + n, []).append(time.perf_counter() - t0); return r
        return g_
    setattr(be, n, mk(f, n))
tr.run(0, 3 * tr.steps)
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
tr.run(3 * tr.steps, 10 * tr.steps)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("epochs 10 steps", 10 * tr.steps, "host us/step", host / (10 * tr.steps) * 1e6, "wall us/step", wall / (10 * tr.steps) * 1e6)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:28s} calls {len(v):4d} total ms {sum(v)*1e3:8.2f} per epoch us {sum(v)/10*1e6:8.1f}")
