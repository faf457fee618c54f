#!/usr/bin/env python3
"""The epoch plan of rank 0 of G alone (sampler share + code packing + mke_oc_plan + the entity-major reference lists), six times in
line, for a rocprofv3 kernel table:   tools/prof.sh r06_plan_c2 30 tools/plan_prof.py [--config c2|c5] [--world 8]"""
import argparse
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
import torch

import oc_rank_compute as T
from multike_amd.distributed_oc import OwnerComputesTrainer
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import xavier_truncated_normal

ap = argparse.ArgumentParser()
ap.add_argument("--config", choices=["c2", "c5"], default="c2")
ap.add_argument("--world", type=int, default=8)
a = ap.parse_args()
cfg = dict(n_ent=200_000, n_rel=550, dim=75, neg=25) if a.config == "c2" else dict(n_ent=2_000_000, n_rel=2000, dim=256, neg=64)
G, B = a.world, 5000
kgs = SyntheticKGs(n_ent=cfg["n_ent"], n_rel=cfg["n_rel"], seed=1234)
ent0 = np.full((cfg["n_ent"], cfg["dim"]), 0.01, dtype=np.float32)
rel0 = xavier_truncated_normal(cfg["n_rel"], cfg["dim"], "cpu", seed=2).numpy()
tr = OwnerComputesTrainer(kgs, ent0, rel0, B, cfg["neg"], 0, G, seed=1, comm=T.LoopbackComm(G, 0, 0.0, 0.0), prefetch=False)
b = tr.bat
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(6):
    tr._compute_plan((b.pos_h, b.pos_r, b.pos_t), b.rng_stream, 1)
e1.record()
torch.cuda.synchronize()
print('{"tool": "plan_prof", "config": "%s", "world": %d, "plan_ms": %.3f, "steps_per_epoch": %d, "refs": %d, "capacity": %d}'
      % (a.config, G, e0.elapsed_time(e1) / 6, tr.steps, tr._em["n_refs_host"], tr._em["capacity"]))
