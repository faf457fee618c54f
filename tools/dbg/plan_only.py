import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import oc_rank_compute as T
from multike_amd.distributed_oc import OwnerComputesTrainer
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import xavier_truncated_normal
G, B = 8, 5000
cfg = dict(n_ent=200_000, n_rel=550, dim=75, neg=25)
kgs = SyntheticKGs(n_ent=cfg["n_ent"], n_rel=cfg["n_rel"], seed=1234)
ent0 = np.zeros((cfg["n_ent"], cfg["dim"]), dtype=np.float32) + 0.01
rel0 = xavier_truncated_normal(cfg["n_rel"], cfg["dim"], "cpu", seed=2).numpy()
tr = OwnerComputesTrainer(kgs, ent0, rel0, B, cfg["neg"], 0, G, seed=1, comm=T.LoopbackComm(G, 0, 0.0, 0.0), prefetch=False)
b = tr.bat
torch.cuda.synchronize()
for _ in range(6):
    b.stage_next_epoch()
    tr._compute_plan((b.pos_h, b.pos_r, b.pos_t), b.rng_stream, 1)
torch.cuda.synchronize()
print('{"done": 1}')
