"""ShardedITC at G = 1 (no collectives) at C2-synth scale: what the sharded epoch driver costs on one GPU, next to the
single-GPU driver's epoch (tools/epoch_bench.py; same default: 10 negatives per positive).  python tools/sharded_epoch_bench.py [n_ent] [neg]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd.distributed_model import ShardedITC
from multike_amd.synthetic import SyntheticKGs
from oracle import attr_cnn_oracle as ao

n_ent = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
neg = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d, n_rel, n_attr, n_lit = 75, 550, 600, 100_000
kgs = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, seed=5)
rng = np.random.default_rng(5)
t = lambda n: (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
tables = {"rv_ent": t(n_ent), "av_ent": t(n_ent), "ent": t(n_ent), "name": t(n_ent), "rel": t(n_rel), "attr": t(n_attr), "lit": t(n_lit)}
cnn = [ao.init_params(d, rng) for _ in range(3)]
ri = lambda hi, n: rng.integers(0, hi, n)
n_at, n_ck = 600_000, int(0.3 * n_ent)
lists = {"attr": list(zip(ri(n_ent, n_at).tolist(), ri(n_attr, n_at).tolist(), ri(n_lit, n_at).tolist(), rng.uniform(0.3, 1, n_at).tolist())),
         "ckge_rel": list(zip(ri(n_ent, n_ck).tolist(), ri(n_rel, n_ck).tolist(), ri(n_ent, n_ck).tolist())),
         "ckgp_rel": [], "ckge_attr": list(zip(ri(n_ent, n_ck).tolist(), ri(n_attr, n_ck).tolist(), ri(n_lit, n_ck).tolist())),
         "ckga_attr": [], "entities": list(range(n_ent))}
m = ShardedITC(kgs, tables, cnn, lists, 0, 1, batch_size=5000, attribute_batch_size=5000, entity_batch_size=5000, neg_triple_num=neg, seed=5)
for i in range(1, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.epoch(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"sharded epoch {i} at G = 1: {dt * 1e3:.1f} ms  (relation steps {m.relation.steps}, attribute steps {-(-n_at // 5000)}, "
          f"ckge-rel {m.ckge_rel.steps}, ckge-attr {-(-n_ck // 5000)}, common {-(-n_ent // 5000)})")

# per phase (synchronised around each)
import time as _t
def timed(label, fn):
    torch.cuda.synchronize(); t0 = _t.perf_counter(); fn(); torch.cuda.synchronize()
    return f"{label} {(_t.perf_counter() - t0) * 1e3:.1f}"
i = 4
parts = [timed("relation", lambda: m._oc_epoch(m.relation)), timed("ckge-rel", lambda: m._oc_epoch(m.ckge_rel)),
         timed("attribute", lambda: m._attr_epoch(m.attr_views[0], "attr", i, 0, 1.0, sampled=False)),
         timed("ckge-attr", lambda: m._attr_epoch(m.attr_views[1], "ckge_attr", i, 1, 2.0, sampled=True)),
         timed("common", lambda: m._common_epoch(i, 3))]
print("phases (ms): " + " | ".join(parts))
