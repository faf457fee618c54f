"""GPU diagnostic: host enqueue cost vs wall time per step of the sharded trainer on one rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from multike_amd.distributed import ShardedRelationTrainer
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import xavier_truncated_normal
kgs = SyntheticKGs()
ent0 = xavier_truncated_normal(kgs.entities_num, 75, "cpu", seed=1).numpy()
rel0 = xavier_truncated_normal(kgs.relations_num, 75, "cpu", seed=2).numpy()
tr = ShardedRelationTrainer(kgs, ent0, rel0, 5000, 25, 0, 1, seed=3)
print("capacity", tr.C)
for i in range(10): tr.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10, 110): tr.step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"100 steps: host enqueue {1e4*(t1-t0):.1f} us/step, wall {1e4*(t2-t0):.1f} us/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(110, 160): tr.step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
