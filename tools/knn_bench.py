"""k-NN refresh of the truncated sampler at DBP-WD scale: python tools/knn_bench.py [n] [dim] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd.base.batch import neighbour_table

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 75
k = int(sys.argv[3]) if len(sys.argv) > 3 else int(0.02 * n)
g = torch.Generator(device="cuda"); g.manual_seed(0)
centers = torch.randn(200, d, device="cuda", generator=g)
e = centers[torch.randint(0, 200, (n,), device="cuda", generator=g)] + 0.8 * torch.randn(n, d, device="cuda", generator=g)
e = torch.nn.functional.normalize(e, dim=1)
ids = list(range(n))
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    table, valid = neighbour_table(e, ids, k, n)
    torch.cuda.synchronize()
    print(f"run {it}: neighbour_table n={n} d={d} k={k}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
