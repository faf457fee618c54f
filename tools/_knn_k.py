import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from multike_amd import _lib
n, d = 100_000, 75
g = torch.Generator(device="cuda"); g.manual_seed(0)
e = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
ep = torch.zeros(n, 80, device="cuda"); ep[:, :d] = e
tau = torch.full((n,), 0.22, device="cuda")   # ~2.8 % of N(0, 1/75) above 0.22
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cand, cnt = _lib.sim_select(ep, 80, 0, n, tau, 8, 512)
    torch.cuda.synchronize()
    print(f"dbg={os.environ.get('MKE_KNN_DBG')} sim_select: {(time.perf_counter() - t0) * 1e3:.2f} ms, mean hits {float(cnt.sum(1).float().mean()):.0f}")
