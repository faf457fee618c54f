import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from multike_amd.base.batch import neighbour_table
from multike_amd.tables import EmbeddingTable
n, d = 210_000, 75
E = EmbeddingTable(n, d, "rv", seed=1)
useful = list(range(0, 105_000))
useful2 = list(range(105_000, 210_000))
def refresh():
    out = []
    for u in (useful, useful2):
        ids = torch.as_tensor(np.asarray(u, dtype=np.int32), device="cuda")
        emb = E.lookup(ids)
        out.append(neighbour_table(emb, u, 2100, n))
    return out
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if it == 2:
        pr = cProfile.Profile(); pr.enable()
    r = refresh()
    torch.cuda.synchronize()
    if it == 2:
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    print(f"refresh {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    del r
