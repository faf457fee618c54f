"""Alignment evaluator at test-set scale: python tools/eval_bench.py [n] [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd.base.alignment import alignment_ranks

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 75
g = torch.Generator(device="cuda"); g.manual_seed(0)
e1 = torch.randn(n, d, device="cuda", generator=g)
e2 = e1 + 0.5 * torch.randn(n, d, device="cuda", generator=g)
if os.environ.get("MKE_EVAL_BENCH_ZERO") == "1":        # tools/sweep_clock.sh: zero operands draw less power, the part clocks higher
    e1, e2 = torch.zeros_like(e1), torch.zeros_like(e2)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rank, best = alignment_ranks(e1, e2)
    torch.cuda.synchronize()
    print(f"run {it}: alignment_ranks n={n} d={d}: {(time.perf_counter() - t0) * 1e3:.2f} ms, hits@1 {float((rank == 0).float().mean()):.3f}")
