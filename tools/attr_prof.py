"""Attribute-view step only (B = 5000, dim 75), for rocprofv3 --kernel-trace: `python tools/attr_prof.py [steps] [B]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd.attr_cnn import AttrCNN
from multike_amd.tables import EmbeddingTable, StepEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d, B = 75, (int(sys.argv[2]) if len(sys.argv) > 2 else 5000)
E = EmbeddingTable(200_000, d, "av", seed=1); A = EmbeddingTable(600, d, "attr", normalize=False, seed=2)
lit = np.random.default_rng(0).standard_normal((100_000, d)).astype(np.float32); lit /= np.linalg.norm(lit, axis=1, keepdims=True)
L = EmbeddingTable(100_000, d, "lit", normalize=False, trainable=False, values=lit)
cnn = AttrCNN(d, seed=3); eng = StepEngine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
def batch():
    return (torch.randint(0, 200_000, (B,), device="cuda", generator=g, dtype=torch.int32), torch.randint(0, 600, (B,), device="cuda", generator=g, dtype=torch.int32),
            torch.randint(0, 100_000, (B,), device="cuda", generator=g, dtype=torch.int32), torch.rand(B, device="cuda", generator=g))
bs = [batch() for _ in range(8)]
for i in range(10): cnn.step(eng, E, A, L, *bs[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps): cnn.step(eng, E, A, L, *bs[i % 8])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"attribute-view CNN step (B={B}, dim={d}): {dt*1e6:.1f} us/step -> {B/dt/1e6:.1f} M triples/s")
