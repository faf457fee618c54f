"""Timings of the secondary paths on MI355X: attribute-view CNN step, alignment evaluator, k-NN refresh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd.attr_cnn import AttrCNN
from multike_amd.base.alignment import alignment_ranks
from multike_amd.base.batch import neighbour_table
from multike_amd.tables import EmbeddingTable, StepEngine

d, B = 75, 5000
E = EmbeddingTable(200_000, d, "av", seed=1); A = EmbeddingTable(600, d, "attr", normalize=False, seed=2)
lit = np.random.default_rng(0).standard_normal((100_000, d)).astype(np.float32); lit /= np.linalg.norm(lit, axis=1, keepdims=True)
L = EmbeddingTable(100_000, d, "lit", normalize=False, trainable=False, values=lit)
cnn = AttrCNN(d, seed=3); eng = StepEngine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
def batch():
    return (torch.randint(0, 200_000, (B,), device="cuda", generator=g, dtype=torch.int32), torch.randint(0, 600, (B,), device="cuda", generator=g, dtype=torch.int32),
            torch.randint(0, 100_000, (B,), device="cuda", generator=g, dtype=torch.int32), torch.rand(B, device="cuda", generator=g))
bs = [batch() for _ in range(8)]
for i in range(5): cnn.step(eng, E, A, L, *bs[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): cnn.step(eng, E, A, L, *bs[i % 8])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"attribute-view CNN step (B={B}, dim={d}): {dt*1e6:.1f} us/step -> {B/dt/1e6:.1f} M triples/s")
for n in (10_000, 60_000):
    e2 = torch.randn(n, d, device="cuda"); e1 = 0.5 * e2 + torch.randn(n, d, device="cuda")
    alignment_ranks(e1[:256], e2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r, b = alignment_ranks(e1, e2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"alignment evaluator {n} x {n} x {d}: {dt*1e3:.1f} ms ({2*n*n*80/dt/1e12:.1f} TFLOP/s f32 MFMA incl. host prep), hits@1 {float((r==0).float().mean())*100:.2f}%")
emb = torch.nn.functional.normalize(torch.randn(100_000, d, device="cuda"), dim=1)
ids = list(range(100_000))
torch.cuda.synchronize(); t0 = time.perf_counter()
tbl, valid = neighbour_table(emb, ids, 2000, 200_000)
torch.cuda.synchronize(); print(f"k-NN refresh 100K x 100K, k=2000: {time.perf_counter()-t0:.2f} s")
