"""GPU diagnostic: where does the per-step wall time go?  (enqueue cost vs GPU time, null vs side stream)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd import _lib
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable
from multike_amd.runner import RelationViewRunner

kgs = SyntheticKGs()
E = EmbeddingTable(kgs.entities_num, 75, "e", seed=1)
R = EmbeddingTable(kgs.relations_num, 75, "r", seed=2)
sides = []
for k in (0, 1):
    t = torch.as_tensor(kgs.triples[k], device="cuda")
    sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, 25, seed=1)
run = RelationViewRunner(E, R, bat)

def epoch(label):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run.run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{label}: enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms for {run.steps} steps -> {1e6*(t2-t0)/run.steps:.1f} us/step")

for _ in range(2): epoch("null stream")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): epoch("side stream")
# tiny launches: host cost per launch
tiny = EmbeddingTable(64, 75, "t", seed=3)
acc = tiny.slot("x")
for label, ctx in (("null", torch.cuda.stream(torch.cuda.default_stream())), ("side", torch.cuda.stream(s))):
    with ctx:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2000):
            _lib.rows_update(tiny.data, acc, tiny.grad, tiny.touched, 5, 75, True, 0, 0.1)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"tiny launches ({label} stream): enqueue {1e6*(t1-t0)/2000:.2f} us each, total {1e6*(t2-t0)/2000:.2f} us each")
