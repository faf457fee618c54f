"""Does sampling the NEXT epoch's negatives on a second stream pay when the training stream has the higher priority?
python tools/prefetch_bench.py   (C2 shape; per-step wall time over whole epochs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd.runner import RelationViewRunner
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable

kgs = SyntheticKGs(n_ent=200_000, n_rel=550, seed=1234)
sides = []
for k in (0, 1):
    t = torch.as_tensor(kgs.triples[k], device="cuda")
    sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))


def run(prefetch, high_priority, n_epochs=6):
    E = EmbeddingTable(kgs.entities_num, 75, "e", seed=1); R = EmbeddingTable(kgs.relations_num, 75, "r", seed=2)
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, 25, seed=1)
    r = RelationViewRunner(E, R, bat, "relation", lr=0.001)
    st = torch.cuda.Stream(priority=-1) if high_priority else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        r.run_epochs(1, prefetch=prefetch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run_epochs(n_epochs, prefetch=prefetch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / (n_epochs * r.steps) * 1e6


for pf, hp in ((False, False), (True, False), (True, True), (False, True), (True, True), (False, False)):
    print(f"prefetch={pf!s:5} training stream high priority={hp!s:5}: {run(pf, hp):6.2f} us/step", flush=True)
