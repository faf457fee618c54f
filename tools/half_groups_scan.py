"""Two groups per wavefront (mke_set_option("score_half_groups")) against one, by row width and negatives per positive:
whole-epoch us per step of the native relation-view runner.  python tools/half_groups_scan.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd import _lib
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable
from multike_amd.runner import RelationViewRunner

kgs = SyntheticKGs()
sides = []
for k in (0, 1):
    t = torch.as_tensor(kgs.triples[k], device="cuda")
    sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
print("| dim | N | one group per wavefront us/step | two groups us/step |\n|---|---|---|---|")
for d in (75, 128, 200):
    for N in (10, 25, 31, 40, 64):
        res = []
        for half in (0, 64):
            _lib.set_option("score_half_groups", half)
            E = EmbeddingTable(kgs.entities_num, d, "e", seed=1); R = EmbeddingTable(kgs.relations_num, d, "r", seed=2)
            bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, N, seed=1)
            run = RelationViewRunner(E, R, bat)
            run.run(); torch.cuda.synchronize()
            t0 = time.perf_counter(); run.run(); run.run(); torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / (2 * run.steps) * 1e6)
        print(f"| {d} | {N} | {res[0]:.1f} | {res[1]:.1f} |")
