"""Launch duration of the fused score kernel against the number of groups (fixed cost of one launch = one wavefront's chain of
dependent round trips).  python tools/score_floor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd import _lib
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable, StepEngine

kgs = SyntheticKGs(n_ent=200_000, n_rel=550, seed=1234)
E = EmbeddingTable(kgs.entities_num, 75, "e", seed=1); R = EmbeddingTable(kgs.relations_num, 75, "r", seed=2)
sides = []
for k in (0, 1):
    t = torch.as_tensor(kgs.triples[k], device="cuda")
    sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
eng = StepEngine()
for N in (10, 25):
    for B in (64, 512, 2048, 5000, 8192):
        bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], B, N, seed=1)
        batches = [bat.batch(s) for s in range(6)]
        ev = []
        for i in range(40):
            pos, neg = batches[i % 6]
            tag, lp = eng._next()
            _lib.count_entity_refs(pos[0], pos[2], neg[0], neg[2], N, E.refcount)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(200000)
            e0.record()
            _lib.triple_score_fwd_bwd_x(E.data, True, R.data, True, 75, pos, None, neg, None, N, 1.0, E.grad, R.grad, E.touched, R.touched,
                                        tag, E.refcount, E.slot("x"), _lib.OPT_ADAGRAD, 0.001, lp)
            e1.record()
            _lib.rows_update_multi([(R.data, R.slot("x"), R.grad, R.touched, True), (E.data, E.slot("x"), E.grad, E.touched, True, E.refcount)],
                                   tag, E.stride, 75, _lib.OPT_ADAGRAD, 0.001)
            ev.append((e0, e1))
        torch.cuda.synchronize()
        ms = np.array([a.elapsed_time(b) for a, b in ev[10:]])
        print(f"N={N:2d} groups={B:5d}: score launch {np.median(ms) * 1e3:6.1f} us (min {ms.min() * 1e3:5.1f})", flush=True)
