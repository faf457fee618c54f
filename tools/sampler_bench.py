"""k_neg_sample at the C2 epoch shape (all positives of an epoch in one launch, as the runner launches it) and at the driver-style
20-step chunk: `sampler_fast` 1 against 0 — same output required bit for bit, launch time by HIP events.
python tools/sampler_bench.py [n_ent]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd import _lib
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher, sample_negatives
from multike_amd.synthetic import SyntheticKGs

n_ent = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
kgs = SyntheticKGs(n_ent=n_ent, n_rel=550, seed=1234)
sides = []
for k in (0, 1):
    t = torch.as_tensor(kgs.triples[k], device="cuda")
    sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, 25, seed=1)
P = bat.pos_h.numel()


def run(N, n_pos, fast, iters=30):
    _lib.set_option("sampler_fast", fast)
    pos = (bat.pos_h[:n_pos], bat.pos_r[:n_pos], bat.pos_t[:n_pos])
    out = tuple(torch.empty(n_pos * N, dtype=torch.int32, device="cuda") for _ in range(3))
    fn = lambda: sample_negatives(pos, sides[0], N, seed=(7, 9), stream_id=3, pos_offset=0, out=out, side1=sides[1], pos_kg=bat.pos_kg[:n_pos])
    for _ in range(3): fn()
    ev = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3, [o.clone() for o in out]


print(f"|E| {n_ent}, {P} positives per epoch (batch 5000)")
for N in (25, 10, 1, 15, 16, 31, 32, 64):
    for n_pos, label in ((P, "epoch launch"), (100_000, "20-step chunk")):
        us0, o0 = run(N, n_pos, 0)
        us1, o1 = run(N, n_pos, 1)
        same = all(torch.equal(a, b) for a, b in zip(o0, o1))
        print(f"neg_per_pos {N:2d}, {label:13s} ({n_pos} positives): {us0:8.1f} us -> {us1:8.1f} us  ({us0 / (n_pos / 5000):.2f} -> {us1 / (n_pos / 5000):.2f} us per "
              f"step of 5000)  outputs {'identical' if same else 'DIFFER'}")
        assert same
_lib.set_option("sampler_fast", 1)
