"""Kernel micro-bench on the C2 shape: isolates the fused triple step (forward only / with scatter, by split
factor) and the row update, timing each launch with HIP events on the launch stream."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd import _lib
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable, StepEngine

ap = argparse.ArgumentParser()
ap.add_argument("--n-ent", type=int, default=200_000); ap.add_argument("--n-rel", type=int, default=550)
ap.add_argument("--dim", type=int, default=75); ap.add_argument("--neg", type=int, default=25)
ap.add_argument("--batch", type=int, default=5000); ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--splits", type=str, default="0,1,2,3,4,5"); ap.add_argument("--copies", type=int, default=1); ap.add_argument("--zipf", type=float, default=0.0)
ap.add_argument("--update-chunk", type=int, default=0)
a = ap.parse_args()
_lib.set_option("update_chunk", a.update_chunk)
kgs = SyntheticKGs(n_ent=a.n_ent, n_rel=a.n_rel, zipf=a.zipf)
d, N = a.dim, a.neg
E = EmbeddingTable(kgs.entities_num, d, "e", seed=1); R = EmbeddingTable(kgs.relations_num, d, "r", seed=2, grad_copies=a.copies)
sides = []
for k in (0, 1):
    t = torch.as_tensor(kgs.triples[k], device="cuda")
    sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], a.batch, N, seed=1)
eng = StepEngine()
batches = [bat.batch(s) for s in range(8)]
T = batches[0][0][0].numel() * (1 + N)
balg = 12 + 24 * d

def timeit(fn, iters=a.iters):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    ev = []
    for i in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(i); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = np.array([x.elapsed_time(y) for x, y in ev])
    return float(np.median(ms)) * 1e3, float(ms.min()) * 1e3

def score(i, bwd=True):
    pos, neg = batches[i % 8]
    tag, lp = eng._next()
    _lib.triple_score_fwd_bwd(E.data, True, R.data, True, d, pos, None, neg, None, N, 1.0, E.grad if bwd else None,
                              R.grad if bwd else None, E.touched, R.touched, tag, lp)
    return tag

def step(i):
    tag = score(i)
    _lib.rows_update_multi([(R.data, R.slot("x"), R.grad, R.touched, True), (E.data, E.slot("x"), E.grad, E.touched, True)],
                           tag, E.stride, d, 0, 0.001)

print(f"shape |E|={a.n_ent} d={d} N={N} P={a.batch}: T={T} triples/launch, B_alg={balg} B/triple")
for s in [int(x) for x in a.splits.split(",")]:
    _lib.set_option("score_splits", s)
    med, mn = timeit(lambda i: score(i, False))
    print(f"splits={s}: score fwd-only   median {med:7.1f} us  min {mn:7.1f}  -> {T*balg/med/1e3:7.1f} GB/s alg")
    # with scatter: the grad buffer fills up (never consumed) — same memory traffic pattern as a real step
    E.grad.zero_(); R.grad.zero_()
    med, mn = timeit(lambda i: score(i, True))
    print(f"splits={s}: score fwd+scatter median {med:7.1f} us  min {mn:7.1f}  -> {T*balg/med/1e3:7.1f} GB/s alg")
_lib.set_option("score_splits", 0)
E.grad.zero_(); R.grad.zero_()
med, mn = timeit(step)
print(f"full step (score + update): median {med:7.1f} us min {mn:7.1f}")
# update alone: touch flags from a scatter, then time only the update
def upd(i):
    pass
ev = []
for i in range(a.iters):
    tag = score(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.rows_update_multi([(R.data, R.slot("x"), R.grad, R.touched, True), (E.data, E.slot("x"), E.grad, E.touched, True)],
                           tag, E.stride, d, 0, 0.001)
    e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
ms = np.array([x.elapsed_time(y) for x, y in ev]) * 1e3
ntouch = int((E.touched == tag).sum())
print(f"update alone: median {np.median(ms):7.1f} us; touched entity rows {ntouch} -> {ntouch*6*E.stride*4/np.median(ms)/1e3:7.1f} GB/s (6 row streams)")

# split: relation table and entity table updated by separate launches (where does the update time go?)
for name, tbl in (("rel", R), ("ent", E)):
    ev = []
    for i in range(a.iters):
        tag = score(i)
        other = E if tbl is R else R
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.rows_update(tbl.data, tbl.slot("x"), tbl.grad, tbl.touched, tag, d, True, 0, 0.001)
        e1.record(); ev.append((e0, e1))
        _lib.rows_update(other.data, other.slot("x"), other.grad, other.touched, tag, d, True, 0, 0.001)
    torch.cuda.synchronize()
    ms = np.array([x.elapsed_time(y) for x, y in ev]) * 1e3
    print(f"update {name} table alone: median {np.median(ms):7.1f} us ({int((tbl.touched == tag).sum())} rows)")
