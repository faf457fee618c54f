"""Where does the +-6 % run-to-run spread of `k_triple_score` at the c5 shape (|E| 2M, dim 256, 64 negatives) come from?
One process: the SAME batches scored on FRESH allocations of the 2 GB table (with spacer allocations of varying size in
between, so that each table lands somewhere else), each allocation timed over 60 launches, then the first allocation again.
If the per-allocation means differ more than the launches inside one allocation do, the spread is placement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd import _lib
from multike_amd.sampling import KGSide, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable, StepEngine, xavier_truncated_normal

n_ent, n_rel, d, N, P = 2_000_000, 2000, 256, 64, 5000
kgs = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, triples_per_entity=1.0, seed=5)
bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None), KGSide(kgs.entities(1), None), P, N, seed=2)
batches = [bat.batch(s) for s in range(12)]
R = EmbeddingTable(n_rel, d, "rel", values=xavier_truncated_normal(n_rel, d, "cpu", seed=6).numpy())
g = torch.Generator(device="cuda"); g.manual_seed(5)
init = torch.randn(n_ent, d, device="cuda", generator=g).clamp_(-2, 2) * float(np.sqrt(2.6 / (n_ent + d)))


def measure(E, reps=5):
    eng = StepEngine()
    ms = []
    for rep in range(reps):
        for pos, neg in batches:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tag, lp = eng._next()
            _lib.count_entity_refs(pos[0], pos[2], neg[0], neg[2], N, E.refcount)
            e0.record()
            _lib.triple_score_fwd_bwd_x(E.data, True, R.data, True, d, pos, None, neg, None, N, 1.0, E.grad, R.grad, E.touched, R.touched,
                                        tag, E.refcount, E.slot("relation"), _lib.OPT_ADAGRAD, 0.001, lp)
            e1.record()
            _lib.rows_update_multi([(R.data, R.slot("relation"), R.grad, R.touched, True),
                                    (E.data, E.slot("relation"), E.grad, E.touched, True, E.refcount)], tag, E.stride, d, _lib.OPT_ADAGRAD, 0.001)
            ms.append((e0, e1))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ms[len(batches):]]) * 1e3      # first pass = warm-up
    return t


def fresh():
    E = EmbeddingTable(n_ent, d, "ent", trainable=False)
    E.trainable = True
    E.data[:, :d] = init
    E.slot("relation")
    return E


keep, rows = [], []
for a in range(6):
    E = fresh()
    t = measure(E)
    rows.append((a, E.data.data_ptr(), t))
    print(f"allocation {a}: table at 0x{E.data.data_ptr():x}: k_triple_score {t.mean():7.1f} us (min {t.min():.1f}, max {t.max():.1f}, sd {t.std():.1f}) over {len(t)} launches", flush=True)
    if a == 0:
        first = E
    else:
        del E
    keep.append(torch.empty((a + 1) * 37_000_000 + 12345, dtype=torch.float32, device="cuda"))    # spacer: shifts the next table
    torch.cuda.empty_cache()
t = measure(first)
print(f"allocation 0 again: {t.mean():7.1f} us (min {t.min():.1f}, max {t.max():.1f}, sd {t.std():.1f})")
means = np.array([r[2].mean() for r in rows])
print(f"between allocations: mean of means {means.mean():.1f} us, spread {means.min():.1f} .. {means.max():.1f} ({100 * (means.max() - means.min()) / means.mean():.1f} %); "
      f"inside an allocation: sd {np.mean([r[2].std() for r in rows]):.1f} us")
