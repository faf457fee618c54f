"""Follow-up of tools/c5_variance.py: the c5 table, its gradient scratch and its Adagrad slot are 2 GiB each (2M rows of 1 KB);
row i of the three arrays is touched together (in-place update: w row + acc row; scatter: grad row).  Does the kernel time
depend on how the three bases are offset against each other?  One big arena, the three arrays carved out of it at chosen row
skews; same batches every time."""
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
E = EmbeddingTable(n_ent, d, "ent", trainable=False)
E.trainable = True
E.slot("relation")
SLACK = 1 << 16                                       # rows of slack per array
arena = torch.zeros(3 * (n_ent + SLACK) * d, dtype=torch.float32, device="cuda")


def carve(k, skew_rows):
    lo = (k * (n_ent + SLACK) + skew_rows) * d
    return arena[lo:lo + n_ent * d].view(n_ent, d)


def measure(reps=4):
    eng = StepEngine()
    ms = []
    for rep in range(reps):
        for pos, neg in batches:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            tag, lp = eng._next()
            _lib.count_entity_refs(pos[0], pos[2], neg[0], neg[2], N, E.refcount)
            e0.record()
            _lib.triple_score_fwd_bwd_x(E.data, True, R.data, True, d, pos, None, neg, None, N, 1.0, E.grad, R.grad, E.touched, R.touched,
                                        tag, E.refcount, E.slot("relation"), _lib.OPT_ADAGRAD, 0.001, lp)
            e1.record()
            _lib.rows_update_multi([(R.data, R.slot("relation"), R.grad, R.touched, True),
                                    (E.data, E.slot("relation"), E.grad, E.touched, True, E.refcount)], tag, E.stride, d, _lib.OPT_ADAGRAD, 0.001)
            e2.record()
            ms.append((e0, e1, e2))
    torch.cuda.synchronize()
    t = np.array([[a.elapsed_time(b), b.elapsed_time(c)] for a, b, c in ms[len(batches):]]) * 1e3
    return t.mean(0), t.std(0)


print(f"arena at 0x{arena.data_ptr():x}; a row is {d * 4} bytes")
for skews in ((0, 0, 0), (0, 1, 2), (0, 4, 8), (0, 5, 11), (0, 16, 32), (0, 21, 43), (0, 64, 128), (0, 69, 139), (0, 256, 512), (0, 1024, 2048),
              (0, 1029, 2059), (0, 4096, 8192), (0, 4101, 8203), (0, 16384, 32768), (0, 16389, 32779), (0, 0, 0)):
    arena.zero_()
    E.data = carve(0, skews[0]); E._grad = carve(1, skews[1]); E.slots["relation"] = carve(2, skews[2])
    E.data.copy_(init); E.slots["relation"].fill_(0.1)
    E.touched.zero_(); E.refcount.zero_()
    m, s = measure()
    rel = [(x.data_ptr() - arena.data_ptr()) % (1 << 31) for x in (E.data, E.grad, E.slots["relation"])]
    print(f"row skews {str(skews):22s} offsets mod 2 GiB {str([hex(r) for r in rel]):40s} k_triple_score {m[0]:6.1f} us (sd {s[0]:.1f})   k_rows_update_multi {m[1]:5.1f} us", flush=True)
