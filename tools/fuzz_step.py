"""Randomised sweep of the relation step and the attribute step against the float64 oracles: shapes the parametrised tests do
not list (every supported row width, 0..64 negatives, tiny and ragged batches, heavy duplicates, weights, SGD / Adagrad,
un-normalised tables).  python tools/fuzz_step.py [cases] [seed]
MKE_FUZZ_ONLY=<case>: draw every case (same random sequence) but run only that relation case, verbosely, in the atomic AND the
deterministic mode (fixed-order double accumulation): a difference that the deterministic mode removes is summation noise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gpu_util import dev_i32, dev_f32, grouped_batch, make_tables
from multike_amd import _lib
from multike_amd.attr_cnn import AttrCNN
from multike_amd.tables import EmbeddingTable, StepEngine
from oracle import attr_cnn_oracle as ao
from oracle import multike_oracle as mo

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DIMS = [d for f in _lib._SUPPORTED_FPL for d in (16 * f, 16 * f - int(rng.integers(1, 15)))]
bad = 0
ONLY = os.environ.get("MKE_FUZZ_ONLY")
ONLY = None if ONLY is None else int(ONLY)
for c in range(cases):
    d = int(rng.choice(DIMS))
    n_ent, n_rel = int(rng.integers(8, 3000)), int(rng.integers(1, 40))
    P, N = int(rng.integers(1, 900)), int(rng.choice([0, 1, 2, 3, 7, 10, 12, 13, 25, 31, 32, 33, 64]))
    ent = mo.xavier_truncated_normal((n_ent, d), rng); rel = mo.xavier_truncated_normal((n_rel, d), rng)
    if rng.random() < 0.2:
        ent[int(rng.integers(n_ent))] = 0.0                      # a zero row: the eps branch of l2_normalize
    pos, neg = grouped_batch(rng, n_ent, n_rel, P, max(N, 1), irregular=bool(rng.random() < 0.5))
    if N == 0:
        neg = None
    ent_norm, rel_norm = bool(rng.random() < 0.85), bool(rng.random() < 0.85)
    opt = "Adagrad" if rng.random() < 0.8 else "SGD"
    pw = rng.uniform(0.2, 1.0, P).astype(np.float32) if rng.random() < 0.3 else None
    scale = float(rng.choice([1.0, 2.0]))
    excl = bool(rng.random() < 0.7)
    if ONLY is not None:
        if c != ONLY:
            continue
        for det in (0, 1):
            _lib.set_option("deterministic", det)
            E, R = make_tables(ent, rel, ent_norm, rel_norm)
            eng = StepEngine()
            e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
            a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
            for step in range(2):
                eng.relation_step(E, R, "relation", tuple(dev_i32(a) for a in pos), None if neg is None else tuple(dev_i32(a) for a in neg),
                                  neg_per_pos=N, lr=0.01, scale=scale, optimizer=opt, exclusive_rows=excl, pos_w=None if pw is None else dev_f32(pw))
                if opt == "Adagrad":
                    mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01, pos_w=None if pw is None else pw.astype(np.float64),
                                                scale=scale, ent_norm=ent_norm, rel_norm=rel_norm)
                else:
                    _, ge, gr = mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01, pos_w=None if pw is None else pw.astype(np.float64),
                                                            scale=scale, ent_norm=ent_norm, rel_norm=rel_norm, update=False)
                    if step == 0:
                        terms = P * (1 + N)
                        print(f"case {c}: d={d} n_ent={n_ent} n_rel={n_rel} P={P} N={N} opt={opt} norm=({ent_norm},{rel_norm}): relation row 0 sums "
                              f"{terms} terms; |gradient row| max {np.abs(gr[0]).max():.3e}, sum of |terms| bound ~ {terms} x O(1)")
                    mo.rows_update_sparse(e64, None, ge, 0.01, normalize=ent_norm, optimizer="SGD")
                    mo.rows_update_sparse(r64, None, gr, 0.01, normalize=rel_norm, optimizer="SGD")
            dr = np.abs(R.raw().cpu().numpy() - r64)
            de = np.abs(E.raw().cpu().numpy() - e64)
            print(f"  deterministic={det}: rel max diff {dr.max():.3e} (|rel| max {np.abs(r64).max():.3e}, ||rel row|| {np.linalg.norm(r64[0]):.3e}); "
                  f"ent max diff {de.max():.3e}")
        _lib.set_option("deterministic", 0)
        sys.exit(0)
    E, R = make_tables(ent, rel, ent_norm, rel_norm)
    eng = StepEngine()
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    ok = True
    msg = ""
    try:
        for step in range(2):
            lp = eng.relation_step(E, R, "relation", tuple(dev_i32(a) for a in pos), None if neg is None else tuple(dev_i32(a) for a in neg),
                                   neg_per_pos=N, lr=0.01, scale=scale, optimizer=opt, exclusive_rows=excl,
                                   pos_w=None if pw is None else dev_f32(pw))
            if opt == "Adagrad":
                L, _, _ = mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01, pos_w=None if pw is None else pw.astype(np.float64),
                                                      scale=scale, ent_norm=ent_norm, rel_norm=rel_norm)
            else:
                L, ge, gr = mo.relation_view_step_dense(e64, r64, a64, b64, pos, neg, 0.01, pos_w=None if pw is None else pw.astype(np.float64),
                                                        scale=scale, ent_norm=ent_norm, rel_norm=rel_norm, update=False)
                mo.rows_update_sparse(e64, None, ge, 0.01, normalize=ent_norm, optimizer="SGD")
                mo.rows_update_sparse(r64, None, gr, 0.01, normalize=rel_norm, optimizer="SGD")
            if abs(float(lp.sum()) - L) > 5e-6 * max(abs(L), 1e-3):
                ok, msg = False, f"loss {float(lp.sum())} vs {L}"
        got = E.raw().cpu().numpy()
        nz = np.linalg.norm(ent, axis=1) > 0
        if not np.allclose(got[nz], e64[nz], rtol=5e-4, atol=5e-6 + 3e-5 * np.abs(e64[nz]).max()):   # fp32 noise scales with the largest entry
            ok, msg = False, msg + f" ent max diff {np.abs(got[nz] - e64[nz]).max():.2e}"
        # a hub row sums tens of thousands of fp32 terms in atomic order: with T = P (1 + N) / n_rel terms per row the rounding of
        # the running sum grows like sqrt(T) * 2^-24 of the sum of |terms|, and SGD / the Jacobian on a normalised row divide by
        # ||w||.  (Round 3's final-tree record run had one such miss, 5.7e-4 on a single-relation table with 26K terms in its
        # row; it belongs to another draw than this script's seed-0 sequence — `MKE_FUZZ_ONLY=17` on this tree: 1.1e-7.)
        hub = P * (1 + N) / n_rel
        if not np.allclose(R.raw().cpu().numpy(), r64, rtol=5e-4, atol=2e-5 + (1e-4 + 2e-6 * np.sqrt(hub)) * np.abs(r64).max()):
            ok, msg = False, msg + f" rel max diff {np.abs(R.raw().cpu().numpy() - r64).max():.2e}"
        if float(E.grad.abs().max()) != 0.0 or float(R.grad.abs().max()) != 0.0 or (E._refcount is not None and int(E.refcount.abs().sum()) != 0):
            ok, msg = False, msg + " scratch not consumed"
    except Exception as ex:  # noqa: BLE001
        ok, msg = False, f"{type(ex).__name__}: {str(ex)[:200]}"
    if not ok:
        bad += 1
        print(f"REL case {c}: d={d} n_ent={n_ent} n_rel={n_rel} P={P} N={N} norm=({ent_norm},{rel_norm}) opt={opt} pw={pw is not None} scale={scale} excl={excl}: {msg}", flush=True)
print(f"relation step: {cases - bad} / {cases} cases agree with the float64 oracle")

bad2 = 0
for c in range(cases // 4):
    d = int(rng.integers(4, 321))
    B = int(rng.integers(1, 1200))
    n_ent, n_attr, n_lit = int(rng.integers(4, 2000)), int(rng.integers(1, 40)), int(rng.integers(2, 500))
    ent = mo.xavier_truncated_normal((n_ent, d), rng); attr = mo.xavier_truncated_normal((n_attr, d), rng)
    lit = rng.standard_normal((n_lit, d)).astype(np.float32); lit /= np.linalg.norm(lit, axis=1, keepdims=True)
    P_ = ao.init_params(d, rng); P_["bias"] = 0.05 * rng.standard_normal(d)
    ih, ia, iv = rng.integers(0, n_ent, B), rng.integers(0, n_attr, B), rng.integers(0, n_lit, B)
    w = rng.uniform(0.2, 1.0, B).astype(np.float32) if rng.random() < 0.6 else None
    scale = float(rng.choice([1.0, 2.0]))
    E = EmbeddingTable(n_ent, d, "av", values=ent); A = EmbeddingTable(n_attr, d, "attr", normalize=False, values=attr)
    L_ = EmbeddingTable(n_lit, d, "lit", normalize=False, trainable=False, values=lit)
    cnn = AttrCNN(d, params=P_); eng = StepEngine()
    p64 = {k: np.asarray(v, dtype=np.float64) for k, v in P_.items()}; acc = {k: np.full_like(v, 0.1) for k, v in p64.items()}
    e64, a64, l64 = ent.astype(np.float64), attr.astype(np.float64), lit.astype(np.float64)
    ae, aa = np.full_like(e64, 0.1), np.full_like(a64, 0.1)
    ok, msg = True, ""
    try:
        for step in range(2):
            lp = cnn.step(eng, E, A, L_, dev_i32(ih), dev_i32(ia), dev_i32(iv), None if w is None else dev_f32(w), scale=scale, lr=0.01)
            Lo, _ = ao.attribute_step_dense(p64, acc, e64, a64, l64, ae, aa, ih, ia, iv, None if w is None else w.astype(np.float64), scale, 0.01)
            if abs(float(lp.sum()) - Lo) > 1e-5 * abs(Lo):
                ok, msg = False, f"loss {float(lp.sum())} vs {Lo}"
        if not np.allclose(E.raw().cpu().numpy(), e64, rtol=1e-3, atol=1e-5):
            ok, msg = False, msg + f" ent max diff {np.abs(E.raw().cpu().numpy() - e64).max():.2e}"
        if not np.allclose(A.raw().cpu().numpy(), a64, rtol=5e-3, atol=5e-5):
            ok, msg = False, msg + f" attr max diff {np.abs(A.raw().cpu().numpy() - a64).max():.2e}"
        for k, v in cnn.numpy_params().items():
            if not np.allclose(v, p64[k], rtol=5e-3, atol=2e-4):
                ok, msg = False, msg + f" {k} max diff {np.abs(v - p64[k]).max():.2e}"
    except Exception as ex:  # noqa: BLE001
        ok, msg = False, f"{type(ex).__name__}: {str(ex)[:200]}"
    if not ok:
        bad2 += 1
        print(f"ATTR case {c}: d={d} B={B} n_ent={n_ent} n_attr={n_attr} n_lit={n_lit} w={w is not None} scale={scale}: {msg}", flush=True)
print(f"attribute step: {cases // 4 - bad2} / {cases // 4} cases agree with the float64 oracle")
sys.exit(1 if bad + bad2 else 0)
