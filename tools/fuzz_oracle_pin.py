"""Pins the CPU oracle on RANDOM shapes — BUILD CONTAINER ONLY (imports the reference's code/losses.py from /root/reference under
the TF -> torch forwarder of tests/golden/make_golden.py; the op SEQUENCE is the reference's, the kernels torch's):
for random table sizes, widths, batch sizes and negative counts, in float64,
  * the reference's eight losses.py functions on gathered rows == oracle/multike_oracle.py's restatements (value to 1e-12);
  * torch autograd through the reference's relation_logistic_loss / logistic_loss_wo_negs / space_mapping_loss / alignment_loss
    == the oracle's closed-form gradients (`logistic_term_grads`, `space_mapping_grads`);
  * two optimizer steps of the relation-view graph (l2_normalize on read, torch.optim.Adagrad with acc0 0.1, eps 0)
    == `relation_view_step_dense` (dense Jacobian + TF1 ApplyAdagrad), tables to 1e-11.
python tools/fuzz_oracle_pin.py [cases] [seed]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference/code"
if not os.path.isdir(REF):
    sys.exit("reference not found (this script only runs in the build container)")
import numpy as np, torch
import make_golden as mg
sys.path.insert(0, REF)
tf = mg.install_tf_forwarder(); mg.install_empty_standins()
ref = importlib.import_module("losses")
from oracle import multike_oracle as mo

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
bad = 0
for c in range(cases):
    E, R, d = int(rng.integers(5, 300)), int(rng.integers(1, 20)), int(rng.integers(1, 130))
    P, N = int(rng.integers(1, 200)), int(rng.choice([1, 2, 5, 10, 25]))
    cs = mg.make_case(rng, E, R, d, P, N)
    msg = ""
    try:
        idx = {k: torch.tensor(cs[k].astype(np.int64)) for k in ("ph", "pr", "pt", "nh", "nr", "nt")}
        ent64, rel64 = cs["ent"].astype(np.float64), cs["rel"].astype(np.float64)
        # ---- two optimizer steps of the relation-view graph ------------------------------------------------
        ent2, rel2 = T(ent64).requires_grad_(True), T(rel64).requires_grad_(True)
        opt = torch.optim.Adagrad([ent2, rel2], lr=0.01, initial_accumulator_value=0.1, eps=0.0)
        e, r = ent64.copy(), rel64.copy(); ae, ar = np.full_like(e, 0.1), np.full_like(r, 0.1)
        for step in range(2):
            opt.zero_grad()
            En, Rn = tf.nn.l2_normalize(ent2, 1), tf.nn.l2_normalize(rel2, 1)
            L = ref.relation_logistic_loss(En[idx["ph"]], Rn[idx["pr"]], En[idx["pt"]], En[idx["nh"]], Rn[idx["nr"]], En[idx["nt"]])
            L.backward(); opt.step()
            Lo, _, _ = mo.relation_view_step_dense(e, r, ae, ar, (cs["ph"], cs["pr"], cs["pt"]), (cs["nh"], cs["nr"], cs["nt"]), 0.01)
            assert abs(L.item() - Lo) <= 1e-11 * max(abs(Lo), 1.0), f"step {step} loss {L.item()} vs {Lo}"
        assert np.allclose(ent2.detach().numpy(), e, rtol=1e-9, atol=1e-11), f"ent after 2 steps: {np.abs(ent2.detach().numpy() - e).max():.2e}"
        assert np.allclose(rel2.detach().numpy(), r, rtol=1e-9, atol=1e-11), f"rel after 2 steps: {np.abs(rel2.detach().numpy() - r).max():.2e}"
        # ---- the losses.py surface on gathered rows: values and gradients -------------------------------------
        En, Rn = mo.l2_normalize_rows(ent64), mo.l2_normalize_rows(rel64)
        rows = [En[cs["ph"]], Rn[cs["pr"]], En[cs["pt"]], En[cs["nh"]], Rn[cs["nr"]], En[cs["nt"]]]
        pw, nw = cs["pw"].astype(np.float64), cs["nw"].astype(np.float64)

        def both(name, fn_ref, fn_or, args, grads=None):
            leaves = [T(a).requires_grad_(True) if np.ndim(a) == 2 else T(a) for a in args]
            v = fn_ref(*leaves)
            vo = fn_or(*args)
            assert abs(v.item() - vo) <= 1e-12 * max(abs(vo), 1.0), f"{name}: {v.item()} vs {vo}"
            if grads is not None:
                v.backward()
                for k, g in grads.items():
                    got = leaves[k].grad.numpy()
                    assert np.allclose(got, g, rtol=1e-9, atol=1e-12), f"{name} grad {k}: {np.abs(got - g).max():.2e}"

        _, gp, _, _ = mo.logistic_term_grads(rows[0], rows[1], rows[2], +1.0)
        _, gn, _, _ = mo.logistic_term_grads(rows[3], rows[4], rows[5], -1.0)
        both("a1", ref.relation_logistic_loss, mo.relation_logistic_loss, rows, {0: gp, 1: gp, 2: -gp, 3: gn, 4: gn, 5: -gn})
        both("a2", ref.relation_logistic_loss_wo_negs, mo.relation_logistic_loss_wo_negs, rows[:3], {0: gp, 2: -gp})
        both("a2b", ref.attribute_logistic_loss_wo_negs, mo.attribute_logistic_loss_wo_negs, rows[:3])
        _, gw, _, _ = mo.logistic_term_grads(rows[0], rows[1], rows[2], +1.0, pw)
        both("a3", ref.logistic_loss_wo_negs, mo.logistic_loss_wo_negs, rows[:3] + [pw], {0: gw, 1: gw, 2: -gw})
        _, gnw, _, _ = mo.logistic_term_grads(rows[3], rows[4], rows[5], -1.0, nw)
        both("a4", ref.attribute_logistic_loss, mo.attribute_logistic_loss, rows[:3] + [pw] + rows[3:] + [nw], {0: gw, 4: gnw, 6: -gnw})
        both("a5", ref.alignment_loss, mo.alignment_loss, [rows[0], rows[2]], {0: 2 * (rows[0] - rows[2]), 1: -2 * (rows[0] - rows[2])})
        M = np.linalg.qr(rng.standard_normal((d, d)))[0] + 0.05 * rng.standard_normal((d, d))
        eye_t, eye = torch.eye(d, dtype=torch.float64), np.eye(d)
        ow = float(rng.choice([0.5, 2.0]))
        Lm, gs, gM = mo.space_mapping_grads(rows[0], rows[2], M, ow)
        both("a6", lambda v, s_, m: ref.space_mapping_loss(v, s_, m, eye_t, ow), lambda v, s_, m: mo.space_mapping_loss(v, s_, m, eye, ow),
             [rows[0], rows[2], M], {1: gs, 2: gM})
        assert abs(Lm - mo.space_mapping_loss(rows[0], rows[2], M, eye, ow)) <= 1e-12 * max(abs(Lm), 1.0)
        both("a6o", lambda m: ref.orthogonal_loss(m, eye_t), lambda m: mo.orthogonal_loss(m, eye), [M])
    except AssertionError as ex:
        msg = f"DIFFERS: {str(ex)[:300]}"
    except Exception as ex:  # noqa: BLE001
        import traceback
        msg = f"{type(ex).__name__}: {str(ex)[:200]} @ {traceback.format_exc().strip().splitlines()[-3][:200]}"
    if msg:
        bad += 1
        print(f"PIN case {c}: E={E} R={R} d={d} P={P} N={N}: {msg}", flush=True)
print(f"oracle vs the reference's losses.py (executed) + torch autograd + torch Adagrad: {cases - bad} / {cases} random cases agree")

# ---- evaluator oracle vs the reference's own greedy_alignment (code/base/alignment.py:8-79,141-163), executed -----------------------
import contextlib, io
ref_al = importlib.import_module("base.alignment")
from oracle import eval_oracle as eo
bad2 = 0
ne = max(cases // 5, 10)
for c in range(ne):
    d = int(rng.integers(2, 120)); n1 = int(rng.integers(2, 400)); n2 = n1 + int(rng.integers(0, 300))
    e2 = rng.standard_normal((n2, d)).astype(np.float32)
    e1 = (float(rng.uniform(0.0, 1.0)) * e2[:n1] + rng.standard_normal((n1, d))).astype(np.float32)
    top_k = sorted({1} | set(int(x) for x in rng.integers(1, min(n2, 60) + 1, 3)))   # the reference asserts 1 in top_k
    msg = ""
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            rest, h1, mr, mrr = ref_al.greedy_alignment(e1, e2, top_k, 1, "inner", True, 0, True)
            _, _, hits_x, _ = ref_al.calculate_rank(list(range(n1)), ref_al.sim(e1, e2, normalize=True), top_k, True, n1)
        rank, best = eo.ranks(e1, e2)
        hits, omr, omrr = eo.metrics(rank, top_k)
        assert np.array_equal(np.round(np.array(hits_x) / n1 * 100, 3), hits), f"hits {hits_x} vs {hits}"
        assert abs(mr - omr) <= 1e-9 * omr and abs(mrr - omrr) <= 1e-9, f"mr {mr} vs {omr}, mrr {mrr} vs {omrr}"
        assert sorted(rest) == sorted((i, int(b)) for i, b in enumerate(best)), "hits1_rest pairs"
    except AssertionError as ex:
        msg = f"DIFFERS: {str(ex)[:300]}"
    except Exception as ex:  # noqa: BLE001
        import traceback
        msg = f"{type(ex).__name__}: {str(ex)[:200]} @ {traceback.format_exc().strip().splitlines()[-3][:200]}"
    if msg:
        bad2 += 1
        print(f"EVAL-PIN case {c}: n1={n1} n2={n2} d={d} top_k={top_k}: {msg}", flush=True)
print(f"evaluator oracle vs the reference's greedy_alignment (executed): {ne - bad2} / {ne} random cases agree")
sys.exit(1 if bad + bad2 else 0)
