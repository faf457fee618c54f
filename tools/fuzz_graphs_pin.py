"""Pins the whole-model oracle to the reference's EXECUTED graphs on RANDOM shapes — BUILD CONTAINER ONLY (imports
/root/reference/code under the eager TensorFlow forwarder of tests/golden/make_golden.py: `MultiKE._define_variables` and the nine
`_define_*_graph` methods, `conv`, `xavier_init` run unmodified; leaf ops of tf.layers and ApplyAdagrad restated).  Per case: random
table sizes, width, batch sizes, negative count, learning rates and weights; every graph's printed loss (1e-10) and every variable
after one optimizer step (1e-8), all other variables unmoved — the comparisons of tests/test_graphs_golden.py on fresh draws.
python tools/fuzz_graphs_pin.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
REF = "/root/reference/code"
if not os.path.isdir(REF):
    sys.exit("reference not found (this script only runs in the build container)")
import numpy as np
import make_golden as mg
sys.path.insert(0, REF)
mg.install_tf_forwarder(); mg.install_empty_standins()
import test_graphs_golden as tg


class Fixture(dict):
    files = property(lambda self: list(self))


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
KEYS = ["relation", "attribute", "ckge_rel", "ckge_attr", "ckga_attr", "ckgp_rel", "common", "mapping"]
bad = 0
for c in range(cases):
    d = int(rng.choice([4, 7, 8, 12, 20, 33, 64, 75, 80, 100]))
    n_ent = int(rng.integers(6, 200))
    sizes = (d, n_ent, int(rng.integers(1, 12)), int(rng.integers(1, 15)), int(rng.integers(2, 60)), int(rng.integers(1, 80)),
             int(rng.integers(1, n_ent)))
    rates = (float(rng.choice([0.001, 0.01, 0.05])), float(rng.choice([0.004, 0.03])), float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.2, 3.0)),
             float(rng.choice([0.0, 0.5, 2.0, 10.0])))
    neg = int(rng.choice([1, 2, 5, 10]))
    g = Fixture()
    mg.graphs_fixture(g, seed=int(rng.integers(1 << 30)), sizes=sizes, neg=neg, rates=rates)
    msg = ""
    for key in KEYS:
        try:
            o = tg._oracle(g)
            loss = tg._run(o, g, key)
            np.testing.assert_allclose(loss, float(g[f"{key}_loss"]), rtol=1e-10)
            lr = float(g[f"{key}_lr"])
            allowed = set(np.asarray(g[f"{key}_var_list"]).tolist()) if int(g[f"{key}_has_var_list"]) else None
            init, now = tg._initial(g), tg._variables(o)
            for name, w0 in init.items():
                gk = f"{key}_g_{name}"
                if gk in g and (allowed is None or name in allowed):
                    grad = g[gk]
                    np.testing.assert_allclose(now[name], w0 - lr * grad / np.sqrt(0.1 + grad * grad), rtol=1e-8, atol=1e-12, err_msg=f"{key}: {name}")
                else:
                    assert np.array_equal(now[name], w0), f"{key}: {name} moved"
        except AssertionError as e:
            msg += f" [{key}: {str(e).strip().splitlines()[-1][:160]}]"
    if msg:
        bad += 1
    if msg or c % 10 == 0 or c == cases - 1:
        print(f"GRAPHS case {c}: sizes={sizes} neg={neg} rates={tuple(round(r, 3) for r in rates)}: {'ok' if not msg else 'MISMATCH' + msg}", flush=True)
print(f"oracle vs executed reference graphs: {cases - bad} / {cases} random configurations agree on all {len(KEYS)} graphs")
sys.exit(1 if bad else 0)
