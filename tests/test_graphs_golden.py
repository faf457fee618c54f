"""The whole-model oracle (oracle/model_oracle.py: one Adagrad step of every graph with dense-table semantics) against the
reference's own graph definitions EXECUTED — `MultiKE._define_variables` and the nine `_define_*_graph` methods of
/root/reference/code/MultiKE_model.py run unmodified over eagerly forwarded TensorFlow calls (tests/golden/make_golden.py
`graphs_fixture` -> tests/golden/graphs_golden.npz): for every graph the loss the loop prints, the loss the optimizer minimises, its
learning rate, the variables it may move, and the float64 autograd gradient w.r.t. every raw variable.  Pins the COMPOSITION the
oracle restates — which tables a graph reads and through which view, the factors (2 x, cv_name_weight, cv_weight), the ITC rate,
the `shared*` variable list of the space-mapping optimizer, the three CNN parameter sets; ApplyAdagrad (acc0 = 0.1, no epsilon) is
applied here as stated in SURVEY.md 9.3.  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import attr_cnn_oracle as ao
from oracle import multike_oracle as mo
from oracle.model_oracle import OracleMultiKE

TABLE = {"rv_ent_embeds": "rv_ent", "rel_embeds": "rel", "av_ent_embeds": "av_ent", "attr_embeds": "attr", "ent_embeds": "ent"}
MAPS = ("nv_mapping", "rv_mapping", "av_mapping")


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "graphs_golden.npz"))


def _oracle(g):
    lr, itc, cvn, cvw, ow = (float(x) for x in g["args"])
    tables = {v: g["raw_" + k] for k, v in TABLE.items()}
    tables.update(name=g["name"], lit=g["lit"])
    cnn = [{n: g[f"cnn{k}_{n}"] for n in ao.PARAM_NAMES} for k in range(3)]
    return OracleMultiKE(tables, cnn, [g["raw_" + m] for m in MAPS], learning_rate=lr, itc_learning_rate=itc, cv_name_weight=cvn, cv_weight=cvw,
                         orthogonal_weight=ow)


def _variables(o):
    out = {k: o.t[v] for k, v in TABLE.items()}
    out.update({m: o.M[i] for i, m in enumerate(MAPS)})
    out.update({f"cnn{k}_{n}": o.cnn[k][n] for k in range(3) for n in ao.PARAM_NAMES})
    return out


def _initial(g):
    out = {k: g["raw_" + k] for k in list(TABLE) + list(MAPS)}
    out.update({f"cnn{k}_{n}": g[f"cnn{k}_{n}"] for k in range(3) for n in ao.PARAM_NAMES})
    return out


def _run(o, g, key):
    f = lambda i: g[f"{key}_feed{i}"]
    B = len(f(0))
    off = np.array([0, B])
    if key == "relation":
        N = len(f(3)) // B
        return o.relation_epoch((f(0), f(1), f(2)), (f(3), f(4), f(5)), off, N) * B
    if key in ("ckge_rel", "ckgp_rel"):
        return o.relation_positives_epoch(key, (f(0), f(1), f(2)), f(3) if key == "ckgp_rel" else None, off, 2.0) * B
    if key in ("attribute", "ckge_attr", "ckga_attr"):
        return o.attribute_epoch(key, (f(0), f(1), f(2)), None if key == "ckge_attr" else f(3), off, 2.0 if key == "ckge_attr" else 1.0) * B
    if key == "common":
        return o.common_space_epoch(f(0), off) * B
    return o.space_mapping_epoch(f(0), off) * B


@pytest.mark.parametrize("key", ["relation", "attribute", "ckge_rel", "ckge_attr", "ckga_attr", "ckgp_rel", "common", "mapping"])
def test_one_step_of_every_graph(G, key):
    g = G
    o = _oracle(g)
    loss = _run(o, g, key)
    np.testing.assert_allclose(loss, float(g[f"{key}_loss"]), rtol=1e-11)      # the figure the reference's loop accumulates
    lr = float(g[f"{key}_lr"])
    allowed = set(g[f"{key}_var_list"].tolist()) if int(g[f"{key}_has_var_list"]) else None
    init, now = _initial(g), _variables(o)
    moved = []
    for name, w0 in init.items():
        gk = f"{key}_g_{name}"
        if gk in g.files and (allowed is None or name in allowed):
            grad = g[gk]
            acc = 0.1 + grad * grad                                            # ApplyAdagrad: accumulate, then step (no epsilon)
            np.testing.assert_allclose(now[name], w0 - lr * grad / np.sqrt(acc), rtol=1e-9, atol=1e-13, err_msg=f"{key}: {name}")
            moved.append(name)
        else:
            assert np.array_equal(now[name], w0), f"{key}: {name} must not move"
    assert moved


def test_what_each_graph_reads_and_moves(G):
    """The facts the product is built on, read off the executed graphs."""
    g = G
    has = lambda key: sorted(f[len(key) + 3:] for f in g.files if f.startswith(key + "_g_"))
    for key in ("relation", "ckge_rel", "ckgp_rel"):
        assert has(key) == ["rel_embeds", "rv_ent_embeds"]
    for k, key in enumerate(("attribute", "ckge_attr", "ckga_attr")):         # one parameter set per conv() call
        assert has(key) == sorted(["attr_embeds", "av_ent_embeds"] + [f"cnn{k}_{n}" for n in ao.PARAM_NAMES])
    assert has("common") == ["av_ent_embeds", "ent_embeds", "rv_ent_embeds"] and float(g["common_lr"]) == float(g["args"][1])
    np.testing.assert_allclose(float(g["common_minimised"]), float(g["args"][3]) * float(g["common_loss"]), rtol=1e-14)
    assert g["mapping_var_list"].tolist() == ["av_mapping", "ent_embeds", "nv_mapping", "rv_mapping"]      # names starting with "shared"
    assert "mapping_g_rv_ent_embeds" in g.files and "mapping_g_av_ent_embeds" in g.files                  # a gradient exists, no update
    # the tables a graph looks rows up in are the NORMALISED views (xavier_init(..., True)); the attribute table is raw
    np.testing.assert_allclose(g["view_rv_ent"], mo.l2_normalize_rows(g["raw_rv_ent_embeds"]), rtol=1e-14)
    assert np.array_equal(g["view_attr"], g["raw_attr_embeds"])
