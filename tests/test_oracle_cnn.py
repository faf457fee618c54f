"""The attribute-CNN oracle (oracle/attr_cnn_oracle.py: forward + hand-derived backward) against (a) torch autograd on an
independent torch restatement of the TF1 semantics and (b) the reference's own attribute-view graph EXECUTED —
`MultiKE._define_attribute_view_graph`, `conv` and `xavier_init` of /root/reference/code run line by line over eagerly forwarded
leaf ops (tests/golden/make_golden.py `cnn_reference_fixture`; keys `n*_ref_*` of tests/golden/cnn_golden.npz).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import attr_cnn_oracle as ao


@pytest.fixture(scope="module")
def cnn_golden():
    return np.load(os.path.join(GOLDEN, "cnn_golden.npz"))


@pytest.mark.parametrize("ci", [0, 1])
def test_forward_and_backward(cnn_golden, ci):
    g = cnn_golden
    pre = f"n{ci}_"
    P = {k: g[pre + "p_" + k] for k in ao.PARAM_NAMES}
    ws = g[pre + "ws"] if (pre + "ws") in g.files else None
    score, _ = ao.forward(P, g[pre + "hs"], g[pre + "as"], g[pre + "vs"])
    np.testing.assert_allclose(score, g[pre + "score"], rtol=1e-10, atol=1e-13)
    loss, grads = ao.loss_and_grads(P, g[pre + "hs"], g[pre + "as"], g[pre + "vs"], ws, float(g[pre + "scale"]))
    np.testing.assert_allclose(loss, g[pre + "loss"], rtol=1e-12)
    for k in ao.PARAM_NAMES + ("hs", "as"):
        np.testing.assert_allclose(grads[k], g[pre + "g_" + k], rtol=1e-8, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("ci", [0, 1])
def test_reference_executed_graph(cnn_golden, ci):
    """(1) the restatement's inputs through the reference's graph: score, loss (the reference multiplies by the weights and sums:
    no scale), every gradient.  (2) tables with repeated / unused rows of any length through the reference's lookups and its
    normalised entity view: the dense table gradients (scatter of the per-triple gradients, Jacobian of l2_normalize on the entity
    rows, the attribute table raw) and the parameter gradients."""
    from oracle import multike_oracle as mo
    g = cnn_golden
    pre, ref = f"n{ci}_", f"n{ci}_ref_"
    P = {k: g[pre + "p_" + k] for k in ao.PARAM_NAMES}
    ws = g[pre + "ws"] if (pre + "ws") in g.files else None
    score, _ = ao.forward(P, g[pre + "hs"], g[pre + "as"], g[pre + "vs"])
    np.testing.assert_allclose(score, g[ref + "score"], rtol=1e-10, atol=1e-13)
    loss, grads = ao.loss_and_grads(P, g[pre + "hs"], g[pre + "as"], g[pre + "vs"], ws, 1.0)
    np.testing.assert_allclose(loss, g[ref + "loss"], rtol=1e-12)
    for k in ao.PARAM_NAMES + ("as",):
        np.testing.assert_allclose(grads[k], g[ref + "g_" + k], rtol=1e-8, atol=1e-12, err_msg=k)
    ent, attr, lit = g[ref + "t_ent"], g[ref + "t_attr"], g[ref + "t_lit"]
    ih, ia, iv, w = g[ref + "t_ih"], g[ref + "t_ia"], g[ref + "t_iv"], g[ref + "t_w"]
    hs = mo.l2_normalize_rows(ent)[ih]
    score, _ = ao.forward(P, hs, attr[ia], lit[iv])
    np.testing.assert_allclose(score, g[ref + "t_score"], rtol=1e-10, atol=1e-13)
    loss, grads = ao.loss_and_grads(P, hs, attr[ia], lit[iv], w, 1.0)
    np.testing.assert_allclose(loss, g[ref + "t_loss"], rtol=1e-12)
    ghat, gattr = np.zeros_like(ent), np.zeros_like(attr)
    np.add.at(ghat, ih, grads["hs"])
    np.add.at(gattr, ia, grads["as"])
    np.testing.assert_allclose(mo.l2_normalize_rows_backward(ent, ghat), g[ref + "t_g_ent"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(gattr, g[ref + "t_g_attr"], rtol=1e-8, atol=1e-12)
    for k in ao.PARAM_NAMES:
        np.testing.assert_allclose(grads[k], g[ref + "t_g_" + k], rtol=1e-8, atol=1e-12, err_msg=k)
    assert not g[ref + "t_g_ent"][-2:].any()              # rows no triple looks up


def test_dense_step_moves_only_touched_rows():
    rng = np.random.default_rng(0)
    d, B = 8, 20
    P = ao.init_params(d, rng)
    acc = {k: np.full_like(v, 0.1) for k, v in P.items()}
    ent = rng.standard_normal((50, d)) * 0.1
    attr = rng.standard_normal((7, d)) * 0.1
    lit = rng.standard_normal((30, d))
    e0, a0 = ent.copy(), attr.copy()
    ih, ia, iv = rng.integers(0, 25, B), rng.integers(0, 7, B), rng.integers(0, 30, B)
    loss, _ = ao.attribute_step_dense(P, acc, ent, attr, lit, np.full_like(ent, 0.1), np.full_like(attr, 0.1), ih, ia, iv,
                                      None, 1.0, 0.01)
    assert loss > 0
    assert np.array_equal(ent[25:], e0[25:]) and not np.array_equal(ent[ih], e0[ih])
    assert not np.array_equal(attr, a0)
