"""The attribute-CNN oracle (oracle/attr_cnn_oracle.py: forward + hand-derived backward) against torch autograd on
an independent torch restatement of the TF1 semantics (tests/golden/cnn_golden.npz).  CPU only."""
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
