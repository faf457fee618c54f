"""The literal auto-encoder oracle (hand-derived backward) against torch autograd, and against the reference's own graph EXECUTED
(`AutoEncoderModel._init_graph` / `_loss_optimizer` / `encoder` / `decoder` of /root/reference/code/literal_encoder.py run
unmodified over eagerly forwarded TensorFlow calls: tests/golden/make_golden.py `ae_graph_fixture`).  float64, CPU."""
import numpy as np
import pytest
import torch

from oracle import literal_oracle as lo


@pytest.mark.parametrize("active,normalize", [("thah", True), ("tanh", True), ("sigmoid", False)])
def test_backward_matches_autograd(active, normalize):
    rng = np.random.default_rng(0)
    dims = [30, 16, 8, 5]
    p = lo.init_params(dims, rng)
    for k in p:
        p[k] *= 0.3
    x = rng.standard_normal((11, 30))
    loss, g = lo.loss_and_grads(p, x, 3, active, normalize)
    T = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    act = {"tanh": torch.tanh, "sigmoid": torch.sigmoid}.get(active, lambda t: t)
    h = torch.tensor(x)
    for i in range(3):
        h = act(h @ T[f"encoder_h{i}"] + T[f"encoder_b{i}"])
    if normalize:
        h = h * torch.rsqrt(torch.clamp_min((h * h).sum(), 1e-12))
    for i in range(3):
        h = act(h @ T[f"decoder_h{i}"] + T[f"decoder_b{i}"])
    L = ((h - torch.tensor(x)) ** 2).mean()
    L.backward()
    np.testing.assert_allclose(loss, L.item(), rtol=1e-12)
    for k in p:
        np.testing.assert_allclose(g[k], T[k].grad.numpy(), rtol=1e-9, atol=1e-13, err_msg=k)


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_reference_executed_graph(ci):
    """Loss and every gradient of the reference's executed graph: the shipped activation string (a linear model), tanh, sigmoid,
    with and without the batch-wide l2_normalize between encoder and decoder."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "graphs_golden.npz"))
    pre = f"ae{ci}_"
    normalize, dims = bool(g[pre + "meta"][0]), [int(v) for v in g[pre + "meta"][1:]]
    active = str(g[pre + "active"])
    p = {k[len(pre) + 2:]: g[k] for k in g.files if k.startswith(pre + "p_")}
    assert set(p) == set(lo.init_params(dims, np.random.default_rng(0)))          # the variables the reference's _init_graph creates
    loss, grads = lo.loss_and_grads(p, g[pre + "x"], len(dims) - 1, active, normalize)
    np.testing.assert_allclose(loss, float(g[pre + "loss"]), rtol=1e-12)
    for k in p:
        np.testing.assert_allclose(grads[k], g[pre + "g_" + k], rtol=1e-9, atol=1e-13, err_msg=k)
