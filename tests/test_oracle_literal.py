"""The literal auto-encoder oracle (hand-derived backward) against torch autograd, float64, CPU."""
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
