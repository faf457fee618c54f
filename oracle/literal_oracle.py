"""CPU ORACLE (test infrastructure, not product code) — the literal auto-encoder (code/literal_encoder.py:19-144).

Restated from the reference + TF1 semantics: encoder 1500 -> 1024 -> 512 -> dim and the mirrored decoder,
`x @ W + b` per layer with an optional sigmoid / tanh (the shipped `encoder_active: "thah"` matches neither branch at
code/literal_encoder.py:75-78, so the shipped model is purely linear); `tf.nn.l2_normalize` with no axis over the whole
code matrix when `encoder_normalize`; loss = mean((decoded - x)^2); one optimizer (Adagrad, acc0 = 0.1, no epsilon) over
all weights and biases.  The final encoding is a NumPy forward over the un-normalised inputs (:114-144).
TensorFlow is unavailable: **parity unpinned at the TF boundary** for the optimizer only — the graph itself (`_init_graph`,
`_loss_optimizer`, `encoder`, `decoder`) is the reference's code EXECUTED over eagerly forwarded calls (every op of it forwards
exactly), float64 autograd through it: tests/golden/make_golden.py `ae_graph_fixture` -> tests/golden/graphs_golden.npz, held to in
tests/test_oracle_literal.py beside the torch-autograd cross-check.
"""
import numpy as np

L2_EPS = 1e-12


def _act(x, active):
    if active == "sigmoid":
        return 1.0 / (1.0 + np.exp(-x))
    if active == "tanh":
        return np.tanh(x)
    return x


def _act_grad(y, active):
    if active == "sigmoid":
        return y * (1.0 - y)
    if active == "tanh":
        return 1.0 - y * y
    return np.ones_like(y)


def init_params(dims, rng, dtype=np.float64):
    """dims = [input, h1, ..., code].  tf.random_normal_initializer: N(0, 1) for weights AND biases (:45-60)."""
    n = len(dims) - 1
    p = {}
    for i in range(n):
        p[f"encoder_h{i}"] = rng.standard_normal((dims[i], dims[i + 1])).astype(dtype)
        p[f"encoder_b{i}"] = rng.standard_normal(dims[i + 1]).astype(dtype)
    for i in range(n):
        j = n - i
        p[f"decoder_h{i}"] = rng.standard_normal((dims[j], dims[j - 1])).astype(dtype)
        p[f"decoder_b{i}"] = rng.standard_normal(dims[j - 1]).astype(dtype)
    return p


def encode(p, x, n_layers, active):
    h = x
    for i in range(n_layers):
        h = _act(h @ p[f"encoder_h{i}"] + p[f"encoder_b{i}"], active)
    return h


def loss_and_grads(p, x, n_layers, active, normalize):
    acts = [x]
    h = x
    for i in range(n_layers):
        h = _act(h @ p[f"encoder_h{i}"] + p[f"encoder_b{i}"], active)
        acts.append(h)
    code = h
    if normalize:
        S = np.sum(code * code)
        inv = 1.0 / np.sqrt(max(S, L2_EPS))
        h = code * inv
    dacts = [h]
    for i in range(n_layers):
        h = _act(h @ p[f"decoder_h{i}"] + p[f"decoder_b{i}"], active)
        dacts.append(h)
    diff = h - x
    loss = np.mean(diff * diff)
    g = {}
    d = 2.0 * diff / diff.size
    for i in reversed(range(n_layers)):
        d = d * _act_grad(dacts[i + 1], active)
        g[f"decoder_h{i}"] = dacts[i].T @ d
        g[f"decoder_b{i}"] = d.sum(0)
        d = d @ p[f"decoder_h{i}"].T
    if normalize:
        out = dacts[0]
        d = inv * (d - out * np.sum(d * out)) if S > L2_EPS else inv * d
    for i in reversed(range(n_layers)):
        d = d * _act_grad(acts[i + 1], active)
        g[f"encoder_h{i}"] = acts[i].T @ d
        g[f"encoder_b{i}"] = d.sum(0)
        d = d @ p[f"encoder_h{i}"].T
    return loss, g


def adagrad_step(p, acc, g, lr):
    for k in p:
        acc[k] += g[k] * g[k]
        p[k] -= lr * g[k] / np.sqrt(acc[k])
