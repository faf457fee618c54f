"""CPU ORACLE (test infrastructure, not product code) — the attribute-view CNN scorer and its loss.

Restates `conv()` of the reference (code/MultiKE_model.py:34-63) and the three graphs that use it
(:134-151 attribute view, :171-185 ckge_attr, :203-221 ckga_attr) from TF1 semantics:

  input  [B,2,d,1]: row 0 = attribute embedding (raw, code/MultiKE_model.py:97), row 1 = literal vector (constant)
  tf.layers.batch_normalization(x, 2)  -> 2nd positional argument is `axis` = 2 (the width axis), inference mode:
         x' = gamma[w] * x / sqrt(moving_var + 1e-3) + beta[w], moving mean 0 / variance 1, gamma/beta trainable
  2 x conv2d(filters=2, kernel [2,4], stride 1, padding SAME, tanh): cross-correlation, pad H bottom 1, W left 1 right 2
  tf.nn.l2_normalize(axis=2): over the width axis per (row, channel), eps 1e-12
  flatten -> index h*2d + w*2 + c ;  dense(4d -> d, tanh)
  tf.nn.l2_normalize(dense) with NO axis: over the whole [B,d] batch ("important!!")
  score = -sum((h - out)^2, 1) ;  loss = scale * sum_i w_i * log(1 + exp(-score_i))

TensorFlow itself is not available: **parity unpinned at the TF boundary** — for the leaf ops (tf.layers batch-norm in
inference mode, conv2d SAME, dense).  The composition above them is pinned by the reference's own `conv` and
`MultiKE._define_attribute_view_graph` EXECUTED over eagerly forwarded TensorFlow calls, with float64 autograd through them
(tests/golden/make_golden.py `cnn_reference_fixture` -> keys `n*_ref_*` of tests/golden/cnn_golden.npz), and by an independent
torch restatement (F.conv2d; `cnn_fixture`) that agrees with it to 1e-10: the forward and the hand-derived backward below are
held to both in tests/test_oracle_cnn.py.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-3
L2_EPS = 1e-12
PARAM_NAMES = ("gamma", "beta", "K1", "b1", "K2", "b2", "W", "bias")


def init_params(dim, rng: np.random.Generator, dtype=np.float64):
    """TF1 defaults: BN gamma 1 / beta 0; conv and dense kernels glorot-uniform, biases 0."""

    def glorot(shape, fan_in, fan_out):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(dtype)

    return {
        "gamma": np.ones(dim, dtype), "beta": np.zeros(dim, dtype),
        "K1": glorot((2, 4, 1, 2), 2 * 4 * 1, 2 * 4 * 2), "b1": np.zeros(2, dtype),
        "K2": glorot((2, 4, 2, 2), 2 * 4 * 2, 2 * 4 * 2), "b2": np.zeros(2, dtype),
        "W": glorot((4 * dim, dim), 4 * dim, dim), "bias": np.zeros(dim, dtype),
    }


def _conv_same(x, K, b):
    """x [B,2,d,Cin], K [2,4,Cin,Cout] -> pre-activation [B,2,d,Cout] (TF SAME padding for a 2x4 kernel)."""
    B, H, Wd, Cin = x.shape
    xp = np.zeros((B, H + 1, Wd + 3, Cin), x.dtype)
    xp[:, :H, 1:Wd + 1] = x
    out = np.zeros((B, H, Wd, K.shape[3]), x.dtype) + b
    for kh in range(2):
        for kw in range(4):
            out += xp[:, kh:kh + H, kw:kw + Wd, :] @ K[kh, kw]
    return out, xp


def _conv_same_backward(dpre, xp, K):
    """-> (dK, db, dx) for out = conv_same(x)."""
    B, H, Wd, Cout = dpre.shape
    dK = np.zeros_like(K)
    dxp = np.zeros_like(xp)
    for kh in range(2):
        for kw in range(4):
            patch = xp[:, kh:kh + H, kw:kw + Wd, :]
            dK[kh, kw] = np.einsum("bhwc,bhwf->cf", patch, dpre)
            dxp[:, kh:kh + H, kw:kw + Wd, :] += dpre @ K[kh, kw].T
    return dK, dpre.sum(axis=(0, 1, 2)), dxp[:, :H, 1:Wd + 1]


def forward(p, hs, as_, vs):
    """hs/as_/vs: gathered rows [B,d] (hs through the normalised view).  Returns (score [B], cache)."""
    s = 1.0 / np.sqrt(1.0 + BN_EPS)
    raw = np.stack([as_, vs], 1)                                   # [B,2,d]
    x = (raw * (p["gamma"] * s) + p["beta"])[..., None]            # [B,2,d,1]
    pre1, xp0 = _conv_same(x, p["K1"], p["b1"])
    c1 = np.tanh(pre1)
    pre2, xp1 = _conv_same(c1, p["K2"], p["b2"])
    c2 = np.tanh(pre2)
    ssq = np.sum(c2 * c2, axis=2, keepdims=True)                   # over width, per (b,h,c)
    n = 1.0 / np.sqrt(np.maximum(ssq, L2_EPS))
    y = c2 * n
    flat = y.reshape(y.shape[0], y.shape[1] * y.shape[2] * y.shape[3])   # index h*2d + w*2 + c (explicit: B may be 0)
    zpre = flat @ p["W"] + p["bias"]
    z = np.tanh(zpre)
    S = np.sum(z * z)
    inv = 1.0 / np.sqrt(max(S, L2_EPS))
    out = z * inv
    diff = hs - out
    score = -np.sum(diff * diff, axis=1)
    cache = dict(s=s, raw=raw, xp0=xp0, c1=c1, xp1=xp1, c2=c2, ssq=ssq, n=n, y=y, flat=flat, z=z, S=S, inv=inv, out=out,
                 diff=diff)
    return score, cache


def dp_tail(c, hs, ws, scale, S_global):
    """Loss tail on a PART of a batch whose batch-wide sum of z^2 is S_global (data-parallel evaluation: the part's own
    contribution is c["S"]).  Returns (loss of the part, g_h, g_out, T_part = sum g_out . out over the part)."""
    inv = 1.0 / np.sqrt(max(S_global, L2_EPS))
    out = c["z"] * inv
    diff = hs - out
    x = np.sum(diff * diff, axis=1)
    w = np.ones_like(x) if ws is None else ws.astype(x.dtype)
    loss = scale * np.sum(w * np.logaddexp(0.0, x))
    coef = scale * w / (1.0 + np.exp(-x))                          # dL/dx
    g_h = 2.0 * coef[:, None] * diff
    g_out = -g_h
    return loss, g_h, g_out, float(np.sum(g_out * out)), dict(inv=inv, out=out)


def dp_backward(p, c, t, g_out, S_global, T_global):
    """Backward of the part from g_out, given the batch-wide scalars.  Returns the parameter gradients of the part (they
    add up over the parts) and 'as' (gradient w.r.t. the part's attribute rows)."""
    if S_global > L2_EPS:
        dz = t["inv"] * (g_out - t["out"] * T_global)
    else:
        dz = t["inv"] * g_out
    dzpre = dz * (1.0 - c["z"] ** 2)
    g = {"W": c["flat"].T @ dzpre, "bias": dzpre.sum(0)}
    dflat = dzpre @ p["W"].T
    dy = dflat.reshape(c["y"].shape)
    dot = np.sum(dy * c["y"], axis=2, keepdims=True)
    dc2 = np.where(c["ssq"] > L2_EPS, c["n"] * (dy - c["y"] * dot), c["n"] * dy)
    dpre2 = dc2 * (1.0 - c["c2"] ** 2)
    g["K2"], g["b2"], dc1 = _conv_same_backward(dpre2, c["xp1"], p["K2"])
    dpre1 = dc1 * (1.0 - c["c1"] ** 2)
    g["K1"], g["b1"], dx = _conv_same_backward(dpre1, c["xp0"], p["K1"])
    dx = dx[..., 0]                                                # [B,2,d]
    g["gamma"] = np.sum(dx * c["raw"] * c["s"], axis=(0, 1))
    g["beta"] = np.sum(dx, axis=(0, 1))
    g["as"] = dx[:, 0, :] * (p["gamma"] * c["s"])
    return g


def loss_and_grads(p, hs, as_, vs, ws=None, scale=1.0):
    """loss = scale * sum w * log(1+exp(-score)); returns (loss, grads) with grads for every parameter plus
    'hs' (gradient w.r.t. the normalised entity rows) and 'as' (w.r.t. the attribute rows)."""
    _, c = forward(p, hs, as_, vs)
    loss, g_h, g_out, T, t = dp_tail(c, hs, ws, scale, c["S"])
    g = dp_backward(p, c, t, g_out, c["S"], T)
    g["hs"] = g_h
    return loss, g


def attribute_step_dense(p, acc, ent, attr, lit, acc_ent, acc_attr, ih, ia, iv, ws, scale, lr, ent_norm=True,
                         attr_norm=False, update=True):
    """One `session.run([loss, optimizer])` of an attribute-view graph with dense-table semantics.  p/acc: parameter
    and Adagrad dicts (modified in place); ent/attr tables and their accumulators likewise; lit is constant."""
    from .multike_oracle import adagrad_dense, l2_normalize_rows, l2_normalize_rows_backward
    E = l2_normalize_rows(ent) if ent_norm else ent
    A = l2_normalize_rows(attr) if attr_norm else attr
    loss, g = loss_and_grads(p, E[ih], A[ia], lit[iv], ws, scale)
    if update:
        ge = np.zeros_like(ent)
        ga = np.zeros_like(attr)
        np.add.at(ge, ih, g["hs"])
        np.add.at(ga, ia, g["as"])
        adagrad_dense(ent, acc_ent, l2_normalize_rows_backward(ent, ge) if ent_norm else ge, lr)
        adagrad_dense(attr, acc_attr, l2_normalize_rows_backward(attr, ga) if attr_norm else ga, lr)
        for k in PARAM_NAMES:
            adagrad_dense(p[k], acc[k], g[k], lr)
    return loss, g
