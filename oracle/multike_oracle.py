"""CPU ORACLE — test infrastructure, NOT product code.

A NumPy restatement of the arithmetic of the MultiKE training hot path, written from the reference's
semantics (nju-websoft/MultiKE, paths below are relative to the reference root).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; the product
(`multike_amd/`) never does and fails loudly when the HIP library is missing.

Pinning status:
  * loss functions: pinned — `tests/golden/make_golden.py` executes the reference's own `code/losses.py`
    (through an op-name shim that maps the dozen `tf.*` calls it makes onto torch) and stores inputs,
    losses and autograd gradients in `tests/golden/losses_*.npz`; `tests/test_oracle_golden.py` checks
    this file against them.
  * graph-level semantics that live in TensorFlow 1.x itself (gradient of `tf.nn.l2_normalize`,
    `tf.train.AdagradOptimizer` = ApplyAdagrad with accumulator 0.1 and no epsilon): TF is not in the
    reference tree and not installable here => **parity unpinned at the TF boundary**; cross-checked
    against torch autograd / torch.optim.Adagrad(initial_accumulator_value=0.1, eps=0) by the same
    golden script.

Everything works in the dtype of its inputs (float64 arrays give the float64 "truth", float32 arrays
give the reference's working precision).
"""
from __future__ import annotations

import numpy as np

L2_EPS = 1e-12  # tf.nn.l2_normalize default epsilon — code/base/initializers.py:26
ADAGRAD_INIT_ACC = 0.1  # tf.train.AdagradOptimizer(initial_accumulator_value=0.1) — code/MultiKE_model.py:17


# ----------------------------------------------------------------------------------------------
# normalise-on-read — code/base/initializers.py:22-26 (xavier_init(..., is_l2_norm=True))
# ----------------------------------------------------------------------------------------------
def l2_normalize_rows(w: np.ndarray) -> np.ndarray:
    """tf.nn.l2_normalize(w, 1): w * rsqrt(max(sum_j w_j^2, 1e-12))."""
    ssq = np.sum(w * w, axis=1, keepdims=True)
    return w / np.sqrt(np.maximum(ssq, w.dtype.type(L2_EPS)))


def l2_normalize_rows_backward(w: np.ndarray, g_hat: np.ndarray) -> np.ndarray:
    """Gradient w.r.t. the raw rows given the gradient w.r.t. the normalised rows.

    For ssq > eps: g = (g_hat - w_hat * (w_hat . g_hat)) / ||w||.  For ssq <= eps the `maximum` clamps
    and its derivative w.r.t. ssq is 0, so g = g_hat * rsqrt(eps).
    """
    ssq = np.sum(w * w, axis=1, keepdims=True)
    inv = 1.0 / np.sqrt(np.maximum(ssq, w.dtype.type(L2_EPS)))
    w_hat = w * inv
    dot = np.sum(w_hat * g_hat, axis=1, keepdims=True)
    proj = np.where(ssq > L2_EPS, w_hat * dot, 0.0).astype(w.dtype)
    return (g_hat - proj) * inv


# ----------------------------------------------------------------------------------------------
# losses.py — the eight public functions, over already-gathered [B, d] arrays
# ----------------------------------------------------------------------------------------------
def _sqdist(hs, rs, ts):
    d = hs + rs - ts
    return d, np.sum(d * d, axis=1)


def _log1pexp(x):
    # the reference writes log(1 + exp(x)) (code/losses.py:9-10); same value, written stably
    return np.logaddexp(0.0, x).astype(x.dtype)


def relation_logistic_loss(phs, prs, pts, nhs, nrs, nts):
    """code/losses.py:4-12."""
    _, x = _sqdist(phs, prs, pts)
    _, y = _sqdist(nhs, nrs, nts)
    return np.sum(_log1pexp(x)) + np.sum(_log1pexp(-y))


def attribute_logistic_loss(phs, pas, pvs, pws, nhs, nas, nvs, nws):
    """code/losses.py:15-27 (dead code in the reference, part of the losses.py surface)."""
    _, x = _sqdist(phs, pas, pvs)
    _, y = _sqdist(nhs, nas, nvs)
    return np.sum(_log1pexp(x) * pws) + np.sum(_log1pexp(-y) * nws)


def relation_logistic_loss_wo_negs(phs, prs, pts):
    """code/losses.py:30-34."""
    _, x = _sqdist(phs, prs, pts)
    return np.sum(_log1pexp(x))


def attribute_logistic_loss_wo_negs(phs, pas, pvs):
    """code/losses.py:37-41."""
    return relation_logistic_loss_wo_negs(phs, pas, pvs)


def logistic_loss_wo_negs(phs, pas, pvs, pws):
    """code/losses.py:44-50."""
    _, x = _sqdist(phs, pas, pvs)
    return np.sum(_log1pexp(x) * pws)


def orthogonal_loss(mapping, eye):
    """code/losses.py:61-63."""
    return np.sum((mapping @ mapping.T - eye) ** 2)


def space_mapping_loss(view_embeds, shared_embeds, mapping, eye, orthogonal_weight, norm_w=0.0001):
    """code/losses.py:53-58.  tf.nn.l2_normalize without an axis normalises over ALL elements."""
    mapped = view_embeds @ mapping
    mapped = mapped / np.sqrt(np.maximum(np.sum(mapped * mapped), L2_EPS))
    map_loss = np.sum((shared_embeds - mapped) ** 2)
    norm_loss = np.sum(mapping * mapping)
    return map_loss + orthogonal_weight * orthogonal_loss(mapping, eye) + norm_w * norm_loss


def alignment_loss(ents1, ents2):
    """code/losses.py:66-69."""
    d = ents1 - ents2
    return np.sum(d * d)


# gradients of the logistic terms w.r.t. the gathered rows
def logistic_term_grads(hs, rs, ts, sign, ws=None):
    """loss = sum w*log(1+exp(sign*||h+r-t||^2)); returns (loss, gh, gr, gt)."""
    d, x = _sqdist(hs, rs, ts)
    z = sign * x
    w = np.ones_like(x) if ws is None else ws.astype(x.dtype)
    loss = np.sum(_log1pexp(z) * w)
    sig = 1.0 / (1.0 + np.exp(-z))
    c = (2.0 * sign) * w * sig
    g = c[:, None] * d
    return loss, g, g, -g


# ----------------------------------------------------------------------------------------------
# graph-level restatement: one relation-view train step with the reference's DENSE semantics
# (code/MultiKE_model.py:114-132 graph + :304-310 session.run)
# ----------------------------------------------------------------------------------------------
def adagrad_dense(var, acc, grad, lr):
    """TF1 ApplyAdagrad on the whole variable: acc += g^2 ; var -= lr * g / sqrt(acc)."""
    acc += grad * grad
    var -= var.dtype.type(lr) * grad / np.sqrt(acc)


def adam_dense(var, m, v, grad, lr, step, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """TF1 ApplyAdam on the whole variable (tf.train.AdamOptimizer defaults; `step` = 1, 2, ...):
    lr_t = lr sqrt(1-b2^t)/(1-b1^t) ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; var -= lr_t m / (sqrt(v) + eps).
    Selectable through args.optimizer (code/MultiKE_model.py:15-25); a zero gradient still moves the weight."""
    dt = var.dtype.type
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    m *= dt(beta1)
    m += dt(1.0 - beta1) * grad
    v *= dt(beta2)
    v += dt(1.0 - beta2) * grad * grad
    var -= dt(lr_t) * m / (np.sqrt(v) + dt(epsilon))


def adadelta_dense(var, accum, accum_update, grad, lr, rho=0.95, epsilon=1e-8):
    """TF1 ApplyAdadelta (tf.train.AdadeltaOptimizer defaults): accum = rho accum + (1-rho) g^2 ;
    u = sqrt(accum_update + eps) / sqrt(accum + eps) * g ; var -= lr u ; accum_update = rho accum_update + (1-rho) u^2."""
    dt = var.dtype.type
    accum *= dt(rho)
    accum += dt(1.0 - rho) * grad * grad
    u = np.sqrt(accum_update + dt(epsilon)) / np.sqrt(accum + dt(epsilon)) * grad
    var -= dt(lr) * u
    accum_update *= dt(rho)
    accum_update += dt(1.0 - rho) * u * u


def relation_view_step_dense(ent, rel, acc_ent, acc_rel, pos, neg, lr, pos_w=None, neg_w=None, scale=1.0,
                             ent_norm=True, rel_norm=True, update=True):
    """One `session.run([loss, optimizer])` of the relation-view graph, whole-table semantics.

    ent/rel/acc_* are modified in place (when update).  pos/neg are (h, r, t) integer arrays (neg may be
    None or empty).  Returns (loss, ghat_ent, ghat_rel): the scalar loss and the dense gradients w.r.t.
    the NORMALISED tables (before the Jacobian), which the scatter kernels are checked against.
    """
    dt = ent.dtype
    E = l2_normalize_rows(ent) if ent_norm else ent
    R = l2_normalize_rows(rel) if rel_norm else rel
    ph, pr, pt = (np.asarray(a, dtype=np.int64) for a in pos)
    ghat_ent = np.zeros_like(ent)
    ghat_rel = np.zeros_like(rel)
    loss_p, gh, gr, gt = logistic_term_grads(E[ph], R[pr], E[pt], +1.0, pos_w)
    np.add.at(ghat_ent, ph, gh)
    np.add.at(ghat_rel, pr, gr)
    np.add.at(ghat_ent, pt, gt)
    loss = loss_p
    if neg is not None and len(neg[0]) > 0:
        nh, nr, nt = (np.asarray(a, dtype=np.int64) for a in neg)
        loss_n, gh, gr, gt = logistic_term_grads(E[nh], R[nr], E[nt], -1.0, neg_w)
        np.add.at(ghat_ent, nh, gh)
        np.add.at(ghat_rel, nr, gr)
        np.add.at(ghat_ent, nt, gt)
        loss = loss + loss_n
    s = dt.type(scale)
    ghat_ent *= s
    ghat_rel *= s
    loss = loss * scale
    if update:
        g_ent = l2_normalize_rows_backward(ent, ghat_ent) if ent_norm else ghat_ent
        g_rel = l2_normalize_rows_backward(rel, ghat_rel) if rel_norm else ghat_rel
        adagrad_dense(ent, acc_ent, g_ent, lr)
        adagrad_dense(rel, acc_rel, g_rel, lr)
    return loss, ghat_ent, ghat_rel


def space_mapping_grads(view_rows, shared_rows, mapping, orthogonal_weight, norm_w=0.0001):
    """Closed-form gradients of `space_mapping_loss` (code/losses.py:53-63) w.r.t. the gathered shared rows and the
    mapping matrix.  Returns (loss, g_shared [B,d], g_mapping [d,d])."""
    d = mapping.shape[0]
    P = view_rows @ mapping
    S = np.sum(P * P)
    inv = 1.0 / np.sqrt(max(S, L2_EPS))
    out = P * inv
    diff = shared_rows - out
    g_out = -2.0 * diff
    dP = inv * (g_out - out * np.sum(g_out * out)) if S > L2_EPS else inv * g_out
    Q = mapping @ mapping.T - np.eye(d, dtype=mapping.dtype)
    gM = view_rows.T @ dP + orthogonal_weight * 4.0 * (Q @ mapping) + norm_w * 2.0 * mapping
    loss = np.sum(diff * diff) + orthogonal_weight * np.sum(Q * Q) + norm_w * np.sum(mapping * mapping)
    return loss, 2.0 * diff, gM


def space_mapping_step_dense(ent, acc_ent, views, mappings, acc_mappings, idx, lr, orthogonal_weight, ent_norm=True, norm_w=0.0001):
    """One `session.run([shared_comb_loss, shared_comb_optimizer])` (code/MultiKE_model.py:241-261,:447-451) with dense-table
    semantics: `views` = [(table, read_through_l2_normalize)] (constants here), `mappings` / `acc_mappings` lists of [d,d]
    arrays and `ent` / `acc_ent` are updated in place (TF1 Adagrad).  Returns the loss."""
    E = l2_normalize_rows(ent) if ent_norm else ent
    F = E[idx]
    total = 0.0
    gF = np.zeros_like(F)
    gMs = []
    for (table, norm), M in zip(views, mappings):
        V = (l2_normalize_rows(table) if norm else table)[idx]
        loss, g_shared, gM = space_mapping_grads(V, F, M, orthogonal_weight, norm_w)
        total += loss
        gF += g_shared
        gMs.append(gM)
    ge = np.zeros_like(ent)
    np.add.at(ge, idx, gF)
    adagrad_dense(ent, acc_ent, l2_normalize_rows_backward(ent, ge) if ent_norm else ge, lr)
    for M, a, g in zip(mappings, acc_mappings, gMs):
        adagrad_dense(M, a, g, lr)
    return total


def alignment_step_dense(table_a, table_b, acc_a, acc_b, ia, ib, lr, weight=1.0, a_norm=True, b_norm=True,
                         update=True):
    """One alignment term  weight * sum ||A^[ia] - B^[ib]||^2  with dense-table semantics
    (code/MultiKE_model.py:229-239; acc_x None = constant table)."""
    A = l2_normalize_rows(table_a) if a_norm else table_a
    B = l2_normalize_rows(table_b) if b_norm else table_b
    ia = np.asarray(ia, dtype=np.int64)
    ib = np.asarray(ib, dtype=np.int64)
    d = A[ia] - B[ib]
    loss = weight * np.sum(d * d)
    g = (2.0 * weight) * d
    ghat_a = np.zeros_like(table_a)
    ghat_b = np.zeros_like(table_b)
    np.add.at(ghat_a, ia, g)
    np.add.at(ghat_b, ib, -g)
    if update:
        if acc_a is not None:
            adagrad_dense(table_a, acc_a, l2_normalize_rows_backward(table_a, ghat_a) if a_norm else ghat_a, lr)
        if acc_b is not None:
            adagrad_dense(table_b, acc_b, l2_normalize_rows_backward(table_b, ghat_b) if b_norm else ghat_b, lr)
    return loss, ghat_a, ghat_b


def rows_update_sparse(table, acc, ghat, lr, normalize=True, optimizer="Adagrad"):
    """The sparse-equivalent of the dense update: only rows whose ghat is non-zero move."""
    rows = np.nonzero(np.any(ghat != 0, axis=1))[0]
    w = table[rows]
    g = l2_normalize_rows_backward(w, ghat[rows]) if normalize else ghat[rows]
    if optimizer == "Adagrad":
        a = acc[rows] + g * g
        acc[rows] = a
        table[rows] = w - table.dtype.type(lr) * g / np.sqrt(a)
    else:  # SGD — tf.train.GradientDescentOptimizer, code/MultiKE_model.py:24
        table[rows] = w - table.dtype.type(lr) * g
    return rows


# ----------------------------------------------------------------------------------------------
# host-side arithmetic of the batch generators
# ----------------------------------------------------------------------------------------------
def kg_batch_split(n1: int, n2: int, batch_size: int):
    """code/base/batch.py:36-37: proportional split of a batch between the two KGs."""
    b1 = int(n1 / (n1 + n2) * batch_size)
    return b1, batch_size - b1


def task_divide(idx, n):
    """code/utils.py:35-49: how step indices are dealt to the n producer processes."""
    total = len(idx)
    if n <= 0 or total == 0 or n > total:
        return [idx]
    if n == total:
        return [[i] for i in idx]
    j = total // n
    out = [idx[k * j:(k + 1) * j] for k in range(n - 1)]
    out.append(idx[(n - 1) * j:])
    return out


def xavier_truncated_normal(shape, rng: np.random.Generator, dtype=np.float32):
    """TF1 xavier_initializer(uniform=False) as used at code/base/initializers.py:24-25:
    truncated normal (+-2 sigma, resampled) with sigma = sqrt(1.3 * 2 / (fan_in + fan_out))."""
    n, d = shape
    sigma = np.sqrt(1.3 * 2.0 / (n + d))
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * sigma).astype(dtype)


def common_space_step_dense(ent, name, rv, av, acc_ent, acc_rv, acc_av, idx, lr, cv_name_weight=1.0, cv_weight=1.0):
    """One common-space learning step (code/MultiKE_model.py:225-239): loss = cv_name_weight * align(ent, name) +
    align(ent, rv) + align(ent, av) over the entities idx; the optimizer minimises cv_weight * loss in ONE step (ent gets the
    sum of its three gradients).  name is a constant read as-is (:88); the others are read through l2_normalize.  Tables and
    accumulators are updated in place; returns cv_weight * loss."""
    w = cv_weight
    l1, g_e1, _ = alignment_step_dense(ent, name, None, None, idx, idx, lr, w * cv_name_weight, True, False, update=False)
    l2, g_e2, g_rv = alignment_step_dense(ent, rv, None, None, idx, idx, lr, w, True, True, update=False)
    l3, g_e3, g_av = alignment_step_dense(ent, av, None, None, idx, idx, lr, w, True, True, update=False)
    adagrad_dense(ent, acc_ent, l2_normalize_rows_backward(ent, g_e1 + g_e2 + g_e3), lr)
    adagrad_dense(rv, acc_rv, l2_normalize_rows_backward(rv, g_rv), lr)
    adagrad_dense(av, acc_av, l2_normalize_rows_backward(av, g_av), lr)
    return l1 + l2 + l3
