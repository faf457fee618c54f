"""GPU parity of the attribute-view CNN step against the golden (torch-autograd) vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import attr_cnn_oracle as ao
from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu


def _tables(hs, as_, vs):
    """Put the gathered rows of a fixture into tables (identity indices; hs rows are unit so normalise = identity)."""
    from multike_amd.tables import EmbeddingTable
    B, d = hs.shape
    ent = EmbeddingTable(B, d, "av_ent", normalize=True, values=hs)
    attr = EmbeddingTable(B, d, "attr", normalize=False, values=as_)
    lit = EmbeddingTable(B, d, "lit", normalize=False, trainable=False, values=vs)
    idx = torch.arange(B, dtype=torch.int32, device="cuda")
    return ent, attr, lit, idx


@pytest.mark.parametrize("ci", [0, 1])
def test_golden_loss_and_every_gradient(ci):
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.tables import StepEngine
    g = np.load(os.path.join(GOLDEN, "cnn_golden.npz"))
    pre = f"n{ci}_"
    d = int(g[pre + "meta"][0])
    P = {k: g[pre + "p_" + k] for k in ao.PARAM_NAMES}
    ws = torch.tensor(g[pre + "ws"], dtype=torch.float32, device="cuda") if (pre + "ws") in g.files else None
    ent, attr, lit, idx = _tables(g[pre + "hs"], g[pre + "as"], g[pre + "vs"])
    cnn = AttrCNN(d, params=P)
    eng = StepEngine()
    lp = cnn.step(eng, ent, attr, lit, idx, idx, idx, ws, scale=float(g[pre + "scale"]), update=False)
    np.testing.assert_allclose(float(lp.sum()), g[pre + "loss"], rtol=5e-6)
    for k in ao.PARAM_NAMES:
        got = cnn.gviews[k].cpu().numpy()
        ref = g[pre + "g_" + k]
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)
    np.testing.assert_allclose(ent.grad[:, :d].cpu().numpy(), g[pre + "g_hs"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(attr.grad[:, :d].cpu().numpy(), g[pre + "g_as"], rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("ci", [0, 1])
def test_reference_executed_attribute_graph(ci):
    """The reference's attribute-view graph EXECUTED (`MultiKE._define_attribute_view_graph` + `conv` + `xavier_init`, eager
    leaf ops: tests/golden/make_golden.py `cnn_reference_fixture`) on tables with repeated and unused rows: the device step's
    loss, the scratch gradients — w.r.t. the normalised entity rows on the device: the Jacobian of the view is applied here, as
    the update launch does —, the attribute table's and every CNN parameter's."""
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.tables import EmbeddingTable, StepEngine
    g = np.load(os.path.join(GOLDEN, "cnn_golden.npz"))
    pre, ref = f"n{ci}_", f"n{ci}_ref_t_"
    d = int(g[pre + "meta"][0])
    P = {k: g[pre + "p_" + k] for k in ao.PARAM_NAMES}
    ent_raw, attr_raw, lit_raw = g[ref + "ent"], g[ref + "attr"], g[ref + "lit"]
    E = EmbeddingTable(ent_raw.shape[0], d, "av_ent", normalize=True, values=ent_raw)
    A = EmbeddingTable(attr_raw.shape[0], d, "attr", normalize=False, values=attr_raw)
    L = EmbeddingTable(lit_raw.shape[0], d, "lit", normalize=False, trainable=False, values=lit_raw)
    t = lambda x: torch.as_tensor(x.astype(np.int32), device="cuda")
    cnn = AttrCNN(d, params=P)
    lp = cnn.step(StepEngine(), E, A, L, t(g[ref + "ih"]), t(g[ref + "ia"]), t(g[ref + "iv"]),
                  torch.as_tensor(g[ref + "w"].astype(np.float32), device="cuda"), scale=1.0, update=False)
    np.testing.assert_allclose(float(lp.sum()), g[ref + "loss"], rtol=5e-6)
    for k in ao.PARAM_NAMES:
        want = g[ref + "g_" + k]
        np.testing.assert_allclose(cnn.gviews[k].cpu().numpy(), want, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(want).max()), err_msg=k)
    g_raw = mo.l2_normalize_rows_backward(ent_raw.astype(np.float64), E.grad[:, :d].cpu().numpy().astype(np.float64))
    np.testing.assert_allclose(g_raw, g[ref + "g_ent"], rtol=1e-3, atol=2e-6)
    np.testing.assert_allclose(A.grad[:, :d].cpu().numpy(), g[ref + "g_attr"], rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("d,B,n_ent,n_attr,n_lit", [(75, 5000, 20000, 300, 8000), (75, 37, 100, 9, 50), (32, 513, 900, 20, 400),
                                                    (130, 64, 200, 11, 90),
                                                    # every width of the fused forward (conv stack + dense layer in one
                                                    # launch: dim <= 80, a quarter-wave per triple, 1..5 positions per lane),
                                                    # its upper edge, and the first width past it
                                                    (8, 100, 60, 7, 30), (40, 300, 500, 13, 200), (64, 215, 300, 10, 100),
                                                    (80, 129, 400, 12, 150),
                                                    # the widths past the fused launches (separate dense / tail-backward / product
                                                    # launches; round 6 built the fused form for 80 < dim <= 128 against these
                                                    # cases, measured it slower and removed it: EXPERIMENTS R6.12)
                                                    (96, 70, 150, 9, 60), (81, 200, 300, 10, 100), (100, 1000, 2000, 40, 700),
                                                    (112, 333, 500, 12, 200), (128, 517, 900, 30, 400), (128, 5000, 20000, 300, 8000)])
def test_three_steps_vs_oracle(d, B, n_ent, n_attr, n_lit):
    """Full step (scatter with duplicate rows, Jacobian + Adagrad on the entity table, plain Adagrad on the raw attribute
    table and on the packed CNN parameters) against the float64 dense oracle."""
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.tables import EmbeddingTable, StepEngine
    rng = np.random.default_rng(d + B)
    P = ao.init_params(d, rng)
    P["bias"] = 0.05 * rng.standard_normal(d)
    P["b1"] = 0.05 * rng.standard_normal(2)
    ent = mo.xavier_truncated_normal((n_ent, d), rng)
    attr = mo.xavier_truncated_normal((n_attr, d), rng)
    lit = rng.standard_normal((n_lit, d)).astype(np.float32)
    lit /= np.linalg.norm(lit, axis=1, keepdims=True)
    E = EmbeddingTable(n_ent, d, "av_ent", normalize=True, values=ent)
    A = EmbeddingTable(n_attr, d, "attr", normalize=False, values=attr)
    L = EmbeddingTable(n_lit, d, "lit", normalize=False, trainable=False, values=lit)
    cnn = AttrCNN(d, params=P)
    eng = StepEngine()
    p64 = {k: v.astype(np.float64) for k, v in P.items()}
    acc = {k: np.full_like(v, 0.1) for k, v in p64.items()}
    e64, a64, l64 = ent.astype(np.float64), attr.astype(np.float64), lit.astype(np.float64)
    ae, aa = np.full_like(e64, 0.1), np.full_like(a64, 0.1)
    for step in range(3):
        ih, ia, iv = rng.integers(0, n_ent, B), rng.integers(0, n_attr, B), rng.integers(0, n_lit, B)
        ws = rng.uniform(0.2, 1.0, B)
        Lo, _ = ao.attribute_step_dense(p64, acc, e64, a64, l64, ae, aa, ih, ia, iv, ws, 2.0, 0.01)
        t = lambda x: torch.as_tensor(x.astype(np.int32), device="cuda")
        lp = cnn.step(eng, E, A, L, t(ih), t(ia), t(iv), torch.as_tensor(ws.astype(np.float32), device="cuda"), scale=2.0,
                      opt_name="attribute", lr=0.01)
        np.testing.assert_allclose(float(lp.sum()), Lo, rtol=1e-5)
    np.testing.assert_allclose(E.raw().cpu().numpy(), e64, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(A.raw().cpu().numpy(), a64, rtol=2e-3, atol=2e-5)
    got = cnn.numpy_params()
    for k in ao.PARAM_NAMES:
        np.testing.assert_allclose(got[k], p64[k], rtol=2e-3, atol=1e-4, err_msg=k)
    assert float(cnn.grads.abs().max()) == 0.0 and float(E.grad.abs().max()) == 0.0 and float(A.grad.abs().max()) == 0.0


@pytest.mark.parametrize("ci", [0, 1])
def test_conv_op_is_differentiable_and_matches_the_golden_gradients(ci):
    """`MultiKE_model.conv` (code/MultiKE_model.py:34-63) as an autograd op on DEVICE rows: the score vector and, through a
    loss built on it with ordinary torch ops, the gradients w.r.t. attr_hs, attr_as and every CNN parameter — against
    cnn_golden.npz (float64 autograd over an independent torch restatement of the TF graph)."""
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.MultiKE_model import conv
    g = np.load(os.path.join(GOLDEN, "cnn_golden.npz"))
    pre = f"n{ci}_"
    d = int(g[pre + "meta"][0])
    cnn = AttrCNN(d, params={k: g[pre + "p_" + k] for k in ao.PARAM_NAMES})
    cnn.params.requires_grad_(True)
    dev = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    hs, as_ = dev(g[pre + "hs"]).requires_grad_(True), dev(g[pre + "as"]).requires_grad_(True)
    vs = dev(g[pre + "vs"])
    score = conv(hs, as_, vs, d, cnn=cnn)
    assert score.shape == (hs.shape[0],) and score.requires_grad and score.cnn is cnn
    np.testing.assert_allclose(score.detach().cpu().numpy(), g[pre + "score"], rtol=2e-5, atol=1e-6)
    per = torch.log(1 + torch.exp(-score))
    if (pre + "ws") in g.files:
        per = per * dev(g[pre + "ws"])
    loss = float(g[pre + "scale"]) * per.sum()
    np.testing.assert_allclose(float(loss.detach()), g[pre + "loss"], rtol=5e-6)
    loss.backward()
    np.testing.assert_allclose(hs.grad.cpu().numpy(), g[pre + "g_hs"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(as_.grad.cpu().numpy(), g[pre + "g_as"], rtol=2e-3, atol=2e-6)
    o = 0
    for name in ao.PARAM_NAMES:
        ref = g[pre + "g_" + name]
        got = cnn.params.grad[o:o + ref.size].view(ref.shape).cpu().numpy()
        o += ref.size
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=name)
    assert o == cnn.params.numel()
    # positional signature of the reference, a fresh parameter set when none is given, loud failure on host tensors
    s2 = conv(hs.detach(), as_.detach(), vs, d, 2, [2, 4], "tanh", 2)
    assert s2.shape == score.shape and s2.cnn is not cnn and s2.cnn.params.requires_grad
    from multike_amd._lib import MultiKEHipError
    with pytest.raises(MultiKEHipError):
        conv(hs.detach().cpu(), as_.detach().cpu(), vs.cpu(), d, cnn=cnn)
    with pytest.raises(MultiKEHipError):
        conv(hs, as_, vs, d, 3, cnn=cnn)


def test_conv_op_gradcheck_against_the_training_step():
    """The op's gradients == what the fused training step (`AttrCNN.step(update=False)`) leaves in the gradient scratch, on a
    batch with an arbitrary upstream gradient shape (the attribute-view loss), dim 75, 1000 rows."""
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.MultiKE_model import conv
    from multike_amd.tables import StepEngine
    rng = np.random.default_rng(9)
    d, B = 75, 1000
    hs = rng.standard_normal((B, d)); hs /= np.linalg.norm(hs, axis=1, keepdims=True)
    as_, vs = 0.3 * rng.standard_normal((B, d)), rng.standard_normal((B, d))
    vs /= np.linalg.norm(vs, axis=1, keepdims=True)
    w = rng.uniform(0.2, 1.0, B).astype(np.float32)
    cnn = AttrCNN(d, seed=4)
    cnn.views["bias"].copy_(torch.tensor(0.05 * rng.standard_normal(d), dtype=torch.float32))
    ent, attr, lit, idx = _tables(hs.astype(np.float32), as_.astype(np.float32), vs.astype(np.float32))
    lp = cnn.step(StepEngine(), ent, attr, lit, idx, idx, idx, torch.tensor(w, device="cuda"), scale=1.0, update=False)
    step_g = cnn.grads.clone()
    cnn.params.requires_grad_(True)
    th, ta = ent.lookup(idx).requires_grad_(True), attr.raw().clone().requires_grad_(True)
    score = conv(th, ta, lit.raw(), d, cnn=cnn)
    loss = (torch.tensor(w, device="cuda") * torch.log(1 + torch.exp(-score))).sum()
    np.testing.assert_allclose(float(loss.detach()), float(lp.sum()), rtol=5e-6)
    loss.backward()
    np.testing.assert_allclose(cnn.params.grad.cpu().numpy(), step_g.cpu().numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(th.grad.cpu().numpy(), ent.grad[:, :d].cpu().numpy(), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), attr.grad[:, :d].cpu().numpy(), rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("tail_opt,bwd_opt", [(1, 1), (0, 1), (1, 0), (0, 0)])
def test_backward_phase_alone_does_not_trust_a_stale_transposed_weight(tail_opt, bwd_opt):
    """mke_attr_step_phases with the loss tail and the backward in SEPARATE calls (the sharded view all-reduces a scalar in
    between).  The fused backward reads W^T from where the tail left it: when the option differed at the tail's call, or no tail of
    this step ran, the backward call must take the unfused path — whatever the option says at ITS call — and give the same
    gradients as the one-call step (round-4 advice)."""
    from multike_amd import _lib
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.tables import StepEngine
    g = np.load(os.path.join(GOLDEN, "cnn_golden.npz"))
    pre = "n0_"
    d = int(g[pre + "meta"][0])
    P = {k: g[pre + "p_" + k] for k in ao.PARAM_NAMES}
    scale = float(g[pre + "scale"])

    def grads(split):
        ent, attr, lit, idx = _tables(g[pre + "hs"], g[pre + "as"], g[pre + "vs"])
        cnn, eng = AttrCNN(d, params=P), StepEngine()
        if not split:
            cnn.step(eng, ent, attr, lit, idx, idx, idx, None, scale=scale, update=False)
        else:
            args, _part = cnn._args(eng, ent, attr, lit, idx, idx, idx, None, int(idx.numel()), scale, "o", 0.01, "Adagrad", False, 1)
            old = _lib.set_option("attr_fused_bwd", tail_opt)
            try:
                _lib.attr_step_phases(args, _lib.ATTR_FWD)
                _lib.attr_step_phases(args, _lib.ATTR_TAIL)
                _lib.set_option("attr_fused_bwd", bwd_opt)
                _lib.attr_step_phases(args, _lib.ATTR_BWD)
            finally:
                _lib.set_option("attr_fused_bwd", old)
        torch.cuda.synchronize()
        return {k: cnn.gviews[k].cpu().numpy().copy() for k in ao.PARAM_NAMES}, attr.grad[:, :d].cpu().numpy().copy()

    one, one_attr = grads(False)
    two, two_attr = grads(True)
    for k in ao.PARAM_NAMES:
        np.testing.assert_allclose(two[k], one[k], rtol=2e-3, atol=2e-5 * max(1.0, np.abs(one[k]).max()), err_msg=k)
    np.testing.assert_allclose(two_attr, one_attr, rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("d", [75, 32, 130])
def test_privatised_attribute_scratch_is_the_same_function(d):
    """The model builds `attr_embeds` with its gradient scratch privatised (triple t adds to copy t % copies; heavy-tailed
    attribute ids pile thousands of same-address atomics on a row otherwise).  Three steps with Zipf-like attribute ids: the same
    tables, parameters and losses as with one copy; every copy back at zero."""
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.tables import EmbeddingTable, StepEngine
    rng = np.random.default_rng(d)
    n_ent, n_attr, n_lit, B = 3000, 40, 500, 1200
    e0, a0 = rng.standard_normal((n_ent, d)).astype(np.float32), rng.standard_normal((n_attr, d)).astype(np.float32) * 0.3
    l0 = rng.standard_normal((n_lit, d)).astype(np.float32)
    pr = np.arange(1, n_attr + 1) ** -1.5
    batches = [(rng.integers(0, n_ent, B), rng.choice(n_attr, B, p=pr / pr.sum()), rng.integers(0, n_lit, B), rng.random(B).astype(np.float32))
               for _ in range(3)]
    out = []
    for copies in (1, 16):
        E = EmbeddingTable(n_ent, d, "e", values=e0)
        A = EmbeddingTable(n_attr, d, "a", normalize=False, values=a0, grad_copies=copies)
        L = EmbeddingTable(n_lit, d, "l", normalize=False, trainable=False, values=l0)
        cnn, eng = AttrCNN(d, seed=5), StepEngine()
        losses = []
        for h, a, v, w in batches:
            t = lambda x, dt=torch.int32: torch.as_tensor(x, device="cuda").to(dt)
            losses.append(float(cnn.step(eng, E, A, L, t(h), t(a), t(v), t(w, torch.float32), lr=0.01).sum()))
        torch.cuda.synchronize()
        assert float(A.grad.abs().max()) == 0.0
        out.append((losses, E.raw().cpu().numpy(), A.raw().cpu().numpy(), {k: v.cpu().numpy().copy() for k, v in cnn.views.items()} if hasattr(cnn, "views") else {}))
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=2e-6)
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out[1][2], out[0][2], rtol=1e-3, atol=1e-5)      # hub attribute rows: thousands of terms in two orders
    for k in out[0][3]:
        np.testing.assert_allclose(out[1][3][k], out[0][3][k], rtol=1e-3, atol=1e-5, err_msg=k)
