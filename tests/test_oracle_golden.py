"""The oracle against the golden vectors produced by executing the reference's own losses.py /
base/batch.py / attr_batch.py / utils.task_divide (tests/golden/make_golden.py).  CPU only."""
import random

import numpy as np
import pytest

from oracle import multike_oracle as mo
from oracle import sampler_oracle as so

from conftest import golden_cases

N_CASES = 5


def _case(g, ci):
    pre = f"c{ci}_"
    c = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
    return c


@pytest.mark.parametrize("ci", range(N_CASES))
def test_case_count(losses_golden, ci):
    assert len(golden_cases(losses_golden)) == N_CASES


@pytest.mark.parametrize("ci", range(N_CASES))
@pytest.mark.parametrize("dt,rtol", [(np.float64, 1e-12), (np.float32, 2e-6)])
def test_losses_py_surface(losses_golden, ci, dt, rtol):
    c = _case(losses_golden, ci)
    tag = "f64" if dt == np.float64 else "f32"
    E = mo.l2_normalize_rows(c["ent"].astype(dt))
    R = mo.l2_normalize_rows(c["rel"].astype(dt))
    rows = [E[c["ph"]], R[c["pr"]], E[c["pt"]], E[c["nh"]], R[c["nr"]], E[c["nt"]]]
    pw, nw = c["pw"].astype(dt), c["nw"].astype(dt)
    np.testing.assert_allclose(mo.relation_logistic_loss(*rows), c["a1_loss_" + tag], rtol=rtol)
    np.testing.assert_allclose(mo.relation_logistic_loss_wo_negs(*rows[:3]), c["a2_loss_" + tag], rtol=rtol)
    np.testing.assert_allclose(mo.attribute_logistic_loss_wo_negs(*rows[:3]), c["a2b_loss_" + tag], rtol=rtol)
    np.testing.assert_allclose(mo.logistic_loss_wo_negs(*rows[:3], pw), c["a3_loss_" + tag], rtol=rtol)
    np.testing.assert_allclose(mo.attribute_logistic_loss(*rows[:3], pw, *rows[3:], nw), c["a4_loss_" + tag], rtol=rtol)
    np.testing.assert_allclose(mo.alignment_loss(rows[0], rows[2]), c["a5_loss_" + tag], rtol=rtol)
    M = c["a6_M"].astype(dt)
    eye = np.eye(M.shape[0], dtype=dt)
    np.testing.assert_allclose(mo.space_mapping_loss(rows[0], rows[2], M, eye, 2.0), c["a6_loss_" + tag], rtol=max(rtol, 2e-5 if dt == np.float32 else 0))
    np.testing.assert_allclose(mo.orthogonal_loss(M, eye), c["a6o_loss_" + tag], rtol=max(rtol, 2e-5 if dt == np.float32 else 0))


@pytest.mark.parametrize("ci", range(N_CASES))
def test_gathered_row_gradients(losses_golden, ci):
    c = _case(losses_golden, ci)
    E = mo.l2_normalize_rows(c["ent"].astype(np.float64))
    R = mo.l2_normalize_rows(c["rel"].astype(np.float64))
    _, gh, gr, gt = mo.logistic_term_grads(E[c["ph"]], R[c["pr"]], E[c["pt"]], +1.0)
    for mine, name in ((gh, "gph"), (gr, "gpr"), (gt, "gpt")):
        np.testing.assert_allclose(mine, c["a1_" + name], rtol=1e-10, atol=1e-14)
    _, gh, gr, gt = mo.logistic_term_grads(E[c["nh"]], R[c["nr"]], E[c["nt"]], -1.0)
    for mine, name in ((gh, "gnh"), (gr, "gnr"), (gt, "gnt")):
        np.testing.assert_allclose(mine, c["a1_" + name], rtol=1e-10, atol=1e-14)
    # weighted variants (a3: positives, a4: both signs)
    pw, nw = c["pw"].astype(np.float64), c["nw"].astype(np.float64)
    _, gh, gr, gt = mo.logistic_term_grads(E[c["ph"]], R[c["pr"]], E[c["pt"]], +1.0, pw)
    for mine, k in ((gh, 0), (gr, 1), (gt, 2)):
        np.testing.assert_allclose(mine, c[f"a3_g{k}"], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(mine, c[f"a4_g{k}"], rtol=1e-10, atol=1e-14)
    _, gh, gr, gt = mo.logistic_term_grads(E[c["nh"]], R[c["nr"]], E[c["nt"]], -1.0, nw)
    for mine, k in ((gh, 4), (gr, 5), (gt, 6)):
        np.testing.assert_allclose(mine, c[f"a4_g{k}"], rtol=1e-10, atol=1e-14)
    d = E[c["ph"]] - E[c["pt"]]
    np.testing.assert_allclose(2 * d, c["a5_g0"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(-2 * d, c["a5_g1"], rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("ci", range(N_CASES))
def test_raw_table_gradient_through_normalise(losses_golden, ci):
    """Jacobian of normalise-on-read with duplicate rows — torch autograd on the reference's loss is the
    expected value (TF itself is not available: unpinned at the TF boundary)."""
    c = _case(losses_golden, ci)
    ent, rel = c["ent"].astype(np.float64), c["rel"].astype(np.float64)
    _, ghat_e, ghat_r = mo.relation_view_step_dense(ent, rel, None, None, (c["ph"], c["pr"], c["pt"]),
                                                    (c["nh"], c["nr"], c["nt"]), 0.0, update=False)
    np.testing.assert_allclose(mo.l2_normalize_rows_backward(ent, ghat_e), c["a1_gent_raw"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(mo.l2_normalize_rows_backward(rel, ghat_r), c["a1_grel_raw"], rtol=1e-9, atol=1e-12)
    # untouched rows have exactly zero gradient (dense Adagrad == sparse Adagrad)
    touched = np.zeros(len(ent), bool)
    for k in ("ph", "pt", "nh", "nt"):
        touched[c[k]] = True
    assert np.all(c["a1_gent_raw"][~touched] == 0)


@pytest.mark.parametrize("ci", range(N_CASES))
def test_three_adagrad_steps(losses_golden, ci):
    c = _case(losses_golden, ci)
    ent, rel = c["ent"].astype(np.float64), c["rel"].astype(np.float64)
    acc_e, acc_r = np.full_like(ent, mo.ADAGRAD_INIT_ACC), np.full_like(rel, mo.ADAGRAD_INIT_ACC)
    losses = []
    for step in range(3):
        L, _, _ = mo.relation_view_step_dense(ent, rel, acc_e, acc_r, (c["ph"], c["pr"], c["pt"]),
                                              (c["nh"], c["nr"], c["nt"]), 0.001)
        losses.append(L)
        if step in (0, 2):
            np.testing.assert_allclose(ent, c[f"a1_ent_after{step + 1}"], rtol=1e-10, atol=1e-14)
            np.testing.assert_allclose(rel, c[f"a1_rel_after{step + 1}"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(losses, c["a1_step_losses_f64"], rtol=1e-12)


def test_sparse_update_equals_dense(losses_golden):
    c = _case(losses_golden, 3)
    ent, rel = c["ent"].astype(np.float64), c["rel"].astype(np.float64)
    _, ghat_e, _ = mo.relation_view_step_dense(ent, rel, None, None, (c["ph"], c["pr"], c["pt"]),
                                               (c["nh"], c["nr"], c["nt"]), 0.0, update=False)
    e_dense, a_dense = ent.copy(), np.full_like(ent, 0.1)
    mo.adagrad_dense(e_dense, a_dense, mo.l2_normalize_rows_backward(ent, ghat_e), 0.001)
    e_sp, a_sp = ent.copy(), np.full_like(ent, 0.1)
    mo.rows_update_sparse(e_sp, a_sp, ghat_e, 0.001)
    assert np.array_equal(e_dense, e_sp) and np.array_equal(a_dense, a_sp)


# ------------------------------- sampler / batch generator ------------------------------------
def _tuples(x):
    return [tuple(t) for t in x]


@pytest.mark.parametrize("run_i", range(4))
def test_mt_restatement_replays_reference_batches(sampler_golden, run_i):
    g = sampler_golden
    run = g["relation_runs"][run_i]
    t1, t2 = _tuples(g["triples1"]), _tuples(g["triples2"])
    k1, k2 = set(_tuples(g["known1"])), set(_tuples(g["known2"]))
    near1 = {int(k): v for k, v in g["near1"].items()} if run["use_near"] else None
    near2 = {int(k): v for k, v in g["near2"].items()} if run["use_near"] else None
    random.seed(run["seed"])
    np.random.seed(run["seed"])
    for step, exp in enumerate(run["steps"]):
        pos, neg = so.mt_relation_batch(t1, t2, k1, k2, g["ents1"], g["ents2"], run["batch_size"], step, near1, near2,
                                        run["neg"])
        assert pos == _tuples(exp["pos"])
        assert neg == _tuples(exp["neg"])


def test_reference_sampler_invariants(sampler_golden):
    """SURVEY §8a-S2 invariants hold on the reference's own output (so they are fair to demand of ours)."""
    g = sampler_golden
    e1 = set(g["ents1"])
    for run in g["relation_runs"]:
        N = run["neg"]
        for st in run["steps"]:
            pos, neg = st["pos"], st["neg"]
            assert len(neg) == N * len(pos)
            for i, (h, r, t) in enumerate(pos):
                for (a, b, c) in neg[i * N:(i + 1) * N]:
                    assert b == r and ((a == h) or (c == t))
                    assert (a in e1) == (h in e1) and (c in e1) == (t in e1)


def test_attribute_batches(sampler_golden):
    g = sampler_golden
    a1, a2 = _tuples(g["attr1"]), _tuples(g["attr2"])
    bs = g["attribute_run"]["batch_size"]
    b1, b2 = mo.kg_batch_split(len(a1), len(a2), bs)
    for step, exp in enumerate(g["attribute_run"]["steps"]):
        pos = so.mt_epoch_slice(a1, b1, step) + so.mt_epoch_slice(a2, b2, step)
        assert pos == _tuples(exp["pos"]) and exp["neg"] == []


def test_task_divide_and_split(sampler_golden):
    for row in sampler_golden["task_divide"]:
        assert [list(x) for x in mo.task_divide(list(range(row["total"])), row["n"])] == row["tasks"]
    for row in sampler_golden["kg_batch_split"]:
        assert mo.kg_batch_split(row["n1"], row["n2"], row["batch"]) == (row["b1"], row["b2"])


def test_dense_optimizer_oracles_agree_with_torch_optim():
    """adam_dense / adadelta_dense restate TF1's ApplyAdam / ApplyAdadelta.  torch.optim.Adadelta is the same rule; torch's
    Adam differs only in where epsilon sits (eps vs eps * sqrt(1 - b2^t)), invisible at |g| >> eps -- an independent
    implementation to pin the restatement against (TensorFlow itself is not installable here)."""
    import torch
    from oracle import multike_oracle as mo
    rng = np.random.default_rng(0)
    w0 = rng.standard_normal(50)
    for kind in ("Adam", "Adadelta"):
        p = torch.tensor(w0, dtype=torch.float64, requires_grad=True)
        opt = (torch.optim.Adam([p], lr=0.01, betas=(0.9, 0.999), eps=1e-8) if kind == "Adam"
               else torch.optim.Adadelta([p], lr=0.5, rho=0.95, eps=1e-8))
        W, s1, s2 = w0.copy(), np.zeros(50), np.zeros(50)
        for step in range(1, 6):
            g = rng.standard_normal(50) + 0.5
            p.grad = torch.tensor(g)
            opt.step()
            (mo.adam_dense(W, s1, s2, g, 0.01, step) if kind == "Adam" else mo.adadelta_dense(W, s1, s2, g, 0.5))
        np.testing.assert_allclose(W, p.detach().numpy(), rtol=(1e-5 if kind == "Adam" else 1e-12), atol=(1e-6 if kind == "Adam" else 0))


def test_space_mapping_gradients_against_autograd():
    """oracle.space_mapping_grads (closed form) vs torch autograd on the loss restatement that is pinned to the reference's
    code/losses.py:53-63 by the golden fixture."""
    import torch
    from oracle import multike_oracle as mo
    rng = np.random.default_rng(1)
    B, d, ow = 40, 9, 2.0
    V, F, M = rng.standard_normal((B, d)), rng.standard_normal((B, d)) * 0.1, rng.standard_normal((d, d)) * 0.4
    loss, gF, gM = mo.space_mapping_grads(V, F, M, ow)
    np.testing.assert_allclose(loss, mo.space_mapping_loss(V, F, M, np.eye(d), ow), rtol=1e-13)
    tV, tF, tM = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (V, F, M))
    P = tV @ tM
    P = P / torch.sqrt(torch.clamp_min((P * P).sum(), 1e-12))
    L = ((tF - P) ** 2).sum() + ow * ((tM @ tM.T - torch.eye(d, dtype=torch.float64)) ** 2).sum() + 1e-4 * (tM * tM).sum()
    L.backward()
    np.testing.assert_allclose(loss, float(L.detach()), rtol=1e-13)
    np.testing.assert_allclose(gF, tF.grad.numpy(), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(gM, tM.grad.numpy(), rtol=1e-10, atol=1e-12)
