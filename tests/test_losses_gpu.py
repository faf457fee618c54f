"""The losses.py op surface (multike_amd/losses.py) against the golden vectors made from the reference's own
code/losses.py — same function names, same positional arguments, torch CUDA tensors in, 0-d tensor out,
differentiable."""
import numpy as np
import pytest
import torch

from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu
RT = 3e-6


def _case(g, ci):
    pre = f"c{ci}_"
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


def _rows(c):
    E = mo.l2_normalize_rows(c["ent"].astype(np.float64))
    R = mo.l2_normalize_rows(c["rel"].astype(np.float64))
    rows = [E[c["ph"]], R[c["pr"]], E[c["pt"]], E[c["nh"]], R[c["nr"]], E[c["nt"]]]
    return [torch.tensor(r, dtype=torch.float32, device="cuda", requires_grad=True) for r in rows]


@pytest.mark.parametrize("ci", range(5))
def test_all_eight_functions(losses_golden, ci):
    from multike_amd import losses as L
    c = _case(losses_golden, ci)
    rows = _rows(c)
    pw = torch.tensor(c["pw"], device="cuda")
    nw = torch.tensor(c["nw"], device="cuda")

    def check(val, key, grads=None):
        assert val.dim() == 0 and val.dtype == torch.float32
        np.testing.assert_allclose(val.item(), c[key + "_loss_f64"], rtol=RT)
        if grads:
            for r_ in rows:
                r_.grad = None
            val.backward()
            for r_, gk in grads:
                np.testing.assert_allclose(r_.grad.cpu().numpy(), c[gk], rtol=1e-4, atol=2e-6)

    check(L.relation_logistic_loss(*rows), "a1", list(zip(rows, ["a1_gph", "a1_gpr", "a1_gpt", "a1_gnh", "a1_gnr", "a1_gnt"])))
    check(L.relation_logistic_loss_wo_negs(*rows[:3]), "a2", list(zip(rows[:3], ["a2_g0", "a2_g1", "a2_g2"])))
    check(L.attribute_logistic_loss_wo_negs(*rows[:3]), "a2b")
    check(L.logistic_loss_wo_negs(*rows[:3], pw), "a3", list(zip(rows[:3], ["a3_g0", "a3_g1", "a3_g2"])))
    check(L.attribute_logistic_loss(*rows[:3], pw, *rows[3:], nw), "a4",
          list(zip(rows, ["a4_g0", "a4_g1", "a4_g2", "a4_g4", "a4_g5", "a4_g6"])))
    check(L.alignment_loss(rows[0], rows[2]), "a5", [(rows[0], "a5_g0"), (rows[2], "a5_g1")])
    M = torch.tensor(c["a6_M"], device="cuda", requires_grad=True)
    eye = torch.eye(M.shape[0], device="cuda")
    v = L.space_mapping_loss(rows[0], rows[2], M, eye, 2.0)
    np.testing.assert_allclose(v.item(), c["a6_loss_f64"], rtol=2e-5)
    for r_ in rows:
        r_.grad = None
    v.backward()
    np.testing.assert_allclose(rows[0].grad.cpu().numpy(), c["a6_g0"], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(M.grad.cpu().numpy(), c["a6_g2"], rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(L.orthogonal_loss(M, eye).item(), c["a6o_loss_f64"], rtol=2e-5)


def test_same_row_used_twice_and_scaling():
    """Gradient accumulates when one tensor is passed as both h and t, and scales with the upstream factor
    (the reference multiplies these losses by 2, code/MultiKE_model.py:168)."""
    from multike_amd import losses as L
    torch.manual_seed(0)
    x = torch.nn.functional.normalize(torch.randn(33, 75, device="cuda"), dim=1).requires_grad_(True)
    r = torch.nn.functional.normalize(torch.randn(33, 75, device="cuda"), dim=1).requires_grad_(True)
    (2 * L.relation_logistic_loss_wo_negs(x, r, x)).backward()
    xr = x.detach().double().cpu().numpy()
    rr = r.detach().double().cpu().numpy()
    _, gh, gr, gt = mo.logistic_term_grads(xr, rr, xr, 1.0)
    np.testing.assert_allclose(x.grad.cpu().numpy(), 2 * (gh + gt), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r.grad.cpu().numpy(), 2 * gr, rtol=1e-4, atol=1e-6)


def test_rejects_cpu_tensors_and_bad_shapes():
    from multike_amd import _lib
    from multike_amd import losses as L
    a = torch.zeros(4, 75)
    with pytest.raises(_lib.MultiKEHipError):
        L.alignment_loss(a, a)
    b = torch.zeros(4, 75, device="cuda")
    with pytest.raises(_lib.MultiKEHipError):
        L.relation_logistic_loss_wo_negs(b, b, torch.zeros(5, 75, device="cuda"))
