"""Edge cases of the native multi-step entry points (empty steps, zero steps, a single view, constant tables) and their error
reporting: they must behave like the corresponding sequence of single steps / raise through mke_last_error."""
import numpy as np
import pytest
import torch

from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu


def _tables(n=500, d=20, seed=0):
    from multike_amd.tables import EmbeddingTable
    rng = np.random.default_rng(seed)
    mk = lambda name, norm=True, train=True: EmbeddingTable(n, d, name, norm, trainable=train, values=mo.xavier_truncated_normal((n, d), rng))
    return mk("ent"), mk("name", False, False), mk("rv"), mk("av")


def test_alignment_steps_with_empty_and_ragged_steps_equal_single_steps():
    from multike_amd.runner import run_alignment_steps
    from multike_amd.tables import StepEngine
    ent, name, rv, av = _tables()
    ent2, name2, rv2, av2 = _tables()
    rng = np.random.default_rng(1)
    idx = torch.as_tensor(rng.integers(0, 500, size=230).astype(np.int32), device="cuda")
    off = np.array([0, 100, 100, 230, 230])                     # steps of 100, 0, 130, 0 rows
    terms = [(0, 1, 0.5), (0, 2, 1.0), (0, 3, 1.0)]
    ring = run_alignment_steps([ent, name, rv, av], terms, idx, idx, off, "cs", 1, 0.01)
    eng = StepEngine()
    tot = []
    for s in range(4):
        sl = idx[off[s]:off[s + 1]]
        t = [(ent2, sl, name2, sl, 0.5), (ent2, sl, rv2, sl, 1.0), (ent2, sl, av2, sl, 1.0)]
        tot.append(float(eng.alignment_step(t, "cs", 0.01)))
    np.testing.assert_allclose(ring.sum(dim=(1, 2)).cpu().numpy(), tot, rtol=1e-6, atol=1e-12)
    for a, b in ((ent, ent2), (rv, rv2), (av, av2), (name, name2)):
        np.testing.assert_allclose(a.raw().cpu().numpy(), b.raw().cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert run_alignment_steps([ent, name, rv, av], terms, idx, idx, np.array([0]), "cs", 100, 0.01).shape[0] == 0


def test_mapping_step_single_view_and_empty_batch():
    from multike_amd import _lib
    from multike_amd.runner import SpaceMappingState, run_space_mapping_steps
    ent, name, rv, av = _tables(seed=2)
    rng = np.random.default_rng(3)
    M0 = np.eye(20) + 0.1 * rng.standard_normal((20, 20))
    st = SpaceMappingState([torch.as_tensor(M0, dtype=torch.float32)], "cuda")
    idx = np.arange(0, 400, 2, dtype=np.int32)
    ring = run_space_mapping_steps(st, ent, [rv], torch.as_tensor(idx, device="cuda"), np.array([0, 200, 200]), "m", 1, 0.01, 2.0)
    E = ent.raw().cpu().numpy()                                   # after the update; recompute the loss from the start state
    e0, _, r0, _ = (t.raw().cpu().numpy().astype(np.float64) for t in _tables(seed=2))
    M64 = M0.astype(np.float32).astype(np.float64)
    L = mo.space_mapping_step_dense(e0, np.full_like(e0, 0.1), [(r0, True)], [M64], [np.full_like(M64, 0.1)], idx, 0.01, 2.0)
    got = ring.sum(dim=(1, 2)).cpu().numpy()
    np.testing.assert_allclose(got[0], L, rtol=2e-5)
    assert got[1] == 0.0                                          # empty step: no loss, nothing moves
    np.testing.assert_allclose(E, e0, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(st.M[0].cpu().numpy(), M64, rtol=2e-4, atol=2e-6)
    big = SpaceMappingState([torch.eye(100)], "cuda")
    from multike_amd.tables import EmbeddingTable
    e100 = EmbeddingTable(50, 100, "e")
    with pytest.raises(_lib.MultiKEHipError, match="dim <= 88"):
        run_space_mapping_steps(big, e100, [e100], torch.arange(10, dtype=torch.int32, device="cuda"), np.array([0, 10]), "m", 1, 0.01, 2.0)


def test_attr_steps_equal_single_steps_with_an_empty_step():
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.tables import EmbeddingTable, StepEngine
    rng = np.random.default_rng(4)
    d, n = 20, 300

    def setup():
        r = np.random.default_rng(5)
        E = EmbeddingTable(200, d, "av", values=mo.xavier_truncated_normal((200, d), r))
        A = EmbeddingTable(30, d, "attr", False, values=mo.xavier_truncated_normal((30, d), r))
        lit = r.standard_normal((80, d)).astype(np.float32)
        L = EmbeddingTable(80, d, "lit", False, trainable=False, values=lit / np.linalg.norm(lit, axis=1, keepdims=True))
        return E, A, L, AttrCNN(d, seed=6), StepEngine()
    ih = torch.as_tensor(rng.integers(0, 200, n).astype(np.int32), device="cuda")
    ia = torch.as_tensor(rng.integers(0, 30, n).astype(np.int32), device="cuda")
    iv = torch.as_tensor(rng.integers(0, 80, n).astype(np.int32), device="cuda")
    w = torch.as_tensor(rng.random(n).astype(np.float32), device="cuda")
    off = np.array([0, 120, 120, 300])
    E1, A1, L1, c1, e1 = setup()
    ring = c1.steps(e1, E1, A1, L1, ih, ia, iv, w, off, lr=0.01)
    E2, A2, L2, c2, e2 = setup()
    single = []
    for s in range(3):
        sl = slice(int(off[s]), int(off[s + 1]))
        single.append(float(c2.step(e2, E2, A2, L2, ih[sl], ia[sl], iv[sl], w[sl], lr=0.01).sum()) if off[s + 1] > off[s] else 0.0)
    np.testing.assert_allclose(ring.sum(dim=1).cpu().numpy(), single, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(E1.raw().cpu().numpy(), E2.raw().cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(c1.params.cpu().numpy(), c2.params.cpu().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_positive_only_steps_use_the_hub_copies_and_stay_the_same_function():
    """The cross-KG loops' positives-only steps (code/MultiKE_model.py:349-369) on Zipf head / tail entities: with hub rows declared
    on the entity table every triple's head / tail gradient of a hub goes to a private copy (mke_relation_plan.hot); the same
    tables as without the declaration, every copy back at zero."""
    import numpy as np
    import torch
    from multike_amd.runner import run_positive_steps
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import EmbeddingTable
    kgs = SyntheticKGs(n_ent=3000, n_rel=20, seed=3, zipf=1.2)
    tr = np.concatenate(kgs.triples)
    B = 2000
    steps = 3            # few sequential steps: two float32 runs that sum a hub's terms in different orders drift apart step by step
                         # (27 steps of 500 at lr 0.01 left single elements 1-3e-5 = 1-2 % apart in 4 runs of 5)
    cols = tuple(torch.as_tensor(np.ascontiguousarray(tr[:steps * B, k]), device="cuda") for k in range(3))
    w = torch.rand(steps * B, device="cuda")
    off = np.arange(steps + 1, dtype=np.int64) * B
    out = []
    for hubs in (False, True):
        E, R = EmbeddingTable(3000, 75, "e", seed=1), EmbeddingTable(20, 75, "r", seed=2, grad_copies=4)
        if hubs:
            deg = np.bincount(tr[:, [0, 2]].reshape(-1), minlength=3000) / steps
            E.set_hot_rows(np.nonzero(deg >= 4)[0], 8)
            assert E.n_hot >= 5
        loss = run_positive_steps(E, R, "ckgp", cols, w, off, 1, 0.01, scale=2.0)
        torch.cuda.synchronize()
        assert float(E._grad_full.abs().max()) == 0.0 and float(R.grad.abs().max()) == 0.0
        out.append((loss.sum(1).cpu().numpy(), E.raw().cpu().numpy(), R.raw().cpu().numpy()))
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=2e-6)
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(out[1][2], out[0][2], rtol=2e-4, atol=1e-5)
