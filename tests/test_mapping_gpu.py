"""The native space-mapping step (mke_mapping_step / _steps: SSL driver, code/MultiKE_model.py:241-261,:439-454) against
the float64 oracle (oracle.space_mapping_step_dense) over several steps, and the model loop on top of it."""
import numpy as np
import pytest
import torch

from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu


def _orth(rng, d):
    q, r = np.linalg.qr(rng.standard_normal((d, d)))
    return q * np.sign(np.diag(r))


@pytest.mark.parametrize("d,n_ent,B", [(75, 3000, 700), (20, 400, 400)])
def test_native_steps_match_oracle(d, n_ent, B):
    from multike_amd.runner import SpaceMappingState, run_space_mapping_steps
    from multike_amd.tables import EmbeddingTable
    rng = np.random.default_rng(5)
    ent0 = mo.xavier_truncated_normal((n_ent, d), rng)
    views0 = [mo.xavier_truncated_normal((n_ent, d), rng) for _ in range(3)]
    lit = views0[0] / np.linalg.norm(views0[0], axis=1, keepdims=True)
    ent = EmbeddingTable(n_ent, d, "ent_embeds", True, values=ent0)
    name = EmbeddingTable(n_ent, d, "name", False, trainable=False, values=lit)
    rv = EmbeddingTable(n_ent, d, "rv", True, values=views0[1])
    av = EmbeddingTable(n_ent, d, "av", True, values=views0[2])
    Ms = [_orth(rng, d) + 0.05 * rng.standard_normal((d, d)) for _ in range(3)]
    st = SpaceMappingState([torch.as_tensor(m, dtype=torch.float32) for m in Ms], "cuda")
    steps = 3
    idx = np.stack([rng.choice(n_ent, size=B, replace=False) for _ in range(steps)]).astype(np.int32)
    ring = run_space_mapping_steps(st, ent, [name, rv, av], torch.as_tensor(idx.reshape(-1), device="cuda"),
                                   np.arange(steps + 1) * B, "shared_comb", 1, 0.01, 2.0)
    got = ring.sum(dim=(1, 2)).cpu().numpy()
    E = ent0.astype(np.float32).astype(np.float64)
    accE = np.full_like(E, 0.1)
    tabs = [(lit.astype(np.float32).astype(np.float64), False), (views0[1].astype(np.float32).astype(np.float64), True),
            (views0[2].astype(np.float32).astype(np.float64), True)]
    M64 = [m.astype(np.float32).astype(np.float64) for m in Ms]
    accM = [np.full_like(m, 0.1) for m in M64]
    for s in range(steps):
        L = mo.space_mapping_step_dense(E, accE, tabs, M64, accM, idx[s], 0.01, 2.0)
        np.testing.assert_allclose(got[s], L, rtol=2e-5)
    np.testing.assert_allclose(ent.raw().cpu().numpy(), E, rtol=2e-4, atol=2e-6)
    for k in range(3):
        np.testing.assert_allclose(st.M[k].cpu().numpy(), M64[k], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(st.accM[k].cpu().numpy(), accM[k], rtol=1e-3, atol=1e-6)
    assert float(st.gM.abs().max()) == 0.0 and float(ent.grad.abs().max()) == 0.0       # gradients consumed
    np.testing.assert_array_equal(rv.raw().cpu().numpy(), views0[1].astype(np.float32))   # views are constants here


def test_model_loop_native_equals_stepwise_torch_path():
    """train_shared_space_mapping_1epo: native epoch vs the step-wise autograd path on the same batches."""
    import contextlib
    import io
    from multike_amd.MultiKE_model import MultiKE
    from multike_amd.synthetic import SyntheticData, synthetic_args

    def model():
        data = SyntheticData(dim=20)
        args = synthetic_args(dim=20, batch_size=700, attribute_batch_size=600, entity_batch_size=800, neg_triple_num=5, learning_rate=0.01)
        m = MultiKE(data, args, data.predicate_align_model)
        m._define_variables(); m._define_name_view_graph(); m._define_space_mapping_graph()
        return m, data.kgs.kg1.entities_list + data.kgs.kg2.entities_list
    a, ents = model()
    b, _ = model()
    b.args.dim = 20
    with contextlib.redirect_stdout(io.StringIO()):
        la = [a.train_shared_space_mapping_1epo(i, ents) for i in (1, 2)]
        import multike_amd.MultiKE_model as mm
        saved = mm._DENSE_OPTS
        mm._DENSE_OPTS = saved + ("Adagrad",)        # force the step-wise torch path for the same optimizer
        try:
            lb = [b.train_shared_space_mapping_1epo(i, ents) for i in (1, 2)]
        finally:
            mm._DENSE_OPTS = saved
    np.testing.assert_allclose(la, lb, rtol=2e-5)
    np.testing.assert_allclose(a.ent_embeds.raw().cpu().numpy(), b.ent_embeds.raw().cpu().numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(a.rv_mapping.cpu().numpy(), b.rv_mapping.detach().cpu().numpy(), rtol=2e-4, atol=2e-6)
