"""Adam / Adadelta (selectable through args.optimizer, code/MultiKE_model.py:15-25): the whole-variable HIP kernels vs the
float64 oracle (oracle/multike_oracle.py:adam_dense / adadelta_dense, themselves pinned to torch.optim in
tests/test_oracle_golden.py), and the model loops running on them."""
import numpy as np
import pytest
import torch

from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["Adam", "Adadelta"])
@pytest.mark.parametrize("normalize", [True, False])
def test_rows_update_dense_matches_oracle(kind, normalize):
    from multike_amd import _lib
    from multike_amd.tables import EmbeddingTable
    rng = np.random.default_rng(3)
    n, d, lr = 300, 75, 0.01
    T = EmbeddingTable(n, d, normalize=normalize, values=rng.standard_normal((n, d)) * 0.3)
    W = T.raw().cpu().numpy().astype(np.float64)
    s1, s2 = np.zeros_like(W), np.zeros_like(W)
    for step in range(1, 4):
        g = rng.standard_normal((n, d)) * 0.1
        g[rng.random(n) < 0.6] = 0.0                          # most rows untouched: they must still move (Adam) / decay
        T.grad[:, :d] = torch.as_tensor(g, dtype=torch.float32, device="cuda")
        a, b, t = T.dense_slots("opt")
        assert t == step
        _lib.rows_update_dense(T.data, a, b, T.grad, d, normalize, _lib.optimizer_struct(kind, lr, t))
        graw = mo.l2_normalize_rows_backward(W, g) if normalize else g
        if kind == "Adam":
            mo.adam_dense(W, s1, s2, graw, lr, step)
        else:
            mo.adadelta_dense(W, s1, s2, graw, lr)
        assert float(T.grad.abs().max()) == 0.0
    np.testing.assert_allclose(T.raw().cpu().numpy(), W, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(a[:, :d].cpu().numpy(), s1, rtol=5e-5, atol=2e-8)
    np.testing.assert_allclose(b[:, :d].cpu().numpy(), s2, rtol=2e-4, atol=1e-10)
    if T.stride > d:
        assert float(T.data[:, d:].abs().max()) == 0.0 and float(a[:, d:].abs().max()) == 0.0


@pytest.mark.parametrize("kind", ["Adam", "Adadelta"])
def test_dense_update_opt_matches_oracle(kind):
    from multike_amd import _lib
    rng = np.random.default_rng(4)
    n, lr = 10_007, 0.003
    w0 = rng.standard_normal(n)
    w = torch.as_tensor(w0, dtype=torch.float32, device="cuda")
    a, b, g = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(w)
    W, s1, s2 = w0.astype(np.float32).astype(np.float64), np.zeros(n), np.zeros(n)
    for step in range(1, 5):
        gg = rng.standard_normal(n) * (rng.random(n) < 0.5)
        g.copy_(torch.as_tensor(gg, dtype=torch.float32))
        _lib.dense_update_opt(w, a, b, g, _lib.optimizer_struct(kind, lr, step))
        gq = gg.astype(np.float32).astype(np.float64)
        (mo.adam_dense(W, s1, s2, gq, lr, step) if kind == "Adam" else mo.adadelta_dense(W, s1, s2, gq, lr))
    np.testing.assert_allclose(w.cpu().numpy(), W, rtol=2e-5, atol=2e-7)
    assert float(g.abs().max()) == 0.0
    with pytest.raises(_lib.MultiKEHipError, match="step >= 1"):
        _lib.dense_update_opt(w, a, b, g, _lib.optimizer_struct("Adam", lr, 0))


@pytest.mark.parametrize("kind", ["Adam", "Adadelta"])
def test_model_loops_run_on_dense_optimizers(kind):
    """args.optimizer = Adam / Adadelta: every loop falls back to its step-wise form on the whole-variable kernels; the
    first relation-view step is checked against the oracle, and Adam must reduce the losses."""
    from multike_amd.MultiKE_model import MultiKE
    from multike_amd.synthetic import SyntheticData, synthetic_args
    data = SyntheticData(dim=20)
    lr = 0.01 if kind == "Adam" else 1.0
    args = synthetic_args(dim=20, batch_size=700, attribute_batch_size=600, entity_batch_size=800, neg_triple_num=5,
                          learning_rate=lr, optimizer=kind)
    m = MultiKE(data, args, data.predicate_align_model)
    for f in ("_define_variables", "_define_name_view_graph", "_define_relation_view_graph", "_define_attribute_view_graph",
              "_define_cross_kg_entity_reference_relation_view_graph", "_define_cross_kg_entity_reference_attribute_view_graph",
              "_define_cross_kg_attribute_reference_graph", "_define_cross_kg_relation_reference_graph",
              "_define_common_space_learning_graph", "_define_space_mapping_graph"):
        getattr(m, f)()
    kgs, pam = data.kgs, data.predicate_align_model
    rel_steps = m._rel_batcher.steps
    attr_steps = int(np.ceil((kgs.kg1.local_attribute_triples_num + kgs.kg2.local_attribute_triples_num) / args.attribute_batch_size))
    ents = kgs.kg1.entities_list + kgs.kg2.entities_list
    hist = []
    for i in range(1, 4):
        hist.append((m.train_relation_view_1epo(i, rel_steps, None, None, None, None),
                     m.train_cross_kg_entity_inference_relation_view_1epo(i, kgs.kg1.sup_relation_triples_list + kgs.kg2.sup_relation_triples_list),
                     m.train_cross_kg_relation_inference_1epo(i, pam.sup_relation_alignment_triples1 + pam.sup_relation_alignment_triples2),
                     m.train_attribute_view_1epo(i, attr_steps, None, None, None, None),
                     m.train_cross_kg_entity_inference_attribute_view_1epo(i, kgs.kg1.sup_attribute_triples_list + kgs.kg2.sup_attribute_triples_list),
                     m.train_common_space_learning_1epo(i, ents)))
    sm = m.train_shared_space_mapping_1epo(1, ents)
    h = np.array(hist)
    assert np.all(np.isfinite(h)) and np.isfinite(sm)
    assert "relation/dense" in m.rv_ent_embeds.slots and "relation" not in m.rv_ent_embeds.slots
    if kind == "Adam":
        assert np.all(h[-1] < h[0])
