"""Dataset folder -> DataModel -> PredicateAlignModel -> driver.run(): the reference's run_ITC.py / run_SSL.py flow
(code/run_ITC.py:14-21) on a folder in the reference's on-disk layout, end to end on the GPU."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _args(folder, word_file, **over):
    from multike_amd.synthetic import synthetic_args
    return synthetic_args(training_data=folder, output=folder + "out/", word2vec_path=word_file, dataset_division="631/",
                          encoder_epoch=3, encoder_active="tanh", encoder_normalize=True, retrain_literal_embeds=False,
                          literal_normalize=True, dim=16, batch_size=120, attribute_batch_size=100, entity_batch_size=64,
                          neg_triple_num=4, learning_rate=0.01, max_epoch=4, shared_learning_max_epoch=2, start_valid=2,
                          eval_freq=2, start_predicate_soft_alignment=1, truncated_freq=2, truncated_epsilon=0.9,
                          is_save=True, **over)


def test_data_model_tables(tmp_path):
    from multike_amd.data_model import LITERAL_EMBEDDINGS_FILE, DataModel
    from multike_amd.synthetic import write_dataset_folder
    folder = str(tmp_path) + "/"
    wf = write_dataset_folder(folder)
    args = _args(folder, wf)
    data = DataModel(args)
    kgs = data.kgs
    n = kgs.entities_num
    assert data.local_name_vectors.shape == (n, args.dim) and data.value_vectors.shape[1] == args.dim
    norms = np.linalg.norm(data.local_name_vectors, axis=1)
    assert np.all((np.abs(norms - 1) < 1e-6) | (norms == 0))
    # attribute triples are now (entity id, attribute id, value id) with ids inside the value table
    for kg in (kgs.kg1, kgs.kg2):
        assert kg.local_attribute_triples_num > 0
        vmax = max(v for _, _, v in kg.local_attribute_triples_list)
        assert vmax < data.value_vectors.shape[0]
        assert all(isinstance(v, int) for _, _, v in kg.sup_attribute_triples_list)
    # aligned entities carry (nearly) the same local name -> their name vectors are each other's nearest neighbours
    t1, t2 = np.array(kgs.test_entities1), np.array(kgs.test_entities2)
    sim = data.local_name_vectors[t1] @ data.local_name_vectors[t2].T
    assert (sim.argmax(1) == np.arange(len(t1))).mean() > 0.7
    # the cache is written and a second DataModel loads it instead of re-training
    assert os.path.exists(folder + LITERAL_EMBEDDINGS_FILE)
    again = DataModel(args)
    np.testing.assert_array_equal(again.local_name_vectors, data.local_name_vectors)
    np.testing.assert_array_equal(again.value_vectors, data.value_vectors)
    assert again.kgs.kg1.local_attribute_triples_list == kgs.kg1.local_attribute_triples_list


@pytest.mark.parametrize("method", ["ITC", "SSL"])
def test_run_from_folder(tmp_path, method):
    from multike_amd.run import main
    from multike_amd.synthetic import write_dataset_folder
    folder = str(tmp_path) + "/"
    wf = write_dataset_folder(folder, n_pairs=150, n_extra=10)
    args = _args(folder, wf)
    cfg = tmp_path / "args.json"
    cfg.write_text(json.dumps(vars(args)))
    res = main(["--method", method, "--training_data", folder.rstrip("/"), "--args", str(cfg), "--set", "max_epoch=4"])
    assert {"nv", "rv", "av", "final"} <= set(res) and all(np.isfinite(v) for v in res.values())
    assert res["nv"] > 0.5
    outs = []
    for root, _, files in os.walk(folder + "out/"):
        outs += files
    assert {"ent_embeds.npy", "rv_ent_embeds.npy", "rel_embeds.npy", "attr_embeds.npy", "kg1_ent_ids", "kg2_rel_ids"} <= set(outs)


def test_views_learn_alignment_from_shared_structure(tmp_path):
    """Functional check of the whole stack (readers -> DataModel -> sampler -> fused steps -> CNN -> common space ->
    evaluator): on two KGs that share 80 % of their structure, the relation and attribute views start at Hits@1 = 0 on
    the held-out test links and must learn to align them."""
    import contextlib
    import io
    from multike_amd.data_model import DataModel
    from multike_amd.MultiKE_CSL import MultiKE_CV
    from multike_amd.MultiKE_Late import test
    from multike_amd.predicate_alignment import PredicateAlignModel
    from multike_amd.synthetic import write_dataset_folder
    folder = str(tmp_path) + "/"
    wf = write_dataset_folder(folder, n_pairs=1500, n_extra=150, n_rel=40, n_attr=30, triples_per_entity=5.0, shared_structure=0.8)
    args = _args(folder, wf, ITC_learning_rate=0.01)
    args.dim, args.batch_size, args.attribute_batch_size, args.entity_batch_size, args.neg_triple_num = 64, 2000, 2000, 2000, 10
    args.encoder_epoch, args.truncated_freq, args.truncated_epsilon, args.start_predicate_soft_alignment = 5, 10, 0.98, 10
    with contextlib.redirect_stdout(io.StringIO()):
        data = DataModel(args)
        m = MultiKE_CV(data, args, PredicateAlignModel(data.kgs, args))
        m._prepare()
        before = {c: float(test(m, embed_choice=c)) for c in ("nv", "rv", "av")}
        for i in range(1, 31):
            m._train_views(i)
            m.train_common_space_learning_1epo(i, m._entity_list)
            m._refresh_neighbours(i)
        after = {c: float(test(m, embed_choice=c)) for c in ("nv", "rv", "av", "final")}
    assert before["rv"] < 0.02 and before["av"] < 0.02
    assert after["nv"] == before["nv"]                       # the name view is a constant table
    assert after["rv"] > 0.8 and after["av"] > 0.5 and after["final"] > 0.7, (before, after)


def test_predicate_refresh_in_hbm_equals_the_host_lists(tmp_path):
    """After DataModel (attribute values are ids: both predicate kinds take the array path), a soft predicate-alignment refresh
    with `PredicateAlignModel.device` set builds its eight lists on the GPU: same triples in the same order, weights = the float32
    of the host's float64, for relations AND attributes; the concatenations the drivers train on stay in HBM."""
    import torch
    from multike_amd.data_model import DataModel
    from multike_amd.predicate_alignment import PredicateAlignModel
    from multike_amd.synthetic import write_dataset_folder
    folder = str(tmp_path) + "/"
    wf = write_dataset_folder(folder)
    args = _args(folder, wf)
    data = DataModel(args)
    host, dev = PredicateAlignModel(data.kgs, args), PredicateAlignModel(data.kgs, args)
    dev.device = torch.device("cuda")
    rng = np.random.default_rng(2)
    rel = rng.standard_normal((data.kgs.relations_num, args.dim))
    attr = rng.standard_normal((data.kgs.attributes_num, args.dim))
    for p in (host, dev):
        p.update_predicate_alignment(rel)
        p.update_predicate_alignment(attr, predicate_type="attribute")
    seen = 0
    for kind in ("relation", "attribute"):
        for name in (f"sup_{kind}_alignment_triples1", f"sup_{kind}_alignment_triples2", f"{kind}_triples_w_weights1",
                     f"{kind}_triples_w_weights2"):
            a, b = getattr(host, name), getattr(dev, name)
            assert a.dev is None and b.dev is not None and len(a) == len(b)
            cols, w = b.dev
            assert cols[0].is_cuda and np.array_equal(torch.stack(cols, dim=1).cpu().numpy(), a.cols.astype(np.int32))
            assert np.array_equal(w.cpu().numpy(), a.w.astype(np.float32))
            seen += len(a)
    assert seen > 0
    both = dev.sup_attribute_alignment_triples1 + dev.sup_attribute_alignment_triples2
    want = host.sup_attribute_alignment_triples1 + host.sup_attribute_alignment_triples2
    assert both.dev is not None and np.array_equal(torch.stack(both.dev[0], dim=1).cpu().numpy(), want.cols.astype(np.int32))
    assert np.array_equal(both.cols, want.cols) and np.array_equal(both.w, want.w)


def test_deferred_predicate_refresh_is_seen_by_the_next_epochs_attribute_view(tmp_path):
    """code/MultiKE_CSL.py:80-87: the soft predicate-alignment refresh at the end of epoch i replaces
    `attribute_triples_w_weights1/2` (code/predicate_alignment.py:170-174), which epoch i + 1's ATTRIBUTE VIEW reads
    (code/MultiKE_model.py:325-328) — not only the two inference loops.  The product defers the refresh's host work under the
    next epoch's first kernels; the attribute view must still be enqueued AFTER it (round 4 enqueued it before: one epoch on
    stale weights).  Asserted: the list objects every phase of epoch i + 1 reads are the refreshed ones, and the deferred
    two-stream run gives the losses of the in-line one-stream run."""
    import contextlib
    import io
    import torch
    from multike_amd.data_model import DataModel
    from multike_amd.MultiKE_CSL import MultiKE_CV
    from multike_amd.predicate_alignment import PredicateAlignModel
    from multike_amd.synthetic import write_dataset_folder
    folder = str(tmp_path) + "/"
    wf = write_dataset_folder(folder, n_pairs=300, n_extra=20, n_rel=12, n_attr=10, shared_structure=0.7)
    args = _args(folder, wf)
    args.start_predicate_soft_alignment = 1

    def run(deferred):
        with contextlib.redirect_stdout(io.StringIO()):
            data = DataModel(args)
            m = MultiKE_CV(data, args, PredicateAlignModel(data.kgs, args))
            m.defer_predicate_update, m.overlap_views = deferred, deferred
            m._prepare()
            pam, seen, losses = m.predicate_align_model, [], []
            for name in ("train_attribute_view_1epo", "train_cross_kg_attribute_inference_1epo", "train_cross_kg_relation_inference_1epo"):
                orig = getattr(m, name)
                def wrapped(i, *a, _orig=orig, _name=name, **k):
                    seen.append((i, _name, pam.attribute_triples_w_weights1, m._ckga_attr_triples, m._ckgp_rel_triples))
                    out = _orig(i, *a, **k)
                    losses.append((i, _name, out))
                    return out
                setattr(m, name, wrapped)
            for i in range(1, 5):
                m._train_views(i)
                m.train_common_space_learning_1epo(i, m._entity_list)
                if i == 2:
                    before = (pam.attribute_triples_w_weights1, m._ckga_attr_triples, m._ckgp_rel_triples)
                    m._update_predicate_alignment()
            m._finish_predicate_update()
            torch.cuda.synchronize()
        return m, seen, losses, before
    m, seen, losses, before = run(True)
    pam = m.predicate_align_model
    assert pam.attribute_triples_w_weights1 is not before[0]               # the refresh did replace the lists
    for i, name, w1, ckga, ckgp in seen:
        if i <= 2:
            assert w1 is before[0] and ckga is before[1] and ckgp is before[2], (i, name)
        else:       # every list-reading phase of the epochs after the refresh — the attribute view included — reads the new lists
            assert w1 is pam.attribute_triples_w_weights1 and ckga is m._ckga_attr_triples and ckgp is m._ckgp_rel_triples, (i, name)
    _, _, ref_losses, _ = run(False)
    assert [(i, n) for i, n, _ in losses if n == "train_attribute_view_1epo"] == [(i, n) for i, n, _ in ref_losses if n == "train_attribute_view_1epo"]
    def val(v):
        with contextlib.redirect_stdout(io.StringIO()):
            return float(v.finish()) if hasattr(v, "finish") else float(v)
    got = {(i, n): val(v) for i, n, v in losses if v is not None}
    want = {(i, n): val(v) for i, n, v in ref_losses if v is not None}
    for k in want:
        assert abs(got[k] - want[k]) <= 2e-4 * abs(want[k]), (k, got[k], want[k])
