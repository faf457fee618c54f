"""On-disk inputs: readers, id assignment, KG containers, literal clean-up and predicate alignment
against tests/golden/data_golden.json, which tests/golden/make_golden.py produced by running the reference's own
code/base/{read,kg,kgs}.py, code/utils.py and code/predicate_alignment.py on the same deterministic folder."""
import json
import os
import types

import numpy as np
import pytest

from multike_amd import predicate_alignment as pa
from multike_amd.base.kgs import (KG, read_attribute_triples, read_kgs_from_folder, read_links, read_relation_triples)
from multike_amd.synthetic import write_dataset_folder
from multike_amd.utils import clear_attribute_triples, is_number, read_local_name, read_word2vec

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "data_golden.json")))


@pytest.fixture(scope="module")
def folder(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("dataset")) + "/"
    write_dataset_folder(d, seed=GOLD["writer"]["seed"], n_pairs=GOLD["writer"]["n_pairs"])
    return d


def _tl(x):
    return [list(t) for t in x]


@pytest.mark.parametrize("mode", ["swapping", "mapping", "sharing"])
def test_ordered_ids_and_containers_match_reference(folder, mode):
    g = GOLD[mode]
    k = read_kgs_from_folder(folder, "631/", mode, True)
    assert k.kg1.entities_id_dict == g["ent_ids1"] and k.kg2.entities_id_dict == g["ent_ids2"]
    assert k.kg1.relations_id_dict == g["rel_ids1"] and k.kg2.relations_id_dict == g["rel_ids2"]
    assert k.kg1.attributes_id_dict == g["attr_ids1"] and k.kg2.attributes_id_dict == g["attr_ids2"]
    assert _tl(k.train_links) == g["train_links"] and _tl(k.valid_links) == g["valid_links"]
    assert _tl(k.test_links) == g["test_links"]
    assert [k.entities_num, k.relations_num, k.attributes_num] == [g["entities_num"], g["relations_num"], g["attributes_num"]]
    for i, kg in ((1, k.kg1), (2, k.kg2)):
        assert _tl(sorted(kg.local_relation_triples_list)) == g[f"local_rel{i}"]
        assert len(kg.local_relation_triples_set) == g[f"local_set_size{i}"]          # the alias grew with the sup triples
        assert [kg.relation_triples_num, kg.local_relation_triples_num, kg.attribute_triples_num,
                kg.local_attribute_triples_num] == g[f"rel_num{i}"]
        assert _tl(sorted(kg.sup_relation_triples_list or [])) == g[f"sup_rel{i}"]
        assert _tl(sorted(kg.sup_attribute_triples_list or [])) == g[f"sup_attr{i}"]
        assert _tl(sorted(kg.local_attribute_triples_list)) == g[f"local_attr{i}"]
        arr = kg.relation_triples_array
        assert arr.dtype == np.int32 and arr.shape == (kg.local_relation_triples_num, 3)
        assert len(kg.known_relation_triples_array) == g[f"local_set_size{i}"]
    assert k.useful_entities_list1 == k.train_entities1 + k.valid_entities1 + k.test_entities1


def test_unordered_layout_is_reference_layout_and_deterministic(folder):
    g = GOLD["unordered"]
    k = read_kgs_from_folder(folder, "631/", "swapping", False)
    k_again = read_kgs_from_folder(folder, "631/", "swapping", False)
    assert k.kg1.entities_id_dict == k_again.kg1.entities_id_dict and k.kg2.entities_id_dict == k_again.kg2.entities_id_dict
    assert [k.kg1.entities_num, k.kg2.entities_num] == [g["n1"], g["n2"]]
    v1, v2 = sorted(k.kg1.entities_id_dict.values()), sorted(k.kg2.entities_id_dict.values())
    assert [v1[0], v1[-1]] == g["ids1_range"] and [v2[0], v2[-1]] == g["ids2_range"]
    assert v1 == list(range(g["n1"])) and v2 == list(range(g["n1"], g["n1"] + g["n2"]))
    assert [len(k.kg1.sup_relation_triples_list), len(k.kg2.sup_relation_triples_list)] == g["sup_rel"]
    # id triples map back to exactly the file's URI triples
    uri, _, _ = read_relation_triples(folder + "rel_triples_1")
    inv_e = {i: u for u, i in k.kg1.entities_id_dict.items()}
    inv_r = {i: u for u, i in k.kg1.relations_id_dict.items()}
    assert {(inv_e[h], inv_r[r], inv_e[t]) for h, r, t in k.kg1.local_relation_triples_list} == set(uri)
    # first-appearance order: the first triple's head gets id 0
    first = next(iter(uri))
    assert k.kg1.entities_id_dict[first[0]] == 0


def test_readers_edge_cases(tmp_path):
    p = tmp_path / "a"
    p.write_text("e1\tp\tv one\textra\t.\n\ne2\tq\n e3 \t p \t\"x\"@en .\n", encoding="utf8")
    t, ents, attrs = read_attribute_triples(str(p))
    assert set(t) == {("e1", "p", "v one extra"), ("e3", "p", '"x"@en')} and set(ents) == {"e1", "e3"} and set(attrs) == {"p"}
    r = tmp_path / "r"
    r.write_text("a\tr\tb \nb\tr\ta\na\tr\tb\n", encoding="utf8")
    t, ents, rels = read_relation_triples(str(r))
    assert list(t) == [("a", "r", "b"), ("b", "r", "a")] and list(ents) == ["a", "b"] and list(rels) == ["r"]
    bad = tmp_path / "bad"
    bad.write_text("a\tb\n", encoding="utf8")
    with pytest.raises(ValueError, match="bad:1"):
        read_relation_triples(str(bad))
    with pytest.raises(ValueError, match="expected 2"):
        read_links(str(r))
    assert read_relation_triples(None) == ({}, {}, {})
    empty = KG(set(), set())
    assert empty.entities_num == 0 and empty.relation_triples_array.shape == (0, 3)


def test_literal_cleanup_names_and_word_vectors(folder):
    k = read_kgs_from_folder(folder, "631/", "swapping", True)
    g = GOLD["clear_attribute_triples"]
    for i, kg in ((1, k.kg1), (2, k.kg2)):
        t, num, st = clear_attribute_triples(kg.local_attribute_triples_list)
        assert _tl(sorted(t)) == g[f"triples{i}"]
        assert sorted(num) == g[f"numbers{i}"] and sorted(st) == g[f"strings{i}"]
    assert read_local_name(folder, set(k.kg1.entities_id_dict), set(k.kg2.entities_id_dict)) == GOLD["local_names"]
    for s, want in GOLD["is_number"].items():
        assert is_number(s) == want, s
    w = read_word2vec(folder + "wiki-news-300d-tiny.vec")
    assert len(w) == GOLD["word2vec"]["n"] and w["amber"].dtype == np.float32
    assert [float(x) for x in w["amber"][:4]] == GOLD["word2vec"]["amber_head"]


def _snap(p):
    return {"relation_alignment_set": sorted(p.relation_alignment_set),
            "attribute_alignment_set": sorted(p.attribute_alignment_set),
            "relation_latent": sorted([a, b, s] for (a, b), s in p.relation_latent_match_pairs_similarity_dict_init.items()),
            "attribute_latent": sorted([a, b, s] for (a, b), s in p.attribute_latent_match_pairs_similarity_dict_init.items()),
            "sup_rel1": sorted(p.sup_relation_alignment_triples1), "sup_rel2": sorted(p.sup_relation_alignment_triples2),
            "sup_attr1": sorted(p.sup_attribute_alignment_triples1), "sup_attr2": sorted(p.sup_attribute_alignment_triples2),
            "rel_w1": sorted(p.relation_triples_w_weights1), "attr_w2": sorted(p.attribute_triples_w_weights2),
            "train_relations1": sorted(p.train_relations1), "train_attributes2": sorted(p.train_attributes2)}


def _close(a, b):
    """nested lists/tuples with floats: equal up to 1e-12 on numbers."""
    if isinstance(a, (list, tuple)):
        assert len(a) == len(b), (len(a), len(b))
        for x, y in zip(a, b):
            _close(x, y)
    elif isinstance(a, float) or isinstance(b, float):
        assert abs(a - b) <= 1e-12, (a, b)
    else:
        assert a == b, (a, b)


def test_predicate_alignment_matches_reference(folder):
    g = GOLD["predicate_alignment"]
    for a, b, want in g["ratios"]:
        assert abs(pa.levenshtein_ratio(a, b) - want) < 1e-15
    # documented value of python-Levenshtein itself (its README): ratio('Hello world!', 'Holly grail!') = 0.58333...
    assert abs(pa.levenshtein_ratio("Hello world!", "Holly grail!") - 7 / 12) < 1e-15
    k = read_kgs_from_folder(folder, "631/", "swapping", True)
    args = types.SimpleNamespace(training_data=folder, predicate_init_sim=0.9, predicate_soft_sim=0.85)
    pam = pa.PredicateAlignModel(k, args)
    got = _snap(pam)
    assert len(got["relation_alignment_set"]) >= 3 and len(got["sup_rel1"]) > 0           # the fixture is not vacuous
    for key, want in g["init"].items():
        _close(got[key], want)
    pam.update_predicate_alignment(np.asarray(g["rel_embed"]))
    pam.update_predicate_alignment(np.asarray(g["attr_embed"]), predicate_type="attribute")
    got = _snap(pam)
    for key, want in g["refreshed"].items():
        _close(got[key], want)


def test_levenshtein_matrix_against_scalar_dp():
    rng = np.random.default_rng(0)
    names1 = ["".join(rng.choice(list("abcde"), size=int(rng.integers(0, 9)))) for _ in range(12)]
    names2 = ["".join(rng.choice(list("abcde"), size=int(rng.integers(0, 11)))) for _ in range(9)]

    def lcs(a, b):
        L = [[0] * (len(b) + 1) for _ in range(len(a) + 1)]
        for i in range(len(a)):
            for j in range(len(b)):
                L[i + 1][j + 1] = L[i][j] + 1 if a[i] == b[j] else max(L[i][j + 1], L[i + 1][j])
        return L[len(a)][len(b)]
    m = pa.levenshtein_ratio_matrix(names1, names2)
    for i, a in enumerate(names1):
        for j, b in enumerate(names2):
            want = 1.0 if len(a) + len(b) == 0 else 2.0 * lcs(a, b) / (len(a) + len(b))
            assert abs(m[i, j] - want) < 1e-15, (a, b)


def test_predicate_refresh_built_on_the_device_equals_the_host_lists(folder):
    """`PredicateAlignModel.device` set (what the single-GPU drivers do): a refresh gathers / compacts its four lists with tensor
    operations (torch CPU tensors here) — same triples in the same order, weights = the float32 of the host's float64, and the
    list still reads as the reference's list (host form on demand) and still matches the reference-made fixture."""
    import torch
    g = GOLD["predicate_alignment"]
    k = read_kgs_from_folder(folder, "631/", "swapping", True)
    args = types.SimpleNamespace(training_data=folder, predicate_init_sim=0.9, predicate_soft_sim=0.85)
    host, dev = pa.PredicateAlignModel(k, args), pa.PredicateAlignModel(k, args)
    dev.device = torch.device("cpu")
    for p in (host, dev):
        p.update_predicate_alignment(np.asarray(g["rel_embed"]))
    for name in ("sup_relation_alignment_triples1", "sup_relation_alignment_triples2", "relation_triples_w_weights1",
                 "relation_triples_w_weights2"):
        a, b = getattr(host, name), getattr(dev, name)
        assert b.dev is not None and a.dev is None and len(a) == len(b) and len(a) > 0
        cols, w = b.dev
        assert np.array_equal(np.stack([c.numpy() for c in cols], axis=1), a.cols.astype(np.int32))
        assert np.array_equal(w.numpy(), a.w.astype(np.float32))
        assert np.array_equal(b.cols, a.cols) and np.array_equal(b.w, a.w) and list(b[:3]) == list(a[:3])
    both = dev.sup_relation_alignment_triples1 + dev.sup_relation_alignment_triples2
    want = host.sup_relation_alignment_triples1 + host.sup_relation_alignment_triples2
    assert both.dev is not None and len(both) == len(want)
    assert np.array_equal(torch.stack(both.dev[0], dim=1).numpy(), want.cols.astype(np.int32)) and np.array_equal(both.cols, want.cols)
    got = _snap(dev)
    for key in ("sup_rel1", "sup_rel2", "rel_w1", "relation_alignment_set"):
        _close(got[key], g["refreshed"][key])
