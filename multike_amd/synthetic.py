"""Synthetic two-KG generator of the benchmark / parity workloads.

Two KGs with disjoint contiguous id ranges — KG1 entities [0, E/2), KG2 [E/2, E); relations
[0, 0.6 R) / [0.6 R, R) — the id layout the reference's `generate_mapping_id(ordered=False)` produces
(code/base/kgs.py:15-20, code/base/read.py:75-84).  Triples are uniform without duplicates, about
`triples_per_entity` per entity (DBP-WD-like 4.6).
"""
from __future__ import annotations

import numpy as np


class SyntheticKGs:
    def __init__(self, n_ent=200_000, n_rel=550, triples_per_entity=4.6, seed=1234, kg1_share=0.5055, zipf=0.0):
        """zipf > 0: head/tail entities drawn with probability ~ rank^-zipf inside each KG (hub entities), the
        contention variant of SURVEY.md §8d; 0 = uniform."""
        rng = np.random.default_rng(seed)
        self.zipf = float(zipf)
        self.entities_num, self.relations_num = int(n_ent), int(n_rel)
        e1 = n_ent // 2
        r1 = max(1, int(round(n_rel * 0.6)))
        self.ent_range = ((0, e1), (e1, n_ent))
        self.rel_range = ((0, r1), (r1, n_rel))
        total = int(n_ent * triples_per_entity)
        counts = (int(total * kg1_share), total - int(total * kg1_share))
        self.triples = []
        for (elo, ehi), (rlo, rhi), n in zip(self.ent_range, self.rel_range, counts):
            self.triples.append(self._uniform_unique(rng, elo, ehi, rlo, max(rhi, rlo + 1), n, self.zipf))

    @staticmethod
    def _uniform_unique(rng, elo, ehi, rlo, rhi, n, zipf=0.0):
        got = np.zeros((0, 3), dtype=np.int32)
        if zipf > 0:
            pr = np.arange(1, ehi - elo + 1, dtype=np.float64) ** (-zipf)
            cdf = np.cumsum(pr / pr.sum())
            perm = rng.permutation(ehi - elo)  # hubs scattered over the id range
            draw = lambda m: elo + perm[np.minimum(np.searchsorted(cdf, rng.random(m)), ehi - elo - 1)]
        else:
            draw = lambda m: rng.integers(elo, ehi, m)
        while len(got) < n:
            m = int((n - len(got)) * 1.1) + 16
            cand = np.stack([draw(m), rng.integers(rlo, rhi, m), draw(m)], 1)
            allt = np.concatenate([got, cand.astype(np.int32)], 0)
            key = (allt[:, 0].astype(np.int64) << 38) | (allt[:, 2].astype(np.int64) << 12) | allt[:, 1].astype(np.int64)
            _, first = np.unique(key, return_index=True)
            got = allt[np.sort(first)]
        return got[:n]

    def entities(self, kg: int) -> np.ndarray:
        lo, hi = self.ent_range[kg]
        return np.arange(lo, hi, dtype=np.int32)


class _KG:
    pass


class SyntheticData:
    """The attributes of the reference's DataModel / KGs / PredicateAlignModel that `MultiKE_model.MultiKE` reads
    (code/data_model.py, code/base/kgs.py, code/predicate_alignment.py), filled with synthetic content:
    relation triples from `SyntheticKGs`, weighted attribute triples, 'swapping' supervision triples generated from
    train links (code/base/read.py:130-146 semantics: e1's triples re-written onto e2 and vice versa)."""

    def __init__(self, n_ent=2000, n_rel=30, n_attr=24, n_values=500, dim=32, triples_per_entity=4.6, attr_per_entity=3.0,
                 link_share=0.3, seed=7):
        rng = np.random.default_rng(seed)
        base = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, triples_per_entity=triples_per_entity, seed=seed)
        self.base = base
        kgs = _KG()
        kgs.entities_num, kgs.relations_num, kgs.attributes_num = n_ent, n_rel, n_attr
        n1 = n_ent // 2
        links = [(i, n1 + i) for i in range(int(n1 * link_share))]
        kgs.train_links = links
        n_tr = len(links)
        n_va = int(n1 * 0.1)
        kgs.valid_entities1 = list(range(n_tr, n_tr + n_va))
        kgs.valid_entities2 = [n1 + i for i in kgs.valid_entities1]
        kgs.test_entities1 = list(range(n_tr + n_va, n1))
        kgs.test_entities2 = [n1 + i for i in kgs.test_entities1]
        a1 = max(1, int(n_attr * 0.5))
        kg_list = []
        for k in (0, 1):
            kg = _KG()
            trip = [tuple(int(v) for v in t) for t in base.triples[k]]
            kg.local_relation_triples_list = list(trip)
            kg.entities_list = [int(e) for e in base.entities(k)]
            kg.entities_num = len(kg.entities_list)
            kg.local_relation_triples_num = len(trip)
            lo, hi = base.ent_range[k]
            alo, ahi = (0, a1) if k == 0 else (a1, n_attr)
            n_at = int(kg.entities_num * attr_per_entity)
            at = {(int(rng.integers(lo, hi)), int(rng.integers(alo, max(ahi, alo + 1))), int(rng.integers(0, n_values)))
                  for _ in range(n_at)}
            kg.local_attribute_triples_list = sorted(at)
            kg.local_attribute_triples_num = len(at)
            kg.entities_id_dict = kg.relations_id_dict = kg.attributes_id_dict = None
            kg_list.append(kg)
        kg1, kg2 = kg_list
        # 'swapping': triples of a linked entity re-written onto its counterpart
        m12 = dict(links)
        m21 = {b: a for a, b in links}
        for kg, other, m in ((kg1, kg2, m12), (kg2, kg1, m21)):
            sup_r = [(m.get(h, h), r, m.get(t, t)) for (h, r, t) in kg.local_relation_triples_list if h in m or t in m]
            sup_a = [(m[h], a, v) for (h, a, v) in kg.local_attribute_triples_list if h in m]
            other._sup_r, other._sup_a = sup_r, sup_a
        for kg in (kg1, kg2):
            kg.sup_relation_triples_list = kg._sup_r
            kg.sup_attribute_triples_list = kg._sup_a
            # the reference's alias: the "known" set handed to the sampler includes the swapped triples
            kg.local_relation_triples_set = set(kg.local_relation_triples_list) | set(kg._sup_r)
        kgs.kg1, kgs.kg2 = kg1, kg2
        kgs.useful_entities_list1, kgs.useful_entities_list2 = kg1.entities_list, kg2.entities_list
        self.kgs = kgs
        v = rng.standard_normal((n_values, dim)).astype(np.float32)
        self.value_vectors = v / np.linalg.norm(v, axis=1, keepdims=True)
        nm = rng.standard_normal((n_ent, dim)).astype(np.float32)
        self.local_name_vectors = nm / np.linalg.norm(nm, axis=1, keepdims=True)
        # PredicateAlignModel attributes the training loops read
        pam = _KG()
        pam.attribute_triples_w_weights1 = [(h, a, v, float(rng.choice([0.2, 0.6, 1.0]))) for (h, a, v) in kg1.local_attribute_triples_list]
        pam.attribute_triples_w_weights2 = [(h, a, v, float(rng.choice([0.2, 0.6, 1.0]))) for (h, a, v) in kg2.local_attribute_triples_list]
        pam.attribute_triples_w_weights_set1 = set(pam.attribute_triples_w_weights1)
        pam.attribute_triples_w_weights_set2 = set(pam.attribute_triples_w_weights2)
        r1 = base.rel_range[0][1]
        pam.sup_relation_alignment_triples1 = [(h, int(rng.integers(r1, n_rel)), t, 0.7) for (h, r, t) in kg1.local_relation_triples_list[:600]]
        pam.sup_relation_alignment_triples2 = [(h, int(rng.integers(0, r1)), t, 0.9) for (h, r, t) in kg2.local_relation_triples_list[:500]]
        pam.sup_attribute_alignment_triples1 = [(h, int(rng.integers(a1, n_attr)), v, 0.8) for (h, a, v) in kg1.local_attribute_triples_list[:400]]
        pam.sup_attribute_alignment_triples2 = [(h, int(rng.integers(0, a1)), v, 0.6) for (h, a, v) in kg2.local_attribute_triples_list[:300]]
        self.predicate_align_model = pam


def synthetic_args(**over):
    """The keys of code/args.json that the hot path reads, with the reference's defaults."""
    from .utils import ARGs
    d = dict(training_data="synthetic/", output="/tmp/multike_out/", alignment_module="swapping", dim=75,
             learning_rate=0.001, optimizer="Adagrad", max_epoch=200, shared_learning_max_epoch=200, batch_size=5000,
             entity_batch_size=5000, attribute_batch_size=5000, neg_triple_num=10, neg_sampling="truncated",
             truncated_epsilon=0.98, truncated_freq=20, batch_threads_num=4, test_threads_num=8, start_valid=100,
             eval_freq=10, top_k=[1, 5, 10, 50], orthogonal_weight=2, cv_name_weight=1, cv_weight=1,
             start_predicate_soft_alignment=10, predicate_soft_sim=0.85, predicate_init_sim=0.90, ITC_learning_rate=0.004,
             seed=0)
    d.update(over)
    return ARGs(d)
