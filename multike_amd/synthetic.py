"""Synthetic two-KG generator of the benchmark / parity workloads.

Two KGs with disjoint contiguous id ranges — KG1 entities [0, E/2), KG2 [E/2, E); relations
[0, 0.6 R) / [0.6 R, R) — the id layout the reference's `generate_mapping_id(ordered=False)` produces
(code/base/kgs.py:15-20, code/base/read.py:75-84).  Triples are uniform without duplicates, about
`triples_per_entity` per entity (DBP-WD-like 4.6).
"""
from __future__ import annotations

import numpy as np


class SyntheticKGs:
    def __init__(self, n_ent=200_000, n_rel=550, triples_per_entity=4.6, seed=1234, kg1_share=0.5055, zipf=0.0, rel_zipf=0.0):
        """zipf > 0: head/tail entities drawn with probability ~ rank^-zipf inside each KG (hub entities), the
        contention variant of SURVEY.md §8d; 0 = uniform."""
        rng = np.random.default_rng(seed)
        self.zipf = float(zipf)
        self.rel_zipf = float(rel_zipf)     # relation ids ~ rank^-rel_zipf inside each KG (real relation frequencies are heavy-tailed too)
        self.entities_num, self.relations_num = int(n_ent), int(n_rel)
        e1 = n_ent // 2
        r1 = max(1, int(round(n_rel * 0.6)))
        self.ent_range = ((0, e1), (e1, n_ent))
        self.rel_range = ((0, r1), (r1, n_rel))
        total = int(n_ent * triples_per_entity)
        counts = (int(total * kg1_share), total - int(total * kg1_share))
        self.triples = []
        for (elo, ehi), (rlo, rhi), n in zip(self.ent_range, self.rel_range, counts):
            self.triples.append(self._uniform_unique(rng, elo, ehi, rlo, max(rhi, rlo + 1), n, self.zipf, self.rel_zipf))

    @staticmethod
    def _uniform_unique(rng, elo, ehi, rlo, rhi, n, zipf=0.0, rel_zipf=0.0):
        got = np.zeros((0, 3), dtype=np.int32)
        if rel_zipf > 0:
            rp = np.arange(1, rhi - rlo + 1, dtype=np.float64) ** (-rel_zipf)
            rcdf = np.cumsum(rp / rp.sum())
            draw_r = lambda m: rlo + np.minimum(np.searchsorted(rcdf, rng.random(m)), rhi - rlo - 1)
        else:
            draw_r = lambda m: rng.integers(rlo, rhi, m)
        if zipf > 0:
            pr = np.arange(1, ehi - elo + 1, dtype=np.float64) ** (-zipf)
            cdf = np.cumsum(pr / pr.sum())
            perm = rng.permutation(ehi - elo)  # hubs scattered over the id range
            draw = lambda m: elo + perm[np.minimum(np.searchsorted(cdf, rng.random(m)), ehi - elo - 1)]
        else:
            draw = lambda m: rng.integers(elo, ehi, m)
        while len(got) < n:
            m = int((n - len(got)) * 1.1) + 16
            cand = np.stack([draw(m), draw_r(m), draw(m)], 1)
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
                 link_share=0.3, seed=7, shared_structure=0.0):
        """shared_structure = p > 0: KG2 is a noisy copy of KG1 — each relation / attribute triple of KG1 is carried over with
        probability p (entity i <-> entity n/2 + i, predicates mapped into KG2's id range), the rest of KG2 stays random — so
        that the relation and attribute views have something to align on (p = 0: independent graphs)."""
        rng = np.random.default_rng(seed)
        base = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, triples_per_entity=triples_per_entity, seed=seed)
        if shared_structure > 0:
            e1 = n_ent // 2
            (rlo1, rhi1), (rlo2, rhi2) = base.rel_range
            t1, t2 = base.triples
            keep = t1[rng.random(len(t1)) < shared_structure]
            keep = keep[(keep[:, 0] + e1 < n_ent) & (keep[:, 2] + e1 < n_ent)]
            copy = np.stack([keep[:, 0] + e1, rlo2 + (keep[:, 1] - rlo1) % max(1, rhi2 - rlo2), keep[:, 2] + e1], 1).astype(np.int32)
            allt = np.concatenate([copy, t2[:max(0, len(t2) - len(copy))]], 0)
            key = (allt[:, 0].astype(np.int64) << 38) | (allt[:, 2].astype(np.int64) << 12) | allt[:, 1].astype(np.int64)
            _, first = np.unique(key, return_index=True)
            base.triples[1] = allt[np.sort(first)]
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
            if k == 1 and shared_structure > 0:      # KG1's attribute triples carried over onto the counterpart entities
                a_hi1 = a1
                at = {t for t in at if rng.random() >= shared_structure}
                at |= {(h + n1, alo + a % max(1, ahi - alo), v) for (h, a, v) in kg_list[0].local_attribute_triples_list
                       if rng.random() < shared_structure and h + n1 < n_ent}
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
    """`utils.default_args()` (the reference's hyper-parameters) pointed at synthetic data, plus a seed."""
    from .utils import default_args
    d = dict(training_data="synthetic/", output="/tmp/multike_out/", seed=0)
    d.update(over)
    return default_args(**d)


# ----------------------------------------------------------------------------------------------------------------
# a dataset *folder* in the reference's on-disk layout (README.md:10-20), for the reader / DataModel tests and demos
# ----------------------------------------------------------------------------------------------------------------
_WORDS = ("amber basalt cedar delta ember fjord garnet harbor indigo jasper kestrel lagoon marble nectar onyx prairie "
          "quartz raven sierra tundra umber valley willow xenon yarrow zephyr atlas bridge canyon dune estuary forest "
          "glacier heath island jungle knoll ledge meadow north oasis peak quarry ridge summit terrace upland vale "
          "wharf yard zenith").split()


def write_dataset_folder(folder, n_pairs=60, n_extra=8, n_rel=6, n_attr=7, seed=11, division="631/", word_dim=300,
                         triples_per_entity=3.0, shared_structure=0.0):
    """Writes rel_triples_{1,2}, attr_triples_{1,2}, entity_local_name_{1,2}, predicate_local_name_{1,2},
    <division>{train,valid,test}_links and a text word-vector file into `folder` (created).  Two KGs over `n_pairs`
    aligned entities (+ `n_extra` unaligned each) with partly matching predicate names, typed / language-tagged /
    multi-field literal values and a few rare attributes, so every branch of the readers is exercised.  Deterministic
    in `seed`.  shared_structure = p > 0: KG2 is a noisy copy of KG1 -- each relation / attribute triple of KG1 is kept
    with probability p (entity i <-> entity i, predicate j <-> predicate j) and the rest is random -- so that the
    relation and attribute views have something to align on (p = 0: the two graphs are independent and only the names
    carry signal).  Returns the path of the word-vector file."""
    import os
    rng = np.random.default_rng(seed)
    folder = folder if folder.endswith("/") else folder + "/"
    os.makedirs(folder + division, exist_ok=True)
    n = n_pairs + n_extra
    names = []
    for i in range(n_pairs):
        w = rng.choice(len(_WORDS), size=2, replace=False)
        names.append(f"{_WORDS[w[0]].capitalize()}_{_WORDS[w[1]]}" + ("_(place)" if i % 7 == 0 else ""))

    def ent(k, i):
        return f"http://kg{k}.example.org/resource/E{i}"

    rel_names = [f"{_WORDS[(3 * j) % len(_WORDS)]}Of" for j in range(n_rel)]
    attr_names = [f"{_WORDS[(5 * j + 1) % len(_WORDS)]}Value" for j in range(n_attr)]
    word_file = folder + "wiki-news-300d-tiny.vec"
    for k in (1, 2):
        # relations: the first n_rel-2 share names across KGs (one with a typo), the rest differ
        rels = {j: f"http://kg{k}.example.org/ontology/{rel_names[j] if j < n_rel - 2 else rel_names[j] + str(k) * 3}"
                for j in range(n_rel)}
        attrs = {j: f"http://kg{k}.example.org/property/{attr_names[j] if j < n_attr - 2 else 'x' * k + attr_names[j]}"
                 for j in range(n_attr)}
        if k == 2:
            rels[0] = rels[0][:-1] + "f"                                  # near-identical name
        n_tri = int(n * triples_per_entity)
        with open(folder + f"rel_triples_{k}", "w", encoding="utf8") as f:
            seen = set()
            if k == 2 and shared_structure > 0:
                seen = {t for t in sorted(kg1_rel) if rng.random() < shared_structure}
            for i in range(n):                                           # every entity occurs at least once
                t = (i, int(rng.integers(n_rel)), int((i + 1 + rng.integers(n - 1)) % n))
                seen.add(t)
            while len(seen) < n_tri:
                seen.add((int(rng.integers(n)), int(rng.integers(n_rel)), int(rng.integers(n))))
            if k == 1:
                kg1_rel = set(seen)
            for (h, r, t) in sorted(seen):
                f.write(f"{ent(k, h)}\t{rels[r]}\t{ent(k, t)} \n" if (h + t) % 5 == 0 else f"{ent(k, h)}\t{rels[r]}\t{ent(k, t)}\n")
        with open(folder + f"attr_triples_{k}", "w", encoding="utf8") as f:
            if k == 1:
                kg1_attr = []
            elif shared_structure > 0:
                for (i, a, v) in kg1_attr:
                    if rng.random() < shared_structure:
                        f.write(f"{ent(k, i)}\t{attrs[a]}\t{v}\n")
            for i in range(n):
                for _ in range(int(rng.integers(1, 5)) if not (k == 2 and shared_structure > 0) else int(rng.integers(0, 2))):
                    a = int(rng.integers(n_attr - 1))                    # the last attribute stays rare (< 10 triples)
                    style = int(rng.integers(5))
                    w = rng.choice(len(_WORDS), size=2, replace=False)
                    if style == 0:
                        v = f'"{_WORDS[w[0]]} {_WORDS[w[1]]}"@en'
                    elif style == 1:
                        v = f'"{int(rng.integers(1, 3000))}.{int(rng.integers(10))}"^^<http://www.w3.org/2001/XMLSchema#double>'
                    elif style == 2:
                        v = f'{_WORDS[w[0]]}_{_WORDS[w[1]]}-(north)\t{_WORDS[w[1]]}\t.'
                    elif style == 3:
                        v = f'http://kg{k}.example.org/resource/E{int(rng.integers(n))}'
                    else:
                        v = f'"{_WORDS[w[0]]}, {_WORDS[w[1]]}/zzunlisted{int(rng.integers(4))}" .'
                    f.write(f"{ent(k, i)}\t{attrs[a]}\t{v}\n")
                    if k == 1:
                        kg1_attr.append((i, a, v))
            f.write(f"{ent(k, 0)}\t{attrs[n_attr - 1]}\t\"rare value\"@en\n")
            f.write(f"{ent(k, 1)}\tshort line\n")
        with open(folder + f"entity_local_name_{k}", "w", encoding="utf8") as f:
            for i in range(n):
                if i < n_pairs:
                    nm = names[i] if (k == 1 or i % 5) else names[i].replace("_", "_the_", 1)
                else:
                    nm = f"{_WORDS[int(rng.integers(len(_WORDS)))]}_{k}{i}"
                if not (k == 2 and i == n - 1):                          # one entity without a local-name line
                    f.write(f"{ent(k, i)}\t{nm}\n")
        with open(folder + f"predicate_local_name_{k}", "w", encoding="utf8") as f:
            for j in range(n_rel):
                f.write(f"{rels[j]}\t{rels[j].rsplit('/', 1)[1]}\n")
            for j in range(n_attr):
                f.write(f"{attrs[j]}\t{attrs[j].rsplit('/', 1)[1]}\n")
    order = rng.permutation(n_pairs)
    n_tr, n_va = int(n_pairs * 0.3), int(n_pairs * 0.1)
    for name, sel in (("train_links", order[:n_tr]), ("valid_links", order[n_tr:n_tr + n_va]), ("test_links", order[n_tr + n_va:])):
        with open(folder + division + name, "w", encoding="utf8") as f:
            for i in sel:
                f.write(f"{ent(1, int(i))}\t{ent(2, int(i))}\n")
    with open(word_file, "w", encoding="utf-8") as f:
        f.write(f"{len(_WORDS)} {word_dim}\n")
        for w in _WORDS + ["the", "place", "north"]:
            vec = rng.standard_normal(word_dim) * 0.3
            f.write(w + " " + " ".join(f"{x:.5f}" for x in vec) + "\n")
            f.write(w.capitalize() + " " + " ".join(f"{x:.5f}" for x in vec * 0.9) + "\n")
    return word_file
