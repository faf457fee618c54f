"""Predicate (relation / attribute) soft alignment: who feeds the weighted and cross-KG inference batches of the hot
path (reference code/predicate_alignment.py:1-224; consumed at code/MultiKE_CSL.py:50-53,81-87 and
code/MultiKE_model.py:373-437).  Host-side bookkeeping, not a kernel.

  * initial matches: mutual best match of predicate local names under the Levenshtein *ratio*
    (2*LCS / (len1+len2), the python-Levenshtein definition: substitutions cost 2), kept when > predicate_init_sim;
  * refresh: mutual nearest neighbour of the predicate embeddings (cosine), blended
    w*name_sim + (1-w)*embedding_sim, kept when > predicate_soft_sim;
  * products: `sup_*_alignment_triples{1,2}` (triples re-stated with the matched predicate, carrying the match
    weight) and `*_triples_w_weights{1,2}` (own triples with weight 0.2 when unmatched, else the zoomed weight).

python-Levenshtein is not a dependency here: `levenshtein_ratio_matrix` is a batched LCS dynamic programme in numpy
(one row of predicates against all of the other side at once).
"""
from __future__ import annotations

import numpy as np

from .base.kgs import TripleArray

UNMATCHED_WEIGHT = 0.2


# ----------------------------------------------------------------------------------------------------------------
# string similarity
# ----------------------------------------------------------------------------------------------------------------
def levenshtein_ratio(a: str, b: str) -> float:
    """Levenshtein.ratio(a, b): (len(a)+len(b) - d) / (len(a)+len(b)) with d the edit distance whose substitution
    cost is 2, i.e. 2*LCS(a,b)/(len(a)+len(b)); 1.0 for two empty strings."""
    return float(levenshtein_ratio_matrix([a], [b])[0, 0])


def levenshtein_ratio_matrix(names1, names2) -> np.ndarray:
    """[len(names1), len(names2)] float64 ratios."""
    n1, n2 = len(names1), len(names2)
    out = np.zeros((n1, n2), dtype=np.float64)
    if n1 == 0 or n2 == 0:
        return out
    len2 = np.array([len(s) for s in names2], dtype=np.int64)
    L2 = int(len2.max()) if n2 else 0
    codes2 = np.full((n2, max(L2, 1)), -1, dtype=np.int64)
    for j, s in enumerate(names2):
        if s:
            codes2[j, :len(s)] = [ord(c) for c in s]
    for i, s in enumerate(names1):
        # lcs[j, q] = LCS(s[:p], names2[j][:q]) rolled over p
        lcs = np.zeros((n2, L2 + 1), dtype=np.int64)
        for ch in s:
            eq = codes2 == ord(ch)                                   # [n2, L2]
            nxt = np.zeros_like(lcs)
            diag = lcs[:, :-1] + eq                                  # take the match
            # nxt[:, q] = max(lcs[:, q], nxt[:, q-1], diag[:, q-1]); the running max along q is a prefix max
            cand = np.maximum(lcs[:, 1:], diag)
            nxt[:, 1:] = np.maximum.accumulate(cand, axis=1)
            lcs = nxt
        common = lcs[np.arange(n2), len2]
        tot = len(s) + len2
        out[i] = np.where(tot > 0, 2.0 * common / np.maximum(tot, 1), 1.0)
    return out


# ----------------------------------------------------------------------------------------------------------------
# matching
# ----------------------------------------------------------------------------------------------------------------
def link2dic(links):
    d1, d2 = {}, {}
    for i, j, w in links:
        d1[i] = (j, w)
        d2[j] = (i, w)
    if len(d1) != len(d2):
        raise ValueError("predicate links are not one-to-one")
    return d1, d2


def zoom_weight(weight, min_w_before, min_w_after=0.5):
    """[min_w_before, 1] -> [min_w_after, 1] linearly (code/predicate_alignment.py:135-137)."""
    return 1.0 - (1.0 - weight) * (1.0 - min_w_after) / (1.0 - min_w_before)


def generate_sup_predicate_triples(predicate_links, triples1, triples2):
    d1, d2 = link2dic(predicate_links)
    s1 = {(s, d1[p][0], o, d1[p][1]) for s, p, o in triples1 if p in d1}
    s2 = {(s, d2[p][0], o, d2[p][1]) for s, p, o in triples2 if p in d2}
    return sorted(s1), sorted(s2)


def add_weights(predicate_links, triples1, triples2, min_w_before):
    d1, d2 = link2dic(predicate_links)
    w1 = {(s, p, o, zoom_weight(d1[p][1], min_w_before) if p in d1 else UNMATCHED_WEIGHT) for s, p, o in triples1}
    w2 = {(s, p, o, zoom_weight(d2[p][1], min_w_before) if p in d2 else UNMATCHED_WEIGHT) for s, p, o in triples2}
    if len(w1) != len(triples1) or len(w2) != len(triples2):
        raise ValueError("duplicate triples in the weighted lists")
    return sorted(w1), sorted(w2), w1, w2


def _first_strict_argmax(sim):
    """Per row: the first column reaching the row maximum, and that maximum; '' semantics of the reference (a row
    whose best similarity is 0 matches nothing) is handled by the caller."""
    j = np.argmax(sim, axis=1)
    return j, sim[np.arange(sim.shape[0]), j]


def init_predicate_alignment(predicate_local_name_dict_1, predicate_local_name_dict_2, predicate_init_sim):
    """Mutual best name matches (code/predicate_alignment.py:47-74).  Returns (set of (p1, p2, sim) with
    sim > predicate_init_sim, dict (p1, p2) -> sim over all mutual matches).  Ties go to the first predicate in
    dict order, as the strict '>' scan of the reference does."""
    p1s, p2s = list(predicate_local_name_dict_1), list(predicate_local_name_dict_2)
    pairs, latent = set(), {}
    if not p1s or not p2s:
        return pairs, latent
    sim = levenshtein_ratio_matrix([predicate_local_name_dict_1[p] for p in p1s],
                                   [predicate_local_name_dict_2[p] for p in p2s])
    j12, s12 = _first_strict_argmax(sim)
    j21, s21 = _first_strict_argmax(sim.T)
    for i, p1 in enumerate(p1s):
        if s12[i] <= 0.0:
            continue                                                  # reference: match '' -> KeyError-free skip
        j = int(j12[i])
        if s21[j] > 0.0 and int(j21[j]) == i:
            latent[(p1, p2s[j])] = float(s12[i])
            if s12[i] > predicate_init_sim:
                pairs.add((p1, p2s[j], float(s12[i])))
    return pairs, latent


def read_predicate_local_name_file(file_path, relation_set):
    """`uri \\t local name`; URIs in `relation_set` are relations, all others attributes
    (code/predicate_alignment.py:77-88)."""
    rel, attr = {}, {}
    with open(file_path, "r", encoding="utf-8") as f:
        for no, line in enumerate(f, 1):
            p = line.rstrip("\n").split("\t")
            if len(p) != 2:
                raise ValueError(f"{file_path}:{no}: expected 2 tab-separated fields")
            (rel if p[0] in relation_set else attr)[p[0]] = p[1]
    return rel, attr


def predicate2id_matched_pairs(predicate_match_pairs_set, predicate_id_dict_1, predicate_id_dict_2):
    return {(predicate_id_dict_1[a], predicate_id_dict_2[b], w) for a, b, w in predicate_match_pairs_set
            if a in predicate_id_dict_1 and b in predicate_id_dict_2}


def find_predicate_alignment_by_embedding(embed, predicate_list1, predicate_list2, predicate_id_dict1=None,
                                          predicate_id_dict2=None):
    """Mutual nearest neighbours across the two predicate id lists under cosine similarity
    (code/predicate_alignment.py:99-132) -> {(id1, id2): sim}."""
    e = np.asarray(embed, dtype=np.float64)
    n = np.sqrt((e * e).sum(axis=1, keepdims=True))
    n[n == 0.0] = 1.0
    e = e / n
    l1, l2 = np.asarray(list(predicate_list1), dtype=np.int64), np.asarray(list(predicate_list2), dtype=np.int64)
    if l1.size == 0 or l2.size == 0:
        return {}
    sim = e[l1] @ e[l2].T
    best12 = np.argmax(sim, axis=1)
    best21 = np.argmax(sim, axis=0)
    return {(int(l1[i]), int(l2[j])): float(sim[i, j]) for i, j in enumerate(best12) if best21[j] == i}


# ----------------------------------------------------------------------------------------------------------------
class PredicateAlignModel:
    def __init__(self, kgs, args):
        self.kgs, self.args = kgs, args
        f = args.training_data
        self.relation_name_dict1, self.attribute_name_dict1 = read_predicate_local_name_file(
            f + "predicate_local_name_1", set(kgs.kg1.relations_id_dict))
        self.relation_name_dict2, self.attribute_name_dict2 = read_predicate_local_name_file(
            f + "predicate_local_name_2", set(kgs.kg2.relations_id_dict))
        self.relation_alignment_set, self.relation_latent_match_pairs_similarity_dict_init = \
            init_predicate_alignment(self.relation_name_dict1, self.relation_name_dict2, args.predicate_init_sim)
        self.attribute_alignment_set, self.attribute_latent_match_pairs_similarity_dict_init = \
            init_predicate_alignment(self.attribute_name_dict1, self.attribute_name_dict2, args.predicate_init_sim)
        self.relation_alignment_set_init = self.relation_alignment_set
        self.attribute_alignment_set_init = self.attribute_alignment_set
        self.update_relation_triples(self.relation_alignment_set)
        self.update_attribute_triples(self.attribute_alignment_set)

    def _int_triples(self, kind, which):
        """The KG's local triples as an int array [n, 3], or None when a column is not integral (attribute values are
        still strings before DataModel has replaced them by value ids)."""
        cache = self.__dict__.setdefault("_arr_cache", {})
        lst = getattr(self.kgs.kg1 if which == 1 else self.kgs.kg2, f"local_{kind}_triples_list")
        key = (kind, which)
        hit = cache.get(key)
        if hit is None or hit[0] is not lst:
            arr = None
            if all(isinstance(x, (int, np.integer)) for x in (lst[0] if lst else (0, 0, 0))):
                arr = np.asarray(lst, dtype=np.int64).reshape(-1, 3)
            hit = (lst, arr)
            cache[key] = hit
        return hit[1]

    def _refresh_arrays(self, kind, id_set, t1, t2):
        """`generate_sup_predicate_triples` + `add_weights` (code/predicate_alignment.py:17-44) as array operations: a
        look-up table per KG from predicate id to (matched predicate, weight)."""
        out = {}
        for which, t, own, other in ((1, t1, 0, 1), (2, t2, 1, 0)):
            n_pred = int(t[:, 1].max()) + 1 if len(t) else 1
            for link in id_set:
                n_pred = max(n_pred, int(link[own]) + 1)
            to = np.full(n_pred, -1, dtype=np.int64)
            wt = np.zeros(n_pred, dtype=np.float64)
            for link in id_set:
                to[link[own]], wt[link[own]] = link[other], link[2]
            p = t[:, 1]
            hit = to[p] >= 0
            out[f"sup{which}"] = TripleArray(np.stack([t[hit, 0], to[p[hit]], t[hit, 2]], axis=1), wt[p[hit]])
            out[f"w{which}"] = TripleArray(t, np.where(hit, zoom_weight(wt[p], self.args.predicate_soft_sim), UNMATCHED_WEIGHT))
        return out

    device = None      # set by a single-GPU driver: refreshes after the first build their lists in HBM (`_refresh_device`)

    def _device_state(self, kind, which, t):
        """Static per (kind, KG): the triples' columns in HBM and the number of triples per predicate on the host."""
        import torch
        st = self.__dict__.setdefault("_dev_state", {})
        key = (kind, which)
        hit = st.get(key)
        if hit is None or hit[0] is not t:
            cols = tuple(torch.as_tensor(np.ascontiguousarray(t[:, k]), dtype=torch.int32, device=self.device) for k in range(3))
            hit = (t, cols, cols[1].long(), np.bincount(t[:, 1]) if len(t) else np.zeros(1, np.int64))
            st[key] = hit
        return hit[1:]

    def _refresh_device(self, kind, id_set, t1, t2):
        """`_refresh_arrays` with the per-triple work in HBM: the host makes the two look-up tables per KG (predicate id ->
        matched predicate, weight: a few hundred entries), knows every list's length from the per-predicate triple counts
        (so nothing waits for the device), and the four lists are gathered / compacted on the device in the triples' own order.
        Weights: the float32 rounding of the float64 value per PREDICATE — what the upload of the host-made list holds.  The
        host form of a list is made on demand by `_refresh_arrays` (nobody reads it during training)."""
        import torch
        dev = self.device
        out, host_cache = {}, {}

        def host(name):
            if not host_cache:
                host_cache.update(self._refresh_arrays(kind, id_set, t1, t2))
            r = host_cache[name]
            return r.cols, r.w

        for which, t, own, other in ((1, t1, 0, 1), (2, t2, 1, 0)):
            cols, p_long, per_pred = self._device_state(kind, which, t)
            n_pred = max([len(per_pred)] + [int(link[own]) + 1 for link in id_set])
            to = np.full(n_pred, -1, dtype=np.int32)
            wt = np.zeros(n_pred, dtype=np.float64)
            for link in id_set:
                to[link[own]], wt[link[own]] = link[other], link[2]
            m = int(per_pred[(to[:len(per_pred)] >= 0)].sum())
            zw = np.where(to >= 0, zoom_weight(wt, self.args.predicate_soft_sim), UNMATCHED_WEIGHT).astype(np.float32)
            to_d = torch.as_tensor(to, device=dev)
            tab = torch.as_tensor(np.concatenate([wt.astype(np.float32), zw]), device=dev)
            wt_d, zw_d = tab[:n_pred], tab[n_pred:]
            if len(t):
                idx = torch.nonzero_static(to_d[p_long] >= 0, size=m).squeeze(1)
                pm = p_long[idx]
                sup = (cols[0][idx], to_d[pm], cols[2][idx])
                out[f"sup{which}"] = TripleArray.on_device(sup, wt_d[pm], lambda n_=f"sup{which}": host(n_))
                out[f"w{which}"] = TripleArray.on_device(cols, zw_d[p_long], lambda n_=f"w{which}": host(n_))
            else:
                r = self._refresh_arrays(kind, id_set, t1, t2)
                out[f"sup{which}"], out[f"w{which}"] = r[f"sup{which}"], r[f"w{which}"]
        return out

    def _refresh(self, kind, alignment_set):
        kg1, kg2 = self.kgs.kg1, self.kgs.kg2
        ids1, ids2 = getattr(kg1, kind + "s_id_dict"), getattr(kg2, kind + "s_id_dict")
        t1, t2 = getattr(kg1, f"local_{kind}_triples_list"), getattr(kg2, f"local_{kind}_triples_list")
        id_set = predicate2id_matched_pairs(alignment_set, ids1, ids2)
        setattr(self, f"{kind}_id_alignment_set", id_set)
        setattr(self, f"train_{kind}s1", [a for a, _, _ in id_set])
        setattr(self, f"train_{kind}s2", [b for _, b, _ in id_set])
        a1, a2 = self._int_triples(kind, 1), self._int_triples(kind, 2)
        if a1 is not None and a2 is not None:
            link2dic(id_set)                                          # one-to-one check, as the reference asserts
            r = (self._refresh_device if self.device is not None else self._refresh_arrays)(kind, id_set, a1, a2)
            setattr(self, f"sup_{kind}_alignment_triples1", r["sup1"])
            setattr(self, f"sup_{kind}_alignment_triples2", r["sup2"])
            setattr(self, f"{kind}_triples_w_weights1", r["w1"])
            setattr(self, f"{kind}_triples_w_weights2", r["w2"])
            self.__dict__.pop(f"_{kind}_wsets", None)                 # the set views are rebuilt on demand
            return
        s1, s2 = generate_sup_predicate_triples(id_set, t1, t2)
        setattr(self, f"sup_{kind}_alignment_triples1", s1)
        setattr(self, f"sup_{kind}_alignment_triples2", s2)
        w1, w2, ws1, ws2 = add_weights(id_set, t1, t2, self.args.predicate_soft_sim)
        setattr(self, f"{kind}_triples_w_weights1", w1)
        setattr(self, f"{kind}_triples_w_weights2", w2)
        self.__dict__[f"_{kind}_wsets"] = (ws1, ws2)

    def _wset(self, kind, which):
        sets = self.__dict__.get(f"_{kind}_wsets")
        if sets is None:
            sets = (set(getattr(self, f"{kind}_triples_w_weights1")), set(getattr(self, f"{kind}_triples_w_weights2")))
            self.__dict__[f"_{kind}_wsets"] = sets
        return sets[which - 1]

    relation_triples_w_weights_set1 = property(lambda self: self._wset("relation", 1))
    relation_triples_w_weights_set2 = property(lambda self: self._wset("relation", 2))
    attribute_triples_w_weights_set1 = property(lambda self: self._wset("attribute", 1))
    attribute_triples_w_weights_set2 = property(lambda self: self._wset("attribute", 2))

    def update_attribute_triples(self, attribute_alignment_set):
        self._refresh("attribute", attribute_alignment_set)

    def update_relation_triples(self, relation_alignment_set):
        self._refresh("relation", relation_alignment_set)

    def update_predicate_alignment(self, embed, predicate_type="relation", w=0.7):
        """code/predicate_alignment.py:188-224."""
        kind = "relation" if predicate_type == "relation" else "attribute"
        kg1, kg2 = self.kgs.kg1, self.kgs.kg2
        ids1, ids2 = getattr(kg1, kind + "s_id_dict"), getattr(kg2, kind + "s_id_dict")
        latent = find_predicate_alignment_by_embedding(embed, getattr(kg1, kind + "s_list"), getattr(kg2, kind + "s_list"))
        refreshed = set()
        for p1, p2, sim_init in getattr(self, kind + "_alignment_set_init"):
            sim = sim_init
            key = (ids1[p1], ids2[p2])
            if key in latent:
                sim = w * sim + (1 - w) * latent[key]
            if sim > self.args.predicate_soft_sim:
                refreshed.add((p1, p2, sim))
        print("update " + kind + " alignment:", len(refreshed))
        setattr(self, kind + "_alignment_set", refreshed)
        self._refresh(kind, refreshed)
