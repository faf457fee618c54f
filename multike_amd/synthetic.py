"""Synthetic two-KG generator of the benchmark / parity workloads.

Two KGs with disjoint contiguous id ranges — KG1 entities [0, E/2), KG2 [E/2, E); relations
[0, 0.6 R) / [0.6 R, R) — the id layout the reference's `generate_mapping_id(ordered=False)` produces
(code/base/kgs.py:15-20, code/base/read.py:75-84).  Triples are uniform without duplicates, about
`triples_per_entity` per entity (DBP-WD-like 4.6).
"""
from __future__ import annotations

import numpy as np


class SyntheticKGs:
    def __init__(self, n_ent=200_000, n_rel=550, triples_per_entity=4.6, seed=1234, kg1_share=0.5055):
        rng = np.random.default_rng(seed)
        self.entities_num, self.relations_num = int(n_ent), int(n_rel)
        e1 = n_ent // 2
        r1 = max(1, int(round(n_rel * 0.6)))
        self.ent_range = ((0, e1), (e1, n_ent))
        self.rel_range = ((0, r1), (r1, n_rel))
        total = int(n_ent * triples_per_entity)
        counts = (int(total * kg1_share), total - int(total * kg1_share))
        self.triples = []
        for (elo, ehi), (rlo, rhi), n in zip(self.ent_range, self.rel_range, counts):
            self.triples.append(self._uniform_unique(rng, elo, ehi, rlo, max(rhi, rlo + 1), n))

    @staticmethod
    def _uniform_unique(rng, elo, ehi, rlo, rhi, n):
        got = np.zeros((0, 3), dtype=np.int32)
        while len(got) < n:
            m = int((n - len(got)) * 1.1) + 16
            cand = np.stack([rng.integers(elo, ehi, m), rng.integers(rlo, rhi, m), rng.integers(elo, ehi, m)], 1)
            allt = np.concatenate([got, cand.astype(np.int32)], 0)
            key = (allt[:, 0].astype(np.int64) << 38) | (allt[:, 2].astype(np.int64) << 12) | allt[:, 1].astype(np.int64)
            _, first = np.unique(key, return_index=True)
            got = allt[np.sort(first)]
        return got[:n]

    def entities(self, kg: int) -> np.ndarray:
        lo, hi = self.ent_range[kg]
        return np.arange(lo, hi, dtype=np.int32)
