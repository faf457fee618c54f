"""attr_batch.py surface of the reference (code/attr_batch.py): attribute-view batches of weighted 4-tuples
(h, a, v, w).  The live call passes neg_triples_num = 0 (code/MultiKE_model.py:331), so this is list slicing plus
the proportional KG split; the (dead) negative sampler is kept for interface completeness."""
from __future__ import annotations

import random

from .sampling import kg_batch_split


def generate_pos_triples(triples, batch_size, step):
    """code/attr_batch.py:4-10."""
    lo = step * batch_size
    return triples[lo:min(lo + batch_size, len(triples))]


def generate_neg_attribute_triples(pos_batch, all_triples_set, entity_list, neg_triples_num, neighbor=None):
    """code/attr_batch.py:13-25: head-only corruption with an unbounded rejection loop (host-side; never called
    with neg_triples_num > 0 by the reference)."""
    neighbor = neighbor or {}
    out = []
    for (h, a, v, w) in pos_batch:
        pool = neighbor.get(h, entity_list)
        for _ in range(neg_triples_num):
            cand = (random.choice(pool), a, v, w)
            while cand in all_triples_set:
                cand = (random.choice(pool), a, v, w)
            out.append(cand)
    return out


def generate_attribute_triple_batch(triple_list1, triple_list2, triple_set1, triple_set2, entity_list1, entity_list2,
                                    batch_size, step, neighbor1, neighbor2, neg_triples_num):
    """code/attr_batch.py:39-50."""
    b1, b2 = kg_batch_split(len(triple_list1), len(triple_list2), batch_size)
    pos1 = generate_pos_triples(triple_list1, b1, step)
    pos2 = generate_pos_triples(triple_list2, b2, step)
    neg1 = generate_neg_attribute_triples(pos1, triple_set1, entity_list1, neg_triples_num, neighbor=neighbor1)
    neg2 = generate_neg_attribute_triples(pos2, triple_set2, entity_list2, neg_triples_num, neighbor=neighbor2)
    return pos1 + pos2, neg1 + neg2


def generate_attribute_triple_batch_queue(triple_list1, triple_list2, triple_set1, triple_set2, entity_list1, entity_list2,
                                          batch_size, steps, out_queue, neighbor1, neighbor2, neg_triples_num):
    """code/attr_batch.py:28-36."""
    for step in steps:
        out_queue.put(generate_attribute_triple_batch(triple_list1, triple_list2, triple_set1, triple_set2, entity_list1,
                                                      entity_list2, batch_size, step, neighbor1, neighbor2, neg_triples_num))
