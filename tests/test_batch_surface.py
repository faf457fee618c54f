"""The PRODUCT's name-for-name batch surface (`multike_amd.base.batch`, `multike_amd.attr_batch`,
`sampling.kg_batch_split`, `utils.task_divide`) against the fixtures made by executing the reference's own
`code/base/batch.py`, `code/attr_batch.py`, `code/utils.py` (tests/golden/sampler_golden.json).

Positives are deterministic (slice arithmetic, code/base/batch.py:36-54, code/attr_batch.py:4-10,39-50): the product
must return the reference's lists exactly.  Negatives come from a different RNG stream (the reference seeds none,
SURVEY §0.6), so they are held to the S2 invariants the reference's own output satisfies
(tests/test_oracle_golden.py::test_reference_sampler_invariants): N per positive in positive order, exactly one of h/t
replaced and never r, the replacement drawn from `neighbor.get(x, entities_list)` of the same KG, no negative inside the
known set unless round 10 had to keep it, distinct candidates inside one round.
"""
import numpy as np
import pytest


def _tuples(x):
    return [tuple(t) for t in x]


def _lists(g):
    return (_tuples(g["triples1"]), _tuples(g["triples2"]), set(_tuples(g["known1"])), set(_tuples(g["known2"])),
            list(g["ents1"]), list(g["ents2"]))


# ------------------------------------------------ CPU: positives, splits ------------------------------------------------
@pytest.mark.parametrize("run_i", range(4))
def test_relation_batch_positives_equal_reference_lists(sampler_golden, run_i):
    """generate_relation_triple_batch with neg_triples_num = 0 touches no device code: pos1 + pos2 list for list."""
    from multike_amd.base import batch as bat
    g = sampler_golden
    run = g["relation_runs"][run_i]
    t1, t2, k1, k2, e1, e2 = _lists(g)
    for step, exp in enumerate(run["steps"]):
        pos, neg = bat.generate_relation_triple_batch(t1, t2, k1, k2, e1, e2, run["batch_size"], step, None, None, 0)
        assert pos == _tuples(exp["pos"]) and neg == []


def test_generate_pos_triples_slices_and_fixed_size():
    from multike_amd.base import batch as bat
    tr = [(i, 0, i + 1) for i in range(11)]
    assert bat.generate_pos_triples(tr, 4, 0) == tr[0:4]
    assert bat.generate_pos_triples(tr, 4, 2) == tr[8:11]            # short last slice (code/base/batch.py:48-50)
    assert bat.generate_pos_triples(tr, 4, 3) == []                  # beyond the end: empty
    assert bat.generate_pos_triples(tr, 4, 2, is_fixed_size=True) == tr[8:11] + tr[0:1]   # :51-53 wraps to the front


def test_attribute_batches_equal_reference_lists(sampler_golden):
    from multike_amd import attr_batch as ab
    g = sampler_golden
    a1, a2 = _tuples(g["attr1"]), _tuples(g["attr2"])
    bs = g["attribute_run"]["batch_size"]
    for step, exp in enumerate(g["attribute_run"]["steps"]):
        pos, neg = ab.generate_attribute_triple_batch(a1, a2, None, None, None, None, bs, step, None, None, 0)
        assert pos == _tuples(exp["pos"]) and neg == exp["neg"] == []

    class Q(list):
        put = list.append
    q = Q()
    steps = list(range(len(g["attribute_run"]["steps"])))
    ab.generate_attribute_triple_batch_queue(a1, a2, None, None, None, None, bs, steps, q, None, None, 0)
    assert [p for p, _ in q] == [_tuples(s["pos"]) for s in g["attribute_run"]["steps"]]


def test_dead_attribute_negative_sampler_contract():
    """code/attr_batch.py:13-25: head-only corruption, unbounded rejection against the known set."""
    import random
    from multike_amd import attr_batch as ab
    random.seed(3)
    pos = [(1, 7, 70, 0.5), (2, 8, 80, 1.0)]
    known = {(1, 7, 70, 0.5), (2, 8, 80, 1.0), (3, 7, 70, 0.5)}
    neg = ab.generate_neg_attribute_triples(pos, known, [1, 2, 3, 4, 5], 4, neighbor={2: [2, 9]})
    assert len(neg) == 8
    for i, (h, a, v, w) in enumerate(pos):
        for c in neg[4 * i:4 * i + 4]:
            assert c[1:] == (a, v, w) and c not in known
    assert all(c[0] == 9 for c in neg[4:])                           # neighbour list honoured; 2 itself is known


def test_kg_batch_split_and_task_divide_tables(sampler_golden):
    from multike_amd.sampling import kg_batch_split
    from multike_amd.utils import task_divide
    for row in sampler_golden["kg_batch_split"]:
        assert kg_batch_split(row["n1"], row["n2"], row["batch"]) == (row["b1"], row["b2"])
    for row in sampler_golden["task_divide"]:
        assert [list(x) for x in task_divide(list(range(row["total"])), row["n"])] == row["tasks"]


# ------------------------------------------------ GPU: the list surface end to end --------------------------------------
def _check_negatives(pos, neg, N, known, ents_of, near_of):
    assert len(neg) == N * len(pos)
    n_known = 0
    for i, (h, r, t) in enumerate(pos):
        grp = neg[i * N:(i + 1) * N]
        for (a, b, c) in grp:
            assert b == r, "the relation is never corrupted"
            assert (a == h) or (c == t), "at most one side replaced"
            if a != h:
                assert a in near_of(h), "corrupt head not from neighbor.get(h, entities_list)"
            if c != t:
                assert c in near_of(t), "corrupt tail not from neighbor.get(t, entities_list)"
            assert a in ents_of(h) and c in ents_of(t), "same KG"
            n_known += (a, b, c) in known
    return n_known


@pytest.mark.gpu
@pytest.mark.parametrize("run_i", range(4))
def test_relation_batch_list_surface_on_device(sampler_golden, run_i):
    from multike_amd.base import batch as bat
    g = sampler_golden
    run = g["relation_runs"][run_i]
    N = run["neg"]
    t1, t2, k1, k2, e1, e2 = _lists(g)
    near1 = {int(k): v for k, v in g["near1"].items()} if run["use_near"] else None
    near2 = {int(k): v for k, v in g["near2"].items()} if run["use_near"] else None
    s1, s2 = set(e1), set(e2)

    def ents_of(x):
        return s1 if x in s1 else s2

    def near_of(x):
        nd = near1 if x in s1 else near2
        return set(nd[x]) if nd and x in nd else ents_of(x)

    total_known = total = 0
    for step, exp in enumerate(run["steps"]):
        pos, neg = bat.generate_relation_triple_batch(t1, t2, k1, k2, e1, e2, run["batch_size"], step, near1, near2, N)
        assert pos == _tuples(exp["pos"])                                       # exactly the reference's positives
        assert all(isinstance(x, tuple) and len(x) == 3 for x in neg)           # list of 3-tuples of ints, as the reference
        total_known += _check_negatives(pos, neg, N, k1 | k2, ents_of, near_of)
        total += len(neg)
        # the reference's own output of this step satisfies the same invariants
        _check_negatives(_tuples(exp["pos"]), _tuples(exp["neg"]), N, k1 | k2, ents_of, near_of)
    # known triples survive only through round 10 (code/base/batch.py:108-111): on these toy KGs that is rare
    ref_known = sum((tuple(x) in (k1 | k2)) for st in run["steps"] for x in st["neg"])
    assert total_known <= max(5, 3 * ref_known + total // 50)


@pytest.mark.gpu
def test_neg_triples_fast_direct_call_and_queue(sampler_golden):
    from multike_amd.base import batch as bat
    g = sampler_golden
    t1, t2, k1, k2, e1, e2 = _lists(g)
    pos = t1[:17]
    neg = bat.generate_neg_triples_fast(pos, k1, e1, 6, neighbor=None, max_try=10, seed=(5, 6))
    again = bat.generate_neg_triples_fast(pos, k1, e1, 6, neighbor=None, max_try=10, seed=(5, 6))
    assert neg == again and len(neg) == 6 * 17                                   # counter-based stream: reproducible
    for i, (h, r, t) in enumerate(pos):
        grp = neg[6 * i:6 * i + 6]
        assert all(b == r and ((a == h) != (c == t) or (a, b, c) == (h, r, t)) for a, b, c in grp)
    assert bat.generate_neg_triples_fast([], k1, e1, 6) == [] and bat.generate_neg_triples_fast(pos, k1, e1, 0) == []

    class Q(list):
        put = list.append
    q = Q()
    bat.generate_relation_triple_batch_queue(t1, t2, k1, k2, e1, e2, 20, [0, 1, 2], q, None, None, 5)
    run = g["relation_runs"][0]
    assert [p for p, _ in q] == [_tuples(s["pos"]) for s in run["steps"][:3]]
    assert all(len(n) == 5 * len(p) for p, n in q)


@pytest.mark.gpu
def test_uniform_over_candidates_chi_square(sampler_golden):
    """Each corrupt entity is uniform over the candidate list (random.sample, code/base/batch.py:96-99)."""
    from multike_amd.base import batch as bat
    g = sampler_golden
    t1, _, k1, _, e1, _ = _lists(g)
    counts = np.zeros(max(e1) + 1)
    n_draws = 0
    for rep in range(40):
        neg = bat.generate_neg_triples_fast(t1, set(), e1, 10, seed=(99, rep))
        for i, (h, r, t) in enumerate(t1):
            for (a, b, c) in neg[10 * i:10 * i + 10]:
                e = a if a != h else c
                if (a, c) != (h, t):
                    counts[e] += 1
                    n_draws += 1
    exp = n_draws / len(e1)
    chi2 = float(((counts[e1] - exp) ** 2 / exp).sum())
    assert chi2 < 2.2 * len(e1), chi2                                            # 49 dof: p ~ 1e-6 at 110


def test_int_triples_accepts_arrays_lists_sets_and_triple_arrays():
    """`sampling.int_triples`: the reference's containers of (h, r, t) tuples (list, set), arrays and `TripleArray`s all become the
    same int32 [n, 3] array; weighted 4-tuples keep their first three columns only through the generic path."""
    from multike_amd.base.kgs import TripleArray
    from multike_amd.sampling import int_triples
    rng = np.random.default_rng(3)
    a = rng.integers(0, 1000, size=(257, 3)).astype(np.int64)
    lst = [tuple(int(x) for x in r) for r in a]
    for form in (a, a.astype(np.int32), lst, TripleArray(a)):
        got = int_triples(form)
        assert got.dtype == np.int32 and got.shape == (257, 3) and np.array_equal(got, a)
    got = int_triples(set(lst))
    assert got.dtype == np.int32 and sorted(map(tuple, got.tolist())) == sorted(set(lst))
    assert int_triples([]).shape == (0, 3) and int_triples(set()).shape == (0, 3)
