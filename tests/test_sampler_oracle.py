"""The statistical link between the reference's negative sampler and the Philox specification the device implements.

`oracle/sampler_oracle.py` holds two restatements of /root/reference/code/base/batch.py:86-116
(`generate_neg_triples_fast`): `mt_negatives` — Mersenne-Twister, pinned list-for-list to the reference executed
(tests/test_oracle_golden.py::test_mt_restatement_replays_reference_batches) — and `philox_negatives`, the
counter-based specification the HIP sampler reproduces bit for bit (tests/test_sampler_gpu.py).  The reference sets
no seed, so the two cannot agree draw for draw; what must agree is the JOINT structure the reference produces:

  * one coin per ROUND: a positive's negatives are all-head or all-tail unless a later round was needed;
  * the rounds a positive uses (re-draws after known triples were dropped);
  * duplicates inside one positive's negatives (each round samples without replacement, rounds are independent);
  * the known-triple leak (only the last round is unfiltered);
  * which entity replaces the head / tail (uniform over the candidate list).

Each is compared by a two-sample chi-square test of homogeneity on >= 2,000 positives, for N in {1, 10, 25}, on a
10K-entity KG whose known set almost never hits (one round nearly always) and on a toy KG whose known set holds half of
all possible triples (re-draws in most positives, the unfiltered last round reached).  A deliberately wrong
specification — one coin per NEGATIVE instead of one per round — must be rejected by the same tests.
"""
import random

import numpy as np
import pytest
from scipy import stats as sps

from oracle import sampler_oracle as so

P_ACCEPT = 1e-3      # the two samplers are the same distribution: p-values are uniform; seeds are fixed
P_REJECT = 1e-9      # the broken specification must be rejected far beyond doubt


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
def _sparse_kg():
    """10,000 entities (the smoke / C1 scale), 100 relations: a candidate is a known triple with probability ~1e-6."""
    rng = np.random.default_rng(11)
    n_ent, n_rel, n_tri = 10_000, 100, 30_000
    tri = {(int(h), int(r), int(t)) for h, r, t in zip(rng.integers(0, n_ent, n_tri), rng.integers(0, n_rel, n_tri),
                                                        rng.integers(0, n_ent, n_tri)) if h != t}
    tri = sorted(tri)
    pos = [tri[i] for i in rng.permutation(len(tri))[:2000]]
    return dict(name="sparse", ents=list(range(n_ent)), known=set(tri), pos=pos, max_try=10, buckets=50)


def _dense_kg():
    """40 entities, 3 relations, HALF of all 4,800 possible triples known: a 25-negative round keeps ~12, so most
    positives need several rounds; with max_try = 3 the unfiltered last round is reached and leaks known triples."""
    rng = np.random.default_rng(12)
    n_ent, n_rel = 40, 3
    every = [(h, r, t) for h in range(n_ent) for r in range(n_rel) for t in range(n_ent)]
    known = {every[i] for i in rng.permutation(len(every))[:len(every) // 2]}
    kl = sorted(known)
    pos = [kl[i] for i in rng.integers(0, len(kl), 2400)]
    return dict(name="dense", ents=list(range(n_ent)), known=known, pos=pos, max_try=3, buckets=40)


_KGS = {"sparse": _sparse_kg, "dense": _dense_kg}
_cache = {}


def _draw(kind, which, N, seed=5, **kw):
    """-> dict of per-positive statistics of one sampler run."""
    key = (kind, which, N, seed, tuple(sorted(kw.items())))
    if key in _cache:
        return _cache[key]
    kg = _cache.setdefault(("kg", kind), _KGS[kind]())
    pos, ents, known, max_try = kg["pos"], kg["ents"], kg["known"], kg["max_try"]
    st = {}
    if which == "mt":
        random.seed(seed)
        np.random.seed(seed)
        neg = so.mt_negatives(pos, known, ents, N, max_try=max_try, stats=st)
    else:
        ph = np.array([p[0] for p in pos]); pr = np.array([p[1] for p in pos]); pt = np.array([p[2] for p in pos])
        nh, nr, nt = so.philox_negatives(ph, pr, pt, N, len(ents), ent_lo=0, known=known, seed=(seed, 77), stream_id=3,
                                         max_try=max_try, stats=st, **kw)
        neg = list(zip(nh.tolist(), nr.tolist(), nt.tolist()))
    assert len(neg) == N * len(pos) and len(st["rounds"]) == len(pos)
    same_side, heads, dups, leaks, repl = [], 0, [], [], []
    for i, (h, r, t) in enumerate(pos):
        grp = neg[i * N:(i + 1) * N]
        # a candidate equal to the entity it replaces leaves the triple unchanged: count it on neither side
        sh = sum(1 for (a, b, c) in grp if a != h)
        stl = sum(1 for (a, b, c) in grp if c != t)
        assert all(b == r for (_, b, _) in grp) and all((a == h) or (c == t) for (a, _, c) in grp)
        same_side.append(int(sh == 0 or stl == 0))
        heads += int(grp[0][0] != h)     # side of the positive's FIRST negative = its first productive round's coin (negatives
                                         # of one round share a side, so per-negative counts would be N-fold over-dispersed)
        dups.append(N - len(set(grp)))
        leaks.append(sum(1 for x in grp if x in known))
        repl.extend(a if a != h else c for (a, b, c) in grp if (a != h) != (c != t))
    out = dict(n_pos=len(pos), n_neg=len(neg), same_side=np.array(same_side), heads=heads, dups=np.array(dups), leaks=np.array(leaks),
               rounds=np.array(st["rounds"]), repl=np.array(repl), buckets=kg["buckets"], n_ent=len(ents))
    _cache[key] = out
    return out


def _homogeneity_p(a_counts, b_counts):
    """Two-sample chi-square test of homogeneity on two count vectors over the same cells; cells with a small expected
    count are pooled so that the asymptotic distribution applies."""
    a, b = np.asarray(a_counts, float), np.asarray(b_counts, float)
    tot = a + b
    keep = tot >= 10
    if (~keep).any():
        a = np.append(a[keep], a[~keep].sum())
        b = np.append(b[keep], b[~keep].sum())
    nz = (a + b) > 0
    a, b = a[nz], b[nz]
    if len(a) < 2:
        return 1.0   # both samples sit in one cell: identical
    return float(sps.chi2_contingency(np.vstack([a, b]), correction=False)[1])


def _hist(x, n):
    return np.bincount(np.asarray(x, int), minlength=n)[:n]


def _all_p(A, B):
    """p-value per compared statistic."""
    N = A["n_neg"] // A["n_pos"]
    ps = {}
    ps["all_same_side"] = _homogeneity_p(_hist(A["same_side"], 2), _hist(B["same_side"], 2))
    ps["coin"] = _homogeneity_p([A["heads"], A["n_pos"] - A["heads"]], [B["heads"], B["n_pos"] - B["heads"]])
    ps["rounds"] = _homogeneity_p(_hist(A["rounds"], 12), _hist(B["rounds"], 12))
    ps["duplicates"] = _homogeneity_p(_hist(A["dups"], N + 1), _hist(B["dups"], N + 1))
    ps["leak"] = _homogeneity_p(_hist(A["leaks"], N + 1), _hist(B["leaks"], N + 1))   # known triples per positive
    w = max(1, A["n_ent"] // A["buckets"])
    ps["replacement_marginal"] = _homogeneity_p(_hist(A["repl"] // w, A["buckets"]), _hist(B["repl"] // w, B["buckets"]))
    return ps


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [1, 10, 25])
@pytest.mark.parametrize("kind", ["sparse", "dense"])
def test_philox_spec_matches_reference_sampler(kind, N):
    """code/base/batch.py:86-116 (as `mt_negatives`) vs the device's specification: same joint structure."""
    A, B = _draw(kind, "mt", N), _draw(kind, "philox", N)
    assert A["n_pos"] >= 2000
    ps = _all_p(A, B)
    bad = {k: v for k, v in ps.items() if v < P_ACCEPT}
    assert not bad, f"{kind} N={N}: the specification differs from the reference in {bad} (all: {ps})"
    # the workload does what it was built for
    if kind == "sparse":
        assert (A["rounds"] == 1).mean() > 0.99 and (B["rounds"] == 1).mean() > 0.99
        if N > 1:
            assert A["same_side"].mean() > 0.99 and B["same_side"].mean() > 0.99   # one coin per round, one round
    else:
        assert (A["rounds"] > 1).mean() > (0.4 if N == 1 else 0.9)      # known triples force re-draws
        assert A["leaks"].sum() > 0 and B["leaks"].sum() > 0                          # the last round is unfiltered
        if N > 1:
            assert A["dups"].sum() > 0 and B["dups"].sum() > 0            # rounds are independent: repeats across rounds


def test_two_reference_runs_agree_with_each_other():
    """Calibration: the same tests on two seeds of the SAME sampler (the reference's) accept."""
    for kind in ("sparse", "dense"):
        ps = _all_p(_draw(kind, "mt", 10, seed=5), _draw(kind, "mt", 10, seed=6))
        assert min(ps.values()) >= P_ACCEPT, (kind, ps)


@pytest.mark.parametrize("kind", ["sparse", "dense"])
def test_coin_per_slot_specification_is_rejected(kind):
    """A specification that flips the coin per negative instead of per round (what a 'uniform corruption' sampler does)
    differs from the reference exactly in the joint structure — the tests above must have the power to see it."""
    A = _draw(kind, "mt", 10)
    W = _draw(kind, "philox", 10, _coin_per_slot=True)
    ps = _all_p(A, W)
    assert ps["all_same_side"] < P_REJECT, ps
    # ... while its MARGINALS (what the chi-square / coin checks of tests/test_sampler_gpu.py look at) still pass:
    assert ps["coin"] >= P_ACCEPT and ps["replacement_marginal"] >= P_ACCEPT, ps


def test_invariants_hold_in_both():
    """relation kept, exactly one side replaced (or the triple unchanged by a self-replacement)."""
    for kind in ("sparse", "dense"):
        for which in ("mt", "philox"):
            d = _draw(kind, which, 25)
            assert d["n_neg"] == 25 * d["n_pos"]
            assert d["repl"].min() >= 0 and d["repl"].max() < d["n_ent"]
