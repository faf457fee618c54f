"""GPU parity of mke_neg_sample / the known-triple set: BIT-EXACT against the Philox specification
(oracle/sampler_oracle.py, oracle/mke_oracle.c) and the reference's invariants."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import sampler_oracle as so

pytestmark = pytest.mark.gpu


def _toy(sampler_golden):
    g = sampler_golden
    t1 = np.array(g["triples1"], np.int32)
    k1 = np.array(g["known1"], np.int32)
    return g, t1, k1


def test_hash_set_membership(sampler_golden):
    from gpu_util import dev_i32
    from multike_amd.sampling import KnownTripleSet
    g, t1, k1 = _toy(sampler_golden)
    ks = KnownTripleSet(dev_i32(k1[:, 0]), dev_i32(k1[:, 1]), dev_i32(k1[:, 2]))
    assert bool(ks.contains(dev_i32(k1[:, 0]), dev_i32(k1[:, 1]), dev_i32(k1[:, 2])).all())
    rng = np.random.default_rng(0)
    q = np.stack([rng.integers(0, 50, 4000), rng.integers(0, 6, 4000), rng.integers(0, 50, 4000)], 1).astype(np.int32)
    known = {tuple(x) for x in k1.tolist()}
    exp = np.array([tuple(x) in known for x in q.tolist()])
    got = ks.contains(dev_i32(q[:, 0]), dev_i32(q[:, 1]), dev_i32(q[:, 2])).cpu().numpy()
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("N,use_known,use_near", [(5, True, False), (10, True, True), (25, False, False),
                                                  (40, True, True), (1, True, False)])
def test_bit_exact_vs_python_spec_small(sampler_golden, N, use_known, use_near):
    from gpu_util import dev_i32
    from multike_amd.sampling import KGSide, KnownTripleSet, sample_negatives
    g, t1, k1 = _toy(sampler_golden)
    known_py = {tuple(x) for x in k1.tolist()} if use_known else None
    ks = KnownTripleSet(dev_i32(k1[:, 0]), dev_i32(k1[:, 1]), dev_i32(k1[:, 2])) if use_known else None
    side = KGSide(g["ents1"], ks)
    cand_table = cand_valid = None
    if use_near:
        rng = np.random.default_rng(4)
        K = 44
        cand_table = np.stack([rng.choice(50, K, replace=False) for _ in range(50)]).astype(np.int32)
        cand_valid = (np.arange(50) % 3 != 0).astype(np.uint8)
        side.set_neighbours(dev_i32(cand_table), torch.as_tensor(cand_valid, device="cuda"))
    pos = tuple(dev_i32(t1[:, k]) for k in range(3))
    got = sample_negatives(pos, side, N, seed=(123, 456), stream_id=7, pos_offset=1000)
    exp = so.philox_negatives(t1[:, 0], t1[:, 1], t1[:, 2], N, 50, ent_lo=0, cand_table=cand_table,
                              cand_valid=cand_valid, known=known_py, seed=(123, 456), stream_id=7, pos_offset=1000)
    for a, b in zip(got, exp):
        assert np.array_equal(a.cpu().numpy(), b)


def test_bit_exact_full_batch_and_invariants():
    """C2 batch shape (2.5K positives of one KG, N=25, 100K candidates, ~900K known triples) against the C
    restatement, plus the reference's invariants on the device output."""
    from gpu_util import dev_i32
    from multike_amd.sampling import KGSide, KnownTripleSet, sample_negatives
    from multike_amd.synthetic import SyntheticKGs
    kgs = SyntheticKGs()
    t = kgs.triples[1]
    ks = KnownTripleSet(dev_i32(t[:, 0]), dev_i32(t[:, 1]), dev_i32(t[:, 2]))
    side = KGSide(kgs.entities(1), ks)
    P, N = 2461, 25
    p = t[5000:5000 + P]
    pos = tuple(dev_i32(p[:, k]) for k in range(3))
    got = [x.cpu().numpy() for x in sample_negatives(pos, side, N, seed=(9, 1), stream_id=3, pos_offset=77)]
    cts = co.TripleSet(t[:, 0], t[:, 1], t[:, 2])
    lo, hi = kgs.ent_range[1]
    exp = co.neg_sample(p[:, 0], p[:, 1], p[:, 2], N, hi - lo, ent_lo=lo, known=cts, seed=(9, 1), stream_id=3,
                        pos_offset=77)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)
    nh, nr, nt = got
    H, R, T = np.repeat(p[:, 0], N), np.repeat(p[:, 1], N), np.repeat(p[:, 2], N)
    assert np.array_equal(nr, R)                                   # relation never corrupted
    assert np.all((nh == H) | (nt == T))                           # at most one side differs
    assert nh.min() >= lo and nh.max() < hi and nt.min() >= lo and nt.max() < hi   # same-KG candidates
    key = (nh.astype(np.int64) << 38) | (nt.astype(np.int64) << 12) | nr
    tk = (t[:, 0].astype(np.int64) << 38) | (t[:, 2].astype(np.int64) << 12) | t[:, 1]
    assert np.isin(key, tk).mean() < 1e-3                          # filtered (only a last round could leak)
    side_head = (nh != H).reshape(P, N)
    assert 0.45 < side_head.mean() < 0.55                          # fair coin
    # uniformity of the corrupted entity over the KG (chi-square on 50 bins)
    corrupted = np.where(nh != H, nh, nt) - lo
    cnt = np.bincount(corrupted * 50 // (hi - lo), minlength=50)
    chi2 = ((cnt - cnt.mean()) ** 2 / cnt.mean()).sum()
    assert chi2 < 100, chi2  # 49 dof: P(chi2 > 100) ~ 2e-5


@pytest.mark.parametrize("N,n", [(30, 33), (12, 14), (15, 16), (16, 19), (32, 34), (50, 52), (63, 64), (64, 66)])
def test_small_population_many_duplicates(N, n):
    """population barely larger than the sample: exercises the without-replacement fix-up path — in every group shape of the
    kernel (16 / 32 / 64 lanes per positive, with and without an idle lane for the coin block)."""
    from gpu_util import dev_i32
    from multike_amd.sampling import KGSide, sample_negatives
    rng = np.random.default_rng(1)
    P = 300
    ph, pr, pt = rng.integers(0, n, P), rng.integers(0, 4, P), rng.integers(0, n, P)
    side = KGSide(np.arange(n), None)
    got = [x.cpu().numpy() for x in sample_negatives(tuple(dev_i32(a) for a in (ph, pr, pt)), side, N, seed=(2, 2))]
    exp = co.neg_sample(ph, pr, pt, N, n, seed=(2, 2))
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)
    nh, nt = got[0].reshape(P, N), got[2].reshape(P, N)
    corrupt = np.where(nh != ph[:, None], nh, nt)
    # no filter => one round => all N drawn without replacement: distinct unless the draw hit the original entity
    for i in range(P):
        side_h = (nh[i] != ph[i]).any()
        vals = nh[i] if side_h else nt[i]
        assert len(set(vals.tolist())) == N


def test_argument_errors():
    from gpu_util import dev_i32
    from multike_amd import _lib
    from multike_amd.sampling import KGSide, sample_negatives
    pos = tuple(dev_i32([1, 2]) for _ in range(3))
    with pytest.raises(_lib.MultiKEHipError, match="smaller than neg_per_pos"):
        sample_negatives(pos, KGSide(np.arange(5), None), 10)   # random.sample would raise ValueError
    with pytest.raises(_lib.MultiKEHipError, match="neg_per_pos"):
        sample_negatives(pos, KGSide(np.arange(500), None), 65)


def test_sample_distinct_bit_exact_and_distinct():
    """mke_sample_distinct (random.sample batching of the cross-KG / common-space loops) vs the oracle restatement,
    plus the properties that define it: in range, distinct inside a step, a full permutation when batch == n,
    steps differ, deterministic."""
    from multike_amd import _lib
    from oracle.sampler_oracle import distinct_sample
    for n, b, steps, seed, stream in ((10, 10, 3, (1, 2), 3), (470_000, 5000, 5, (7, 0x4D4B45), 1), (7, 3, 5, (0, 0), 0),
                                      (1, 1, 2, (5, 5), 9), (2 ** 20 + 1, 1000, 2, (123, 456), 77), (65_537, 4096, 3, (9, 9), 2)):
        got = _lib.sample_distinct(n, b, steps, seed, stream).cpu().numpy()
        want = distinct_sample(n, b, steps, seed, stream)
        np.testing.assert_array_equal(got, want)
        assert got.min() >= 0 and got.max() < n
        assert all(len(set(row.tolist())) == b for row in got)
        again = _lib.sample_distinct(n, b, steps, seed, stream).cpu().numpy()
        np.testing.assert_array_equal(got, again)
    full = _lib.sample_distinct(200_000, 200_000, 2, (3, 4), 5).cpu().numpy()
    assert np.array_equal(np.sort(full[0]), np.arange(200_000)) and np.array_equal(np.sort(full[1]), np.arange(200_000))
    assert (full[0] != full[1]).mean() > 0.99
    # uniform: every position about equally likely over many steps (chi-square-ish bound)
    many = _lib.sample_distinct(64, 8, 4000, (11, 12), 13).cpu().numpy()
    cnt = np.bincount(many.ravel(), minlength=64)
    assert abs(cnt - 500).max() < 5 * np.sqrt(500)
    with pytest.raises(_lib.MultiKEHipError, match="batch <= n"):
        _lib.sample_distinct(5, 6, 1)
    assert _lib.sample_distinct(5, 0, 3).shape == (3, 0)


def test_fast_and_plain_kernel_forms_are_the_same_stream():
    """`sampler_fast` (coin block in the draw evaluation's idle lane, LDS duplicate table, 16-lane groups) is a
    performance option: the output is the plain form's, bit for bit, at every group shape, with and without the known-triple
    filter and the truncated-sampling candidate table."""
    from gpu_util import dev_i32
    from multike_amd import _lib
    from multike_amd.sampling import KGSide, KnownTripleSet, sample_negatives
    rng = np.random.default_rng(11)
    n, P = 700, 1999
    tri = np.stack([rng.integers(0, n, 6000), rng.integers(0, 9, 6000), rng.integers(0, n, 6000)], axis=1).astype(np.int32)
    ks = KnownTripleSet(dev_i32(tri[:, 0]), dev_i32(tri[:, 1]), dev_i32(tri[:, 2]))
    pos = tuple(dev_i32(tri[:P, k]) for k in range(3))
    try:
        for N in (1, 7, 15, 16, 25, 31, 32, 33, 63, 64):
            for near in (False, True):
                side = KGSide(np.arange(n), ks)
                if near:
                    K = 70
                    tab = np.stack([rng.choice(n, K, replace=False) for _ in range(n)]).astype(np.int32)
                    side.set_neighbours(dev_i32(tab), torch.as_tensor((np.arange(n) % 4 != 0).astype(np.uint8), device="cuda"))
                outs = []
                for fast in (0, 1):
                    _lib.set_option("sampler_fast", fast)
                    outs.append([x.cpu().numpy() for x in sample_negatives(pos, side, N, seed=(5, N), stream_id=2, pos_offset=31)])
                for a, b in zip(*outs):
                    assert np.array_equal(a, b), (N, near)
    finally:
        _lib.set_option("sampler_fast", 1)
