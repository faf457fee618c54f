"""GPU parity of the MFMA alignment evaluator (mke_align_rank) and the k-NN refresh."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import eval_oracle as eo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ci", [0, 1])
def test_golden_metrics_from_reference(ci):
    from multike_amd.base.alignment import greedy_alignment
    from multike_amd.base import evaluation as eva
    g = np.load(os.path.join(GOLDEN, "eval_golden.npz"))
    pre = f"e{ci}_"
    top_k = g[pre + "top_k"].tolist()
    rest, hits1, mr, mrr = greedy_alignment(g[pre + "e1"], g[pre + "e2"], top_k, 8, "inner", True, 0, True)
    assert hits1 == g[pre + "hits1"]
    np.testing.assert_allclose(mr, g[pre + "mr"], rtol=1e-9)
    np.testing.assert_allclose(mrr, g[pre + "mrr"], rtol=1e-9)
    assert np.array_equal(np.array(sorted(rest)), g[pre + "rest"])
    h1, mrr2 = eva.valid(g[pre + "e1"], g[pre + "e2"], None, top_k, 8, normalize=True)
    assert h1 == g[pre + "hits1"]


@pytest.mark.parametrize("n1,n2,d", [(1000, 1777, 75), (33, 33, 4), (4097, 6000, 256), (500, 501, 100)])
def test_ranks_vs_oracle(n1, n2, d):
    """Ranks are integers: exact except where two similarities are within fp32 rounding of each other."""
    from multike_amd.base.alignment import alignment_ranks
    rng = np.random.default_rng(n1 + d)
    e2 = rng.standard_normal((n2, d)).astype(np.float32)
    e1 = (0.5 * e2[:n1] + rng.standard_normal((n1, d))).astype(np.float32)
    rank, best = alignment_ranks(e1, e2)
    r64, b64 = eo.ranks(e1.astype(np.float64), e2.astype(np.float64))
    got = rank.cpu().numpy()
    assert np.mean(got == r64) > 0.995 and np.max(np.abs(got - r64)) <= 2
    assert np.mean(best.cpu().numpy() == b64) > 0.995
    # a row's own similarity never counts (gold taken from the same MFMA computation): perfect alignment -> rank 0
    rank0, best0 = alignment_ranks(e2[:n1], e2)
    assert int(rank0.max()) == 0 and np.array_equal(best0.cpu().numpy(), np.arange(n1))


def test_rows_wider_than_the_widest_instantiation():
    """dim 257..320 (the tables' widest stride, MKE_MAX_STRIDE): the 320-float instantiation of the same MFMA sweep — no
    library GEMM anywhere in the evaluator; wider rows cannot come from a table of this package and are an error."""
    from multike_amd import _lib
    from multike_amd.base.alignment import alignment_counts
    rng = np.random.default_rng(3)
    for d in (257, 300, 320):
        e2 = rng.standard_normal((900, d)).astype(np.float32)
        e1 = (0.3 * e2[:700] + rng.standard_normal((700, d))).astype(np.float32)
        e2[5] = e2[4]                                                          # golds 4 and 5 tie with each other
        greater, ties, best = alignment_counts(e1, e2)
        r64, b64 = eo.ranks(e1.astype(np.float64), e2.astype(np.float64))
        assert np.mean(greater.cpu().numpy() == r64) > 0.99 and np.max(np.abs(greater.cpu().numpy() - r64)) <= 2
        assert int(ties[4]) == 2 and int(ties[5]) == 2 and int((ties != 1).sum()) == 2
        assert np.mean(best.cpu().numpy() == b64) > 0.99
        rank0, _, best0 = alignment_counts(e2[:700], e2)                       # gold from the same fma chain: never counts itself
        assert int(rank0.max()) == 0
    with pytest.raises(_lib.MultiKEHipError):
        alignment_counts(np.zeros((4, 321), np.float32), np.zeros((4, 321), np.float32))


def test_neighbour_table_of_320_float_rows():
    """k-NN refresh at the widest supported row (k_sim_sample / k_sim_select at kpad 320): exact top-k sets vs float64."""
    from multike_amd.base.batch import neighbour_table
    rng = np.random.default_rng(9)
    n, d, k = 600, 300, 12
    e = rng.standard_normal((n, d)).astype(np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    table, valid = neighbour_table(e, list(range(n)), k, n)
    sim = e.astype(np.float64) @ e.astype(np.float64).T
    got = np.sort(table.cpu().numpy(), axis=1)
    exp = np.sort(np.argpartition(-sim, k - 1, axis=1)[:, :k], axis=1)
    assert np.mean(got == exp) > 0.995 and int(valid.sum()) == n


@pytest.mark.parametrize("rows,n,k", [(3, 70_000, 2000), (40, 5000, 1), (7, 4097, 4097), (5, 300, 17), (2, 100_003, 64)])
def test_topk_long_is_the_exact_top_k_in_column_order(rows, n, k):
    """mke_topk_long (whole similarity rows, any length): the k largest columns, the k-th value's ties taken in column order,
    written in column order — against a stable float64 sort; rows with heavy ties and with negative / zero values."""
    from multike_amd import _lib
    rng = np.random.default_rng(rows * 1000 + k)
    v = rng.standard_normal((rows, n)).astype(np.float32)
    v[0] = np.round(v[0], 1)                       # many exact ties around the threshold
    if rows > 1:
        v[1] = 0.0                                 # every column ties: the first k columns win
    out = _lib.topk_long(torch.as_tensor(v, device="cuda"), k).cpu().numpy()
    for r in range(rows):
        order = np.argsort(-v[r].astype(np.float64), kind="stable")[:k]      # largest first, ties by column
        assert np.array_equal(out[r], np.sort(order)), r
    # a strided view (rows of a wider matrix) and the id map
    wide = torch.as_tensor(np.concatenate([v, v], 1), device="cuda")
    ids = torch.arange(n, dtype=torch.int32, device="cuda") * 3 + 1
    out2 = _lib.topk_long(wide[:, :n], k, id_map=ids).cpu().numpy()
    assert np.array_equal(out2, out * 3 + 1)


def test_ties_are_ranked_at_mid_rank():
    """The reference's argsort leaves the gold at an arbitrary place among the columns that tie with it; the evaluator
    reports the mid-rank: a zero row (similarity 0 to every column) lands in the middle, not at Hits@1; a duplicated gold
    column shares ranks 0 and 1."""
    from multike_amd.base.alignment import alignment_ranks, greedy_alignment
    rng = np.random.default_rng(5)
    n, d = 200, 75
    e2 = rng.standard_normal((n, d)).astype(np.float32)
    e1 = e2.copy()
    e1[7] = 0.0                                   # no name vector: sim = 0 against all n columns
    e2[11] = e2[10]                               # column 11 duplicates column 10: row 10's gold ties with column 11
    e1[11] = e2[10]                               # ... and row 11's gold with column 10
    rank, _ = alignment_ranks(e1, e2)
    r = rank.cpu().numpy()
    assert r[7] == (n - 1) / 2.0                  # n columns tie (the gold among them): mid-rank
    assert r[10] == 0.5                           # two columns tie for the first place
    assert r[3] == 0.0
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        _, hits1, mr, mrr = greedy_alignment(e1, e2, [1, 10], 1, "inner", True, 0, True)
    # expected values over a random order of the tied columns: rows 10 and 11 are half a Hits@1 each (their golds tie with one
    # other column), row 7 is 1/n of one; every other row is a whole one
    exp_hits1 = (n - 3 + 0.5 + 0.5 + 1.0 / n) / n * 100
    assert hits1 == round(exp_hits1, 3)
    H = lambda m: sum(1.0 / i for i in range(1, m + 1))
    exp_mrr = (n - 3 + 2 * (1 + 0.5) / 2 + H(n) / n) / n
    assert abs(mrr - exp_mrr) < 1e-12
    exp_mr = (n - 3 + 2 * 1.5 + (n + 1) / 2) / n
    assert abs(mr - exp_mr) < 1e-12


def test_neighbour_table_matches_reference_definition():
    """top-k inner products per row, self included, unordered (code/base/batch.py:143-150)."""
    from multike_amd.base.batch import generate_neighbours, neighbour_table
    rng = np.random.default_rng(0)
    n, d, k = 700, 20, 15
    e = rng.standard_normal((n, d)).astype(np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    ids = (np.arange(n) * 3 + 5).tolist()
    table, valid = neighbour_table(e, ids, k, n_ent_total=3 * n + 10)
    sim = e.astype(np.float64) @ e.astype(np.float64).T
    t = table.cpu().numpy()
    srt = np.sort(sim, axis=1)[:, ::-1]
    clear = np.nonzero(srt[:, k - 1] - srt[:, k] > 2e-5)[0]     # rows whose k-th and (k+1)-th similarities are not a float32 tie
    assert len(clear) > 0.98 * n
    for i in clear:                                              # EXACT sets (the reference-run table: tests/test_pins_golden.py)
        exp = set(np.asarray(ids)[np.argpartition(-sim[i], k)[:k]].tolist())
        got = set(t[ids[i]].tolist())
        assert got == exp and ids[i] in got
    assert int(valid.sum()) == n
    dic = generate_neighbours(e, ids, k, 4)
    assert set(dic.keys()) == set(ids) and len(dic[ids[0]]) == k


def test_neighbour_table_threshold_path_is_exact():
    """Long rows, k = 2 %: the sample-threshold + fused similarity/compaction + short top-k path must return exactly the
    top-k set of a full-width top-k (rows of clustered data so that similarities are far from uniform), including the
    fallback rows."""
    import torch
    from multike_amd.base.batch import neighbour_table
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    n, d, k = 40_000, 32, 800
    centers = torch.randn(50, d, device="cuda", generator=g)
    e = centers[torch.randint(0, 50, (n,), device="cuda", generator=g)] + 0.7 * torch.randn(n, d, device="cuda", generator=g)
    e = torch.nn.functional.normalize(e, dim=1)
    ids = list(range(n))
    table, valid = neighbour_table(e, ids, k, n)
    assert int(valid.sum()) == n
    table2, _ = neighbour_table(e, ids, k, n)
    assert torch.equal(table, table2)                      # same input, same table (order included)
    table3, _ = neighbour_table(e, ids, k, n, rows_per_launch=16_384)
    assert torch.equal(table, table3)                      # several row ranges per refresh (KGs beyond 2^29 / cap rows)
    tol = 3e-6                                             # the kernel's f32 fma chain vs the float64 reference
    for lo in range(0, n, 10_000):                         # EVERY row
        sim = e[lo:lo + 10_000].double() @ e.double().t()
        kth = torch.topk(sim, k, dim=1).values[:, -1:]
        t = table[lo:lo + 10_000].long()
        ts = t.sort(dim=1).values
        assert bool((ts[:, 1:] != ts[:, :-1]).all())                                 # k distinct columns
        got = sim.gather(1, t)
        assert float((kth - got).max()) <= tol                                       # nothing below the k-th value
        assert torch.equal((got > kth + tol).sum(1), (sim > kth + tol).sum(1))       # everything clearly above it
        assert bool((t == torch.arange(lo, lo + t.shape[0], device="cuda")[:, None]).any(dim=1).all())   # self included


@pytest.mark.parametrize("rows,n_seg,seg_cap,k", [(37, 1, 4096, 123), (64, 4, 512, 700), (5, 8, 64, 1), (9, 2, 100, 200)])
def test_topk_rows_exact_with_ties_and_status(rows, n_seg, seg_cap, k):
    """mke_topk_rows: the k largest of a segmented short list, ties at the k-th value broken by list order, output in
    list order, k-th value reported; short / overflowed rows flagged."""
    import torch
    from multike_amd import _lib
    g = torch.Generator(device="cuda"); g.manual_seed(rows * 7 + k)
    vals = torch.randn(rows, n_seg, seg_cap, device="cuda", generator=g)
    vals = torch.round(vals * 8) / 8                      # plenty of exact ties
    idx = torch.randint(0, 1 << 30, (rows, n_seg, seg_cap), device="cuda", generator=g, dtype=torch.int32)
    cnt = torch.randint(max(1, seg_cap // 2), seg_cap + 1, (rows, n_seg), device="cuda", generator=g, dtype=torch.int32)
    cnt[0] = seg_cap
    if rows > 3:
        cnt[1, 0] = seg_cap + 5                           # overflow -> status 2
        cnt[2] = 0                                        # nothing -> status 1
    out, kth, status = _lib.topk_rows(vals, k, idx=idx, seg_count=cnt, want_kth=True)
    st = status.cpu().numpy()
    for r in range(rows):
        c = cnt[r].cpu().numpy()
        if (c > seg_cap).any():
            assert st[r] == 2
            continue
        lst_v = np.concatenate([vals[r, s, :c[s]].cpu().numpy() for s in range(n_seg)])
        lst_i = np.concatenate([idx[r, s, :c[s]].cpu().numpy() for s in range(n_seg)])
        if len(lst_v) < k:
            assert st[r] == 1
            continue
        assert st[r] == 0
        kv = np.sort(lst_v)[::-1][k - 1]
        assert float(kth[r]) == float(kv)
        take = lst_v > kv
        ties = np.nonzero(lst_v == kv)[0][:k - int(take.sum())]
        take[ties] = True
        assert np.array_equal(out[r].cpu().numpy(), lst_i[take])
    # plain list (no idx, no counts): positions, and the threshold-only mode
    v2 = torch.randn(11, 4096, device="cuda", generator=g)
    out2, kth2, st2 = _lib.topk_rows(v2, 50, want_kth=True)
    ref = torch.topk(v2, 50, dim=1)
    assert int(st2.abs().sum()) == 0 and torch.equal(kth2, ref.values[:, -1])
    assert torch.equal(out2.long().sort(dim=1).values, ref.indices.sort(dim=1).values)
    _, kth3, _ = _lib.topk_rows(v2, 50, want_idx=False, want_kth=True)
    assert torch.equal(kth3, kth2)


@pytest.mark.parametrize("n,d,lo,hi,n_seg", [(5000, 75, 0, 5000, 4), (3333, 20, 1000, 1777, 1), (9000, 256, 128, 1000, 2),
                                              (700, 100, 0, 700, 8), (2500, 120, 300, 2500, 2), (1500, 200, 0, 1500, 1)])
def test_sim_select_candidates(n, d, lo, hi, n_seg):
    """mke_sim_select: per row the columns with similarity above the row's threshold, per column segment, in column
    order, with their similarities (f32 MFMA chain vs a float64 product: 1e-5 band around the threshold)."""
    import torch
    from multike_amd import _lib
    g = torch.Generator(device="cuda"); g.manual_seed(n + d)
    e = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
    kpad = min(x for x in _lib.SIM_SELECT_KPADS if x >= d)
    ep = torch.zeros(n, kpad, device="cuda")
    ep[:, :d] = e
    sim = (e[lo:hi].double() @ e.double().t())
    tau = sim.float().quantile(0.97, dim=1).contiguous()
    seg_cap = 512 // n_seg
    cand, cnt = _lib.sim_select(ep, kpad, lo, hi, tau, n_seg, seg_cap)
    cidx, csim = cand[..., 0].contiguous(), cand[..., 1].contiguous().view(torch.float32)
    bn = 64 if kpad <= 208 else 32
    tiles = (n + bn - 1) // bn
    per = (tiles + n_seg - 1) // n_seg * bn
    cidx, csim, cnt = cidx.cpu().numpy(), csim.cpu().numpy(), cnt.cpu().numpy()
    sim_np, tau_np = sim.cpu().numpy(), tau.cpu().numpy().astype(np.float64)
    for r in list(range(0, hi - lo, 37)) + [hi - lo - 1]:
        for s in range(n_seg):
            a, b = s * per, min(n, (s + 1) * per)
            if a >= b:
                assert cnt[r, s] == 0
                continue
            sure = np.nonzero(sim_np[r, a:b] > tau_np[r] + 1e-5)[0] + a
            maybe = np.nonzero(sim_np[r, a:b] > tau_np[r] - 1e-5)[0] + a
            c = int(cnt[r, s])
            assert len(sure) <= c <= len(maybe)
            got = cidx[r, s, :min(c, seg_cap)]
            assert np.all(np.diff(got) > 0)                                   # column order, no duplicates
            if c <= seg_cap:
                assert set(sure.tolist()) <= set(got.tolist()) <= set(maybe.tolist())
            np.testing.assert_allclose(csim[r, s, :len(got)], sim_np[r, got], rtol=0, atol=2e-6)


@pytest.mark.parametrize("n,d,lo,hi,ns", [(3000, 75, 0, 3000, 4096), (1000, 20, 100, 777, 100), (900, 256, 0, 900, 65), (500, 120, 0, 500, 64)])
def test_sim_sample_matches_product_and_select(n, d, lo, hi, ns):
    """mke_sim_sample = rows x sample-rows similarities (float64 product within 2e-6), and bit-identical to what
    mke_sim_select computes for the same (row, column) pairs (same fma chains)."""
    import torch
    from multike_amd import _lib
    g = torch.Generator(device="cuda"); g.manual_seed(n + ns)
    e = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
    kpad = min(x for x in _lib.SIM_SELECT_KPADS if x >= d)
    ep = torch.zeros(n, kpad, device="cuda")
    ep[:, :d] = e
    cols = torch.randint(0, n, (ns,), device="cuda", generator=g)
    out = _lib.sim_sample(ep, kpad, lo, hi, ep[cols].contiguous())
    ref = e[lo:hi].double() @ e[cols].double().t()
    assert out.shape == (hi - lo, ns)
    assert float((out.double() - ref).abs().max()) < 2e-6
    # the main pass reproduces these numbers exactly: select everything (tau = -2) in one segment, compare by column
    if n <= 1000:
        cand, cnt = _lib.sim_select(ep, kpad, lo, hi, torch.full((hi - lo,), -2.0, device="cuda"), 1, 1024)
        assert int(cnt.min()) == n and int(cnt.max()) == n
        sims = cand[:, 0, :n, 1].contiguous().view(torch.float32)           # column order = column index
        assert torch.equal(sims[:, cols], out)


@pytest.mark.timeout(600)
def test_evaluator_at_full_size_properties_and_spot_checks():
    """SURVEY f2's size: 60K x 60K x 75 (the matrix the reference materialises as 14 GB).  Size-independent properties —
    perfect alignment ranks every row first; a rotation of the columns changes no rank (what the rank-sharded evaluator of
    multike_amd/distributed_run.py relies on) — plus 64 rows checked against the float64 oracle on all 60K columns."""
    from multike_amd.base.alignment import alignment_counts
    rng = np.random.default_rng(60)
    n, d = 60_000, 75
    e2 = rng.standard_normal((n, d)).astype(np.float32)
    greater, ties, best = alignment_counts(e2, e2)
    assert int(greater.max()) == 0 and int(ties.max()) == 1 and np.array_equal(best.cpu().numpy(), np.arange(n))
    e1 = (0.35 * e2 + rng.standard_normal((n, d))).astype(np.float32)
    g1, t1, b1 = alignment_counts(e1, e2)
    lo, hi = 17_000, 24_500                                   # a block of rows against the rotated columns
    rot = np.concatenate([e2[lo:hi], e2[:lo], e2[hi:]])
    g2, t2, _ = alignment_counts(e1[lo:hi], rot)
    assert torch.equal(g2, g1[lo:hi]) and torch.equal(t2, t1[lo:hi])
    rows = rng.choice(n, 64, replace=False)
    a = eo.normalize_rows(e1[rows].astype(np.float64))
    b = eo.normalize_rows(e2.astype(np.float64))
    sim = a @ b.T
    exp = np.sum(sim > sim[np.arange(64), rows][:, None], axis=1)
    got = g1.cpu().numpy()[rows]
    assert np.mean(got == exp) >= 0.95 and np.max(np.abs(got - exp)) <= 2          # integer ranks; fp32 near-ties may flip one
    assert 0.05 < float((g1 == 0).double().mean()) < 0.95                                # neither trivial nor hopeless


@pytest.mark.timeout(900)
def test_knn_refresh_at_full_size_and_its_sharded_form():
    """SURVEY f3's size: 100K entities, k = int(0.02 * 100K) = 2000 neighbours.  Every row contains itself; 24 rows checked
    against the float64 definition (top-k inner products, unordered); the four slices of `neighbour_table(part=(r, 4))` — what
    each of four ranks computes in the multi-GPU refresh — reassemble the single-GPU table exactly."""
    from multike_amd.base.batch import neighbour_table
    rng = np.random.default_rng(100)
    n, d, k = 100_000, 75, 2000
    centres = rng.standard_normal((400, d))
    e = (centres[rng.integers(0, 400, n)] + 0.8 * rng.standard_normal((n, d))).astype(np.float32)     # clustered: real neighbourhoods
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    ids = np.arange(n) + 7
    table, valid = neighbour_table(e, ids, k, n + 7)
    t = table.cpu().numpy()
    assert int(valid.sum()) == n and int(valid[:7].sum()) == 0
    assert bool((table[7:] == torch.arange(7, n + 7, device=table.device, dtype=torch.int32)[:, None]).any(dim=1).all())
    e64 = e.astype(np.float64)
    for i in rng.choice(n, 24, replace=False):
        sim = e64 @ e64[i]
        chosen = t[i + 7].astype(np.int64) - 7
        assert len(set(chosen.tolist())) == k
        out = np.ones(n, dtype=bool)
        out[chosen] = False
        assert sim[chosen].min() >= sim[out].max() - 2e-6, i                                          # a true top-k set up to fp32 ties
    full = torch.zeros_like(table)
    seen = torch.zeros(n + 7, dtype=torch.int32, device=table.device)
    for r in range(4):
        tp, vp, ip = neighbour_table(e, ids, k, n + 7, part=(r, 4))
        full[ip] = tp[ip]
        seen[ip] += 1
        assert int(vp.sum()) == ip.numel()
    assert int(seen[7:].min()) == 1 and int(seen[7:].max()) == 1 and int(seen[:7].sum()) == 0
    # the same top-k SETS (the order inside a row is unspecified, and identical here because the kernels are deterministic)
    assert torch.equal(torch.sort(full[7:], dim=1).values, torch.sort(table[7:], dim=1).values)
