"""base/alignment.py surface of the reference (code/base/alignment.py:8-79): `greedy_alignment` — Hits@k / MR / MRR of
the gold counterpart under the (normalised) inner-product similarity — on the f32 matrix cores via `mke_align_rank`.
The n1 x n2 similarity matrix is never materialised (the reference holds 60K x 60K fp32 = 14 GB and argsorts its rows
in `nums_threads` worker processes)."""
from __future__ import annotations

import time

import numpy as np
import torch

from .. import _lib


def _prep(x, device, normalize):
    t = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(device) if not isinstance(x, torch.Tensor) else x.to(device).float()
    if normalize:  # sklearn.preprocessing.normalize: zero rows stay zero (code/base/similarity.py:30-32)
        n = torch.linalg.norm(t, dim=1, keepdim=True)
        t = t / torch.where(n == 0, torch.ones_like(n), n)
    return t


def alignment_counts(embed1, embed2, normalize=True, device="cuda"):
    """(greater [n1] int64, ties [n1] int64, best [n1] int64): greater_i = #{j: sim_ij > sim_ii}, ties_i = #{j: sim_ij ==
    sim_ii} (the gold column included, so >= 1), best_i = argmax_j sim_ij."""
    a, b = _prep(embed1, device, normalize), _prep(embed2, device, normalize)
    n1, d = a.shape
    n2 = b.shape[0]
    if n2 < n1:
        raise _lib.MultiKEHipError("greedy_alignment: gold column = row index needs len(embed2) >= len(embed1)")
    if d > _lib.SIM_SELECT_KPADS[-1]:      # wider than the widest table the package supports (MKE_MAX_STRIDE): no second backend
        raise _lib.MultiKEHipError(f"greedy_alignment: rows of {d} floats exceed the widest k_align_rank instantiation "
                                   f"({_lib.SIM_SELECT_KPADS[-1]} = MKE_MAX_STRIDE)")
    kpad = min(x for x in _lib.SIM_SELECT_KPADS if x >= d)
    ap = torch.zeros(n1, kpad, dtype=torch.float32, device=device)
    ap[:, :d] = a
    bp = torch.zeros(n2, kpad, dtype=torch.float32, device=device)
    bp[:, :d] = b
    rank = torch.zeros(n1, dtype=torch.int32, device=device)
    ties = torch.zeros(n1, dtype=torch.int32, device=device)
    best = torch.zeros(n1, dtype=torch.int64, device=device)
    _lib.align_rank(ap, bp, kpad, n1, n2, rank, best, ties)
    col = 0xFFFFFFFF - (best & 0xFFFFFFFF)
    return rank.long(), ties.long().clamp_min(1), col


def alignment_ranks(embed1, embed2, normalize=True, device="cuda"):
    """(rank [n1] float64, best [n1] int64): rank_i = greater_i + (ties_i - 1) / 2 — the gold's EXPECTED 0-based position when
    the columns that tie with it are ordered at random.  The reference's argsort / argpartition leaves the gold at an arbitrary
    position among them (code/base/alignment.py:152-160).  Without ties this is the reference's rank exactly."""
    greater, ties, col = alignment_counts(embed1, embed2, normalize, device)
    return greater.double() + (ties.double() - 1.0) * 0.5, col


def tie_aware_metrics(greater, ties, top_k):
    """Expected Hits@k counts / MR / MRR over a uniformly random order of the columns tied with the gold: the gold's position
    is uniform on [greater, greater + ties - 1], so P(position < k) = clamp((k - greater) / ties, 0, 1) — a gold tied with one
    other column is HALF a Hits@1, not a whole one (a threshold on the mid-rank 0.5 < 1 would count it fully) — E[position + 1]
    = greater + (ties + 1) / 2, E[1 / (position + 1)] = (H(greater + ties) - H(greater)) / ties.  Degenerate inputs (a zero
    name vector ties with every column; duplicated embeddings) therefore score their chance level.  Without ties
    (ties == 1) these are exactly the reference's integer counts (code/base/alignment.py:141-163)."""
    g, t = greater.double(), ties.double()
    ks = torch.as_tensor([float(k) for k in top_k], dtype=torch.float64, device=g.device)
    hits = ((ks[:, None] - g[None, :]) / t[None, :]).clamp(0.0, 1.0).sum(1)
    mr = (g + (t + 1.0) * 0.5).mean()
    rr = 1.0 / (g + 1.0)                                                  # ties == 1: the reference's 1 / (rank + 1) exactly
    tied = ties > 1
    out = torch.cat([hits, mr[None], rr.sum()[None], tied.sum()[None].double()]).cpu().tolist()   # one read-back for all of them
    mrr_sum = out[-2]
    if out[-1] > 0:       # rows whose gold ties with other columns (degenerate inputs): harmonic-number difference, those rows only
        gt, tt = g[tied], t[tied]
        mrr_sum += float(((_harmonic(gt + tt) - _harmonic(gt)) / tt - rr[tied]).sum())
    return out[:-3], out[-3], mrr_sum / max(g.numel(), 1)


_EULER_GAMMA = 0.57721566490153286


def _harmonic(n):
    """H(n) = sum_{j=1..n} 1/j for a float64 tensor of non-negative integers: exact partial sums below 32, the asymptotic series
    ln n + gamma + 1/(2n) - 1/(12 n^2) + 1/(120 n^4) above (next term 1/(252 n^6) < 4e-12).  In place of
    torch.special.digamma(n + 1) + gamma, which this PyTorch build compiles at its first use in a process (0.1-1.7 s on a fresh box)."""
    table = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.float64), 1.0 / torch.arange(1, 32, dtype=torch.float64)]), 0).to(n.device)
    small = n < 32
    m = torch.where(small, torch.full_like(n, 32.0), n)
    i2 = 1.0 / (m * m)
    big = torch.log(m) + _EULER_GAMMA + 0.5 / m - i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0))
    return torch.where(small, table[torch.where(small, n, torch.zeros_like(n)).long()], big)


def greedy_alignment(embed1, embed2, top_k, nums_threads, metric, normalize, csls_k, accurate, want_pairs=True):
    """code/base/alignment.py:8-79.  Returns (alignment_rest, hits1, mr, mrr).  `nums_threads` is accepted and ignored
    (one kernel launch).  Only the path the reference uses is built: inner product (or cosine == inner product of
    normalised rows), csls_k == 0.  want_pairs = False (base.evaluation.valid, which drops them): alignment_rest is None —
    the set of (row, best column) tuples is a Python object per row."""
    if csls_k and csls_k > 0:
        raise _lib.MultiKEHipError("greedy_alignment: CSLS re-scoring is not built (the reference never enables it)")
    if not (metric == "inner" or (metric == "cosine" and normalize)):
        raise _lib.MultiKEHipError(f"greedy_alignment: metric {metric!r} is not built (the reference uses 'inner')")
    assert 1 in top_k
    t = time.time()
    greater, ties, best = alignment_counts(embed1, embed2, normalize)
    num = greater.numel()
    hits, mr, mrr = tie_aware_metrics(greater, ties, top_k)
    hits = np.round(np.array(hits) / num * 100, 3)
    alignment_rest = set(zip(range(num), best.cpu().tolist())) if want_pairs else None
    cost = time.time() - t
    if accurate:
        print("accurate results: hits@{} = {}%, mr = {:.3f}, mrr = {:.6f}, time = {:.3f} s ".format(top_k, hits, mr, mrr, cost))
    else:
        print("quick results: hits@{} = {}%, time = {:.3f} s ".format(top_k, hits, cost))
    return alignment_rest, hits[0], mr, mrr
