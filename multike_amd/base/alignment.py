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


def alignment_ranks(embed1, embed2, normalize=True, device="cuda"):
    """(rank [n1] float64, best [n1] int64): rank_i = #{j: sim_ij > sim_ii} + (#{j: sim_ij == sim_ii} - 1) / 2; best_i =
    argmax_j sim_ij.  Ties: the reference's argsort / argpartition leaves the gold at an arbitrary position among the columns
    that tie with it (code/base/alignment.py:152-160); the MID-rank is reported here, so degenerate inputs — a zero name
    vector has similarity 0 to every column, duplicated embeddings — do not count as Hits@1 (counting only strictly
    greater columns would resolve every tie in the gold's favour).  Without ties this is the reference's rank exactly."""
    a, b = _prep(embed1, device, normalize), _prep(embed2, device, normalize)
    n1, d = a.shape
    n2 = b.shape[0]
    if n2 < n1:
        raise _lib.MultiKEHipError("greedy_alignment: gold column = row index needs len(embed2) >= len(embed1)")
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
    return rank.double() + (ties.double() - 1.0).clamp_min(0.0) * 0.5, col


def greedy_alignment(embed1, embed2, top_k, nums_threads, metric, normalize, csls_k, accurate):
    """code/base/alignment.py:8-79.  Returns (alignment_rest, hits1, mr, mrr).  `nums_threads` is accepted and ignored
    (one kernel launch).  Only the path the reference uses is built: inner product (or cosine == inner product of
    normalised rows), csls_k == 0."""
    if csls_k and csls_k > 0:
        raise _lib.MultiKEHipError("greedy_alignment: CSLS re-scoring is not built (the reference never enables it)")
    if not (metric == "inner" or (metric == "cosine" and normalize)):
        raise _lib.MultiKEHipError(f"greedy_alignment: metric {metric!r} is not built (the reference uses 'inner')")
    assert 1 in top_k
    t = time.time()
    rank, best = alignment_ranks(embed1, embed2, normalize)
    num = rank.numel()
    hits = np.array([float((rank < k).sum()) for k in top_k]) / num * 100
    hits = np.round(hits, 3)
    mr = float((rank + 1).mean())
    mrr = float((1.0 / (rank + 1)).mean())
    alignment_rest = set(zip(range(num), best.cpu().tolist()))
    cost = time.time() - t
    if accurate:
        print("accurate results: hits@{} = {}%, mr = {:.3f}, mrr = {:.6f}, time = {:.3f} s ".format(top_k, hits, mr, mrr, cost))
    else:
        print("quick results: hits@{} = {}%, time = {:.3f} s ".format(top_k, hits, cost))
    return alignment_rest, hits[0], mr, mrr
