"""CPU ORACLE (test infrastructure, not product code) — the alignment evaluator.

Restates code/base/evaluation.py:6-14 -> code/base/alignment.py:8-79,141-163 -> code/base/similarity.py:30-34:
row-normalise both embedding sets, sim = E1 . E2^T, gold column of row i = i, rank = position of the gold in the
descending order, Hits@k = share of rows with rank < k (in percent, rounded to 3 decimals), MR = mean(rank+1),
MRR = mean(1/(rank+1)).  Pinned by tests/golden/eval_golden.npz, produced by executing the reference's own
greedy_alignment (tests/golden/make_golden.py).
"""
import numpy as np


def normalize_rows(x):
    """sklearn.preprocessing.normalize (l2): zero rows stay zero."""
    n = np.linalg.norm(x, axis=1, keepdims=True)
    return x / np.where(n == 0, 1.0, n)


def ranks(embed1, embed2, normalize=True):
    a = normalize_rows(embed1) if normalize else embed1
    b = normalize_rows(embed2) if normalize else embed2
    sim = a @ b.T
    gold = sim[np.arange(len(a)), np.arange(len(a))]
    return np.sum(sim > gold[:, None], axis=1), np.argmax(sim, axis=1)


def metrics(rank, top_k):
    hits = np.array([np.sum(rank < k) for k in top_k]) / len(rank) * 100
    return np.round(hits, 3), float(np.mean(rank + 1)), float(np.mean(1.0 / (rank + 1)))
