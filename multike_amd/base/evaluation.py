"""base/evaluation.py surface of the reference (code/base/evaluation.py:6-27)."""
import numpy as np

from .alignment import greedy_alignment


def valid(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False, csls_k=0, accurate=False):
    if mapping is not None:
        embeds1 = np.matmul(embeds1, mapping)
    _, hits1_12, mr_12, mrr_12 = greedy_alignment(embeds1, embeds2, top_k, threads_num, metric, normalize, csls_k, accurate)
    return hits1_12, mrr_12


def test(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False, csls_k=0, accurate=True):
    if mapping is not None:
        embeds1 = np.matmul(embeds1, mapping)
    alignment_rest_12, hits1_12, mr_12, mrr_12 = greedy_alignment(embeds1, embeds2, top_k, threads_num, metric, normalize,
                                                                  csls_k, accurate)
    return alignment_rest_12, hits1_12, mrr_12
