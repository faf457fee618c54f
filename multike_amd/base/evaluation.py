"""base/evaluation.py surface of the reference (code/base/evaluation.py:6-27): `valid` and `test` — thin wrappers that
optionally push the first embedding set through a mapping matrix and hand over to the MFMA evaluator."""
import numpy as np

from .alignment import greedy_alignment


def _evaluate(source, target, mapping, top_k, threads_num, metric, normalize, csls_k, accurate, want_pairs=True):
    """Shared body: Hits@k / MR / MRR of `source` rows against `target` rows (gold = same index).  NumPy arrays or device
    tensors (the drivers hand over rows gathered on the device)."""
    projected = source if mapping is None else (np.matmul(source, mapping) if isinstance(source, np.ndarray) else source @ mapping)
    return greedy_alignment(projected, target, top_k, threads_num, metric, normalize, csls_k, accurate, want_pairs=want_pairs)


def valid(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False, csls_k=0, accurate=False):
    """-> (hits@1, MRR); quick mode by default, as the reference's validation."""
    _pairs, hits1, _mr, mrr = _evaluate(embeds1, embeds2, mapping, top_k, threads_num, metric, normalize, csls_k, accurate,
                                        want_pairs=False)
    return hits1, mrr


def test(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False, csls_k=0, accurate=True):
    """-> (aligned pairs, hits@1, MRR); accurate mode by default, as the reference's test."""
    pairs, hits1, _mr, mrr = _evaluate(embeds1, embeds2, mapping, top_k, threads_num, metric, normalize, csls_k, accurate)
    return pairs, hits1, mrr
