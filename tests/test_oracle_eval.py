"""The evaluator oracle against the reference's own greedy_alignment / calculate_rank output (eval_golden.npz)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import eval_oracle as eo


@pytest.mark.parametrize("ci", [0, 1])
def test_ranks_and_metrics(ci):
    g = np.load(os.path.join(GOLDEN, "eval_golden.npz"))
    pre = f"e{ci}_"
    rank, best = eo.ranks(g[pre + "e1"], g[pre + "e2"])
    hits, mr, mrr = eo.metrics(rank, g[pre + "top_k"])
    assert np.array_equal(hits, g[pre + "hits_accurate"])
    assert np.array_equal(hits, g[pre + "hits_quick"])       # argpartition mode gives the same Hits@k
    assert hits[0] == g[pre + "hits1"]
    np.testing.assert_allclose(mr, g[pre + "mr"], rtol=1e-12)
    np.testing.assert_allclose(mrr, g[pre + "mrr"], rtol=1e-12)
    rest = np.stack([np.arange(len(best)), best], 1)
    assert np.array_equal(rest, g[pre + "rest"])              # hits1_rest = {(gold, argmax)}
