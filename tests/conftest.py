import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def losses_golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "losses_golden.npz"))


@pytest.fixture(scope="session")
def sampler_golden():
    import json
    with open(os.path.join(GOLDEN, "sampler_golden.json")) as f:
        return json.load(f)


def golden_cases(g):
    n = 0
    while f"c{n}_meta" in g.files:
        n += 1
    return list(range(n))
