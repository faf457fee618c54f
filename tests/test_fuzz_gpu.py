"""A seeded slice of tools/fuzz_step.py in the suite: random row widths (every supported stride, ragged dims), 0..64 negatives,
tiny tables with heavy duplicates, zero rows, weights, SGD / Adagrad, un-normalised tables, exclusive-row path on and off —
two consecutive steps each, against the float64 oracles.  (The 2,000 + 500 case sweep of round 3 is in profiles/r03_fuzz.log.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_random_shapes_agree_with_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_step.py"), "120", "11"], capture_output=True, text=True,
                       timeout=550)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "relation step: 120 / 120" in r.stdout and "attribute step: 30 / 30" in r.stdout


@pytest.mark.timeout(600)
def test_random_shapes_of_the_surfaces_around_the_step():
    """tools/fuzz_aux.py: sampler bit-exact (random populations, id offsets, non-contiguous entity lists, neighbour tables,
    filter on / off), evaluator counts inside the fp32 band of the float64 ranks (row widths 1..320, duplicated gold columns),
    exact k-NN sets, common-space and space-mapping steps."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_aux.py"), "60", "13"], capture_output=True, text=True,
                       timeout=550)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    for line in ("sampler: 60 / 60", "evaluator: 30 / 30", "k-NN refresh: 15 / 15", "common-space step: 30 / 30",
                 "space-mapping step: 15 / 15"):
        assert line in r.stdout, r.stdout[-2000:]


@pytest.mark.timeout(900)
def test_random_datasets_and_hyper_parameters_track_the_whole_model_oracle():
    """tools/fuzz_model.py: ITC / SSL drivers on random synthetic datasets and hyper-parameters (batch sizes smaller and larger
    than the data, 1..25 negatives, gates, uniform / truncated sampling); every phase's epoch loss within 1e-4 of the float64
    whole-model oracle replaying the recorded batches."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_model.py"), "8", "17"], capture_output=True, text=True,
                       timeout=850)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "whole schedule: 8 / 8" in r.stdout
