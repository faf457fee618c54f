"""Placement by trial of big row arrays (multike_amd/tables.py placed_rows; the probe kernel mke_probe_rows)."""
import numpy as np
import pytest
import torch


def test_small_arrays_and_cpu_arrays_are_plain_allocations():
    from multike_amd import tables
    log = []
    x = tables.placed_rows(7, 16, "cpu", 0.1, log)
    assert x.shape == (7, 16) and float(x.min()) == float(x.max()) == np.float32(0.1) and log == []


@pytest.mark.gpu
def test_probe_reads_the_rows_it_is_given():
    from multike_amd import _lib
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    a, b, c = (torch.randn(500, 80, device="cuda", generator=g) for _ in range(3))
    idx = torch.randint(0, 500, (3000,), device="cuda", generator=g, dtype=torch.int32)
    out = torch.empty(3000, device="cuda")
    _lib.probe_rows(a, None, None, idx, out)
    np.testing.assert_allclose(out.cpu().numpy(), a[idx.long()].sum(1).cpu().numpy(), rtol=1e-4, atol=1e-4)
    _lib.probe_rows(a, b, c, idx, out)
    np.testing.assert_allclose(out.cpu().numpy(), (a + b + c)[idx.long()].sum(1).cpu().numpy(), rtol=1e-4, atol=2e-4)


@pytest.mark.gpu
def test_placed_rows_keeps_one_candidate_and_fills_it(monkeypatch):
    """The threshold lowered to 1 MB so that a 4 MB array goes through the search: candidates are probed, exactly one survives,
    it holds the fill value, and an EmbeddingTable built on it trains like any other (same values as an unplaced one)."""
    from multike_amd import tables
    monkeypatch.setenv("MKE_PLACE_MIN_MB", "1")
    monkeypatch.setenv("MKE_PLACE_TRIES", "4")
    log = []
    x = tables.placed_rows(16384, 80, "cuda", 0.1, log)
    assert x.shape == (16384, 80) and float(x.min()) == float(x.max()) == np.float32(0.1)
    assert len(log) == 1 and 1 <= len(log[0]["probe_us"]) <= 4 and 0 <= log[0]["kept"] < len(log[0]["probe_us"])
    assert all(t > 0 for t in log[0]["probe_us"])
    vals = np.random.default_rng(0).standard_normal((16384, 75)).astype(np.float32)
    placed = tables.EmbeddingTable(16384, 75, "t", values=vals)
    monkeypatch.setenv("MKE_PLACE", "0")
    plain = tables.EmbeddingTable(16384, 75, "t", values=vals)
    assert torch.equal(placed.data, plain.data) and torch.equal(placed.slot("o"), plain.slot("o")) and torch.equal(placed.grad, plain.grad)
