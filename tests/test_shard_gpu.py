"""GPU checks of the row-set bookkeeping kernels of the sharded step (index work: exact)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G,n_ent,lens", [(1, 5000, (700, 700, 7000, 7000)), (8, 200_000, (5000, 5000, 125_000, 125_000)),
                                          (3, 1000, (0, 10, 0, 3)), (8, 64, (200, 200, 200, 200))])
def test_rowset_build_remap_gather_scatter(G, n_ent, lens):
    from multike_amd import _lib
    rng = np.random.default_rng(sum(lens) + G)
    streams_np = [rng.integers(0, n_ent, n).astype(np.int32) for n in lens]
    streams = [torch.as_tensor(x, device="cuda") for x in streams_np]
    uniq = np.unique(np.concatenate(streams_np))
    per_owner = [uniq[uniq % G == o] for o in range(G)]
    C = max(len(x) for x in per_owner) + 5
    i32 = dict(dtype=torch.int32, device="cuda")
    flags, id_map, counts, overflow = torch.zeros(n_ent, **i32), torch.zeros(n_ent, **i32), torch.zeros(G, **i32), torch.zeros(1, **i32)
    req = torch.full((G * C,), -1, **i32)
    _lib.rowset_build(streams, flags, counts, req, id_map, overflow, G, C)
    assert int(overflow) == 0
    assert counts.cpu().tolist() == [len(x) for x in per_owner]
    rq = req.cpu().numpy().reshape(G, C)
    im = id_map.cpu().numpy()
    for o in range(G):
        n = len(per_owner[o])
        assert set((rq[o, :n].astype(np.int64) * G + o).tolist()) == set(per_owner[o].tolist())   # each distinct id once
        assert np.all(rq[o, n:] == -1)
    # id_map inverts req
    slots = im[uniq]
    assert np.array_equal(rq.reshape(-1)[slots].astype(np.int64) * G + slots // C, uniq)
    # remap + flags cleared
    outs = [torch.empty_like(x) for x in streams]
    probe_req, probe_cnt = torch.ones(37, **i32), torch.ones(G, **i32)
    _lib.rowset_remap(streams, outs, id_map, flags, probe_req, probe_cnt)
    assert int((probe_req != -1).sum()) == 0 and int(probe_cnt.abs().sum()) == 0
    for x_np, out in zip(streams_np, outs):
        assert np.array_equal(out.cpu().numpy(), im[x_np])
    assert int(flags.abs().sum()) == 0
    # owner side: gather padded rows / scatter-add them back
    stride, dim = 80, 75
    table = torch.randn(n_ent // G + 1, stride, device="cuda")
    table[:, dim:] = 0
    want = req  # with G ranks each owner would receive its own column; here rank 0 plays every owner
    rows = torch.empty(G * C, stride, device="cuda")
    scratch = torch.ones(G * C, stride, device="cuda")
    _lib.rows_gather_padded(table, want, rows, scratch)
    assert float(scratch.abs().max()) == 0.0
    w = want.cpu().numpy()
    exp = np.zeros((G * C, stride), np.float32)
    exp[w >= 0] = table.cpu().numpy()[w[w >= 0]]
    assert np.array_equal(rows.cpu().numpy(), exp)
    grad = torch.zeros_like(table)
    touched = torch.zeros(table.shape[0], **i32)
    want_before = want.clone()
    _lib.rows_scatter_add(want_before, rows, dim, grad, touched, 7, req, counts)
    assert int((req != -1).sum()) == 0 and int(counts.abs().sum()) == 0      # re-initialised for the next step
    ge = np.zeros((table.shape[0], stride), np.float32)
    np.add.at(ge, w[w >= 0], exp[w >= 0])
    np.testing.assert_allclose(grad.cpu().numpy(), ge, rtol=1e-6, atol=1e-6)
    tt = np.zeros(table.shape[0], np.int32)
    tt[w[w >= 0]] = 7
    assert np.array_equal(touched.cpu().numpy(), tt)


def test_rowset_overflow_is_flagged():
    from multike_amd import _lib
    i32 = dict(dtype=torch.int32, device="cuda")
    ids = torch.arange(0, 100, **i32)
    flags, id_map, counts, overflow = torch.zeros(100, **i32), torch.zeros(100, **i32), torch.zeros(2, **i32), torch.zeros(1, **i32)
    req = torch.full((2 * 10,), -1, **i32)
    _lib.rowset_build([ids], flags, counts, req, id_map, overflow, 2, 10)
    assert int(overflow) == 1 and int(id_map.max()) < 20
