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
    # ids beyond an owner's capacity go to the PAD row behind the compact row set (index G*C), never to another entity's slot
    m = id_map.cpu().numpy()
    assert int(overflow) == 1 and int(m.max()) == 20 and int((m == 20).sum()) == 80
    real = np.sort(m[m < 20])
    assert np.array_equal(real, np.arange(20))                      # the 20 slots that exist are each used exactly once


@pytest.mark.parametrize("G,n_local,C,dim", [(1, 3000, 1500, 75), (8, 2500, 900, 75), (3, 700, 700, 20), (16, 64, 40, 256)])
def test_owner_reduce_update_equals_scatter_then_update(G, n_local, C, dim):
    """mke_rowset_remap's inversion + mke_rows_update_multi(slot_of) == mke_rows_scatter_add + touched-row update:
    each owner row sums what the G ranks sent for it and is updated once; slots are consumed (reset to -1)."""
    from multike_amd import _lib
    from multike_amd.tables import ADAGRAD_INIT_ACC
    rng = np.random.default_rng(G * 1000 + n_local)
    stride = _lib.stride_for(dim)
    want_np = np.full((G, C), -1, np.int32)
    for g in range(G):                                     # every rank asks for a distinct subset of the local rows
        k = int(rng.integers(0, min(C, n_local) + 1))
        want_np[g, rng.permutation(C)[:k]] = rng.permutation(n_local)[:k]
    want = torch.as_tensor(want_np.reshape(-1), device="cuda")
    rows = torch.randn(G * C, stride, device="cuda") * 0.1
    rows[:, dim:] = 0
    rows[want < 0] = 0
    t0 = torch.randn(n_local, stride, device="cuda")
    t0[:, dim:] = 0
    i32 = dict(dtype=torch.int32, device="cuda")
    # --- two-kernel path
    ta, aa = t0.clone(), torch.full_like(t0, ADAGRAD_INIT_ACC)
    grad, touched = torch.zeros_like(t0), torch.zeros(n_local, **i32)
    _lib.rows_scatter_add(want, rows, dim, grad, touched, 5)
    _lib.rows_update(ta, aa, grad, touched, 5, dim, True, _lib.OPT_ADAGRAD, 0.05)
    # --- inverted requests, one launch
    slot_of = torch.full((n_local * G,), -1, **i32)
    _lib.rowset_remap([], [], None, None, None, None, want, slot_of, G, C)
    so = slot_of.cpu().numpy().reshape(n_local, G)
    for g in range(G):
        k = np.nonzero(want_np[g] >= 0)[0]
        assert np.array_equal(so[want_np[g, k], g], k)
    assert int((so >= 0).sum()) == int((want_np >= 0).sum())
    tb, ab = t0.clone(), torch.full_like(t0, ADAGRAD_INIT_ACC)
    _lib.rows_update_multi([dict(table=tb, acc=ab, normalize=True, src_rows=rows, slot_of=slot_of, n_ranks=G, capacity=C)],
                           5, stride, dim, _lib.OPT_ADAGRAD, 0.05)
    assert int((slot_of != -1).sum()) == 0
    hit = np.zeros(n_local, bool)
    hit[want_np[want_np >= 0]] = True
    assert np.array_equal(tb.cpu().numpy()[~hit], t0.cpu().numpy()[~hit])        # rows nobody asked for: bit-identical
    np.testing.assert_allclose(tb.cpu().numpy(), ta.cpu().numpy(), rtol=2e-5, atol=2e-6)   # summation order differs
    np.testing.assert_allclose(ab.cpu().numpy(), aa.cpu().numpy(), rtol=2e-5, atol=2e-6)
