"""Three more pieces pinned to the reference EXECUTED in the build container (tests/golden/make_golden.py `pins_fixture`,
round-5 review item 5): the k-NN refresh (code/base/batch.py:119-150), the weighted view averaging weights
(code/MultiKE_Late.py:64-88) and the NumPy final encode of the literal auto-encoder (code/literal_encoder.py:114-144).
CPU part: the oracle restatements and the host-side NumPy path; GPU part: the kernels through the C-ABI."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pins():
    return np.load(os.path.join(HERE, "golden", "pins_golden.npz"))


# ---- CPU: oracle / host logic against the reference's outputs ---------------------------------------------------------------
def test_oracle_encode_is_the_references_final_encode(pins):
    """oracle/literal_oracle.encode == AutoEncoderModel.encoder_multi_batches run on injected weights."""
    from oracle import literal_oracle as lo
    p = {}
    for i in range(3):
        p[f"encoder_h{i}"], p[f"encoder_b{i}"] = pins[f"enc_w{i}"].astype(np.float64), pins[f"enc_b{i}"].astype(np.float64)
    for rows in (25, 30):
        for act in ("sigmoid", "tanh"):
            got = lo.encode(p, pins[f"enc_x{rows}"].astype(np.float64), 3, act)
            np.testing.assert_allclose(got, pins[f"enc_out{rows}_{act}"], rtol=2e-6, atol=1e-6)   # the reference computes in float32 (its tanh is (e^x - e^-x) / (e^x + e^-x): ~3e-7 absolute near zero)


def test_wva_weights_host_path(pins):
    """multike_amd.MultiKE_Late.wva on NumPy views == the reference's wva (mean cosine of a view with the views' average)."""
    import contextlib
    import io
    from multike_amd.MultiKE_Late import wva
    with contextlib.redirect_stdout(io.StringIO()):
        w = wva(*(pins[f"wva_view{i}"] for i in range(3)))
    np.testing.assert_allclose(np.asarray(w, dtype=np.float64), pins["wva_weights"], rtol=2e-6)


def test_knn_fixture_has_no_boundary_ties(pins):
    """The expected neighbour sets are exact: every row's k-th and (k+1)-th similarities are >= 2e-5 apart, and the float64 top-k
    of the stored matrix is the reference's table."""
    e, k = pins["knn_embeds"].astype(np.float64), int(pins["knn_k"])
    sim = e @ e.T
    srt = np.sort(sim, axis=1)[:, ::-1]
    assert float((srt[:, k - 1] - srt[:, k]).min()) > 2e-5
    ids = pins["knn_ids"]
    top = np.sort(ids[np.argsort(-sim, axis=1)[:, :k]], axis=1)
    assert np.array_equal(top, pins["knn_table"])


# ---- GPU: the kernels --------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_neighbour_table_is_the_references_generate_neighbours(pins):
    """mke_knn.hip through multike_amd.base.batch (neighbour_table and the dict-returning generate_neighbours): the SAME neighbour
    set as base.batch.generate_neighbours for every one of the 700 rows — exact, no slack."""
    from multike_amd.base.batch import generate_neighbours, neighbour_table
    e, ids, k = pins["knn_embeds"], pins["knn_ids"].tolist(), int(pins["knn_k"])
    table, valid = neighbour_table(e, ids, k, n_ent_total=max(ids) + 7)
    t = table.cpu().numpy()
    assert int(valid.sum()) == len(ids)
    got = np.sort(t[np.asarray(ids)], axis=1)
    assert np.array_equal(got, pins["knn_table"])
    dic = generate_neighbours(e, ids, k, 4)
    assert sorted(dic.keys()) == sorted(ids)
    assert all(sorted(dic[i]) == pins["knn_table"][r].tolist() for r, i in enumerate(ids))


@pytest.mark.gpu
def test_wva_weights_device_path(pins):
    import contextlib
    import io
    import torch
    from multike_amd.MultiKE_Late import wva
    with contextlib.redirect_stdout(io.StringIO()):
        w = wva(*(torch.as_tensor(pins[f"wva_view{i}"], device="cuda") for i in range(3)))
    np.testing.assert_allclose(np.asarray([float(x) for x in w]), pins["wva_weights"], rtol=5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("act", ["sigmoid", "tanh"])
@pytest.mark.parametrize("rows", [25, 30])
def test_encoder_multi_batches_is_the_references(pins, act, rows):
    """AutoEncoderModel.encoder_multi_batches (mke_ae_encode: the MFMA GEMM with bias + activation in its epilogue) with the
    reference's weights injected == the reference's NumPy encode, batch boundaries included (30 rows = 3 full batches of 10: the
    reference's last batch is then EMPTY)."""
    import contextlib
    import io
    import types
    from multike_amd.literal_encoder import AutoEncoderModel
    args = types.SimpleNamespace(dim=4, batch_size=10, encoder_active=act, encoder_normalize=False, optimizer="Adagrad", learning_rate=0.01)
    x = pins[f"enc_x{rows}"]
    with contextlib.redirect_stdout(io.StringIO()):
        m = AutoEncoderModel(x, args, input_dimension=12, hidden_dimensions=[8, 6, 4], seed=0)
        m.set_params({**{f"encoder_h{i}": pins[f"enc_w{i}"] for i in range(3)}, **{f"encoder_b{i}": pins[f"enc_b{i}"] for i in range(3)}})
        got = m.encoder_multi_batches(x)
    assert got.dtype == np.float64 and got.shape == (rows, 4)
    np.testing.assert_allclose(got, pins[f"enc_out{rows}_{act}"], rtol=2e-5, atol=2e-6)
