"""GPU parity of the literal auto-encoder (native steps on the hand-written MFMA GEMMs with fused epilogues + HIP
optimizer step) against the float64 oracle (oracle/literal_oracle.py: hand-derived gradients, pinned to torch autograd in
tests/test_oracle_literal.py)."""
import numpy as np
import pytest

from oracle import literal_oracle as lo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("active", ["thah", "tanh"])
def test_training_steps_and_encoding(active):
    from multike_amd.literal_encoder import AutoEncoderModel
    from multike_amd.synthetic import synthetic_args
    rng = np.random.default_rng(1)
    dims = [60, 32, 16, 8]
    L, bs = 23, 10
    x = rng.standard_normal((L, 60)).astype(np.float32)
    args = synthetic_args(dim=8, batch_size=bs, learning_rate=0.05, encoder_active=active, encoder_normalize=True,
                          encoder_epoch=2)
    m = AutoEncoderModel(x.reshape(L, 3, 20), args, input_dimension=60, hidden_dimensions=[32, 16, 8])
    p = lo.init_params(dims, rng)
    for k in p:
        p[k] *= 0.4
    m.set_params(p)
    xn = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float64)
    acc = {k: np.full_like(v, 0.1) for k, v in p.items()}
    for epoch in range(2):
        got = m.train_one_epoch(epoch + 1)
        tot = 0.0
        for i in range(L // bs + 1):
            b = xn[i * bs:(i + 1) * bs]
            if len(b) == 0:
                continue
            l, g = lo.loss_and_grads(p, b, 3, active, True)
            lo.adagrad_step(p, acc, g, 0.05)
            tot += l
        np.testing.assert_allclose(got, tot + bs, rtol=2e-5)       # the reference's printed value adds batch_size
    mine = m.numpy_params()
    for k in p:
        np.testing.assert_allclose(mine[k], p[k], rtol=2e-3, atol=2e-5, err_msg=k)
    enc = m.encoder_multi_batches(x.reshape(L, 3, 20))
    assert enc.dtype == np.float64 and enc.shape == (L, 8)
    np.testing.assert_allclose(enc, lo.encode(p, x.astype(np.float64), 3, active), rtol=5e-3, atol=5e-4)


def test_literal_encoder_pipeline():
    from multike_amd.literal_encoder import LiteralEncoder
    from multike_amd.synthetic import synthetic_args
    rng = np.random.default_rng(2)
    words = [f"w{i}" for i in range(40)]
    w2v = {w: rng.standard_normal(12).astype(np.float32) for w in words}
    lits = [" ".join(rng.choice(words, rng.integers(1, 7))) for _ in range(57)] + ["unknown token"]
    args = synthetic_args(dim=6, batch_size=20, learning_rate=0.01, encoder_active="thah", encoder_normalize=True,
                          encoder_epoch=3)
    enc = LiteralEncoder(lits, w2v, args, tokens_max_len=5, word2vec_dimension=12)
    assert enc.encoded_literal_vector.shape == (58, 6) and np.all(np.isfinite(enc.encoded_literal_vector))


@pytest.mark.parametrize("dims,L,active,normalize", [([128, 64, 32, 16], 256, "thah", True),     # every product on the 16-byte path
                                                     ([128, 64, 32, 16], 256, "tanh", True),
                                                     ([128, 64, 32, 16], 256, "sigmoid", False),
                                                     ([150, 70, 33, 9], 77, "tanh", True),        # ragged: dword-load kernel
                                                     ([64, 20], 300, "thah", True),                # one layer
                                                     ([96, 64, 48, 32, 12], 130, "sigmoid", True)])  # four layers
def test_native_gradients_vs_oracle(dims, L, active, normalize):
    """Gradients of ONE batch left in the packed buffer by mke_ae_train_steps(update = 0): every weight and bias against
    the float64 oracle; loss; the zero-invariant scratch (partials) restored."""
    import torch
    from multike_amd import _lib
    from multike_amd.literal_encoder import AutoEncoderModel
    from multike_amd.synthetic import synthetic_args
    rng = np.random.default_rng(len(dims) * 1000 + L)
    n = len(dims) - 1
    x = rng.standard_normal((L, dims[0])).astype(np.float32)
    args = synthetic_args(dim=dims[-1], batch_size=L, learning_rate=0.05, encoder_active=active, encoder_normalize=normalize,
                          encoder_epoch=1)
    m = AutoEncoderModel(x, args, input_dimension=dims[0], hidden_dimensions=dims[1:])
    p = lo.init_params(dims, rng)
    for k in p:
        p[k] *= 0.3
    m.set_params(p)
    xin = m.word_vec_list                                   # row-normalised when encoder_normalize (as the model feeds it)
    x64 = xin.double().cpu().numpy()
    loss, g = lo.loss_and_grads(p, x64, n, active, normalize)
    m._ensure_scratch(L)
    m._plan.optimizer, m._plan.update, m._plan.lr = _lib.OPT_SGD, 0, 0.0
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    _lib.ae_train_steps(m._plan, xin, L, out)
    np.testing.assert_allclose(float(out[0]), loss, rtol=2e-5)
    for k in g:
        got = m._gviews[k].cpu().numpy()
        scale = np.abs(g[k]).max() + 1e-30
        assert np.abs(got - g[k]).max() / scale < 2e-4, (k, np.abs(got - g[k]).max() / scale)
    assert float(m._partials.abs().max()) == 0.0
    # pad entries of the packed buffer never receive a gradient
    total = float(m.grads.abs().sum())
    named = sum(float(v.abs().sum()) for v in m._gviews.values())
    assert abs(total - named) <= 1e-6 * max(1.0, total)


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_native_gradients_vs_the_reference_graph_executed(ci):
    """The device step's loss and gradients against the reference's own auto-encoder graph EXECUTED (tests/golden/make_golden.py
    `ae_graph_fixture`: code/literal_encoder.py:41-91 run unmodified, eager leaf ops, float64 autograd)."""
    import os
    import torch
    from conftest import GOLDEN
    from multike_amd import _lib
    from multike_amd.literal_encoder import AutoEncoderModel
    from multike_amd.synthetic import synthetic_args
    g = np.load(os.path.join(GOLDEN, "graphs_golden.npz"))
    pre = f"ae{ci}_"
    normalize, dims = bool(g[pre + "meta"][0]), [int(v) for v in g[pre + "meta"][1:]]
    active = str(g[pre + "active"])
    x = g[pre + "x"]
    L = x.shape[0]
    args = synthetic_args(dim=dims[-1], batch_size=L, learning_rate=0.05, encoder_active=active, encoder_normalize=False, encoder_epoch=1)
    # the graph is fed the rows as they are (the row normalisation of the inputs is the constructor's: code/literal_encoder.py:33-35)
    m = AutoEncoderModel(x.astype(np.float32), args, input_dimension=dims[0], hidden_dimensions=dims[1:])
    m._plan.normalize = int(normalize)                      # the graph's own batch-wide l2_normalize of the code matrix (:65-66)
    m.set_params({k[len(pre) + 2:]: g[k] for k in g.files if k.startswith(pre + "p_")})
    m._ensure_scratch(L)
    m._plan.optimizer, m._plan.update, m._plan.lr = _lib.OPT_SGD, 0, 0.0
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    _lib.ae_train_steps(m._plan, m.word_vec_list, L, out)
    np.testing.assert_allclose(float(out[0]), float(g[pre + "loss"]), rtol=2e-5)
    for k, got in m._gviews.items():
        want = g[pre + "g_" + k]
        scale = np.abs(want).max() + 1e-30
        assert np.abs(got.cpu().numpy() - want).max() / scale < 2e-4, (k, np.abs(got.cpu().numpy() - want).max() / scale)


def test_full_shape_step_runs_and_learns():
    """The reference's shape (code/literal_encoder.py:26-31: 1500 -> 1024 -> 512 -> 75, batches of 5000): a few native
    epochs on 10,300 literals (2 full batches + a short one) lower the loss; the encoding has the right shape."""
    import torch
    from multike_amd.literal_encoder import AutoEncoderModel
    from multike_amd.synthetic import synthetic_args
    g = torch.Generator(device="cpu"); g.manual_seed(0)
    base = torch.randn(40, 1500, generator=g)
    x = (base[torch.randint(0, 40, (10300,), generator=g)] + 0.05 * torch.randn(10300, 1500, generator=g)).numpy()
    args = synthetic_args(dim=75, batch_size=5000, learning_rate=0.01, encoder_active="thah", encoder_normalize=True, encoder_epoch=4)
    m = AutoEncoderModel(x.reshape(10300, 5, 300), args, seed=3)
    with torch.no_grad():
        m.params.mul_(0.02)                                  # N(0,1) weights at width 1500 overflow nothing but learn slowly
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        first = m.train_one_epoch(1)
        for e in range(2, 6):
            last = m.train_one_epoch(e)
    assert np.isfinite(first) and np.isfinite(last) and last < first
    with contextlib.redirect_stdout(io.StringIO()):
        enc = m.encoder_multi_batches(x[:6000].reshape(6000, 5, 300))
    assert enc.shape == (6000, 75) and enc.dtype == np.float64 and np.all(np.isfinite(enc))
    ref = m.encoder(torch.as_tensor(x[:64], device="cuda")).double().cpu().numpy()
    np.testing.assert_allclose(enc[:64], ref, rtol=1e-6, atol=1e-6)
