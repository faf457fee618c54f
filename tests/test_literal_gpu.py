"""GPU parity of the literal auto-encoder (library GEMMs + HIP optimizer step) against the float64 oracle."""
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
