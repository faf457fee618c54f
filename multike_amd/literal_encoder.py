"""literal_encoder.py surface of the reference (code/literal_encoder.py): `AutoEncoderModel`, `LiteralEncoder` — the
dense literal auto-encoder that produces the `[#literals, dim]` value vectors of the attribute view (pre-processing,
runs once; §8 row M1).

This is the one GEMM-shaped part of the system: 1500 -> 1024 -> 512 -> dim -> 512 -> 1024 -> 1500 on 5000-row batches
(~126 GFLOP per training step).  They are plain dense GEMMs, so they go to the library (rocBLAS / hipBLASLt through
torch: fp32 on the matrix cores); the optimizer step over the packed parameter buffer is the HIP `mke_dense_update`
kernel (TF1 Adagrad: acc0 = 0.1, no epsilon).  Parameters live in ONE packed float32 buffer, like the CNN's.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import _lib
from .tables import ADAGRAD_INIT_ACC

_OPT = {"Adagrad": _lib.OPT_ADAGRAD, "SGD": _lib.OPT_SGD}


class AutoEncoderModel:
    def __init__(self, word_vec_list, args, input_dimension=1500, hidden_dimensions=None, device="cuda", seed=None):
        """code/literal_encoder.py:19-39."""
        self.args = args
        self.session = None
        self.device = torch.device(device)
        self.input_dimension = input_dimension
        hidden = list(hidden_dimensions) if hidden_dimensions is not None else [1024, 512, self.args.dim]
        self.layer_num = len(hidden)
        self.hidden_dimensions = [input_dimension] + hidden
        x = np.reshape(np.asarray(word_vec_list, dtype=np.float32), [len(word_vec_list), input_dimension])
        if self.args.encoder_normalize:  # sklearn preprocessing.normalize: zero rows stay zero (:33-35)
            n = np.linalg.norm(x, axis=1, keepdims=True)
            x = x / np.where(n == 0, 1.0, n)
        self.word_vec_list = torch.as_tensor(x, device=self.device)
        if self.args.optimizer not in _OPT and self.args.optimizer not in _lib.DENSE_OPTS:
            raise _lib.MultiKEHipError(f"optimizer {self.args.optimizer!r}: Adagrad, SGD, Adam or Adadelta")
        self._init_graph(seed)

    def _init_graph(self, seed):
        """code/literal_encoder.py:41-61: every weight and bias ~ N(0, 1) (tf.random_normal_initializer)."""
        hds, n = self.hidden_dimensions, self.layer_num
        shapes = []
        for i in range(n):
            shapes += [(f"encoder_h{i}", (hds[i], hds[i + 1])), (f"encoder_b{i}", (hds[i + 1],))]
        for i in range(n):
            j = n - i
            shapes += [(f"decoder_h{i}", (hds[j], hds[j - 1])), (f"decoder_b{i}", (hds[j - 1],))]
        total = sum(int(np.prod(s)) for _, s in shapes)
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(int(seed))
        self.params = torch.randn(total, generator=g).to(self.device)
        self.grads = torch.zeros_like(self.params)
        self.acc = torch.full_like(self.params, ADAGRAD_INIT_ACC)
        self.weights, self.biases, self._gviews = {}, {}, {}
        o = 0
        for name, shape in shapes:
            k = int(np.prod(shape))
            view = self.params[o:o + k].view(shape).requires_grad_(False)
            (self.weights if "_h" in name else self.biases)[name] = view
            self._gviews[name] = self.grads[o:o + k].view(shape)
            o += k

    def set_params(self, p: dict):
        for k, v in p.items():
            (self.weights if "_h" in k else self.biases)[k].copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32))

    def numpy_params(self) -> dict:
        return {k: v.detach().cpu().numpy().copy() for k, v in {**self.weights, **self.biases}.items()}

    def _act(self, x):
        if self.args.encoder_active == 'sigmoid':
            return torch.sigmoid(x)
        if self.args.encoder_active == 'tanh':
            return torch.tanh(x)
        return x  # any other string (the shipped "thah") selects no activation (:75-78)

    def encoder(self, input_data, W=None, B=None):
        W, B = W or self.weights, B or self.biases
        h = input_data
        for i in range(self.layer_num):
            h = self._act(torch.addmm(B[f"encoder_b{i}"], h, W[f"encoder_h{i}"]))
        return h

    def decoder(self, input_data, W=None, B=None):
        W, B = W or self.weights, B or self.biases
        h = input_data
        for i in range(self.layer_num):
            h = self._act(torch.addmm(B[f"decoder_b{i}"], h, W[f"decoder_h{i}"]))
        return h

    def train_step(self, batch: torch.Tensor) -> torch.Tensor:
        """loss + optimizer of one batch (code/literal_encoder.py:63-69): library GEMMs forward and backward, HIP
        update over the packed parameters."""
        leaves = {k: v.detach().requires_grad_(True) for k, v in {**self.weights, **self.biases}.items()}
        W = {k: v for k, v in leaves.items() if "_h" in k}
        B = {k: v for k, v in leaves.items() if "_b" in k}
        code = self.encoder(batch, W, B)
        if self.args.encoder_normalize:  # tf.nn.l2_normalize with no axis: the whole matrix (:65-66)
            code = code * torch.rsqrt(torch.clamp_min(torch.sum(code * code), 1e-12))
        dec = self.decoder(code, W, B)
        loss = torch.mean(torch.square(dec - batch))
        names = list(leaves)
        grads = torch.autograd.grad(loss, [leaves[k] for k in names])
        for k, g in zip(names, grads):
            self._gviews[k].copy_(g)
        if self.args.optimizer in _lib.DENSE_OPTS:
            if getattr(self, "_dense", None) is None:
                self._dense = [torch.zeros_like(self.params), torch.zeros_like(self.params), 0]
            self._dense[2] += 1
            _lib.dense_update_opt(self.params, self._dense[0], self._dense[1], self.grads,
                                  _lib.optimizer_struct(self.args.optimizer, float(self.args.learning_rate), self._dense[2]))
            return loss.detach()
        opt = _OPT[self.args.optimizer]
        _lib.dense_update(self.params, self.acc if opt == _lib.OPT_ADAGRAD else None, self.grads, opt,
                          float(self.args.learning_rate))
        return loss.detach()

    def train_one_epoch(self, epoch):
        """code/literal_encoder.py:93-112.  `num_batch = L // batch_size + 1`, so the last slice is empty when L is a
        multiple of the batch size; TF would feed it and produce a NaN loss — it is skipped here.  The printed value
        keeps the reference's `loss_sum += batch_size` (:108)."""
        start_time = time.time()
        bs = self.args.batch_size
        L = self.word_vec_list.shape[0]
        loss_sum = torch.zeros((), device=self.device)
        for i in range(L // bs + 1):
            batch = self.word_vec_list[i * bs:(i + 1) * bs]
            if batch.shape[0] == 0:
                continue
            loss_sum = loss_sum + self.train_step(batch)
        loss_sum = float(loss_sum) + self.args.batch_size
        print('epoch {} of literal encoder, loss: {:.4f}, time: {:.4f}s'.format(epoch, loss_sum, time.time() - start_time))
        return loss_sum

    def encoder_multi_batches(self, input_data):
        """code/literal_encoder.py:114-144: forward of the encoder only, on the inputs AS GIVEN (not row-normalised),
        no output normalisation, float64 result."""
        print('encode literal embeddings...', len(input_data))
        x = torch.as_tensor(np.reshape(np.asarray(input_data, dtype=np.float32), [len(input_data), self.input_dimension]),
                            device=self.device)
        bs = self.args.batch_size
        out = [self.encoder(x[i:i + bs]) for i in range(0, x.shape[0], bs)]
        res = torch.cat(out, 0).double().cpu().numpy() if out else np.zeros((0, self.args.dim))
        print("encoded literal embeddings", res.shape)
        return res


class LiteralEncoder:
    """code/literal_encoder.py:159-180: literal -> up to `tokens_max_len` word vectors -> flattened 1500-d input ->
    auto-encoder trained for `encoder_epoch` epochs -> encoded vectors.  Words missing from `word2vec` get the vector
    `char_embedder(word)` when a callable is supplied (the reference trains character embeddings with gensim for them,
    code/utils.py:94-230 — host-side, out of scope) and a zero vector otherwise."""

    def __init__(self, literal_list, word2vec, args, tokens_max_len=5, word2vec_dimension=300, char_embedder=None,
                 device="cuda"):
        self.args = args
        self.literal_list = literal_list
        self.word2vec = word2vec
        self.tokens_max_len = tokens_max_len
        self.word2vec_dimension = word2vec_dimension
        vecs = np.zeros((len(literal_list), tokens_max_len, word2vec_dimension), dtype=np.float32)
        for li, literal in enumerate(literal_list):
            words = literal.split(' ')
            for i in range(min(tokens_max_len, len(words))):
                v = word2vec.get(words[i])
                if v is None and char_embedder is not None:
                    v = char_embedder(words[i])
                if v is not None:
                    vecs[li, i] = v
        model = AutoEncoderModel(vecs, args, input_dimension=tokens_max_len * word2vec_dimension, device=device)
        for i in range(args.encoder_epoch):
            model.train_one_epoch(i + 1)
        self.encoder_model = model
        self.encoded_literal_vector = model.encoder_multi_batches(vecs)
