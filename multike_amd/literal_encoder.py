"""literal_encoder.py surface of the reference (code/literal_encoder.py): `AutoEncoderModel`, `LiteralEncoder` — the
dense literal auto-encoder that produces the `[#literals, dim]` value vectors of the attribute view (pre-processing,
runs once; §8 row M1).

This is the one GEMM-shaped part of the system: 1500 -> 1024 -> 512 -> dim -> 512 -> 1024 -> 1500 on 5000-row batches
(~126 GFLOP per training step).  Forward, hand-derived backward and the optimizer step of a whole epoch are ONE native
call (`mke_ae_train_steps`): every product runs on the hand-written f32 MFMA GEMM of `csrc/mke_gemm.hip` with the layer's
bias / activation / loss / normalisation partial sums / bias-gradient column sums fused into its epilogue — no
torch.autograd, no library GEMM.  Parameters live in ONE packed float32 buffer (tensor starts padded to 16 bytes).
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
        """code/literal_encoder.py:41-61: every weight and bias ~ N(0, 1) (tf.random_normal_initializer).  Packed
        buffer: encoder_h0 | encoder_b0 | ... | decoder_h0 | decoder_b0 | ..., each tensor's start padded to a multiple of
        4 floats (pad entries are zero and have a zero gradient for ever)."""
        hds, n = self.hidden_dimensions, self.layer_num
        if n > _lib.AE_MAX_LAYERS:
            raise _lib.MultiKEHipError(f"at most {_lib.AE_MAX_LAYERS} encoder layers")
        shapes = []
        for i in range(n):
            shapes += [(f"encoder_h{i}", (hds[i], hds[i + 1])), (f"encoder_b{i}", (hds[i + 1],))]
        for i in range(n):
            j = n - i
            shapes += [(f"decoder_h{i}", (hds[j], hds[j - 1])), (f"decoder_b{i}", (hds[j - 1],))]
        pad4 = lambda v: (int(v) + 3) // 4 * 4
        offs, o = {}, 0
        for name, shape in shapes:              # weight [in][out] stored with row stride pad4(out); bias [out]
            offs[name] = o
            o += shape[0] * pad4(shape[1]) if len(shape) == 2 else pad4(shape[0])
        total = o
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(int(seed))
        host = torch.zeros(total)
        self.params = host.to(self.device)
        self.grads = torch.zeros_like(self.params)
        self.acc = torch.full_like(self.params, ADAGRAD_INIT_ACC)
        self.weights, self.biases, self._gviews = {}, {}, {}

        def view_of(buf, name, shape):
            if len(shape) == 2:
                ld = pad4(shape[1])
                return buf[offs[name]:offs[name] + shape[0] * ld].view(shape[0], ld)[:, :shape[1]]
            return buf[offs[name]:offs[name] + shape[0]]

        for name, shape in shapes:
            view = view_of(self.params, name, shape)
            view.copy_(torch.randn(shape, generator=g).to(self.device))
            (self.weights if "_h" in name else self.biases)[name] = view
            self._gviews[name] = view_of(self.grads, name, shape)
        # native plan
        p = _lib.AEPlanStruct()
        p.n_layers = n
        for i, w in enumerate(hds):
            p.dims[i] = int(w)
        p.act = {"sigmoid": _lib.ACT_SIGMOID, "tanh": _lib.ACT_TANH}.get(self.args.encoder_active, _lib.ACT_NONE)  # :75-78
        p.normalize = int(bool(self.args.encoder_normalize))
        f32 = torch.float32
        p.params, p.grads, p.acc = (_lib.ptr(t, f32, "ae") for t in (self.params, self.grads, self.acc))
        p.n_params = total
        for i in range(n):
            p.w_off[i], p.b_off[i] = offs[f"encoder_h{i}"], offs[f"encoder_b{i}"]
            p.w_off[n + i], p.b_off[n + i] = offs[f"decoder_h{i}"], offs[f"decoder_b{i}"]
        self._partials = torch.zeros(3 * _lib.LOSS_PARTIALS, dtype=torch.float64, device=self.device)
        self._scalars = torch.zeros(4, dtype=f32, device=self.device)
        p.partials, p.scalars = _lib.ptr(self._partials, torch.float64, "partials"), _lib.ptr(self._scalars, f32, "scalars")
        self._plan, self._scratch = p, None
        self._dense = None

    def _ensure_scratch(self, rows: int):
        need = _lib.ae_scratch_floats(self._plan, rows)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float32, device=self.device)
        self._plan.scratch = _lib.ptr(self._scratch, torch.float32, "scratch")
        self._plan.scratch_floats = self._scratch.numel()

    def set_params(self, p: dict):
        for k, v in p.items():
            (self.weights if "_h" in k else self.biases)[k].copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32))

    def numpy_params(self) -> dict:
        return {k: v.detach().cpu().numpy().copy() for k, v in {**self.weights, **self.biases}.items()}

    def _layer(self, x, w, b):
        ld = (w.shape[1] + 3) // 4 * 4           # row stride a multiple of 16 bytes: the 16-byte-load kernel applies
        out = torch.empty(x.shape[0], ld, dtype=torch.float32, device=self.device)[:, :w.shape[1]]
        if x.shape[0]:
            if x.stride(1) != 1:
                x = x.contiguous()
            _lib.dense_layer_fwd(x, w, b, self._plan.act, out)
        return out

    def encoder(self, input_data):
        """code/literal_encoder.py:71-80 on an arbitrary [n, input_dimension] device tensor."""
        h = input_data
        for i in range(self.layer_num):
            h = self._layer(h, self.weights[f"encoder_h{i}"], self.biases[f"encoder_b{i}"])
        return h

    def decoder(self, input_data):
        """code/literal_encoder.py:82-91."""
        h = input_data
        for i in range(self.layer_num):
            h = self._layer(h, self.weights[f"decoder_h{i}"], self.biases[f"decoder_b{i}"])
        return h

    def train_steps(self, x: torch.Tensor, batch_rows: int) -> torch.Tensor:
        """loss + optimizer over the rows of x in batches of batch_rows (code/literal_encoder.py:63-69,98-107) as one
        native call; returns the per-batch losses (float64, device).  Adam / Adadelta: the native call leaves the gradients
        and the whole-variable update kernel follows, batch by batch."""
        p = self._plan
        n_b = (x.shape[0] + batch_rows - 1) // batch_rows
        losses = torch.zeros(max(1, n_b), dtype=torch.float64, device=self.device)
        if x.shape[0] == 0:
            return losses[:0]
        self._ensure_scratch(min(batch_rows, x.shape[0]))
        p.lr = float(self.args.learning_rate)
        if self.args.optimizer in _lib.DENSE_OPTS:
            if self._dense is None:
                self._dense = [torch.zeros_like(self.params), torch.zeros_like(self.params), 0]
            p.optimizer, p.update = _lib.OPT_SGD, 0
            for b in range(n_b):
                _lib.ae_train_steps(p, x[b * batch_rows:(b + 1) * batch_rows], batch_rows, losses[b:b + 1])
                self._dense[2] += 1
                _lib.dense_update_opt(self.params, self._dense[0], self._dense[1], self.grads,
                                      _lib.optimizer_struct(self.args.optimizer, p.lr, self._dense[2]))
            return losses[:n_b]
        p.optimizer, p.update = _OPT[self.args.optimizer], 1
        _lib.ae_train_steps(p, x, batch_rows, losses)
        return losses[:n_b]

    def train_step(self, batch: torch.Tensor) -> torch.Tensor:
        """One batch (code/literal_encoder.py:63-69); returns its loss as a device scalar."""
        return self.train_steps(batch, max(1, batch.shape[0]))[0]

    def train_one_epoch(self, epoch):
        """code/literal_encoder.py:93-112.  `num_batch = L // batch_size + 1`, so the last slice is empty when L is a
        multiple of the batch size; TF would feed it and produce a NaN loss — it is skipped here.  The printed value
        keeps the reference's `loss_sum += batch_size` (:108)."""
        start_time = time.time()
        losses = self.train_steps(self.word_vec_list, self.args.batch_size)
        loss_sum = float(losses.sum()) + self.args.batch_size
        print('epoch {} of literal encoder, loss: {:.4f}, time: {:.4f}s'.format(epoch, loss_sum, time.time() - start_time))
        return loss_sum

    def encoder_multi_batches(self, input_data):
        """code/literal_encoder.py:114-144: forward of the encoder only, on the inputs AS GIVEN (not row-normalised),
        no output normalisation, float64 result."""
        print('encode literal embeddings...', len(input_data))
        x = torch.as_tensor(np.reshape(np.asarray(input_data, dtype=np.float32), [len(input_data), self.input_dimension]),
                            device=self.device)
        bs = self.args.batch_size
        code = torch.empty(x.shape[0], self.hidden_dimensions[-1], dtype=torch.float32, device=self.device)
        for i in range(0, x.shape[0], bs):
            self._ensure_scratch(min(bs, x.shape[0] - i))
            _lib.ae_encode(self._plan, x[i:i + bs], code[i:i + bs])
        res = code.double().cpu().numpy() if x.shape[0] else np.zeros((0, self.args.dim))
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
