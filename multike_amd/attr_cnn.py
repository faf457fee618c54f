"""AttrCNN — one parameter set of the attribute-view CNN scorer and its training step.

The reference calls `conv()` (code/MultiKE_model.py:34-63) from three graphs; only the first call is inside a
`variable_scope`, so TF1's layer auto-naming gives THREE independent parameter sets: one AttrCNN per
graph.  Parameters live in one packed float32 buffer (layout in include/multike_hip.h §8); each optimizer keeps its
own Adagrad accumulator for it, like every other variable.

The step is HIP kernels for the conv stack and the loss tail, and three library GEMMs (rocBLAS through torch.matmul)
for the dense layer — the only contractions on the path (flat[n,4d] @ W[4d,d] and its two gradients).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .tables import ADAGRAD_INIT_ACC, EmbeddingTable

_OPT = {"Adagrad": _lib.OPT_ADAGRAD, "SGD": _lib.OPT_SGD}


class AttrCNN:
    def __init__(self, dim: int, device="cuda", seed=None, params: dict | None = None):
        self.dim = dim
        self.device = torch.device(device)
        self.n_conv = _lib.cnn_conv_params(dim)
        self.n_params = _lib.cnn_params(dim)
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.slots: dict[str, torch.Tensor] = {}
        d = dim
        o = 0
        self.views, self.gviews = {}, {}
        for name, shape in (("gamma", (d,)), ("beta", (d,)), ("K1", (2, 4, 1, 2)), ("b1", (2,)), ("K2", (2, 4, 2, 2)),
                            ("b2", (2,)), ("W", (4 * d, d)), ("bias", (d,))):
            n = int(np.prod(shape))
            self.views[name] = self.params[o:o + n].view(shape)
            self.gviews[name] = self.grads[o:o + n].view(shape)
            o += n
        assert o == self.n_params
        if params is None:
            self._init_tf_defaults(seed)
        else:
            for k, v in params.items():
                self.views[k].copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32))

    def _init_tf_defaults(self, seed):
        """tf.layers defaults: BN gamma 1 / beta 0; conv and dense kernels glorot-uniform; biases 0."""
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(int(seed))
        d = self.dim

        def glorot(shape, fan_in, fan_out):
            lim = float(np.sqrt(6.0 / (fan_in + fan_out)))
            return (torch.rand(shape, generator=g) * 2 - 1) * lim

        self.views["gamma"].fill_(1.0)
        self.views["K1"].copy_(glorot((2, 4, 1, 2), 8, 16))
        self.views["K2"].copy_(glorot((2, 4, 2, 2), 16, 16))
        self.views["W"].copy_(glorot((4 * d, d), 4 * d, d))

    def slot(self, opt_name: str) -> torch.Tensor:
        s = self.slots.get(opt_name)
        if s is None:
            s = torch.full_like(self.params, ADAGRAD_INIT_ACC)
            self.slots[opt_name] = s
        return s

    def numpy_params(self) -> dict:
        return {k: v.detach().cpu().numpy().copy() for k, v in self.views.items()}

    # ------------------------------------------------------------------------------------------------
    def step(self, eng, ent: EmbeddingTable, attr: EmbeddingTable, lit: EmbeddingTable, ih, ia, iv, weights=None,
             scale: float = 1.0, opt_name: str = "attribute", lr: float = 0.001, optimizer: str = "Adagrad",
             update: bool = True) -> torch.Tensor:
        """loss + optimizer of one attribute-view graph:  scale * sum w * log(1 + exp(-conv(h, a, v))).
        Returns the loss partials (`.sum()` is the loss).  With update=False the gradients are left in
        `self.grads`, `ent.grad`, `attr.grad` for inspection."""
        d, n = self.dim, ih.numel()
        dev = self.device
        tag, lp = eng._next()
        flat = torch.empty(n, 4 * d, dtype=torch.float32, device=dev)
        _lib.attr_conv_fwd(attr.data, attr.normalize, lit.data, d, ia, iv, self.params, flat)
        z = torch.matmul(flat, self.views["W"])                                  # library GEMM [n,4d] x [4d,d]
        ssq = torch.empty(_lib.LOSS_PARTIALS, dtype=torch.float64, device=dev)
        dot = torch.empty(_lib.LOSS_PARTIALS, dtype=torch.float64, device=dev)
        _lib.attr_tail_z(z, self.views["bias"], ssq)
        gout = torch.empty(n, d, dtype=torch.float32, device=dev)
        _lib.attr_tail_loss(z, ssq, ent.data, ent.normalize, ih, weights, scale, gout, dot,
                            ent.grad if ent.trainable else None, ent.touched if ent.trainable else None, tag, lp)
        _lib.attr_tail_bwd(z, gout, ssq, dot)                                    # gout is now dL/dzpre
        torch.matmul(flat.t(), gout, out=self.gviews["W"])
        torch.sum(gout, 0, out=self.gviews["bias"])
        dflat = torch.matmul(gout, self.views["W"].t())
        _lib.attr_conv_bwd(attr.data, attr.normalize, lit.data, d, ia, iv, self.params, dflat, self.grads,
                           attr.grad if attr.trainable else None, attr.touched if attr.trainable else None, tag)
        if update:
            if optimizer not in _OPT:
                raise _lib.MultiKEHipError(f"optimizer {optimizer!r} not supported by the HIP path (Adagrad, SGD)")
            if ent.trainable:
                eng._apply(ent, opt_name, optimizer, lr, tag)
            if attr.trainable:
                eng._apply(attr, opt_name, optimizer, lr, tag)
            _lib.dense_update(self.params, self.slot(opt_name) if optimizer == "Adagrad" else None, self.grads,
                              _OPT[optimizer], lr)
        return lp
