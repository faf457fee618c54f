"""AttrCNN — one parameter set of the attribute-view CNN scorer and its training step.

The reference calls `conv()` (code/MultiKE_model.py:34-63) from three graphs; only the first call is inside a
`variable_scope`, so TF1's layer auto-naming gives THREE independent parameter sets: one AttrCNN per
graph.  Parameters live in one packed float32 buffer (layout in include/multike_hip.h §8); each optimizer keeps its
own Adagrad accumulator for it, like every other variable.

The training step is ONE native call (`mke_attr_step` / `mke_attr_steps`, multike_amd/csrc/mke_attr_cnn.hip): hand-written
HIP kernels for the conv stack, the loss tail and the updates; the dense layer's three products — the only contractions on
the path, flat[n,4d] @ W[4d,d] and its two gradients — on the library's own f32 MFMA tiles (mke_gemm.hip).  No library GEMM,
no autograd.
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
        self._scratch = None
        self._dense = {}         # optimizer name -> [slot1, slot2, step] of an Adam / Adadelta instance
        self._workspace = None   # zero-invariant reduction workspace of mke_attr_conv_bwd
        self._partials = torch.zeros(8, 3 * _lib.LOSS_PARTIALS, dtype=torch.float64, device=self.device)
        self._ring = 0
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

    def _args(self, eng, ent, attr, lit, ih, ia, iv, weights, max_n, scale, opt_name, lr, optimizer, update, n_tags):
        """Fill an mke_attr_step_args for a step (or a run of steps) of at most `max_n` triples; reserves n_tags tags."""
        if optimizer not in _OPT:
            raise _lib.MultiKEHipError(f"optimizer {optimizer!r} not supported by the HIP path (Adagrad, SGD)")
        d, n = self.dim, int(max_n)
        need = _lib.attr_scratch_floats(n, d)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(max(need, 1), dtype=torch.float32, device=self.device)
        tag, _ = eng._next()
        eng.tag += n_tags - 1
        part = self._partials[self._ring]
        self._ring = (self._ring + 1) % self._partials.shape[0]
        adagrad = optimizer == "Adagrad"
        a = _lib.AttrStepArgs()
        f32, i32 = torch.float32, torch.int32
        a.ent_table, a.n_ent, a.ent_stride, a.ent_normalize = _lib.ptr(ent.data, f32, "ent"), ent.n_rows, ent.stride, int(ent.normalize)
        a.ent_acc = _lib.ptr(ent.slot(opt_name), f32, "acc") if (adagrad and ent.trainable and update) else None
        a.ent_grad = _lib.ptr(ent.grad, f32, "grad") if ent.trainable else None
        a.ent_touched = _lib.ptr(ent.touched, i32, "touched") if ent.trainable else None
        a.attr_table, a.n_attr, a.attr_stride = _lib.ptr(attr.data, f32, "attr"), attr.n_rows, attr.stride
        a.attr_normalize = int(attr.normalize)
        a.attr_acc = _lib.ptr(attr.slot(opt_name), f32, "acc") if (adagrad and attr.trainable and update) else None
        a.attr_grad = _lib.ptr(attr.grad, f32, "grad") if attr.trainable else None
        a.attr_grad_copies = attr.grad_copies          # a wholly privatised scratch ([copies][rows][stride]): hub attributes
        a.attr_touched = _lib.ptr(attr.touched, i32, "touched") if attr.trainable else None
        a.lit_table, a.lit_stride, a.dim = _lib.ptr(lit.data, f32, "lit"), lit.stride, d
        a.ih, a.ia, a.iv = _lib.ptr(ih, i32, "ih"), _lib.ptr(ia, i32, "ia"), _lib.ptr(iv, i32, "iv")
        a.weights = _lib.ptr(weights, f32, "weights") if weights is not None else None
        a.n, a.scale = n, float(scale)
        a.params, a.param_grads = _lib.ptr(self.params, f32, "params"), _lib.ptr(self.grads, f32, "grads")
        a.param_acc = _lib.ptr(self.slot(opt_name), f32, "acc") if (adagrad and update) else None
        a.scratch, a.partials = _lib.ptr(self._scratch, f32, "scratch"), _lib.ptr(part, torch.float64, "partials")
        a.optimizer, a.lr, a.tag, a.update = _OPT[optimizer], float(lr), tag, int(update)
        if self._workspace is None:
            self._workspace = torch.zeros(_lib.cnn_workspace_floats(d), dtype=torch.float32, device=self.device)
        a.workspace = _lib.ptr(self._workspace, f32, "workspace")
        return a, part

    def steps(self, eng, ent: EmbeddingTable, attr: EmbeddingTable, lit: EmbeddingTable, ih, ia, iv, weights, step_off,
              scale: float = 1.0, opt_name: str = "attribute", lr: float = 0.001, optimizer: str = "Adagrad") -> torch.Tensor:
        """`len(step_off) - 1` consecutive steps as ONE native call (`mke_attr_steps`): step s trains positions
        [step_off[s], step_off[s+1]) of the epoch-ordered index arrays.  Returns the loss partials
        [n_steps, LOSS_PARTIALS] (sum of row s = loss of step s)."""
        off = np.ascontiguousarray(step_off, dtype=np.int64)
        n_steps = len(off) - 1
        ring = torch.zeros(max(1, n_steps), _lib.LOSS_PARTIALS, dtype=torch.float64, device=self.device)
        if n_steps <= 0:
            return ring[:0]
        a, _ = self._args(eng, ent, attr, lit, ih, ia, iv, weights, int(np.diff(off).max()), scale, opt_name, lr, optimizer,
                          True, n_steps)
        _lib.attr_steps(a, off, ring)
        return ring

    # ------------------------------------------------------------------------------------------------
    def step(self, eng, ent: EmbeddingTable, attr: EmbeddingTable, lit: EmbeddingTable, ih, ia, iv, weights=None,
             scale: float = 1.0, opt_name: str = "attribute", lr: float = 0.001, optimizer: str = "Adagrad",
             update: bool = True) -> torch.Tensor:
        """loss + optimizer of one attribute-view graph:  scale * sum w * log(1 + exp(-conv(h, a, v))), as ONE native
        call (`mke_attr_step`: conv stack, the dense layer's three products on the MFMA GEMM kernel, loss tail,
        backward, row updates, dense update).  Returns the loss partials (`.sum()` is the loss).  With update=False the
        gradients are left in `self.grads`, `ent.grad`, `attr.grad` for inspection."""
        if optimizer in _lib.DENSE_OPTS and update:
            # Adam / Adadelta move every weight: run the fused forward + backward without its (touched-rows) update,
            # then the whole-variable kernels over both tables and the packed parameters
            a, part = self._args(eng, ent, attr, lit, ih, ia, iv, weights, int(ih.numel()), scale, opt_name, lr, "SGD", False, 1)
            _lib.attr_step(a)
            for tb in (ent, attr):
                if tb.trainable:
                    eng._apply(tb, opt_name, optimizer, lr, a.tag)
            st = self._dense.setdefault(opt_name, [torch.zeros_like(self.params), torch.zeros_like(self.params), 0])
            st[2] += 1
            _lib.dense_update_opt(self.params, st[0], st[1], self.grads, _lib.optimizer_struct(optimizer, lr, st[2]))
            return part[:_lib.LOSS_PARTIALS]
        a, part = self._args(eng, ent, attr, lit, ih, ia, iv, weights, int(ih.numel()), scale, opt_name, lr, optimizer, update, 1)
        _lib.attr_step(a)
        return part[:_lib.LOSS_PARTIALS]


class _ConvScore(torch.autograd.Function):
    """score_i = -|| h_i - out_i ||^2 with out = l2_normalize(tanh([flat, 1] @ [W; bias])) over the WHOLE batch and flat = the
    conv stack of (a_i, v_i) — `conv()` of code/MultiKE_model.py:34-63 as a differentiable op on gathered DEVICE rows.

    Forward: `mke_attr_conv_fwd` (BN affine, 2 x conv + tanh, width l2-norm) -> the dense layer on the library's own f32 MFMA
    GEMM (`mke_dense_layer_fwd`) -> `mke_attr_tail_z` (bias, tanh, sum z^2).  Backward (hand-derived, the derivation of
    oracle/attr_cnn_oracle.py): `mke_attr_tail_bwd` (through the batch-wide normalisation and tanh), the dense layer's two
    gradient products on `mke_gemm_f32` (dW split over K), `mke_attr_conv_bwd` (recomputes the stack, back-propagates it into
    the attribute rows and the 52 + 2 dim conv / BN parameters).  No host copy, no library GEMM, no autograd inside.
    Gradients flow to attr_hs, attr_as and the packed parameter buffer; attr_vs (the literal vectors, a constant in the
    reference: code/MultiKE_model.py:88) gets none."""

    @staticmethod
    def forward(ctx, attr_hs, attr_as, attr_vs, params, dim):
        dev = attr_hs.device
        hs, as_, vs = (t.detach().contiguous().float() for t in (attr_hs, attr_as, attr_vs))
        B = hs.shape[0]
        if not (hs.shape == as_.shape == vs.shape == (B, dim)):
            raise _lib.MultiKEHipError(f"conv: row shapes {tuple(hs.shape)} {tuple(as_.shape)} {tuple(vs.shape)} != [B, {dim}]")
        if attr_vs.requires_grad:
            raise _lib.MultiKEHipError("conv: attr_vs is a constant (literal vectors); its gradient is not built")
        p = params.detach()
        nconv = _lib.cnn_conv_params(dim)
        W, bias = p[nconv:nconv + 4 * dim * dim].view(4 * dim, dim), p[nconv + 4 * dim * dim:]
        idx = torch.arange(B, dtype=torch.int32, device=dev)
        flat = torch.empty(B, 4 * dim, dtype=torch.float32, device=dev)
        z = torch.empty(B, dim, dtype=torch.float32, device=dev)
        ssq = torch.zeros(_lib.LOSS_PARTIALS, dtype=torch.float64, device=dev)
        if B:
            _lib.attr_conv_fwd(as_, False, vs, dim, idx, idx, p, flat)
            _lib.dense_layer_fwd(flat, W, None, _lib.ACT_NONE, z)
            _lib.attr_tail_z(z, bias, ssq)
        inv = torch.rsqrt(torch.clamp_min(ssq.sum(), 1e-12)).float()
        diff = hs - z * inv
        ctx.dim = dim
        ctx.save_for_backward(hs, as_, vs, p, flat, z, ssq, idx)
        return -(diff * diff).sum(1)

    @staticmethod
    def backward(ctx, gs):
        hs, as_, vs, p, flat, z, ssq, idx = ctx.saved_tensors
        dim, dev, B = ctx.dim, hs.device, hs.shape[0]
        nconv = _lib.cnn_conv_params(dim)
        W = p[nconv:nconv + 4 * dim * dim].view(4 * dim, dim)
        inv = torch.rsqrt(torch.clamp_min(ssq.sum(), 1e-12)).float()
        diff = hs - z * inv
        g2 = (2.0 * gs.float()).unsqueeze(1) * diff        # dL/dout = +2 gs (h - out);  dL/dh = -that
        g_h = -g2 if ctx.needs_input_grad[0] else None
        if not (ctx.needs_input_grad[1] or ctx.needs_input_grad[3]) or B == 0:
            return g_h, None, None, None, None
        gout = g2.contiguous()
        dot = torch.zeros(_lib.LOSS_PARTIALS, dtype=torch.float64, device=dev)
        dot[0] = (gout.double() * z.double()).sum()
        gp = torch.zeros_like(p)
        _lib.attr_tail_bwd(z, gout, ssq, dot, grad_bias=gp[nconv + 4 * dim * dim:])      # gout <- dL/dzpre, dbias += column sums
        gW = gp[nconv:nconv + 4 * dim * dim].view(4 * dim, dim)
        _lib.gemm_f32(flat, gout, gW, transpose_a=True, splits=max(1, min(32, B // 160)), accumulate=True)   # gW is zero: split-K adds into it
        dflat = torch.empty_like(flat)
        _lib.gemm_f32(gout, W, dflat, transpose_b=True)
        g_as = torch.zeros_like(as_)              # the kernel scatters with the attribute rows' own stride
        touched = torch.zeros(B, dtype=torch.int32, device=dev)
        _lib.attr_conv_bwd(as_, False, vs, dim, idx, idx, p, dflat, gp, g_as, touched, 1)
        return g_h, (g_as if ctx.needs_input_grad[1] else None), None, (gp if ctx.needs_input_grad[3] else None), None


def conv_score(attr_hs, attr_as, attr_vs, dim, cnn: AttrCNN):
    """Differentiable score vector [B] of `cnn` on gathered device rows (see `_ConvScore`)."""
    for t, nm in ((attr_hs, "attr_hs"), (attr_as, "attr_as"), (attr_vs, "attr_vs")):
        if not t.is_cuda:
            raise _lib.MultiKEHipError(f"conv: {nm} must be a CUDA/HIP tensor (multike_amd has no CPU path)")
    return _ConvScore.apply(attr_hs, attr_as, attr_vs, cnn.params, int(dim))
