"""losses.py — the reference's loss-op surface (code/losses.py), name for name and argument for argument,
over torch CUDA tensors, backed by the HIP kernels of libmultike_hip.so.

Every function takes ALREADY GATHERED `[B, dim]` float32 rows (what `tf.nn.embedding_lookup` returned in
the reference), returns a 0-d float32 tensor and is differentiable w.r.t. the row arguments.  Forward and
the gradient rows are produced by one kernel launch (`mke_gathered_logistic_fwd_bwd` /
`mke_gathered_alignment_fwd_bwd`); `backward` only scales them by the incoming gradient.
The training loops of `MultiKE_model.py` do not go through this un-fused surface — they use the fused
index-stream kernels (tables.StepEngine); this module exists so code written against `losses.py` runs
unchanged, and it is parity-tested against the golden vectors made from the reference's own losses.py.

The two dense-matrix losses (`space_mapping_loss`, `orthogonal_loss`, code/losses.py:53-63) are a 75x75
GEMM and elementwise ops — plain torch (rocBLAS) as SURVEY.md §2.1 prescribes.
"""
from __future__ import annotations

import torch

from . import _lib


def _prep(x: torch.Tensor, name: str) -> torch.Tensor:
    if not x.is_cuda:
        raise _lib.MultiKEHipError(f"{name}: expected a CUDA/HIP tensor (multike_amd has no CPU path)")
    return x.contiguous().float()


class _LogisticTerm(torch.autograd.Function):
    """sum_i w_i * log(1 + exp(sign * ||h_i + r_i - t_i||^2))"""

    @staticmethod
    def forward(ctx, hs, rs, ts, ws, sign):
        hs, rs, ts = _prep(hs, "hs"), _prep(rs, "rs"), _prep(ts, "ts")
        if not (hs.shape == rs.shape == ts.shape and hs.dim() == 2):
            raise _lib.MultiKEHipError(f"row shapes differ: {tuple(hs.shape)} {tuple(rs.shape)} {tuple(ts.shape)}")
        ws_ = None if ws is None else _prep(ws, "ws").reshape(-1)
        if ws_ is not None and ws_.numel() != hs.shape[0]:
            raise _lib.MultiKEHipError("weights length differs from the number of rows")
        need_grad = any(ctx.needs_input_grad[:3])
        lp = torch.empty(_lib.LOSS_PARTIALS, dtype=torch.float64, device=hs.device)
        g = torch.empty_like(hs) if need_grad else None
        # the kernel writes gh, gr (= gh) and gt (= -gh); one buffer is kept, the other two are scratch
        gr = torch.empty_like(hs) if need_grad else None
        gt = torch.empty_like(hs) if need_grad else None
        _lib.gathered_logistic_fwd_bwd(hs, rs, ts, ws_, sign, g, gr, gt, lp)
        ctx.save_for_backward(g)
        return lp.sum().float()

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        if g is None:
            return None, None, None, None, None
        gg = g * gout
        return gg, gg, -gg, None, None


class _AlignmentTerm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _prep(a, "ents1"), _prep(b, "ents2")
        if a.shape != b.shape or a.dim() != 2:
            raise _lib.MultiKEHipError("alignment_loss: shapes differ")
        need_grad = any(ctx.needs_input_grad)
        lp = torch.empty(_lib.LOSS_PARTIALS, dtype=torch.float64, device=a.device)
        ga = torch.empty_like(a) if need_grad else None
        gb = torch.empty_like(a) if need_grad else None
        _lib.gathered_alignment_fwd_bwd(a, b, ga, gb, lp)
        ctx.save_for_backward(ga)
        return lp.sum().float()

    @staticmethod
    def backward(ctx, gout):
        (ga,) = ctx.saved_tensors
        if ga is None:
            return None, None
        gg = ga * gout
        return gg, -gg


def relation_logistic_loss(phs, prs, pts, nhs, nrs, nts):
    """code/losses.py:4-12."""
    return _LogisticTerm.apply(phs, prs, pts, None, 1) + _LogisticTerm.apply(nhs, nrs, nts, None, -1)


def attribute_logistic_loss(phs, pas, pvs, pws, nhs, nas, nvs, nws):
    """code/losses.py:15-27."""
    return _LogisticTerm.apply(phs, pas, pvs, pws, 1) + _LogisticTerm.apply(nhs, nas, nvs, nws, -1)


def relation_logistic_loss_wo_negs(phs, prs, pts):
    """code/losses.py:30-34."""
    return _LogisticTerm.apply(phs, prs, pts, None, 1)


def attribute_logistic_loss_wo_negs(phs, pas, pvs):
    """code/losses.py:37-41."""
    return _LogisticTerm.apply(phs, pas, pvs, None, 1)


def logistic_loss_wo_negs(phs, pas, pvs, pws):
    """code/losses.py:44-50."""
    return _LogisticTerm.apply(phs, pas, pvs, pws, 1)


def orthogonal_loss(mapping, eye):
    """code/losses.py:61-63."""
    return torch.sum(torch.pow(mapping @ mapping.T - eye, 2))


def space_mapping_loss(view_embeds, shared_embeds, mapping, eye, orthogonal_weight, norm_w=0.0001):
    """code/losses.py:53-58.  `tf.nn.l2_normalize(x)` with no axis normalises over the WHOLE matrix."""
    mapped = view_embeds @ mapping
    mapped = mapped * torch.rsqrt(torch.clamp_min(torch.sum(mapped * mapped), 1e-12))
    map_loss = torch.sum(torch.square(shared_embeds - mapped))
    norm_loss = torch.sum(torch.square(mapping))
    return map_loss + orthogonal_weight * orthogonal_loss(mapping, eye) + norm_w * norm_loss


def alignment_loss(ents1, ents2):
    """code/losses.py:66-69."""
    return _AlignmentTerm.apply(ents1, ents2)
