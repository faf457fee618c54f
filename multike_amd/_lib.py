"""ctypes binding of libmultike_hip.so (the C-ABI declared in include/multike_hip.h).

This is the only place the product touches native code.  There is NO fallback: if the library is
missing or a tensor is not on the GPU the call raises.  PyTorch is used for device memory and streams
only — every pointer handed to the library is `tensor.data_ptr()` of a CUDA(HIP) tensor and the stream
is torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libmultike_hip.so")

LOSS_PARTIALS = 2048  # MKE_LOSS_PARTIALS
MAX_STRIDE = 320  # MKE_MAX_STRIDE
OPT_ADAGRAD, OPT_SGD, OPT_ADAM, OPT_ADADELTA = 0, 1, 2, 3
DENSE_OPTS = {"Adam": OPT_ADAM, "Adadelta": OPT_ADADELTA}   # TF1 rules that move zero-gradient weights: whole-variable kernels
_SUPPORTED_FPL = (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 16, 20)

# every symbol include/multike_hip.h declares (tests/test_abi.py checks the .so exports each of them)
SYMBOLS = (
    "mke_version", "mke_last_error", "mke_set_option", "mke_triple_score_fwd_bwd", "mke_triple_score_fwd_bwd_x",
    "mke_count_entity_refs", "mke_triple_score_fwd_bwd_xc", "mke_triple_score_fwd_bwd_xch", "mke_triple_score_fwd_bwd_det", "mke_stage_reduce", "mke_rows_update", "mke_rows_update_multi", "mke_rows_update_multi_count",
    "mke_neg_sample", "mke_tripleset_build", "mke_tripleset_query", "mke_gathered_logistic_fwd_bwd",
    "mke_gathered_alignment_fwd_bwd", "mke_align_fwd_bwd", "mke_gather_rows", "mke_relation_steps",
    "mke_rowset_build", "mke_rowset_remap", "mke_rows_gather_padded", "mke_rows_scatter_add",
    "mke_attr_conv_fwd", "mke_attr_conv_bwd", "mke_attr_tail_z", "mke_attr_tail_loss", "mke_attr_tail_bwd",
    "mke_dense_update", "mke_align_rank", "mke_gemm_f32", "mke_attr_scratch_floats", "mke_attr_step", "mke_attr_steps", "mke_attr_step_phases",
    "mke_sample_distinct", "mke_neg_sample_at", "mke_rows_update_dense", "mke_dense_update_opt", "mke_align_steps", "mke_sim_select", "mke_sim_sample", "mke_topk_rows", "mke_topk_candidates", "mke_mapping_scratch_floats", "mke_mapping_step", "mke_mapping_step_phases", "mke_mapping_steps",
    "mke_ae_scratch_floats", "mke_ae_train_steps", "mke_ae_step_phases", "mke_ae_encode", "mke_dense_layer_fwd",
    "mke_topk_long", "mke_probe_rows", "mke_oc_block_floats", "mke_oc_pack_codes", "mke_oc_plan", "mke_oc_bases", "mke_oc_count", "mke_oc_score", "mke_oc_apply", "mke_oc_run",
    "mke_oc_em_plan_temp_bytes", "mke_oc_em_plan", "mke_oc_pass2", "mke_oc_steps",
    "mke_tuning_init", "mke_triple_score_fwd_bwd_t", "mke_rows_update_multi_t",
)
ACT_NONE, ACT_TANH, ACT_SIGMOID = 0, 1, 2
AE_MAX_LAYERS = 4


class AttrStepArgs(C.Structure):
    """mke_attr_step_args"""
    _fields_ = [
        ("ent_table", C.c_void_p), ("n_ent", C.c_int64), ("ent_stride", C.c_int), ("ent_normalize", C.c_int),
        ("ent_acc", C.c_void_p), ("ent_grad", C.c_void_p), ("ent_touched", C.c_void_p),
        ("attr_table", C.c_void_p), ("n_attr", C.c_int64), ("attr_stride", C.c_int), ("attr_normalize", C.c_int),
        ("attr_acc", C.c_void_p), ("attr_grad", C.c_void_p), ("attr_touched", C.c_void_p),
        ("lit_table", C.c_void_p), ("lit_stride", C.c_int), ("dim", C.c_int),
        ("ih", C.c_void_p), ("ia", C.c_void_p), ("iv", C.c_void_p), ("weights", C.c_void_p), ("n", C.c_int64),
        ("scale", C.c_float), ("params", C.c_void_p), ("param_grads", C.c_void_p), ("param_acc", C.c_void_p),
        ("scratch", C.c_void_p), ("partials", C.c_void_p), ("optimizer", C.c_int), ("lr", C.c_float), ("tag", C.c_int32),
        ("update", C.c_int), ("workspace", C.c_void_p), ("attr_grad_copies", C.c_int), ("tuning", C.c_void_p),
    ]


class AlignTableStruct(C.Structure):
    """mke_align_table"""
    _fields_ = [("table", C.c_void_p), ("acc", C.c_void_p), ("grad", C.c_void_p), ("touched", C.c_void_p), ("n_rows", C.c_int64),
                ("normalize", C.c_int)]


class AlignTermStruct(C.Structure):
    """mke_align_term"""
    _fields_ = [("a", C.c_int), ("b", C.c_int), ("weight", C.c_float)]


class AlignPlanStruct(C.Structure):
    """mke_align_plan"""
    _fields_ = [("tables", AlignTableStruct * 4), ("n_tables", C.c_int), ("terms", AlignTermStruct * 4), ("n_terms", C.c_int),
                ("stride", C.c_int), ("dim", C.c_int), ("ia", C.c_void_p), ("ib", C.c_void_p),
                ("step_off", C.POINTER(C.c_int64)), ("n_steps", C.c_int), ("optimizer", C.c_int), ("lr", C.c_float),
                ("tag_base", C.c_int32), ("loss_partials", C.c_void_p)]


class MappingViewStruct(C.Structure):
    """mke_mapping_view"""
    _fields_ = [("table", C.c_void_p), ("normalize", C.c_int)]


class MappingStepArgs(C.Structure):
    """mke_mapping_step_args"""
    _fields_ = [("ent_table", C.c_void_p), ("n_ent", C.c_int64), ("ent_normalize", C.c_int), ("ent_acc", C.c_void_p),
                ("ent_grad", C.c_void_p), ("ent_touched", C.c_void_p), ("views", MappingViewStruct * 3), ("n_views", C.c_int),
                ("stride", C.c_int), ("dim", C.c_int), ("idx", C.c_void_p), ("n", C.c_int64), ("M", C.c_void_p), ("gM", C.c_void_p),
                ("accM", C.c_void_p), ("orthogonal_weight", C.c_float), ("norm_w", C.c_float), ("scratch", C.c_void_p),
                ("partials", C.c_void_p), ("optimizer", C.c_int), ("lr", C.c_float), ("tag", C.c_int32), ("update", C.c_int)]


class HotRowsStruct(C.Structure):
    """mke_hot_rows"""
    _fields_ = [("slot", C.c_void_p), ("n_hot", C.c_int32), ("copies", C.c_int32), ("row0", C.c_int64)]


class OcStepStruct(C.Structure):
    """mke_oc_step"""
    _fields_ = [("ent", C.c_void_p), ("ent_acc", C.c_void_p), ("ent_grad", C.c_void_p), ("ent_touched", C.c_void_p),
                ("ref_count", C.c_void_p), ("n_local", C.c_int64),
                ("rel", C.c_void_p), ("rel_acc", C.c_void_p), ("rel_grad", C.c_void_p), ("rel_grad_copies", C.c_int), ("rel_touched", C.c_void_p),
                ("n_rel", C.c_int64), ("stride", C.c_int), ("dim", C.c_int), ("rank", C.c_int), ("n_ranks", C.c_int),
                ("pos_h", C.c_void_p), ("pos_r", C.c_void_p), ("pos_t", C.c_void_p), ("n_pos", C.c_int64), ("per", C.c_int64),
                ("slot_h", C.c_void_p), ("slot_t", C.c_void_p), ("own_h", C.c_void_p), ("n_own_h", C.c_int64),
                ("own_t", C.c_void_p), ("n_own_t", C.c_int64), ("neg_per_pos", C.c_int), ("capacity", C.c_int64),
                ("codes", C.c_void_p), ("code_off", C.c_int64 * 16),
                ("optimizer", C.c_int), ("lr", C.c_float), ("scale", C.c_float), ("tag", C.c_int32),
                ("n_peers", C.c_int), ("peer_v", C.c_void_p * 16), ("peer_g", C.c_void_p * 16), ("pos_w", C.c_void_p),
                ("hot", HotRowsStruct),
                # version 105: entity-major second pass
                ("em_coef", C.c_void_p), ("em_pos0", C.c_int64), ("em_refs", C.c_void_p), ("em_rows", C.c_void_p), ("em_off", C.c_void_p),
                ("em_n_rows", C.c_int64), ("em_chunks", C.c_int), ("em_block_floats", C.c_int64), ("em_v", C.c_void_p * 4),
                ("em_gv", C.c_void_p * 4),
                ("em_part", C.c_void_p), ("em_long_rows", C.c_void_p), ("em_long_part0", C.c_void_p), ("em_n_long", C.c_int64),
                ("em_part0", C.c_int64), ("em_partials", C.c_void_p), ("em_mode", C.c_int), ("tuning", C.c_void_p)]


TUNE_DEFAULT = -2
TUNING_FIELDS = ("score_splits", "score_half_groups", "score_offsets32", "score_lane_ids", "count_in_score", "update_chunk",
                 "oc_score_quarter", "attr_fused_bwd", "sampler_fast")


class TuningStruct(C.Structure):
    """mke_tuning: the performance knobs of ONE plan / call (a field at TUNE_DEFAULT follows mke_set_option's process default)."""
    _fields_ = [(f, C.c_int) for f in TUNING_FIELDS] + [("reserved", C.c_int * 7)]


def tuning(**knobs) -> TuningStruct:
    """A mke_tuning with the given knobs set and every other field at the process default.  The caller keeps it alive for as long
    as a plan points at it."""
    t = TuningStruct()
    _check(lib().mke_tuning_init(C.byref(t)), "mke_tuning_init")
    for k, v in knobs.items():
        if k not in TUNING_FIELDS:
            raise MultiKEHipError(f"unknown tuning knob {k!r}")
        setattr(t, k, int(v))
    return t


def tuning_ptr(t) -> int | None:
    return C.addressof(t) if t is not None else None


OC_EM_MAX_CHUNKS = 4
OC_EM_WAVES = 32768


class OcEmPlanArgs(C.Structure):
    """mke_oc_em_plan_args"""
    _fields_ = [("pos_h", C.c_void_p), ("pos_r", C.c_void_p), ("pos_t", C.c_void_p), ("codes", C.c_void_p), ("neg_per_pos", C.c_int),
                ("slot_h", C.c_void_p), ("slot_t", C.c_void_p), ("step_lo", C.c_void_p), ("n_steps", C.c_int), ("chunks", C.c_int),
                ("n_all", C.c_int64), ("max_step", C.c_int64), ("n_ranks", C.c_int), ("rank", C.c_int), ("n_local", C.c_int64), ("n_rel", C.c_int64),
                ("keys", C.c_void_p), ("keys_alt", C.c_void_p), ("capacity", C.c_int64), ("vals_alt", C.c_void_p), ("scratch8", C.c_void_p), ("wave_scratch", C.c_void_p),
                ("refs", C.c_void_p), ("rows", C.c_void_p), ("off", C.c_void_p), ("flags", C.c_void_p), ("scan", C.c_void_p),
                ("step_row0", C.c_void_p), ("n_refs", C.c_void_p),
                ("item_row", C.c_void_p), ("item_off", C.c_void_p), ("item_part", C.c_void_p), ("long_row", C.c_void_p), ("long_part0", C.c_void_p),
                ("step_item0", C.c_void_p), ("step_long0", C.c_void_p), ("step_part0", C.c_void_p),
                ("temp", C.c_void_p), ("temp_bytes", C.c_int64)]


OC_COMM_NCCL, OC_COMM_CALLBACK, OC_COMM_LOOPBACK = 0, 1, 2
OC_CB_MOVE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)     # all_gather / reduce_scatter callbacks
OC_CB_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)              # all_reduce callback


class OcCommStruct(C.Structure):
    """mke_oc_comm"""
    _fields_ = [("kind", C.c_int), ("ctx", C.c_void_p), ("all_gather", C.c_void_p), ("reduce_scatter", C.c_void_p), ("all_reduce", C.c_void_p),
                ("world", C.c_int), ("rank", C.c_int), ("wire_gbps", C.c_float), ("latency_us", C.c_float)]


class OcLoopStruct(C.Structure):
    """mke_oc_loop"""
    _fields_ = [("parts", C.c_void_p), ("step_part0", C.c_void_p), ("n_steps", C.c_int), ("chunks", C.c_int),
                ("send", C.c_void_p * 4), ("v_all", C.c_void_p * 4), ("g_all", C.c_void_p * 4), ("gv", C.c_void_p * 4), ("block_floats", C.c_int64),
                ("loss_ring", C.c_void_p), ("loss_stride", C.c_int64), ("tag_base", C.c_int32), ("comm", C.c_void_p), ("comm_stream", C.c_void_p),
                ("overlap_rs", C.c_int)]


class AEPlanStruct(C.Structure):
    """mke_ae_plan"""
    _fields_ = [("n_layers", C.c_int), ("dims", C.c_int * 5), ("act", C.c_int), ("normalize", C.c_int),
                ("params", C.c_void_p), ("grads", C.c_void_p), ("acc", C.c_void_p), ("n_params", C.c_int64),
                ("w_off", C.c_int64 * 8), ("b_off", C.c_int64 * 8), ("optimizer", C.c_int), ("lr", C.c_float),
                ("update", C.c_int), ("scratch", C.c_void_p), ("scratch_floats", C.c_int64), ("partials", C.c_void_p),
                ("scalars", C.c_void_p)]


class OptimizerStruct(C.Structure):
    """mke_optimizer (TF1 defaults)"""
    _fields_ = [("kind", C.c_int), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float),
                ("rho", C.c_float), ("step", C.c_int64)]


def optimizer_struct(name: str, lr: float, step: int = 1) -> OptimizerStruct:
    return OptimizerStruct(DENSE_OPTS[name], float(lr), 0.9, 0.999, 1e-8, 0.95, int(step))


class KGSideStruct(C.Structure):
    """mke_kg_side"""
    _fields_ = [("ent_list", C.c_void_p), ("ent_lo", C.c_int32), ("n_ent", C.c_int32), ("cand_table", C.c_void_p),
                ("cand_valid", C.c_void_p), ("cand_k", C.c_int32), ("known_keys", C.c_void_p),
                ("known_capacity", C.c_uint64)]


class UpdateTableStruct(C.Structure):
    """mke_update_table"""
    _fields_ = [("table", C.c_void_p), ("acc", C.c_void_p), ("grad", C.c_void_p), ("touched", C.c_void_p),
                ("n_rows", C.c_int64), ("normalize", C.c_int), ("grad_copies", C.c_int), ("ref_count", C.c_void_p),
                ("src_rows", C.c_void_p), ("slot_of", C.c_void_p), ("n_ranks", C.c_int), ("capacity", C.c_int64),
                ("hot", HotRowsStruct)]


class RelationPlanStruct(C.Structure):
    """mke_relation_plan"""
    _fields_ = [
        ("ent_table", C.c_void_p), ("n_ent", C.c_int64), ("ent_normalize", C.c_int),
        ("rel_table", C.c_void_p), ("n_rel", C.c_int64), ("rel_normalize", C.c_int),
        ("ent_acc", C.c_void_p), ("rel_acc", C.c_void_p), ("ent_grad", C.c_void_p), ("rel_grad", C.c_void_p),
        ("rel_grad_copies", C.c_int),
        ("ent_touched", C.c_void_p), ("rel_touched", C.c_void_p), ("ent_ref_count", C.c_void_p), ("overlap", C.c_int),
        ("neg_chunk_capacity", C.c_int64), ("stride", C.c_int), ("dim", C.c_int),
        ("pos_h", C.c_void_p), ("pos_r", C.c_void_p), ("pos_t", C.c_void_p), ("pos_kg", C.c_void_p),
        ("step_off", C.POINTER(C.c_int64)), ("n_steps", C.c_int), ("sides", KGSideStruct * 2),
        ("neg_per_pos", C.c_int), ("max_try", C.c_int), ("sample_chunk", C.c_int), ("negatives_ready", C.c_int),
        ("neg_h", C.c_void_p), ("neg_r", C.c_void_p), ("neg_t", C.c_void_p),
        ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("stream_id", C.c_uint32),
        ("optimizer", C.c_int), ("lr", C.c_float), ("scale", C.c_float),
        ("loss_partials", C.c_void_p), ("loss_ring", C.c_int), ("tag_base", C.c_int32), ("pos_w", C.c_void_p),
        ("hot", HotRowsStruct), ("tuning", C.c_void_p),
    ]

_lib = None


class MultiKEHipError(RuntimeError):
    pass


def lib():
    """Load the library (once).  Raises if it has not been built — there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise MultiKEHipError(
                f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C multike_amd/csrc`.  multike_amd has no CPU fallback.")
        L = C.CDLL(SO_PATH)
        L.mke_version.restype = C.c_int
        L.mke_last_error.restype = C.c_char_p
        for name in SYMBOLS[2:]:
            getattr(L, name).restype = C.c_int
        L.mke_attr_scratch_floats.restype = C.c_int64
        L.mke_mapping_scratch_floats.restype = C.c_int64
        L.mke_ae_scratch_floats.restype = C.c_int64
        L.mke_oc_block_floats.restype = C.c_int64
        L.mke_oc_em_plan_temp_bytes.restype = C.c_int64
        _lib = L
        # MKE_OPTIONS="name=value,name=value": mke_set_option calls applied at load (performance knobs for experiments)
        for kv in filter(None, os.environ.get("MKE_OPTIONS", "").split(",")):
            k, _, v = kv.partition("=")
            if L.mke_set_option(k.strip().encode(), C.c_int(int(v)), None):
                raise MultiKEHipError(f"MKE_OPTIONS: {L.mke_last_error().decode()}")
    return _lib


def stride_for(dim: int) -> int:
    """Smallest supported row stride (floats) >= dim: a multiple of 16 whose /16 the kernels instantiate."""
    need = (dim + 15) // 16
    for f in _SUPPORTED_FPL:
        if f >= need:
            return f * 16
    raise MultiKEHipError(f"dim {dim} exceeds the largest supported stride {MAX_STRIDE}")


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().mke_last_error().decode("utf-8", "replace")
        raise MultiKEHipError(f"{what} failed (code {rc}): {msg}")


def _dev(t: torch.Tensor | None, dtype, name: str):
    """data_ptr of a contiguous CUDA tensor of the given dtype (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise MultiKEHipError(f"{name}: expected a CUDA/HIP tensor, got device {t.device} (no CPU path exists)")
    if t.dtype != dtype:
        raise MultiKEHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise MultiKEHipError(f"{name}: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


_pinned_stream = None  # raw hipStream_t handle pinned by a hot loop (saves a torch.cuda.current_stream() per launch)


def pin_stream(handle: int | None):
    """Pin the stream every launch goes to (a raw `stream.cuda_stream` handle), or None to follow torch's current
    stream again.  Returns the previous pin."""
    global _pinned_stream
    old, _pinned_stream = _pinned_stream, handle
    return old


def _stream():
    if _pinned_stream is not None:
        return C.c_void_p(_pinned_stream)
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def version() -> int:
    return lib().mke_version()


def set_option(name: str, value: int) -> int:
    old = C.c_int(0)
    _check(lib().mke_set_option(name.encode(), C.c_int(value), C.byref(old)), "mke_set_option")
    return old.value


def triple_score_fwd_bwd_det(ent, ent_normalize, rel, rel_normalize, dim, pos, pos_w, neg, neg_w, neg_per_pos, scale, grad_ent,
                             grad_rel, touched_ent, touched_rel, tag, ref_count, ent_acc, optimizer, lr, stage_rows, stage_keys,
                             loss_partials):
    ph, pr, pt = pos
    nh, nr, nt = neg if neg is not None else (None, None, None)
    i32, f32 = torch.int32, torch.float32
    rc = lib().mke_triple_score_fwd_bwd_det(
        _dev(ent, f32, "ent"), C.c_int64(ent.shape[0]), C.c_int(int(ent_normalize)), _dev(rel, f32, "rel"), C.c_int64(rel.shape[0]),
        C.c_int(int(rel_normalize)), C.c_int(ent.shape[1]), C.c_int(dim), _dev(ph, i32, "ph"), _dev(pr, i32, "pr"), _dev(pt, i32, "pt"),
        _dev(pos_w, f32, "pos_w"), C.c_int64(ph.numel()), _dev(nh, i32, "nh"), _dev(nr, i32, "nr"), _dev(nt, i32, "nt"),
        _dev(neg_w, f32, "neg_w"), C.c_int64(0 if nh is None else nh.numel()), C.c_int(neg_per_pos), C.c_float(scale),
        _dev(grad_ent, f32, "grad_ent"), _dev(grad_rel, f32, "grad_rel"), _dev(touched_ent, i32, "touched"),
        _dev(touched_rel, i32, "touched"), C.c_int32(tag), _dev(ref_count, i32, "ref_count"), _dev(ent_acc, f32, "acc"),
        C.c_int(optimizer), C.c_float(lr), _dev(stage_rows, f32, "stage_rows"), _dev(stage_keys, torch.int64, "stage_keys"),
        C.c_int64(stage_keys.numel()), _dev(loss_partials, torch.float64, "loss"), _stream())
    _check(rc, "mke_triple_score_fwd_bwd_det")


def stage_reduce(stage_rows, sorted_keys, order, grad_ent, grad_rel, touched_ent, touched_rel, tag):
    rc = lib().mke_stage_reduce(_dev(stage_rows, torch.float32, "stage_rows"), _dev(sorted_keys, torch.int64, "keys"),
                                _dev(order, torch.int64, "order"), C.c_int64(sorted_keys.numel()), C.c_int(stage_rows.shape[1]),
                                _dev(grad_ent, torch.float32, "grad_ent"), _dev(grad_rel, torch.float32, "grad_rel"),
                                _dev(touched_ent, torch.int32, "touched"), _dev(touched_rel, torch.int32, "touched"), C.c_int32(tag),
                                _stream())
    _check(rc, "mke_stage_reduce")


def get_option(name: str) -> int:
    old = set_option(name, 0)
    set_option(name, old)
    return old


def triple_score_fwd_bwd(ent, ent_normalize, rel, rel_normalize, dim, pos, pos_w, neg, neg_w, neg_per_pos, scale,
                         grad_ent, grad_rel, touched_ent, touched_rel, tag, loss_partials):
    """mke_triple_score_fwd_bwd.  pos/neg = (h, r, t) int32 CUDA tensors (neg may be None).
    grad_rel may be [K, n_rel, stride] (K privatised copies) or [n_rel, stride]."""
    rel_copies = 1 if grad_rel is None or grad_rel.dim() == 2 else grad_rel.shape[0]
    ph, pr, pt = pos
    n_pos = ph.numel()
    if neg is None:
        nh = nr = nt = None
        n_neg = 0
    else:
        nh, nr, nt = neg
        n_neg = nh.numel()
    rc = lib().mke_triple_score_fwd_bwd(
        _dev(ent, torch.float32, "ent_table"), C.c_int64(ent.shape[0]), C.c_int(int(ent_normalize)),
        _dev(rel, torch.float32, "rel_table"), C.c_int64(rel.shape[0]), C.c_int(int(rel_normalize)),
        C.c_int(ent.shape[1]), C.c_int(dim),
        _dev(ph, torch.int32, "pos_h"), _dev(pr, torch.int32, "pos_r"), _dev(pt, torch.int32, "pos_t"),
        _dev(pos_w, torch.float32, "pos_w"), C.c_int64(n_pos),
        _dev(nh, torch.int32, "neg_h"), _dev(nr, torch.int32, "neg_r"), _dev(nt, torch.int32, "neg_t"),
        _dev(neg_w, torch.float32, "neg_w"), C.c_int64(n_neg), C.c_int(neg_per_pos), C.c_float(scale),
        _dev(grad_ent, torch.float32, "grad_ent"), _dev(grad_rel, torch.float32, "grad_rel"), C.c_int(rel_copies),
        _dev(touched_ent, torch.int32, "touched_ent"), _dev(touched_rel, torch.int32, "touched_rel"), C.c_int32(tag),
        _dev(loss_partials, torch.float64, "loss_partials"), _stream())
    _check(rc, "mke_triple_score_fwd_bwd")


def count_entity_refs(pos_h, pos_t, neg_h, neg_t, neg_per_pos, ref_count):
    rc = lib().mke_count_entity_refs(_dev(pos_h, torch.int32, "pos_h"), _dev(pos_t, torch.int32, "pos_t"),
                                     C.c_int64(pos_h.numel()), _dev(neg_h, torch.int32, "neg_h"),
                                     _dev(neg_t, torch.int32, "neg_t"), C.c_int64(0 if neg_h is None else neg_h.numel()),
                                     C.c_int(neg_per_pos), _dev(ref_count, torch.int32, "ref_count"), _stream())
    _check(rc, "mke_count_entity_refs")


def triple_score_fwd_bwd_x(ent, ent_normalize, rel, rel_normalize, dim, pos, pos_w, neg, neg_w, neg_per_pos, scale, grad_ent,
                           grad_rel, touched_ent, touched_rel, tag, ref_count, ent_acc, optimizer, lr, loss_partials, hot=None,
                           tuning=None):
    """mke_triple_score_fwd_bwd_x: the fused step with the exclusive-row fast path (ref_count filled by
    count_entity_refs for the same batch).  hot (HotRowsStruct of the entity table, `EmbeddingTable.hot_struct()`): the hub
    rows' flushes go to their private copies (mke_triple_score_fwd_bwd_xch); the update must then get the same struct."""
    ph, pr, pt = pos
    nh, nr, nt = neg
    rel_copies = 1 if grad_rel.dim() == 2 else grad_rel.shape[0]
    if tuning is not None:       # this call's knobs (TuningStruct): mke_triple_score_fwd_bwd_t
        rc = lib().mke_triple_score_fwd_bwd_t(
            _dev(ent, torch.float32, "ent_table"), C.c_int64(ent.shape[0]), C.c_int(int(ent_normalize)),
            _dev(rel, torch.float32, "rel_table"), C.c_int64(rel.shape[0]), C.c_int(int(rel_normalize)),
            C.c_int(ent.shape[1]), C.c_int(dim),
            _dev(ph, torch.int32, "pos_h"), _dev(pr, torch.int32, "pos_r"), _dev(pt, torch.int32, "pos_t"),
            _dev(pos_w, torch.float32, "pos_w"), C.c_int64(ph.numel()),
            _dev(nh, torch.int32, "neg_h"), _dev(nr, torch.int32, "neg_r"), _dev(nt, torch.int32, "neg_t"),
            _dev(neg_w, torch.float32, "neg_w"), C.c_int64(nh.numel()), C.c_int(neg_per_pos), C.c_float(scale),
            _dev(grad_ent, torch.float32, "grad_ent"), _dev(grad_rel, torch.float32, "grad_rel"), C.c_int(rel_copies),
            _dev(touched_ent, torch.int32, "touched_ent"), _dev(touched_rel, torch.int32, "touched_rel"), C.c_int32(tag),
            _dev(ref_count, torch.int32, "ref_count"), _dev(ent_acc, torch.float32, "ent_acc"), C.c_int(optimizer), C.c_float(lr),
            None, (C.byref(hot) if hot is not None and hot.n_hot > 0 else None), C.byref(tuning),
            _dev(loss_partials, torch.float64, "loss_partials"), _stream())
        _check(rc, "mke_triple_score_fwd_bwd_t")
        return
    if hot is not None and hot.n_hot > 0:
        rc = lib().mke_triple_score_fwd_bwd_xch(
            _dev(ent, torch.float32, "ent_table"), C.c_int64(ent.shape[0]), C.c_int(int(ent_normalize)),
            _dev(rel, torch.float32, "rel_table"), C.c_int64(rel.shape[0]), C.c_int(int(rel_normalize)),
            C.c_int(ent.shape[1]), C.c_int(dim),
            _dev(ph, torch.int32, "pos_h"), _dev(pr, torch.int32, "pos_r"), _dev(pt, torch.int32, "pos_t"),
            _dev(pos_w, torch.float32, "pos_w"), C.c_int64(ph.numel()),
            _dev(nh, torch.int32, "neg_h"), _dev(nr, torch.int32, "neg_r"), _dev(nt, torch.int32, "neg_t"),
            _dev(neg_w, torch.float32, "neg_w"), C.c_int64(nh.numel()), C.c_int(neg_per_pos), C.c_float(scale),
            _dev(grad_ent, torch.float32, "grad_ent"), _dev(grad_rel, torch.float32, "grad_rel"), C.c_int(rel_copies),
            _dev(touched_ent, torch.int32, "touched_ent"), _dev(touched_rel, torch.int32, "touched_rel"), C.c_int32(tag),
            _dev(ref_count, torch.int32, "ref_count"), _dev(ent_acc, torch.float32, "ent_acc"), C.c_int(optimizer), C.c_float(lr),
            None, C.byref(hot), _dev(loss_partials, torch.float64, "loss_partials"), _stream())
        _check(rc, "mke_triple_score_fwd_bwd_xch")
        return
    rc = lib().mke_triple_score_fwd_bwd_x(
        _dev(ent, torch.float32, "ent_table"), C.c_int64(ent.shape[0]), C.c_int(int(ent_normalize)),
        _dev(rel, torch.float32, "rel_table"), C.c_int64(rel.shape[0]), C.c_int(int(rel_normalize)),
        C.c_int(ent.shape[1]), C.c_int(dim),
        _dev(ph, torch.int32, "pos_h"), _dev(pr, torch.int32, "pos_r"), _dev(pt, torch.int32, "pos_t"),
        _dev(pos_w, torch.float32, "pos_w"), C.c_int64(ph.numel()),
        _dev(nh, torch.int32, "neg_h"), _dev(nr, torch.int32, "neg_r"), _dev(nt, torch.int32, "neg_t"),
        _dev(neg_w, torch.float32, "neg_w"), C.c_int64(nh.numel()), C.c_int(neg_per_pos), C.c_float(scale),
        _dev(grad_ent, torch.float32, "grad_ent"), _dev(grad_rel, torch.float32, "grad_rel"), C.c_int(rel_copies),
        _dev(touched_ent, torch.int32, "touched_ent"), _dev(touched_rel, torch.int32, "touched_rel"), C.c_int32(tag),
        _dev(ref_count, torch.int32, "ref_count"), _dev(ent_acc, torch.float32, "ent_acc"), C.c_int(optimizer), C.c_float(lr),
        _dev(loss_partials, torch.float64, "loss_partials"), _stream())
    _check(rc, "mke_triple_score_fwd_bwd_x")


def rows_update(table, acc, grad, touched, tag, dim, normalize, optimizer, lr):
    copies = 1 if grad.dim() == 2 else grad.shape[0]
    rc = lib().mke_rows_update(
        _dev(table, torch.float32, "table"), _dev(acc, torch.float32, "acc"), _dev(grad, torch.float32, "grad"),
        C.c_int(copies), _dev(touched, torch.int32, "touched"), C.c_int32(tag), C.c_int64(table.shape[0]), C.c_int(table.shape[1]),
        C.c_int(dim), C.c_int(int(normalize)), C.c_int(optimizer), C.c_float(lr), _stream())
    _check(rc, "mke_rows_update")


def ptr(t: torch.Tensor | None, dtype, name: str) -> int | None:
    """Raw device address (for the plan / side structs)."""
    return _dev(t, dtype, name).value


def rows_update_multi(tables, tag, stride, dim, optimizer, lr, tuning=None):
    """tables: list of (data, acc, grad, touched, normalize[, ref_count[, hot]]) (hot: the table's HotRowsStruct); or, for the owner side of the sharded
    step, a dict(table=, acc=, normalize=, src_rows=, slot_of=, n_ranks=, capacity=) (mke_update_table.slot_of)."""
    arr = (UpdateTableStruct * len(tables))()
    for k, tpl in enumerate(tables):
        if isinstance(tpl, dict):
            arr[k].table = ptr(tpl["table"], torch.float32, "table")
            arr[k].acc = ptr(tpl["acc"], torch.float32, "acc")
            arr[k].n_rows = tpl["table"].shape[0]
            arr[k].normalize = int(tpl["normalize"])
            arr[k].grad_copies = 1
            arr[k].src_rows = ptr(tpl["src_rows"], torch.float32, "src_rows")
            arr[k].slot_of = ptr(tpl["slot_of"], torch.int32, "slot_of")
            arr[k].n_ranks = int(tpl["n_ranks"])
            arr[k].capacity = int(tpl["capacity"])
            if tpl["slot_of"].numel() < arr[k].n_rows * arr[k].n_ranks or tpl["src_rows"].shape[0] < arr[k].n_ranks * arr[k].capacity:
                raise ValueError("slot_of / src_rows are smaller than n_rows * n_ranks / n_ranks * capacity")
            continue
        data, acc, grad, touched, normalize = tpl[:5]
        arr[k].ref_count = ptr(tpl[5], torch.int32, "ref_count") if len(tpl) > 5 and tpl[5] is not None else None
        if len(tpl) > 6 and tpl[6] is not None:
            arr[k].hot = tpl[6]
        arr[k].table = ptr(data, torch.float32, "table")
        arr[k].acc = ptr(acc, torch.float32, "acc")
        arr[k].grad = ptr(grad, torch.float32, "grad")
        arr[k].touched = ptr(touched, torch.int32, "touched") if touched is not None else None
        arr[k].n_rows = data.shape[0]
        arr[k].normalize = int(normalize)
        arr[k].grad_copies = 1 if grad.dim() == 2 else grad.shape[0]
    if tuning is not None:       # this call's knobs (TuningStruct)
        rc = lib().mke_rows_update_multi_t(arr, C.c_int(len(tables)), C.c_int32(tag), C.c_int(stride), C.c_int(dim),
                                           C.c_int(optimizer), C.c_float(lr), None, C.byref(tuning), _stream())
        _check(rc, "mke_rows_update_multi_t")
        return
    rc = lib().mke_rows_update_multi(arr, C.c_int(len(tables)), C.c_int32(tag), C.c_int(stride), C.c_int(dim),
                                     C.c_int(optimizer), C.c_float(lr), _stream())
    _check(rc, "mke_rows_update_multi")


def neg_sample(pos, pos_offset, pos_kg, sides, neg_per_pos, max_try, seed, stream_id, neg_out):
    """sides: (KGSideStruct * 2) host array; pos_kg: uint8 CUDA tensor or None (all KG 0)."""
    ph, pr, pt = pos
    nh, nr, nt = neg_out
    rc = lib().mke_neg_sample(
        _dev(ph, torch.int32, "pos_h"), _dev(pr, torch.int32, "pos_r"), _dev(pt, torch.int32, "pos_t"),
        C.c_int64(ph.numel()), C.c_int64(pos_offset), _dev(pos_kg, torch.uint8, "pos_kg"), sides,
        C.c_int(neg_per_pos), C.c_int(max_try), C.c_uint32(seed[0] & 0xFFFFFFFF), C.c_uint32(seed[1] & 0xFFFFFFFF),
        C.c_uint32(stream_id & 0xFFFFFFFF), _dev(nh, torch.int32, "neg_h"), _dev(nr, torch.int32, "neg_r"),
        _dev(nt, torch.int32, "neg_t"), _stream())
    _check(rc, "mke_neg_sample")


def neg_sample_at(pos, pos_index, pos_kg, sides, neg_per_pos, max_try, seed, stream_id, neg_out):
    """mke_neg_sample_at: like neg_sample, with an explicit int32 epoch position per positive."""
    ph, pr, pt = pos
    nh, nr, nt = neg_out
    rc = lib().mke_neg_sample_at(
        _dev(ph, torch.int32, "pos_h"), _dev(pr, torch.int32, "pos_r"), _dev(pt, torch.int32, "pos_t"),
        C.c_int64(ph.numel()), _dev(pos_index, torch.int32, "pos_index"), _dev(pos_kg, torch.uint8, "pos_kg"), sides,
        C.c_int(neg_per_pos), C.c_int(max_try), C.c_uint32(seed[0] & 0xFFFFFFFF), C.c_uint32(seed[1] & 0xFFFFFFFF),
        C.c_uint32(stream_id & 0xFFFFFFFF), _dev(nh, torch.int32, "neg_h"), _dev(nr, torch.int32, "neg_r"),
        _dev(nt, torch.int32, "neg_t"), _stream())
    _check(rc, "mke_neg_sample_at")


SIM_SELECT_KPADS = (16, 32, 48, 64, 80, 96, 112, 128, 160, 192, 208, 256, 320)   # instantiations of k_sim_select / k_sim_sample / k_align_rank: up to MAX_STRIDE


def sim_select(emb: torch.Tensor, kpad: int, row_lo: int, row_hi: int, tau: torch.Tensor, n_seg: int, seg_cap: int):
    """mke_sim_select -> (cand int32 [rows, n_seg, seg_cap, 2] = (column, similarity bits) pairs, seg_count int32 [rows, n_seg])."""
    if emb.dim() != 2 or emb.stride(1) != 1 or emb.dtype != torch.float32:
        raise MultiKEHipError("sim_select: emb must be a row-major float32 matrix")
    rows = row_hi - row_lo
    if tau.numel() != rows:
        raise MultiKEHipError("sim_select: one threshold per row")
    cand = torch.empty(rows, n_seg, seg_cap, 2, dtype=torch.int32, device=emb.device)
    cnt = torch.empty(rows, n_seg, dtype=torch.int32, device=emb.device)
    rc = lib().mke_sim_select(C.c_void_p(emb.data_ptr()), C.c_int(emb.stride(0)), C.c_int(kpad), C.c_int64(emb.shape[0]),
                              C.c_int64(row_lo), C.c_int64(row_hi), _dev(tau, torch.float32, "tau"), C.c_int(n_seg),
                              C.c_int(seg_cap), _dev(cand, torch.int32, "cand"), _dev(cnt, torch.int32, "seg_count"), _stream())
    _check(rc, "mke_sim_select")
    return cand, cnt


def sim_sample(emb: torch.Tensor, kpad: int, row_lo: int, row_hi: int, samp: torch.Tensor):
    """mke_sim_sample -> float32 [row_hi - row_lo, n_samp] similarities of the rows to the sample rows."""
    out = torch.empty(row_hi - row_lo, samp.shape[0], dtype=torch.float32, device=emb.device)
    rc = lib().mke_sim_sample(_dev(emb, torch.float32, "emb"), C.c_int(emb.stride(0)), C.c_int(kpad), C.c_int64(emb.shape[0]),
                              C.c_int64(row_lo), C.c_int64(row_hi), _dev(samp, torch.float32, "samp"), C.c_int(samp.stride(0)),
                              C.c_int(samp.shape[0]), _dev(out, torch.float32, "out"), _stream())
    _check(rc, "mke_sim_sample")
    return out


def topk_candidates(cand: torch.Tensor, seg_count: torch.Tensor, k: int, id_map=None):
    """mke_topk_candidates over sim_select's pairs -> (out_idx int32 [rows, k], status int32 [rows])."""
    rows, n_seg, seg_cap, _ = cand.shape
    out = torch.empty(rows, k, dtype=torch.int32, device=cand.device)
    status = torch.empty(rows, dtype=torch.int32, device=cand.device)
    rc = lib().mke_topk_candidates(_dev(cand, torch.int32, "cand"), _dev(seg_count, torch.int32, "seg_count"), C.c_int64(rows),
                                   C.c_int(n_seg), C.c_int(seg_cap), C.c_int(k), _dev(id_map, torch.int32, "id_map"),
                                   _dev(out, torch.int32, "out_idx"), C.c_void_p(0), _dev(status, torch.int32, "status"), _stream())
    _check(rc, "mke_topk_candidates")
    return out, status


def topk_rows(vals: torch.Tensor, k: int, idx=None, seg_count=None, id_map=None, want_idx=True, want_kth=False):
    """mke_topk_rows over vals [rows, n_seg, seg_cap] (or [rows, seg_cap]) -> (out_idx int32 [rows, k] | None,
    kth float32 [rows] | None, status int32 [rows])."""
    if vals.dim() == 2:
        vals = vals.unsqueeze(1)
    if not vals.is_contiguous():
        raise MultiKEHipError("topk_rows: vals must be contiguous")
    rows, n_seg, seg_cap = vals.shape
    out = torch.empty(rows, k, dtype=torch.int32, device=vals.device) if want_idx else None
    kth = torch.empty(rows, dtype=torch.float32, device=vals.device) if want_kth else None
    status = torch.empty(rows, dtype=torch.int32, device=vals.device)
    rc = lib().mke_topk_rows(_dev(vals, torch.float32, "vals"), _dev(idx, torch.int32, "idx"),
                             _dev(seg_count, torch.int32, "seg_count"), C.c_int64(rows), C.c_int(n_seg), C.c_int(seg_cap),
                             C.c_int(k), _dev(id_map, torch.int32, "id_map"), _dev(out, torch.int32, "out_idx"),
                             _dev(kth, torch.float32, "out_kth"), _dev(status, torch.int32, "status"), _stream())
    _check(rc, "mke_topk_rows")
    return out, kth, status


def topk_long(vals: torch.Tensor, k: int, id_map=None) -> torch.Tensor:
    """mke_topk_long over whole rows vals [rows, n] (any n) -> int32 [rows, k], column order."""
    if vals.dim() != 2 or vals.stride(1) != 1:
        raise MultiKEHipError("topk_long: vals must be [rows, n] with unit column stride")
    rows, n = vals.shape
    out = torch.empty(rows, k, dtype=torch.int32, device=vals.device)
    rc = lib().mke_topk_long(C.c_void_p(vals.data_ptr()) if vals.is_cuda and vals.dtype == torch.float32 else _dev(vals, torch.float32, "vals"),
                             C.c_int64(rows), C.c_int64(n), C.c_int64(vals.stride(0) if rows > 1 else n), C.c_int(k),
                             _dev(id_map, torch.int32, "id_map"), _dev(out, torch.int32, "out_idx"), _stream())
    _check(rc, "mke_topk_long")
    return out


def mapping_scratch_floats(n: int, dim: int) -> int:
    return int(lib().mke_mapping_scratch_floats(C.c_int64(n), C.c_int(dim)))


MAP_FWD, MAP_TAIL, MAP_BWD, MAP_UPD, MAP_ALL = 1, 2, 4, 8, 15
MAPPING_MAX_VIEWS = 3


def mapping_step_phases(args: MappingStepArgs, loss_partials: torch.Tensor, phases: int):
    rc = lib().mke_mapping_step_phases(C.byref(args), _dev(loss_partials, torch.float64, "loss_partials"), C.c_int(phases), _stream())
    _check(rc, "mke_mapping_step_phases")


def mapping_step(args: MappingStepArgs, loss_partials: torch.Tensor):
    rc = lib().mke_mapping_step(C.byref(args), _dev(loss_partials, torch.float64, "loss_partials"), _stream())
    _check(rc, "mke_mapping_step")


def mapping_steps(args: MappingStepArgs, step_off: np.ndarray, loss_ring: torch.Tensor):
    """loss_ring: float64 [ring, 4, LOSS_PARTIALS]."""
    off = np.ascontiguousarray(step_off, dtype=np.int64)
    rc = lib().mke_mapping_steps(C.byref(args), off.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int(len(off) - 1),
                                 _dev(loss_ring, torch.float64, "loss_ring"), C.c_int(loss_ring.shape[0]), _stream())
    _check(rc, "mke_mapping_steps")


def align_steps(plan: AlignPlanStruct):
    rc = lib().mke_align_steps(C.byref(plan), _stream())
    _check(rc, "mke_align_steps")


def relation_steps(plan: RelationPlanStruct, step_begin: int, step_end: int):
    rc = lib().mke_relation_steps(C.byref(plan), C.c_int(step_begin), C.c_int(step_end), _stream())
    _check(rc, "mke_relation_steps")


def tripleset_build(h, r, t, keys):
    rc = lib().mke_tripleset_build(_dev(h, torch.int32, "h"), _dev(r, torch.int32, "r"), _dev(t, torch.int32, "t"),
                                   C.c_int64(h.numel()), _dev(keys, torch.int64, "keys"), C.c_uint64(keys.numel()),
                                   _stream())
    _check(rc, "mke_tripleset_build")


def tripleset_query(h, r, t, keys, out):
    rc = lib().mke_tripleset_query(_dev(h, torch.int32, "h"), _dev(r, torch.int32, "r"), _dev(t, torch.int32, "t"),
                                   C.c_int64(h.numel()), _dev(keys, torch.int64, "keys"), C.c_uint64(keys.numel()),
                                   _dev(out, torch.uint8, "out"), _stream())
    _check(rc, "mke_tripleset_query")


def gathered_logistic_fwd_bwd(hs, rs, ts, ws, sign, gh, gr, gt, loss_partials):
    n, dim = hs.shape
    rc = lib().mke_gathered_logistic_fwd_bwd(
        _dev(hs, torch.float32, "hs"), _dev(rs, torch.float32, "rs"), _dev(ts, torch.float32, "ts"),
        _dev(ws, torch.float32, "ws"), C.c_int64(n), C.c_int(dim), C.c_int(dim), C.c_int(sign),
        _dev(gh, torch.float32, "gh"), _dev(gr, torch.float32, "gr"), _dev(gt, torch.float32, "gt"),
        _dev(loss_partials, torch.float64, "loss_partials"), _stream())
    _check(rc, "mke_gathered_logistic_fwd_bwd")


def gathered_alignment_fwd_bwd(a, b, ga, gb, loss_partials):
    n, dim = a.shape
    rc = lib().mke_gathered_alignment_fwd_bwd(
        _dev(a, torch.float32, "a"), _dev(b, torch.float32, "b"), C.c_int64(n), C.c_int(dim), C.c_int(dim),
        _dev(ga, torch.float32, "ga"), _dev(gb, torch.float32, "gb"),
        _dev(loss_partials, torch.float64, "loss_partials"), _stream())
    _check(rc, "mke_gathered_alignment_fwd_bwd")


def align_fwd_bwd(table_a, a_normalize, table_b, b_normalize, dim, ia, ib, weight, grad_a, touched_a, grad_b, touched_b,
                  tag, loss_partials):
    rc = lib().mke_align_fwd_bwd(
        _dev(table_a, torch.float32, "table_a"), C.c_int(int(a_normalize)), _dev(table_b, torch.float32, "table_b"),
        C.c_int(int(b_normalize)), C.c_int(table_a.shape[1]), C.c_int(dim), _dev(ia, torch.int32, "ia"),
        _dev(ib, torch.int32, "ib"), C.c_int64(ia.numel()), C.c_float(weight), _dev(grad_a, torch.float32, "grad_a"),
        _dev(touched_a, torch.int32, "touched_a"), _dev(grad_b, torch.float32, "grad_b"),
        _dev(touched_b, torch.int32, "touched_b"), C.c_int32(tag), _dev(loss_partials, torch.float64, "loss_partials"),
        _stream())
    _check(rc, "mke_align_fwd_bwd")


def gather_rows(table, normalize, dim, idx, out):
    n = out.shape[0]
    rc = lib().mke_gather_rows(_dev(table, torch.float32, "table"), C.c_int(int(normalize)), C.c_int(table.shape[1]),
                               C.c_int(dim), _dev(idx, torch.int32, "idx"), C.c_int64(n),
                               _dev(out, torch.float32, "out"), _stream())
    _check(rc, "mke_gather_rows")


def probe_rows(a, b, c, idx, out):
    """mke_probe_rows: rows idx of a (and b, c when given) read together; out [len(idx)] float32."""
    rc = lib().mke_probe_rows(_dev(a, torch.float32, "a"), _dev(b, torch.float32, "b"), _dev(c, torch.float32, "c"),
                              C.c_int(a.shape[1]), _dev(idx, torch.int32, "idx"), C.c_int64(idx.numel()),
                              _dev(out, torch.float32, "out"), _stream())
    _check(rc, "mke_probe_rows")


def rowset_build(streams, flags, counts, req, id_map, overflow, n_ranks, capacity):
    """streams: up to four int32 id tensors."""
    args = []
    for k in range(4):
        t = streams[k] if k < len(streams) else None
        args += [_dev(t, torch.int32, f"ids{k}"), C.c_int64(0 if t is None else t.numel())]
    rc = lib().mke_rowset_build(*args, _dev(flags, torch.int32, "flags"), _dev(counts, torch.int32, "counts"),
                                _dev(req, torch.int32, "req"), _dev(id_map, torch.int32, "id_map"),
                                _dev(overflow, torch.int32, "overflow"), C.c_int(n_ranks), C.c_int(capacity), _stream())
    _check(rc, "mke_rowset_build")


def rowset_remap(streams, outs, id_map, flags, reset_req=None, reset_counts=None, want=None, slot_of=None, n_ranks=0,
                 capacity=0):
    args = []
    for k in range(4):
        t = streams[k] if k < len(streams) else None
        o = outs[k] if k < len(outs) else None
        args += [_dev(t, torch.int32, f"ids{k}"), _dev(o, torch.int32, f"out{k}"), C.c_int64(0 if t is None else t.numel())]
    rc = lib().mke_rowset_remap(*args, _dev(id_map, torch.int32, "id_map"), _dev(flags, torch.int32, "flags"),
                                _dev(reset_req, torch.int32, "reset_req"), C.c_int64(0 if reset_req is None else reset_req.numel()),
                                _dev(reset_counts, torch.int32, "reset_counts"),
                                C.c_int(0 if reset_counts is None else reset_counts.numel()),
                                _dev(want, torch.int32, "want"), _dev(slot_of, torch.int32, "slot_of"), C.c_int(n_ranks),
                                C.c_int(capacity), _stream())
    _check(rc, "mke_rowset_remap")


def rows_gather_padded(table, idx, out, zero_rows=None):
    rc = lib().mke_rows_gather_padded(_dev(table, torch.float32, "table"), C.c_int(table.shape[1]),
                                      _dev(idx, torch.int32, "idx"), C.c_int64(idx.numel()),
                                      _dev(out, torch.float32, "out"), _dev(zero_rows, torch.float32, "zero_rows"), _stream())
    _check(rc, "mke_rows_gather_padded")


def rows_scatter_add(idx, rows, dim, grad, touched, tag, reset_req=None, reset_counts=None):
    rc = lib().mke_rows_scatter_add(_dev(idx, torch.int32, "idx"), _dev(rows, torch.float32, "rows"),
                                    C.c_int64(idx.numel()), C.c_int(grad.shape[1]), C.c_int(dim),
                                    _dev(grad, torch.float32, "grad"), _dev(touched, torch.int32, "touched"), C.c_int32(tag),
                                    _dev(reset_req, torch.int32, "reset_req"), _dev(reset_counts, torch.int32, "reset_counts"),
                                    C.c_int(0 if reset_counts is None else reset_counts.numel()), _stream())
    _check(rc, "mke_rows_scatter_add")


def cnn_conv_params(dim: int) -> int:
    return 2 * dim + 52


def cnn_params(dim: int) -> int:
    return cnn_conv_params(dim) + 4 * dim * dim + dim


def attr_conv_fwd(attr, attr_normalize, lit, dim, ia, iv, params, flat):
    rc = lib().mke_attr_conv_fwd(_dev(attr, torch.float32, "attr_table"), C.c_int(attr.shape[1]), C.c_int(int(attr_normalize)),
                                 _dev(lit, torch.float32, "lit_table"), C.c_int(lit.shape[1]), C.c_int(dim),
                                 _dev(ia, torch.int32, "ia"), _dev(iv, torch.int32, "iv"), C.c_int64(ia.numel()),
                                 _dev(params, torch.float32, "params"), _dev(flat, torch.float32, "flat"),
                                 C.c_int(flat.shape[1]), _stream())
    _check(rc, "mke_attr_conv_fwd")


def cnn_workspace_floats(dim: int) -> int:
    """MKE_CNN_WORKSPACE_FLOATS(dim)"""
    return 32 * (2 * dim + 64)


def attr_conv_bwd(attr, attr_normalize, lit, dim, ia, iv, params, dflat, grad_params, grad_attr, touched_attr, tag,
                  workspace=None):
    rc = lib().mke_attr_conv_bwd(_dev(attr, torch.float32, "attr_table"), C.c_int(attr.shape[1]), C.c_int(int(attr_normalize)),
                                 _dev(lit, torch.float32, "lit_table"), C.c_int(lit.shape[1]), C.c_int(dim),
                                 _dev(ia, torch.int32, "ia"), _dev(iv, torch.int32, "iv"), C.c_int64(ia.numel()),
                                 _dev(params, torch.float32, "params"), _dev(dflat, torch.float32, "dflat"),
                                 _dev(grad_params, torch.float32, "grad_params"), _dev(grad_attr, torch.float32, "grad_attr"),
                                 _dev(touched_attr, torch.int32, "touched_attr"), C.c_int32(tag),
                                 _dev(workspace, torch.float32, "workspace"), _stream())
    _check(rc, "mke_attr_conv_bwd")


def attr_tail_z(z, bias, sumsq_partials):
    n, dim = z.shape
    rc = lib().mke_attr_tail_z(_dev(z, torch.float32, "z"), _dev(bias, torch.float32, "bias"), C.c_int64(n), C.c_int(dim),
                               _dev(sumsq_partials, torch.float64, "sumsq_partials"), _stream())
    _check(rc, "mke_attr_tail_z")


def attr_tail_loss(z, sumsq_partials, ent, ent_normalize, ih, weights, scale, gout, dot_partials, grad_ent, touched_ent, tag,
                   loss_partials):
    n, dim = z.shape
    rc = lib().mke_attr_tail_loss(_dev(z, torch.float32, "z"), _dev(sumsq_partials, torch.float64, "sumsq"),
                                  _dev(ent, torch.float32, "ent_table"), C.c_int(ent.shape[1]), C.c_int(int(ent_normalize)),
                                  _dev(ih, torch.int32, "ih"), _dev(weights, torch.float32, "weights"), C.c_float(scale),
                                  C.c_int64(n), C.c_int(dim), _dev(gout, torch.float32, "gout"),
                                  _dev(dot_partials, torch.float64, "dot_partials"), _dev(grad_ent, torch.float32, "grad_ent"),
                                  _dev(touched_ent, torch.int32, "touched_ent"), C.c_int32(tag),
                                  _dev(loss_partials, torch.float64, "loss_partials"), _stream())
    _check(rc, "mke_attr_tail_loss")


def attr_tail_bwd(z, gout, sumsq_partials, dot_partials, grad_bias=None):
    n, dim = z.shape
    rc = lib().mke_attr_tail_bwd(_dev(z, torch.float32, "z"), _dev(gout, torch.float32, "gout"),
                                 _dev(sumsq_partials, torch.float64, "sumsq"), _dev(dot_partials, torch.float64, "dot"),
                                 C.c_int64(n), C.c_int(dim), _dev(grad_bias, torch.float32, "grad_bias"), _stream())
    _check(rc, "mke_attr_tail_bwd")


def dense_update(param, acc, grad, optimizer, lr):
    rc = lib().mke_dense_update(_dev(param, torch.float32, "param"), _dev(acc, torch.float32, "acc"),
                                _dev(grad, torch.float32, "grad"), C.c_int64(param.numel()), C.c_int(optimizer), C.c_float(lr),
                                _stream())
    _check(rc, "mke_dense_update")


def oc_block_floats(capacity: int, stride: int) -> int:
    return int(lib().mke_oc_block_floats(C.c_int64(capacity), C.c_int(stride)))


def oc_pack_codes(pos_h, neg_h, neg_t, neg_per_pos: int, codes):
    rc = lib().mke_oc_pack_codes(_dev(pos_h, torch.int32, "pos_h"), _dev(neg_h, torch.int32, "neg_h"),
                                 _dev(neg_t, torch.int32, "neg_t"), C.c_int64(pos_h.numel()), C.c_int(neg_per_pos),
                                 _dev(codes, torch.int32, "codes"), _stream())
    _check(rc, "mke_oc_pack_codes")


OC_NEED_HR, OC_NEED_RT, OC_CODE_MASK = 0x40000000, 0x80000000, 0x3FFFFFFF


def oc_plan(pos_h, pos_t, codes, neg_per_pos: int, part_lo, n_parts: int, n_ranks: int, rank: int, slot_h, slot_t, own_h, own_t, counts):
    """The epoch's slots / owned lists / per-(part, owner) counts in one launch (mke_oc_plan); `codes`: the whole epoch's codes
    by epoch position (their group flags say which vector a positive needs), None when neg_per_pos == 0."""
    i32 = torch.int32
    rc = lib().mke_oc_plan(_dev(pos_h, i32, "pos_h"), _dev(pos_t, i32, "pos_t"),
                           _dev(codes, i32, "codes") if neg_per_pos else None, C.c_int(neg_per_pos), _dev(part_lo, torch.int64, "part_lo"),
                           C.c_int(n_parts), C.c_int(n_ranks), C.c_int(rank), _dev(slot_h, i32, "slot_h"), _dev(slot_t, i32, "slot_t"),
                           _dev(own_h, i32, "own_h"), _dev(own_t, i32, "own_t"), _dev(counts, i32, "counts"), _stream())
    _check(rc, "mke_oc_plan")


def oc_bases(step: OcStepStruct, send_block):
    rc = lib().mke_oc_bases(C.byref(step), _dev(send_block, torch.float32, "send_block"), _stream())
    _check(rc, "mke_oc_bases")


def oc_count(step: OcStepStruct):
    rc = lib().mke_oc_count(C.byref(step), _stream())
    _check(rc, "mke_oc_count")


def oc_score(step: OcStepStruct, v_all, block_floats: int, g_all, loss_partials):
    rc = lib().mke_oc_score(C.byref(step), _dev(v_all, torch.float32, "v_all"), C.c_int64(block_floats),
                            _dev(g_all, torch.float32, "g_all"), _dev(loss_partials, torch.float64, "loss_partials"), _stream())
    _check(rc, "mke_oc_score")


def oc_apply(step: OcStepStruct, gv):
    rc = lib().mke_oc_apply(C.byref(step), _dev(gv, torch.float32, "gv"), _stream())
    _check(rc, "mke_oc_apply")


OC_BASES, OC_COUNT, OC_SCORE, OC_APPLY, OC_UPDATE, OC_PASS2 = 1, 2, 4, 8, 16, 32


def oc_em_plan_temp_bytes(capacity: int) -> int:
    n = lib().mke_oc_em_plan_temp_bytes(C.c_int64(capacity))
    if n < 0:
        raise MultiKEHipError("mke_oc_em_plan_temp_bytes: capacity out of range")
    return int(n)


def oc_em_plan(args: OcEmPlanArgs):
    """mke_oc_em_plan: the epoch's references to this rank's rows, sorted by (step, row) — struct of raw device addresses."""
    _check(lib().mke_oc_em_plan(C.byref(args), _stream()), "mke_oc_em_plan")


def oc_steps(loop: OcLoopStruct, step_begin: int, step_end: int):
    """mke_oc_steps: global steps [step_begin, step_end) of the current epoch enqueued by one native call."""
    _check(lib().mke_oc_steps(C.byref(loop), C.c_int(step_begin), C.c_int(step_end), _stream()), "mke_oc_steps")


def oc_pass2(step: OcStepStruct):
    _check(lib().mke_oc_pass2(C.byref(step), _stream()), "mke_oc_pass2")


def oc_run(step: OcStepStruct, phases: int, send, v_all, block_floats: int, g_all, gv, loss_partials):
    """mke_oc_run with RAW device addresses (ints / None) — the caller validated the tensors once per epoch."""
    rc = lib().mke_oc_run(C.byref(step), C.c_int(phases), C.c_void_p(send), C.c_void_p(v_all), C.c_int64(block_floats),
                          C.c_void_p(g_all), C.c_void_p(gv), C.c_void_p(loss_partials), _stream())
    _check(rc, "mke_oc_run")


def ae_scratch_floats(plan: AEPlanStruct, rows: int) -> int:
    n = lib().mke_ae_scratch_floats(C.byref(plan), C.c_int64(rows))
    if n < 0:
        raise MultiKEHipError("mke_ae_scratch_floats: bad plan")
    return int(n)


def ae_train_steps(plan: AEPlanStruct, x: torch.Tensor, batch_rows: int, loss_out: torch.Tensor):
    """mke_ae_train_steps over the rows of x (float32 [n, ldx] CUDA, row-major); loss_out: float64 [n_batches]."""
    rc = lib().mke_ae_train_steps(C.byref(plan), _dev(x, torch.float32, "x"), C.c_int64(x.shape[0]), C.c_int64(x.stride(0)),
                                  C.c_int64(batch_rows), _dev(loss_out, torch.float64, "loss_out"), _stream())
    _check(rc, "mke_ae_train_steps")


AE_ENC, AE_DEC, AE_BWD, AE_UPD, AE_ALL = 1, 2, 4, 8, 15


def ae_step_phases(plan: AEPlanStruct, x, rows: int, ldx: int, global_rows: int, phases: int, loss_out: torch.Tensor):
    """mke_ae_step_phases: one batch of the auto-encoder cut at its batch-wide sums (x: float32 [rows, ldx] CUDA or None)."""
    rc = lib().mke_ae_step_phases(C.byref(plan), (_dev(x, torch.float32, "x") if rows else None), C.c_int64(rows), C.c_int64(ldx),
                                  C.c_int64(global_rows), C.c_int(phases), _dev(loss_out, torch.float64, "loss_out"), _stream())
    _check(rc, "mke_ae_step_phases")


def ae_encode(plan: AEPlanStruct, x: torch.Tensor, out: torch.Tensor):
    rc = lib().mke_ae_encode(C.byref(plan), _dev(x, torch.float32, "x"), C.c_int64(x.shape[0]), C.c_int64(x.stride(0)),
                             _dev(out, torch.float32, "out"), C.c_int64(out.stride(0)), _stream())
    _check(rc, "mke_ae_encode")


def dense_layer_fwd(x: torch.Tensor, w: torch.Tensor, b, act: int, out: torch.Tensor):
    """out = act(x @ w + b) on the hand-written MFMA GEMM (bias / activation in its epilogue)."""
    M, K = x.shape
    K2, N = w.shape
    if K != K2 or tuple(out.shape) != (M, N) or x.stride(1) != 1 or w.stride(1) != 1 or out.stride(1) != 1:
        raise MultiKEHipError(f"dense_layer_fwd: shapes {tuple(x.shape)} x {tuple(w.shape)} -> {tuple(out.shape)} (row-major)")
    for t, nm in ((x, "x"), (w, "w"), (out, "out")):       # row-major with any row stride (padded rows are not "contiguous")
        if not t.is_cuda or t.dtype != torch.float32:
            raise MultiKEHipError(f"dense_layer_fwd: {nm} must be a float32 CUDA tensor")
    rc = lib().mke_dense_layer_fwd(C.c_void_p(x.data_ptr()), C.c_int64(x.stride(0)), C.c_void_p(w.data_ptr()),
                                   C.c_int64(w.stride(0)), _dev(b, torch.float32, "b"), C.c_int(act),
                                   C.c_void_p(out.data_ptr()), C.c_int64(out.stride(0)), C.c_int(M), C.c_int(N), C.c_int(K),
                                   _stream())
    _check(rc, "mke_dense_layer_fwd")


def align_rank(emb1, emb2, kpad, n1, n2, rank, best, ties=None):
    """mke_align_rank over row-major padded emb1 [n1, ld1] / emb2 [n2, ld2]."""
    rc = lib().mke_align_rank(_dev(emb1, torch.float32, "emb1"), C.c_int(emb1.shape[1]), _dev(emb2, torch.float32, "emb2"),
                              C.c_int(emb2.shape[1]), C.c_int(kpad), C.c_int64(n1), C.c_int64(n2),
                              _dev(rank, torch.int32, "rank"), _dev(ties, torch.int32, "ties"), _dev(best, torch.int64, "best"),
                              _stream())
    _check(rc, "mke_align_rank")


def gemm_f32(lhs, rhs, out, transpose_a=False, transpose_b=False, splits=1, accumulate=False):
    """out (=|+=) op(lhs) @ op(rhs) for 2-D float32 CUDA tensors of any strides; `out` row-major."""
    a = lhs.t() if transpose_a else lhs
    b = rhs.t() if transpose_b else rhs
    M, K = a.shape
    K2, N = b.shape
    if K != K2 or tuple(out.shape) != (M, N) or out.stride(1) != 1:
        raise MultiKEHipError(f"gemm_f32: shapes {tuple(a.shape)} x {tuple(b.shape)} -> {tuple(out.shape)}")
    for t, nm in ((lhs, "lhs"), (rhs, "rhs"), (out, "out")):
        if not t.is_cuda or t.dtype != torch.float32:
            raise MultiKEHipError(f"gemm_f32: {nm} must be a float32 CUDA tensor")
    rc = lib().mke_gemm_f32(C.c_void_p(a.data_ptr()), C.c_int64(a.stride(0)), C.c_int64(a.stride(1)),
                            C.c_void_p(b.data_ptr()), C.c_int64(b.stride(0)), C.c_int64(b.stride(1)),
                            C.c_void_p(out.data_ptr()), C.c_int64(out.stride(0)), C.c_int(M), C.c_int(N), C.c_int(K),
                            C.c_int(splits), C.c_int(int(accumulate)), _stream())
    _check(rc, "mke_gemm_f32")


def rows_update_dense(table, slot1, slot2, grad, dim, normalize, opt: OptimizerStruct):
    """mke_rows_update_dense: Adam / Adadelta over EVERY row of the table (grad consumed)."""
    rc = lib().mke_rows_update_dense(_dev(table, torch.float32, "table"), _dev(slot1, torch.float32, "slot1"),
                                     _dev(slot2, torch.float32, "slot2"), _dev(grad, torch.float32, "grad"),
                                     C.c_int64(table.shape[0]), C.c_int(table.shape[1]), C.c_int(dim), C.c_int(int(normalize)),
                                     C.byref(opt), _stream())
    _check(rc, "mke_rows_update_dense")


def dense_update_opt(param, slot1, slot2, grad, opt: OptimizerStruct):
    """mke_dense_update_opt: Adam / Adadelta over a flat parameter buffer (grad consumed)."""
    rc = lib().mke_dense_update_opt(_dev(param, torch.float32, "param"), _dev(slot1, torch.float32, "slot1"),
                                    _dev(slot2, torch.float32, "slot2"), _dev(grad, torch.float32, "grad"),
                                    C.c_int64(param.numel()), C.byref(opt), _stream())
    _check(rc, "mke_dense_update_opt")


def attr_scratch_floats(n: int, dim: int) -> int:
    return int(lib().mke_attr_scratch_floats(C.c_int64(n), C.c_int(dim)))


def attr_steps(args: AttrStepArgs, step_off: np.ndarray, loss_ring: torch.Tensor):
    """mke_attr_steps: `step_off` host int64 [n_steps + 1]; loss_ring float64 [ring, LOSS_PARTIALS] on the device."""
    off = np.ascontiguousarray(step_off, dtype=np.int64)
    rc = lib().mke_attr_steps(C.byref(args), off.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int(len(off) - 1),
                              _dev(loss_ring, torch.float64, "loss_ring"), C.c_int(loss_ring.shape[0]), _stream())
    _check(rc, "mke_attr_steps")


def sample_distinct(n: int, batch: int, n_steps: int, seed=(0, 0), stream_id: int = 0, device="cuda") -> torch.Tensor:
    """mke_sample_distinct -> int32 [n_steps, batch]: `batch` distinct positions of range(n) per step."""
    out = torch.empty(n_steps, batch, dtype=torch.int32, device=device)
    rc = lib().mke_sample_distinct(C.c_int64(n), C.c_int(batch), C.c_int(n_steps), C.c_uint32(seed[0] & 0xFFFFFFFF),
                                   C.c_uint32(seed[1] & 0xFFFFFFFF), C.c_uint32(stream_id & 0xFFFFFFFF),
                                   _dev(out, torch.int32, "out"), _stream())
    _check(rc, "mke_sample_distinct")
    return out


ATTR_FWD, ATTR_TAIL, ATTR_BWD, ATTR_UPD = 1, 2, 4, 8


def attr_step_phases(args: AttrStepArgs, phases: int):
    rc = lib().mke_attr_step_phases(C.byref(args), C.c_int(phases), _stream())
    _check(rc, "mke_attr_step_phases")


def attr_step(args: AttrStepArgs):
    rc = lib().mke_attr_step(C.byref(args), _stream())
    _check(rc, "mke_attr_step")
