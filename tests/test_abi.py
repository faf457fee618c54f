"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/multike_hip.h declares, argument validation returns error codes (no compute without a GPU), and the
ctypes structs match the header's layout."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _header():
    with open(os.path.join(ROOT, "include", "multike_hip.h")) as f:
        return f.read()


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from multike_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        g.build()
    return _lib.lib()


def test_header_symbols_are_exported_and_bound(lib):
    from multike_amd import _lib
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(mke_\w+)\s*\(", _header(), flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    raw = C.CDLL(_lib.SO_PATH)
    for sym in declared:
        assert getattr(raw, sym) is not None


def test_version_and_constants(lib):
    from multike_amd import _lib
    h = _header()
    assert lib.mke_version() == int(re.search(r"#define MKE_VERSION (\d+)", h).group(1))
    assert _lib.LOSS_PARTIALS == int(re.search(r"#define MKE_LOSS_PARTIALS (\d+)", h).group(1))
    assert _lib.MAX_STRIDE == int(re.search(r"#define MKE_MAX_STRIDE (\d+)", h).group(1))


def test_struct_layouts_match_header():
    """Field order of each ctypes.Structure == member order in the header."""
    from multike_amd import _lib
    h = _header()
    for cname, st in (("mke_kg_side", _lib.KGSideStruct), ("mke_update_table", _lib.UpdateTableStruct),
                      ("mke_relation_plan", _lib.RelationPlanStruct), ("mke_oc_step", _lib.OcStepStruct),
                      ("mke_ae_plan", _lib.AEPlanStruct), ("mke_oc_em_plan_args", _lib.OcEmPlanArgs), ("mke_oc_comm", _lib.OcCommStruct),
                      ("mke_oc_loop", _lib.OcLoopStruct), ("mke_tuning", _lib.TuningStruct), ("mke_attr_step_args", _lib.AttrStepArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), h, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.search(r"(\w+)\s*(?:\[[^\]]*\])?$", part.strip()).group(1))
        assert names == [f[0] for f in st._fields_], (cname, names)


def test_argument_validation_without_gpu(lib):
    """Bad arguments are rejected before any launch: error code < 0 and a message."""
    null = C.c_void_p(0)
    rc = lib.mke_rows_update(null, null, null, C.c_int(1), null, C.c_int32(1), C.c_int64(4), C.c_int(80), C.c_int(75), C.c_int(1),
                             C.c_int(0), C.c_float(0.1), null)
    assert rc == -1 and b"NULL" in lib.mke_last_error()
    one = C.c_void_p(16)
    rc = lib.mke_rows_update(one, one, one, C.c_int(1), one, C.c_int32(1), C.c_int64(4), C.c_int(75), C.c_int(75), C.c_int(1),
                             C.c_int(0), C.c_float(0.1), null)
    assert rc == -2 and b"stride" in lib.mke_last_error()
    rc = lib.mke_rows_update(one, one, one, C.c_int(1), one, C.c_int32(1), C.c_int64(4), C.c_int(80), C.c_int(75), C.c_int(1),
                             C.c_int(7), C.c_float(0.1), null)
    assert rc == -3
    rc = lib.mke_gathered_logistic_fwd_bwd(one, one, one, null, C.c_int64(3), C.c_int(75), C.c_int(75), C.c_int(2), null,
                                           null, null, one, null)
    assert rc == -2 and b"sign" in lib.mke_last_error()
    # empty work is a no-op success without touching the device
    assert lib.mke_rows_update(one, one, one, C.c_int(1), one, C.c_int32(1), C.c_int64(0), C.c_int(80), C.c_int(75), C.c_int(1),
                               C.c_int(0), C.c_float(0.1), null) == 0


def test_product_refuses_cpu_tensors():
    import torch
    from multike_amd import _lib
    from multike_amd import losses
    with pytest.raises(_lib.MultiKEHipError, match="no CPU path"):
        losses.alignment_loss(torch.zeros(2, 75), torch.zeros(2, 75))


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under multike_amd/ may reference it."""
    pkg = os.path.join(ROOT, "multike_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
                assert "libmke_oracle" not in src, fn


def test_stride_for():
    from multike_amd import _lib
    assert _lib.stride_for(75) == 80 and _lib.stride_for(256) == 256 and _lib.stride_for(4) == 16
    assert _lib.stride_for(100) == 112 and _lib.stride_for(129) == 160
    with pytest.raises(_lib.MultiKEHipError):
        _lib.stride_for(400)


def test_round6_entry_points_validate_without_gpu(lib):
    """ABI 105: the entity-major plan / second pass, the native step loop and the per-call tuning reject bad arguments before any
    launch (NULL structs, a step that is not entity-major, chunk counts, a communicator without entry points, a hub-row table
    mixed with a slot_of table in one update call)."""
    from multike_amd import _lib
    null = C.c_void_p(0)
    assert lib.mke_oc_em_plan(null, null) == -1 and b"NULL" in lib.mke_last_error()
    assert lib.mke_oc_pass2(null, null) == -1
    assert lib.mke_oc_steps(null, C.c_int(0), C.c_int(1), null) == -1
    assert lib.mke_tuning_init(null) == -1
    a = _lib.OcEmPlanArgs()
    a.n_ranks, a.rank, a.n_local, a.n_rel, a.chunks, a.capacity = 2, 0, 10, 3, 9, 100
    assert lib.mke_oc_em_plan(C.byref(a), null) == -2 and b"chunks" in lib.mke_last_error()
    a.chunks = 1
    assert lib.mke_oc_em_plan(C.byref(a), null) == -1 and b"NULL output" in lib.mke_last_error()
    st = _lib.OcStepStruct()
    assert lib.mke_oc_pass2(C.byref(st), null) == -3 and b"entity-major" in lib.mke_last_error()
    lp = _lib.OcLoopStruct()
    assert lib.mke_oc_steps(C.byref(lp), C.c_int(0), C.c_int(0), null) == -1 and b"NULL parts" in lib.mke_last_error()
    t = _lib.tuning(score_splits=2)
    assert [getattr(t, f) for f in _lib.TUNING_FIELDS] == [2] + [_lib.TUNE_DEFAULT] * (len(_lib.TUNING_FIELDS) - 1)
    # a hub-row table and a slot_of table in one update call: refused (the launch instantiates one form or the other)
    arr = (_lib.UpdateTableStruct * 2)()
    for k in (0, 1):
        arr[k].table, arr[k].acc, arr[k].grad, arr[k].n_rows, arr[k].normalize, arr[k].grad_copies = 16, 16, 16, 8, 1, 1
    arr[0].slot_of, arr[0].src_rows, arr[0].n_ranks, arr[0].capacity = 16, 16, 2, 4
    arr[1].hot.slot, arr[1].hot.n_hot, arr[1].hot.copies, arr[1].hot.row0 = 16, 2, 4, 8
    rc = lib.mke_rows_update_multi(arr, C.c_int(2), C.c_int32(1), C.c_int(80), C.c_int(75), C.c_int(0), C.c_float(0.1), null)
    assert rc == -3 and b"cannot share" in lib.mke_last_error()
    arr[0].slot_of = None
    arr[1].hot.row0 = 2                       # copies inside the table's own rows
    rc = lib.mke_rows_update_multi(arr, C.c_int(2), C.c_int32(1), C.c_int(80), C.c_int(75), C.c_int(0), C.c_float(0.1), null)
    assert rc == -2 and b"hub rows" in lib.mke_last_error()
