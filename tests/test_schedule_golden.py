"""f1 pinned against the reference's OWN loops: `tests/golden/schedule_golden.json` holds the call traces of the reference's
`MultiKE_CV.run` (code/MultiKE_CSL.py:36-107) and `MultiKE_Late.run` (code/MultiKE_Late.py:201-280), executed unmodified by
`tests/golden/make_golden.py` on a recording stand-in model (tests/schedule_mock.py).  The product's drivers, run on the same
stand-in, must emit the same trace event for event: phase order, the `i > start_predicate_soft_alignment`, `i % 10`,
`i % eval_freq`, `i % truncated_freq` gates, step counts and task splits, the list VERSION every phase receives after a soft
predicate-alignment refresh, the neighbour tables handed to the relation view, the early `break`, `save` and the closing
tests.  Needs no GPU (nothing is trained)."""
import contextlib
import io
import json
import os
from unittest import mock

import pytest

import schedule_mock as sm
from conftest import GOLDEN


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "schedule_golden.json")) as f:
        return json.load(f)["schedules"]


def _product_trace(method, scenario):
    from multike_amd import MultiKE_CSL as p_csl
    from multike_amd import MultiKE_Late as p_late
    cls = p_csl.MultiKE_CV if method == "ITC" else p_late.MultiKE_Late
    trace = []
    model = object.__new__(cls)
    fns = sm.instrument(model, scenario, trace)
    patches = [mock.patch.object(p_late, "valid", fns["valid"]), mock.patch.object(p_late, "test", fns["test"]),
               mock.patch.object(p_late, "valid_WVA", fns["valid_WVA"]), mock.patch.object(p_late, "test_WVA", fns["test_WVA"]),
               mock.patch.object(p_late, "neighbour_table", fns["neighbours"]),
               mock.patch.object(p_csl, "valid", fns["valid"]), mock.patch.object(p_csl, "test", fns["test"])]
    with contextlib.ExitStack() as st, contextlib.redirect_stdout(io.StringIO()):
        for p in patches:
            st.enter_context(p)
        model.run()
    return json.loads(json.dumps(trace))          # tuples -> lists, as the fixture went through JSON


@pytest.mark.parametrize("method", ["ITC", "SSL"])
@pytest.mark.parametrize("scenario", sorted(sm.SCENARIOS))
def test_driver_emits_the_reference_trace(golden, method, scenario):
    exp = golden[f"{method}/{scenario}"]
    got = _product_trace(method, scenario)
    for k, (a, b) in enumerate(zip(got, exp)):
        assert a == b, f"{method}/{scenario}: event {k} differs: product {a!r:.200} != reference {b!r:.200}"
    assert len(got) == len(exp), (len(got), len(exp), got[len(exp):][:3], exp[len(got):][:3])


def test_fixture_covers_every_gate(golden):
    """The scenarios really exercise what they claim (a fixture that never reaches a gate pins nothing)."""
    names = lambda key: [e[0] for e in golden[key]]
    t = golden["ITC/default_gates_30_epochs"]
    upd = [k for k, e in enumerate(t) if e[0] == "update_predicate_alignment"]
    assert len(upd) == 4                                        # relation + attribute after epochs 10 and 20; epoch 30 breaks first
    soft = [e[1] for e in t if e[0] == "train_cross_kg_relation_inference_1epo"]
    assert soft[0] == 11 and soft[-1] == 30                     # i > 10
    versions = sorted({e[2][0][1] for e in t if e[0] == "train_cross_kg_relation_inference_1epo"})
    assert versions == [101, 102]                               # the lists were rebuilt after epochs 10 and 20
    assert [e[4] for e in t if e[0] == "train_relation_view_1epo"][19:22] == [None, "nb1", "nb1"]   # refresh after epoch 20
    assert names("ITC/early_stop_after_second_validation").count("valid") == 6      # two rounds, then the break
    assert "valid" not in names("ITC/valid_never_reached") and names("ITC/valid_never_reached")[-5:] == ["save", "test", "test", "test", "test"]
    assert "generate_neighbours" not in names("ITC/uniform_sampling")
    s = names("SSL/small_gates")
    assert s.count("train_shared_space_mapping_1epo") == 5 and "valid_WVA" in s and s[-1] == "test" and s[-2] == "test_WVA"
    assert "train_common_space_learning_1epo" not in s and "train_common_space_learning_1epo" in names("ITC/small_gates")
