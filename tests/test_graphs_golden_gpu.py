"""The device model (multike_amd/MultiKE_model.py: the reference's method names over the HIP step loops) against the reference's own
graph definitions EXECUTED (tests/golden/graphs_golden.npz, written by tests/golden/make_golden.py `graphs_fixture`: `MultiKE.
_define_variables` + the `_define_*_graph` methods of /root/reference/code/MultiKE_model.py run unmodified over eagerly forwarded
TensorFlow calls, float64 autograd).  One epoch of one step of every loop that takes its batch as an argument — the four cross-KG
inference loops, common-space learning, the space mapping — from the fixture's tables and batch: the printed loss and every variable
after the step, which must be w - lr g / sqrt(0.1 + g^2) for the variables the reference's optimizer moves and unchanged for all
others (ApplyAdagrad as stated in SURVEY.md 9.3).  (The relation and attribute VIEW loops draw their own batches: their graphs are
held to the same fixture through the oracle, tests/test_graphs_golden.py, and to the executed attribute graph in
tests/test_attr_cnn_gpu.py.)"""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import attr_cnn_oracle as ao

pytestmark = pytest.mark.gpu

TABLES = ("rv_ent_embeds", "rel_embeds", "av_ent_embeds", "attr_embeds", "ent_embeds")
MAPS = ("nv_mapping", "rv_mapping", "av_mapping")
CNN_OF = {"attribute": "_attr_cnn", "ckge_attr": "_ckge_attr_cnn", "ckga_attr": "_ckga_attr_cnn"}


def _model(g, entity_batch_size):
    from multike_amd.MultiKE_model import MultiKE
    from multike_amd.synthetic import synthetic_args
    lr, itc, cvn, cvw, ow = (float(x) for x in g["args"])
    n_ent, d = g["raw_ent_embeds"].shape
    kg = types.SimpleNamespace(entities_list=list(range(n_ent)))
    kgs = types.SimpleNamespace(entities_num=n_ent, relations_num=g["raw_rel_embeds"].shape[0], attributes_num=g["raw_attr_embeds"].shape[0],
                                kg1=kg, kg2=kg)
    data = types.SimpleNamespace(kgs=kgs, value_vectors=g["lit"].astype(np.float32), local_name_vectors=g["name"].astype(np.float32))
    args = synthetic_args(dim=d, batch_size=1000, attribute_batch_size=1000, entity_batch_size=entity_batch_size, learning_rate=lr,
                          ITC_learning_rate=itc, cv_name_weight=cvn, cv_weight=cvw, orthogonal_weight=ow)
    m = MultiKE(data, args, None)
    m._define_variables()
    m._define_attribute_view_graph()
    m._define_cross_kg_entity_reference_relation_view_graph()
    m._define_cross_kg_entity_reference_attribute_view_graph()
    m._define_cross_kg_attribute_reference_graph()
    m._define_cross_kg_relation_reference_graph()
    m._define_common_space_learning_graph()
    for name in TABLES:                                    # the fixture's variables
        getattr(m, name).raw().copy_(torch.as_tensor(g["raw_" + name].astype(np.float32)))
    for name in MAPS:
        getattr(m, name).data.copy_(torch.as_tensor(g["raw_" + name].astype(np.float32)))
    for k, attr in enumerate(CNN_OF.values()):
        cnn = getattr(m, attr)
        for n in ao.PARAM_NAMES:
            cnn.views[n].copy_(torch.as_tensor(g[f"cnn{k}_{n}"].astype(np.float32)))
    m._define_space_mapping_graph()                        # after the mappings hold the fixture's values
    return m


def _state(m):
    out = {n: getattr(m, n).raw().double().cpu().numpy() for n in TABLES}
    out.update({n: getattr(m, n).detach().double().cpu().numpy() for n in MAPS})
    for k, attr in enumerate(CNN_OF.values()):
        out.update({f"cnn{k}_{n}": getattr(m, attr).views[n].double().cpu().numpy().copy() for n in ao.PARAM_NAMES})
    return out


@pytest.mark.parametrize("key", ["ckge_rel", "ckgp_rel", "ckge_attr", "ckga_attr", "common", "mapping"])
def test_one_step_of_a_loop_equals_the_executed_reference_graph(key):
    g = np.load(os.path.join(GOLDEN, "graphs_golden.npz"))
    f = lambda i: g[f"{key}_feed{i}"]
    B = len(f(0))
    m = _model(g, B)                                       # one step: the whole list is the batch (code/MultiKE_model.py:443-444, 462-463)
    if key == "ckge_rel":
        shown = m.train_cross_kg_entity_inference_relation_view_1epo(1, [tuple(int(v) for v in t) for t in zip(f(0), f(1), f(2))])
    elif key == "ckgp_rel":
        shown = m.train_cross_kg_relation_inference_1epo(1, [(int(a), int(b), int(c), float(w)) for a, b, c, w in zip(f(0), f(1), f(2), f(3))])
    elif key == "ckge_attr":
        shown = m.train_cross_kg_entity_inference_attribute_view_1epo(1, [tuple(int(v) for v in t) for t in zip(f(0), f(1), f(2))])
    elif key == "ckga_attr":
        shown = m.train_cross_kg_attribute_inference_1epo(1, [(int(a), int(b), int(c), float(w)) for a, b, c, w in zip(f(0), f(1), f(2), f(3))])
    elif key == "common":
        shown = m.train_common_space_learning_1epo(1, [int(v) for v in f(0)])
    else:
        shown = m.train_shared_space_mapping_1epo(1, [int(v) for v in f(0)])
    torch.cuda.synchronize()
    np.testing.assert_allclose(shown * B, float(g[f"{key}_loss"]), rtol=2e-5)     # the loops print loss / trained samples
    lr = float(g[f"{key}_lr"])
    allowed = set(g[f"{key}_var_list"].tolist()) if int(g[f"{key}_has_var_list"]) else None
    now = _state(m)
    init = {n: g["raw_" + n] for n in TABLES + MAPS}
    init.update({f"cnn{k}_{n}": g[f"cnn{k}_{n}"] for k in range(3) for n in ao.PARAM_NAMES})
    moved = 0
    for name, w0 in init.items():
        gk = f"{key}_g_{name}"
        if gk in g.files and (allowed is None or name in allowed):
            grad = g[gk]
            want = w0 - lr * grad / np.sqrt(0.1 + grad * grad)
            np.testing.assert_allclose(now[name], want, rtol=3e-4, atol=3e-6, err_msg=f"{key}: {name}")
            assert np.abs(want - w0).max() > 1e-4
            moved += 1
        else:
            np.testing.assert_array_equal(now[name], w0.astype(np.float32).astype(np.float64), err_msg=f"{key}: {name} must not move")
    assert moved
