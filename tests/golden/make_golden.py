#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by EXECUTING THE REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py [--reference /root/reference]

What runs, and how:
  * `code/losses.py` is imported unmodified.  Its only dependency is `import tensorflow as tf`, which is
    not installable here, so a module named `tensorflow` is placed in `sys.modules` that forwards the
    dozen op NAMES losses.py uses (reduce_sum, square, log, exp, add, multiply, pow, matmul,
    nn.l2_normalize) to torch.  The op SEQUENCE is therefore the reference's; TF's own kernels are not
    reproduced ("parity unpinned at the TF boundary").
  * `code/base/batch.py` and `code/attr_batch.py` are imported unmodified with empty stand-in modules
    for the imports they never use on this path (`tensorflow`, `gensim`), and run under fixed
    `random.seed` / `np.random.seed`.
  * the attribute-view graph (round 6): `MultiKE._define_attribute_view_graph`, `conv` (code/MultiKE_model.py) and `xavier_init`
    (code/base/initializers.py) are EXECUTED on an instance made without `__init__`, the graph-building calls forwarded eagerly;
    only the leaf ops (tf.layers batch-norm / conv2d / dense, embedding_lookup) are restated — see `cnn_reference_fixture`.
  * graph-level pieces that exist only inside TF (gradient of l2_normalize, ApplyAdagrad) are produced
    with torch autograd and `torch.optim.Adagrad(initial_accumulator_value=0.1, eps=0)` on top of the
    reference's loss functions.

Only data (inputs and expected outputs) is written; no reference source text is stored.
"""
import argparse
import importlib
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------
def install_tf_forwarder():
    import importlib.machinery
    tf = types.ModuleType("tensorflow")
    tf.__spec__ = importlib.machinery.ModuleSpec("tensorflow", None)  # torch probes find_spec("tensorflow")
    tf.reduce_sum = lambda x, axis=None: torch.sum(x) if axis is None else torch.sum(x, dim=axis)
    tf.square = torch.square
    tf.log = torch.log
    tf.exp = torch.exp
    tf.add = torch.add
    tf.multiply = torch.mul
    tf.pow = torch.pow
    tf.matmul = lambda a, b, transpose_b=False: a @ (b.T if transpose_b else b)
    nn = types.ModuleType("tensorflow.nn")

    def l2_normalize(x, axis=None, epsilon=1e-12, dim=None):
        axis = dim if axis is None else axis
        ssq = torch.sum(x * x) if axis is None else torch.sum(x * x, dim=axis, keepdim=True)
        return x * torch.rsqrt(torch.clamp_min(ssq, epsilon))

    nn.l2_normalize = l2_normalize
    nn.tanh = torch.tanh          # bound as a DEFAULT ARGUMENT when code/MultiKE_model.py is imported (`conv(..., activation=tf.nn.tanh)`)
    tf.nn = nn
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.nn"] = nn
    return tf


def install_empty_standins():
    for name in ("gensim", "gensim.models", "gensim.models.word2vec", "Levenshtein"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["gensim.models.word2vec"].Word2Vec = object


def T(a, dtype):
    return torch.tensor(np.asarray(a), dtype=dtype)


# ------------------------------------------------------------------------------------------------
def make_case(rng, E, R, d, P, N):
    """Tables + one grouped batch with duplicates, a few arbitrary negatives and weights."""
    sigma_e = np.sqrt(2.6 / (E + d))
    ent = (rng.standard_normal((E, d)).clip(-2, 2) * sigma_e).astype(np.float32)
    rel = (rng.standard_normal((R, d)).clip(-2, 2) * np.sqrt(2.6 / (R + d))).astype(np.float32)
    ph = rng.integers(0, E, P)
    pr = rng.integers(0, R, P)
    pt = rng.integers(0, E, P)
    if P > 3:  # force duplicates: same entity as head, tail and in several positives
        ph[1] = ph[0]
        pt[2] = ph[0]
    nh = np.repeat(ph, N)
    nr = np.repeat(pr, N)
    nt = np.repeat(pt, N)
    side = rng.integers(0, 2, P * N).astype(bool)
    corrupt = rng.integers(0, E, P * N)
    nh = np.where(side, corrupt, nh)
    nt = np.where(side, nt, corrupt)
    # a few irregular negatives: identical to the positive, both sides changed, other relation
    if P * N > 6:
        nh[0], nt[0] = ph[0], pt[0]
        nh[3], nt[3] = rng.integers(0, E), rng.integers(0, E)
        nr[5] = (nr[5] + 1) % R
    pw = rng.uniform(0.2, 1.0, P).astype(np.float32)
    nw = rng.uniform(0.2, 1.0, P * N).astype(np.float32)
    return dict(ent=ent, rel=rel, ph=ph.astype(np.int32), pr=pr.astype(np.int32), pt=pt.astype(np.int32),
                nh=nh.astype(np.int32), nr=nr.astype(np.int32), nt=nt.astype(np.int32), pw=pw, nw=nw)


def losses_fixture(ref_losses, tf, out):
    cases = [(0, 40, 6, 4, 7, 1), (1, 40, 6, 4, 7, 10), (2, 40, 6, 4, 64, 10), (0, 80, 9, 75, 16, 10),
             (1, 80, 9, 75, 7, 1)]
    lr = 0.001
    for ci, (seed, E, R, d, P, N) in enumerate(cases):
        rng = np.random.default_rng(1000 + seed + 17 * ci)
        c = make_case(rng, E, R, d, P, N)
        pre = f"c{ci}_"
        for k, v in c.items():
            out[pre + k] = v
        out[pre + "meta"] = np.array([seed, E, R, d, P, N], dtype=np.int64)
        idx = {k: torch.tensor(c[k].astype(np.int64)) for k in ("ph", "pr", "pt", "nh", "nr", "nt")}
        M_np = (np.linalg.qr(rng.standard_normal((d, d)))[0] + 0.05 * rng.standard_normal((d, d))).astype(np.float32)
        out[pre + "a6_M"] = M_np
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            ent = T(c["ent"], dt).requires_grad_(True)
            rel = T(c["rel"], dt).requires_grad_(True)

            def gathered():
                En = tf.nn.l2_normalize(ent, 1)
                Rn = tf.nn.l2_normalize(rel, 1)
                return [En[idx["ph"]], Rn[idx["pr"]], En[idx["pt"]], En[idx["nh"]], Rn[idx["nr"]], En[idx["nt"]]]

            # a1 on gathered rows: loss, grads w.r.t. gathered rows and w.r.t. the raw tables
            rows = gathered()
            for r_ in rows:
                r_.retain_grad()
            loss = ref_losses.relation_logistic_loss(*rows)
            loss.backward()
            out[pre + "a1_loss_" + tag] = np.float64(loss.item())
            if tag == "f64":
                for name, r_ in zip(("gph", "gpr", "gpt", "gnh", "gnr", "gnt"), rows):
                    out[pre + "a1_" + name] = r_.grad.numpy()
                out[pre + "a1_gent_raw"] = ent.grad.numpy().copy()
                out[pre + "a1_grel_raw"] = rel.grad.numpy().copy()
            # three optimizer steps on the same batch (TF1 Adagrad: acc0 = 0.1, no epsilon)
            ent2 = T(c["ent"], dt).requires_grad_(True)
            rel2 = T(c["rel"], dt).requires_grad_(True)
            opt = torch.optim.Adagrad([ent2, rel2], lr=lr, initial_accumulator_value=0.1, eps=0.0)
            step_losses = []
            for step in range(3):
                opt.zero_grad()
                En = tf.nn.l2_normalize(ent2, 1)
                Rn = tf.nn.l2_normalize(rel2, 1)
                L = ref_losses.relation_logistic_loss(En[idx["ph"]], Rn[idx["pr"]], En[idx["pt"]], En[idx["nh"]],
                                                      Rn[idx["nr"]], En[idx["nt"]])
                L.backward()
                opt.step()
                step_losses.append(L.item())
                if step in (0, 2) and tag == "f64":
                    out[pre + f"a1_ent_after{step + 1}"] = ent2.detach().numpy().copy()
                    out[pre + f"a1_rel_after{step + 1}"] = rel2.detach().numpy().copy()
            out[pre + "a1_step_losses_" + tag] = np.array(step_losses)

            # the other losses.py functions on gathered rows (loss + grads w.r.t. inputs), f64 grads only
            with torch.no_grad():
                base = [x.detach().clone() for x in gathered()]
            pw, nw = T(c["pw"], dt), T(c["nw"], dt)

            def run(name, fn, args, store_grads=True):
                leaves = [a.clone().requires_grad_(True) if a.is_floating_point() and a.dim() == 2 else a for a in args]
                val = fn(*leaves)
                out[pre + name + "_loss_" + tag] = np.float64(val.item())
                if tag == "f64" and store_grads:
                    val.backward()
                    for k, a in enumerate(leaves):
                        if a.requires_grad:
                            out[pre + f"{name}_g{k}"] = a.grad.numpy()

            run("a2", ref_losses.relation_logistic_loss_wo_negs, base[:3])
            run("a2b", ref_losses.attribute_logistic_loss_wo_negs, base[:3], store_grads=False)
            run("a3", ref_losses.logistic_loss_wo_negs, base[:3] + [pw])
            run("a4", ref_losses.attribute_logistic_loss, base[:3] + [pw] + base[3:] + [nw])
            run("a5", ref_losses.alignment_loss, [base[0], base[2]])
            M = T(M_np, dt)
            eye = torch.eye(d, dtype=dt)
            run("a6", lambda v, s, m: ref_losses.space_mapping_loss(v, s, m, eye, 2.0), [base[0], base[2], M])
            run("a6o", lambda m: ref_losses.orthogonal_loss(m, eye), [M])


def toy_kgs(rng):
    n1, n2 = 50, 50
    ents1, ents2 = list(range(n1)), list(range(n1, n1 + n2))

    def triples(ents, rels, n):
        s = set()
        while len(s) < n:
            s.add((int(rng.choice(ents)), int(rng.choice(rels)), int(rng.choice(ents))))
        return sorted(s)

    t1 = triples(ents1, list(range(6)), 130)
    t2 = triples(ents2, list(range(6, 11)), 110)
    # "known" sets are supersets of the positives (the reference's alias includes swapped triples)
    k1 = set(t1) | set(triples(ents1, list(range(6)), 60))
    k2 = set(t2) | set(triples(ents2, list(range(6, 11)), 60))
    return ents1, ents2, t1, t2, k1, k2


def sampler_fixture(ref_batch, ref_attr_batch, out_json):
    rng = np.random.default_rng(77)
    ents1, ents2, t1, t2, k1, k2 = toy_kgs(rng)
    out_json["ents1"], out_json["ents2"] = ents1, ents2
    out_json["triples1"], out_json["triples2"] = t1, t2
    out_json["known1"], out_json["known2"] = sorted(k1), sorted(k2)
    # truncated-sampling neighbour dicts for some entities (code/base/batch.py:94-95 neighbor.get(x, all))
    near1 = {e: [int(x) for x in rng.choice(ents1, 12, replace=False)] for e in ents1[::3]}
    near2 = {e: [int(x) for x in rng.choice(ents2, 12, replace=False)] for e in ents2[::2]}
    out_json["near1"] = {str(k): v for k, v in near1.items()}
    out_json["near2"] = {str(k): v for k, v in near2.items()}
    runs = []
    for seed, bs, N, use_near in ((0, 20, 5, False), (1, 20, 5, True), (2, 48, 10, False), (3, 240, 3, True)):
        steps = int(np.ceil((len(t1) + len(t2)) / bs))
        random.seed(seed)
        np.random.seed(seed)
        per_step = []
        for step in range(steps + 1):  # one step past the end: empty slices
            pos, neg = ref_batch.generate_relation_triple_batch(t1, t2, k1, k2, ents1, ents2, bs, step,
                                                                near1 if use_near else None,
                                                                near2 if use_near else None, N)
            per_step.append({"pos": [list(x) for x in pos], "neg": [list(x) for x in neg]})
        runs.append({"seed": seed, "batch_size": bs, "neg": N, "use_near": use_near, "steps": per_step})
    out_json["relation_runs"] = runs
    # attribute batches (weights ride along; the live call passes neg_triples_num = 0 — MultiKE_model.py:331)
    a1 = [(h, r, t, round(0.2 + 0.01 * i, 4)) for i, (h, r, t) in enumerate(t1[:70])]
    a2 = [(h, r, t, round(0.5 + 0.005 * i, 4)) for i, (h, r, t) in enumerate(t2[:45])]
    out_json["attr1"], out_json["attr2"] = [list(x) for x in a1], [list(x) for x in a2]
    arun = []
    for step in range(5):
        pos, neg = ref_attr_batch.generate_attribute_triple_batch(a1, a2, set(a1), set(a2), ents1, ents2, 30, step,
                                                                  None, None, 0)
        arun.append({"pos": [list(x) for x in pos], "neg": [list(x) for x in neg]})
    out_json["attribute_run"] = {"batch_size": 30, "steps": arun}


def host_fixture(ref_utils, out_json):
    td = []
    for total, n in ((183, 4), (3, 4), (4, 4), (0, 4), (10, 3), (7, 0), (9, 2)):
        td.append({"total": total, "n": n, "tasks": [list(map(int, x)) for x in ref_utils.task_divide(list(range(total)), n)]})
    out_json["task_divide"] = td
    sp = []
    for n1, n2, B in ((460000, 450000, 5000), (130, 110, 20), (1, 999, 5000), (105000, 95000, 5000), (7, 7, 3)):
        b1 = int(n1 / (n1 + n2) * B)  # code/base/batch.py:36-37 evaluated here as the expected value
        sp.append({"n1": n1, "n2": n2, "batch": B, "b1": b1, "b2": B - b1})
    out_json["kg_batch_split"] = sp


def cnn_fixture(out):
    """The attribute-view CNN scorer (code/MultiKE_model.py:34-63), restated: an implementation of its TF1 semantics that shares
    nothing with the reference's code (rounds 1-5: the only pin; since round 6 `cnn_reference_fixture` below EXECUTES the
    reference's `conv` over forwarded leaf ops and asserts that the two agree).  Its TF1 semantics
    are restated here with torch.nn.functional ops — an implementation independent of oracle/attr_cnn_oracle.py —
    and differentiated by autograd in float64.  Pins the oracle's forward and hand-derived backward."""
    import torch.nn.functional as F
    for ci, (d, B, weighted, scale) in enumerate(((8, 13, False, 2.0), (75, 37, True, 1.0))):
        rng = np.random.default_rng(500 + ci)
        P = {"gamma": 1 + 0.2 * rng.standard_normal(d), "beta": 0.1 * rng.standard_normal(d),
             "K1": 0.5 * rng.standard_normal((2, 4, 1, 2)), "b1": 0.1 * rng.standard_normal(2),
             "K2": 0.5 * rng.standard_normal((2, 4, 2, 2)), "b2": 0.1 * rng.standard_normal(2),
             "W": rng.standard_normal((4 * d, d)) * np.sqrt(6.0 / (5 * d)), "bias": 0.1 * rng.standard_normal(d)}
        hs = rng.standard_normal((B, d)); hs /= np.linalg.norm(hs, axis=1, keepdims=True)
        as_ = 0.3 * rng.standard_normal((B, d))
        vs = rng.standard_normal((B, d)); vs /= np.linalg.norm(vs, axis=1, keepdims=True)
        ws = rng.uniform(0.2, 1.0, B) if weighted else None
        T_ = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
        th, ta = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (hs, as_))
        tv = torch.tensor(vs, dtype=torch.float64)
        x = torch.stack([ta, tv], 1)                                            # [B,2,d]
        x = x * (T_["gamma"] / np.sqrt(1.0 + 1e-3)) + T_["beta"]                 # BN inference, axis = width
        x = x[:, None]                                                          # NCHW [B,1,2,d]
        for K, b in ((T_["K1"], T_["b1"]), (T_["K2"], T_["b2"])):
            x = F.pad(x, (1, 2, 0, 1))                                          # SAME for a 2x4 kernel
            x = torch.tanh(F.conv2d(x, K.permute(3, 2, 0, 1), b))
        x = x.permute(0, 2, 3, 1)                                               # NHWC [B,2,d,2]
        x = x * torch.rsqrt(torch.clamp_min((x * x).sum(2, keepdim=True), 1e-12))   # l2_normalize(axis=2)
        flat = x.reshape(B, -1)
        z = torch.tanh(flat @ T_["W"] + T_["bias"])
        o = z * torch.rsqrt(torch.clamp_min((z * z).sum(), 1e-12))              # l2_normalize, no axis
        score = -((th - o) ** 2).sum(1)
        per = torch.log(1 + torch.exp(-score))
        loss = scale * ((per * torch.tensor(ws)) if ws is not None else per).sum()
        loss.backward()
        pre = f"n{ci}_"
        out[pre + "meta"] = np.array([d, B, int(weighted)], dtype=np.int64)
        out[pre + "scale"] = np.float64(scale)
        for k, v in P.items():
            out[pre + "p_" + k] = v
            out[pre + "g_" + k] = T_[k].grad.numpy()
        out[pre + "hs"], out[pre + "as"], out[pre + "vs"] = hs, as_, vs
        if ws is not None:
            out[pre + "ws"] = ws
        out[pre + "score"] = score.detach().numpy()
        out[pre + "loss"] = np.float64(loss.item())
        out[pre + "g_hs"], out[pre + "g_as"] = th.grad.numpy(), ta.grad.numpy()


def _eager_tf(queue):
    """TensorFlow's graph-BUILDING calls as eager torch calls (a placeholder is its fed value, a variable the tensor handed over
    under its scoped name): what lets `MultiKE._define_*_graph` (code/MultiKE_model.py) run unmodified.  Returns the pieces a
    fixture needs.  Leaf ops restated here: tf.layers batch-norm (inference) / conv2d / dense, embedding_lookup, constant."""
    import contextlib
    import math
    from unittest import mock
    import torch.nn.functional as F
    tf = sys.modules["tensorflow"]
    tf.__getattr__ = lambda name: mock.MagicMock(name="tf." + name)
    tf.nn.__getattr__ = lambda name: mock.MagicMock(name="tf.nn." + name)

    class _Shape(tuple):
        def as_list(self):
            return list(self)

    class _T(torch.Tensor):                       # a tensor whose .shape has TensorShape's as_list()
        @property
        def shape(self):
            return _Shape(torch.Tensor.shape.__get__(self))

    wrap = lambda t: t.as_subclass(_T)
    layers = types.ModuleType("tensorflow.layers")

    def batch_normalization(inputs, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, training=False, **kw):
        assert not training and center and scale
        p = queue["params"].pop(0)
        shp = [1] * inputs.dim()
        shp[axis] = -1
        return wrap(inputs * (p["gamma"].reshape(shp) / math.sqrt(1.0 + epsilon)) + p["beta"].reshape(shp))

    def conv2d(inputs, filters, kernel_size, strides, padding, activation=None):
        p = queue["params"].pop(0)
        kh, kw = kernel_size
        assert padding == "same" and list(strides) == [1, 1] and tuple(p["K"].shape) == (kh, kw, inputs.shape[3], filters)
        x = inputs.permute(0, 3, 1, 2)                                          # NHWC -> NCHW
        x = F.pad(x, ((kw - 1) // 2, kw - 1 - (kw - 1) // 2, (kh - 1) // 2, kh - 1 - (kh - 1) // 2))
        y = F.conv2d(x, p["K"].permute(3, 2, 0, 1), p["b"]).permute(0, 2, 3, 1)
        return wrap(activation(y) if activation is not None else y)

    def dense(inputs, units, activation=None):
        p = queue["params"].pop(0)
        assert p["W"].shape[1] == units
        y = inputs @ p["W"] + p["bias"]
        return wrap(activation(y) if activation is not None else y)

    layers.batch_normalization, layers.conv2d, layers.dense = batch_normalization, conv2d, dense
    tf.layers = layers
    scope = []

    @contextlib.contextmanager
    def variable_scope(name, *a, **k):
        scope.append(name)
        try:
            yield
        finally:
            scope.pop()

    def get_variable(name, shape=None, dtype=None, initializer=None):
        v = queue["vars"][name]
        assert shape is None or list(v.shape) == list(shape), (name, shape, v.shape)
        # what tf.trainable_variables() lists (code/MultiKE_model.py:257 filters on the scoped name)
        queue["trainable"].append(types.SimpleNamespace(name="/".join(scope + [name]) + ":0", key=name, tensor=v))
        return v

    tf.reshape = lambda x, shape: wrap(torch.reshape(x, tuple(shape)))
    tf.concat = lambda xs, axis: wrap(torch.cat(list(xs), dim=axis))
    tf.placeholder = lambda dtype, shape=None: queue["feeds"].pop(0)
    tf.constant = lambda value, dtype=None, name=None: torch.tensor(np.asarray(value), dtype=torch.float64)
    tf.get_variable = get_variable
    tf.trainable_variables = lambda: list(queue["trainable"])
    tf.variable_scope = variable_scope
    tf.name_scope = lambda *a, **k: contextlib.nullcontext()      # names ops, not variables
    tf.reduce_mean = lambda x, axis=None: torch.mean(x) if axis is None else torch.mean(x, dim=axis)
    tf.nn.embedding_lookup = lambda table, ids: wrap(table[ids])
    tf.nn.sigmoid = torch.sigmoid
    return tf


def cnn_reference_fixture(out):
    """The attribute-view graph of the reference EXECUTED: `MultiKE._define_attribute_view_graph` (code/MultiKE_model.py:134-151)
    on an instance made with `object.__new__`, with its tables built by the reference's own `xavier_init(..., is_l2_norm)`
    (code/base/initializers.py:23-27: the normalised view of the entity table, the raw attribute table) and its scorer by the
    reference's own `conv` (code/MultiKE_model.py:34-63).  TensorFlow's graph-building calls are forwarded to torch EAGERLY — a
    placeholder is its fed value — so every line of the reference's composition runs: the reshapes and the concat, which axis
    batch-norm and the two l2_normalize calls work on, the NHWC flatten order into the dense layer, the embedding lookups, the
    loss.  What is restated here (and only here) are the LEAF ops: tf.layers.batch_normalization in inference mode with its
    never-updated moving statistics (0, 1), tf.layers.conv2d (HWIO kernel, SAME padding of an even kernel: the extra column on
    the right / row at the bottom), tf.layers.dense, embedding_lookup — "unpinned at the TF boundary" still applies to those.
    Gradients: torch autograd through the executed graph, float64.  Same inputs as cnn_fixture, plus a table with repeated rows."""
    from unittest import mock
    queue = {"params": [], "feeds": [], "vars": {}, "trainable": []}
    _eager_tf(queue)
    ref_model = importlib.import_module("MultiKE_model")
    ref_init = importlib.import_module("base.initializers")
    assert ref_model.conv.__defaults__[2] is torch.tanh           # the default activation was bound to the forwarder's tanh
    for ci, (d, B, weighted, scale) in enumerate(((8, 13, False, 2.0), (75, 37, True, 1.0))):
        rng = np.random.default_rng(500 + ci)                     # the draws of cnn_fixture, in its order
        P = {"gamma": 1 + 0.2 * rng.standard_normal(d), "beta": 0.1 * rng.standard_normal(d),
             "K1": 0.5 * rng.standard_normal((2, 4, 1, 2)), "b1": 0.1 * rng.standard_normal(2),
             "K2": 0.5 * rng.standard_normal((2, 4, 2, 2)), "b2": 0.1 * rng.standard_normal(2),
             "W": rng.standard_normal((4 * d, d)) * np.sqrt(6.0 / (5 * d)), "bias": 0.1 * rng.standard_normal(d)}
        hs = rng.standard_normal((B, d)); hs /= np.linalg.norm(hs, axis=1, keepdims=True)
        as_ = 0.3 * rng.standard_normal((B, d))
        vs = rng.standard_normal((B, d)); vs /= np.linalg.norm(vs, axis=1, keepdims=True)
        ws = rng.uniform(0.2, 1.0, B) if weighted else None
        t64 = lambda a, g=True: torch.tensor(a, dtype=torch.float64, requires_grad=g)
        pre = f"n{ci}_ref_"

        def run(ent_raw, attr_raw, lit, ih, ia, iv, w):
            T_ = {k: t64(v) for k, v in P.items()}
            te, ta = t64(ent_raw), t64(attr_raw)
            queue["params"] = [{"gamma": T_["gamma"], "beta": T_["beta"]}, {"K": T_["K1"], "b": T_["b1"]}, {"K": T_["K2"], "b": T_["b2"]},
                               {"W": T_["W"], "bias": T_["bias"]}]
            queue["vars"] = {"av_ent_embeds": te, "attr_embeds": ta}
            queue["feeds"] = [torch.as_tensor(ih), torch.as_tensor(ia), torch.as_tensor(iv), torch.tensor(w, dtype=torch.float64)]
            m = object.__new__(ref_model.MultiKE)
            m.args = argparse.Namespace(dim=d, learning_rate=0.01, optimizer="Adagrad")
            # the reference's own initialiser wrappers: the normalised VIEW of the entity table, the raw attribute table ("False important!")
            m.av_ent_embeds = ref_init.xavier_init([ent_raw.shape[0], d], "av_ent_embeds", True)
            m.attr_embeds = ref_init.xavier_init([attr_raw.shape[0], d], "attr_embeds", False)
            m.literal_embeds = torch.tensor(lit, dtype=torch.float64)
            score_box = []
            real_conv = ref_model.conv

            def spy(*a, **k):
                score_box.append(real_conv(*a, **k))
                return score_box[-1]
            with mock.patch.object(ref_model, "conv", spy), mock.patch.object(ref_model, "generate_optimizer", lambda *a, **k: None):
                m._define_attribute_view_graph()
            assert not queue["params"] and not queue["feeds"]
            m.attribute_loss.backward()
            return m.attribute_loss.item(), score_box[0].detach().numpy(), te.grad.numpy(), ta.grad.numpy(), {k: v.grad.numpy() for k, v in T_.items()}

        # (1) cnn_fixture's case: row i of every table is triple i's row; the entity rows are unit vectors (their normalised view is
        # themselves up to rounding).  Score and loss must equal the independent restatement's
        w1 = ws if ws is not None else np.ones(B)
        loss, score, _, g_as, gp = run(hs, as_, vs, np.arange(B), np.arange(B), np.arange(B), w1)
        np.testing.assert_allclose(score, out[f"n{ci}_score"], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(loss * scale, out[f"n{ci}_loss"], rtol=1e-12)
        np.testing.assert_allclose(g_as * scale, out[f"n{ci}_g_as"], rtol=1e-8, atol=1e-12)
        for k in P:
            np.testing.assert_allclose(gp[k] * scale, out[f"n{ci}_g_{k}"], rtol=1e-8, atol=1e-12, err_msg=k)
        out[pre + "score"], out[pre + "loss"], out[pre + "g_as"] = score, np.float64(loss), g_as
        for k in P:
            out[pre + "g_" + k] = gp[k]
        # (2) tables with repeated and unused rows, entity rows of any length: the lookups' scatter and the normalised view's Jacobian
        n_ent, n_attr, n_lit = B // 2 + 3, 5, B // 3 + 2
        ent_raw = rng.standard_normal((n_ent, d)) * rng.uniform(0.3, 3.0, (n_ent, 1))
        attr_raw = 0.3 * rng.standard_normal((n_attr, d))
        lit = rng.standard_normal((n_lit, d)); lit /= np.linalg.norm(lit, axis=1, keepdims=True)
        ih, ia, iv = rng.integers(0, n_ent - 2, B), rng.integers(0, n_attr, B), rng.integers(0, n_lit, B)
        loss, score, g_ent, g_attr, gp = run(ent_raw, attr_raw, lit, ih, ia, iv, w1)
        out[pre + "t_ent"], out[pre + "t_attr"], out[pre + "t_lit"] = ent_raw, attr_raw, lit
        out[pre + "t_ih"], out[pre + "t_ia"], out[pre + "t_iv"], out[pre + "t_w"] = ih, ia, iv, w1
        out[pre + "t_score"], out[pre + "t_loss"], out[pre + "t_g_ent"], out[pre + "t_g_attr"] = score, np.float64(loss), g_ent, g_attr
        for k in P:
            out[pre + "t_g_" + k] = gp[k]


def graphs_fixture(out, seed=77, sizes=(12, 40, 5, 6, 15, 17, 9), neg=3, rates=(0.01, 0.03, 0.7, 1.5, 2.0)):
    """EVERY graph of the reference's model EXECUTED once (eagerly: `_eager_tf`): `MultiKE._define_variables` and the nine
    `_define_*_graph` methods (code/MultiKE_model.py:86-261) run unmodified on an instance made with `object.__new__`, in the order
    the drivers call them (code/MultiKE_CSL.py:21-31, MultiKE_Late.py:184-196), with `generate_optimizer` replaced by a recorder
    of (loss, learning rate, var_list).  For every graph: the loss attribute the training loop prints, the loss the optimizer
    minimises, its learning rate, which variables it may move, and the gradient of the minimised loss w.r.t. EVERY raw variable
    (float64 autograd through the executed graph: lookups, normalised views, losses.py, conv).  What this pins is the reference's
    composition — which tables a graph reads, through which view, with which factor, rate and variable list; the leaf ops of
    tf.layers and ApplyAdagrad stay restated (oracle/)."""
    from unittest import mock
    queue = {"params": [], "feeds": [], "vars": {}, "trainable": []}
    _eager_tf(queue)
    ref_model = importlib.import_module("MultiKE_model")
    rng = np.random.default_rng(seed)                   # (tools/fuzz_graphs_pin.py calls this with random sizes)
    d, n_ent, n_rel, n_attr, n_lit, B, EB = sizes
    raw = {"rv_ent_embeds": rng.standard_normal((n_ent, d)) * rng.uniform(0.4, 2.5, (n_ent, 1)), "rel_embeds": rng.standard_normal((n_rel, d)),
           "av_ent_embeds": rng.standard_normal((n_ent, d)) * rng.uniform(0.4, 2.5, (n_ent, 1)), "attr_embeds": 0.3 * rng.standard_normal((n_attr, d)),
           "ent_embeds": rng.standard_normal((n_ent, d)) * rng.uniform(0.4, 2.5, (n_ent, 1)),
           "nv_mapping": np.linalg.qr(rng.standard_normal((d, d)))[0] + 0.05 * rng.standard_normal((d, d)),
           "rv_mapping": np.linalg.qr(rng.standard_normal((d, d)))[0] + 0.05 * rng.standard_normal((d, d)),
           "av_mapping": np.linalg.qr(rng.standard_normal((d, d)))[0] + 0.05 * rng.standard_normal((d, d))}
    lit = rng.standard_normal((n_lit, d)); lit /= np.linalg.norm(lit, axis=1, keepdims=True)
    name = rng.standard_normal((n_ent, d)); name /= np.linalg.norm(name, axis=1, keepdims=True)
    cnn = [{"gamma": 1 + 0.2 * rng.standard_normal(d), "beta": 0.1 * rng.standard_normal(d),
            "K1": 0.5 * rng.standard_normal((2, 4, 1, 2)), "b1": 0.1 * rng.standard_normal(2),
            "K2": 0.5 * rng.standard_normal((2, 4, 2, 2)), "b2": 0.1 * rng.standard_normal(2),
            "W": rng.standard_normal((4 * d, d)) * np.sqrt(6.0 / (5 * d)), "bias": 0.1 * rng.standard_normal(d)} for _ in range(3)]
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    V = {k: t64(v) for k, v in raw.items()}
    C_ = [{k: t64(v) for k, v in p.items()} for p in cnn]
    queue["vars"] = V
    for p in C_:      # one conv() call per attribute-type graph, in definition order: batch-norm, conv, conv, dense
        queue["params"] += [{"gamma": p["gamma"], "beta": p["beta"]}, {"K": p["K1"], "b": p["b1"]}, {"K": p["K2"], "b": p["b2"]}, {"W": p["W"], "bias": p["bias"]}]
    ids = lambda hi, n=B: rng.integers(0, hi, n)
    N = neg
    feeds = {                                  # placeholders in the order each method creates them
        "relation": [ids(n_ent), ids(n_rel), ids(n_ent), ids(n_ent, B * N), ids(n_rel, B * N), ids(n_ent, B * N)],
        "attribute": [ids(n_ent), ids(n_attr), ids(n_lit), rng.uniform(0.2, 1.0, B)],
        "ckge_rel": [ids(n_ent), ids(n_rel), ids(n_ent)],
        "ckge_attr": [ids(n_ent), ids(n_attr), ids(n_lit)],
        "ckga_attr": [ids(n_ent), ids(n_attr), ids(n_lit), rng.uniform(0.2, 1.0, B)],
        "ckgp_rel": [ids(n_ent), ids(n_rel), ids(n_ent), rng.uniform(0.2, 1.0, B)],
        "common": [ids(n_ent)],
        "mapping": [rng.permutation(n_ent)[:EB]],
    }
    m = object.__new__(ref_model.MultiKE)
    m.args = argparse.Namespace(dim=d, learning_rate=rates[0], ITC_learning_rate=rates[1], optimizer="Adagrad", cv_name_weight=rates[2],
                                cv_weight=rates[3], orthogonal_weight=rates[4], entity_batch_size=EB)
    m.data = types.SimpleNamespace(value_vectors=lit, local_name_vectors=name)
    m.kgs = types.SimpleNamespace(entities_num=n_ent, relations_num=n_rel, attributes_num=n_attr)
    recorded = []
    order = [("name_view", "_define_name_view_graph", None), ("relation", "_define_relation_view_graph", "relation_loss"),
             ("attribute", "_define_attribute_view_graph", "attribute_loss"),
             ("ckge_rel", "_define_cross_kg_entity_reference_relation_view_graph", "ckge_relation_loss"),
             ("ckge_attr", "_define_cross_kg_entity_reference_attribute_view_graph", "ckge_attribute_loss"),
             ("ckga_attr", "_define_cross_kg_attribute_reference_graph", "ckga_attribute_loss"),
             ("ckgp_rel", "_define_cross_kg_relation_reference_graph", "ckgp_relation_loss"),
             ("common", "_define_common_space_learning_graph", "cross_name_loss"), ("mapping", "_define_space_mapping_graph", "shared_comb_loss")]
    with mock.patch.object(ref_model, "generate_optimizer",
                           lambda loss, learning_rate, var_list=None, opt="SGD": recorded.append((loss, learning_rate, var_list, opt))):
        m._define_variables()
        for key, meth, attr in order:
            n0 = len(recorded)
            queue["feeds"] = [torch.as_tensor(x) if np.asarray(x).dtype.kind == "i" else torch.tensor(x, dtype=torch.float64) for x in feeds.get(key, [])]
            getattr(m, meth)()
            assert not queue["feeds"], key
            if attr is None:
                assert len(recorded) == n0
                continue
            assert len(recorded) == n0 + 1
            minimised, lr, var_list, opt = recorded[-1]
            assert opt == "Adagrad"
            leaves = {**V, **{f"cnn{k}_{n}": t for k, p in enumerate(C_) for n, t in p.items()}}
            grads = torch.autograd.grad(minimised, list(leaves.values()), allow_unused=True, retain_graph=True)
            out[f"{key}_loss"] = np.float64(getattr(m, attr).item())
            out[f"{key}_minimised"] = np.float64(minimised.item())
            out[f"{key}_lr"] = np.float64(lr)
            out[f"{key}_var_list"] = np.array([] if var_list is None else sorted(v.key for v in var_list))      # empty: every variable the loss depends on
            out[f"{key}_has_var_list"] = np.int64(var_list is not None)
            for (n, _), g in zip(leaves.items(), grads):
                if g is not None and bool((g != 0).any()):
                    out[f"{key}_g_{n}"] = g.numpy()
            for i, x in enumerate(feeds[key]):
                out[f"{key}_feed{i}"] = np.asarray(x)
    assert not queue["params"]
    out["trainable_names"] = np.array(sorted(v.name for v in queue["trainable"]))
    for k, v in raw.items():
        out["raw_" + k] = v
    out["lit"], out["name"] = lit, name
    for k, p in enumerate(cnn):
        for n, v in p.items():
            out[f"cnn{k}_{n}"] = v
    out["args"] = np.array(rates, dtype=np.float64)          # learning_rate, ITC_learning_rate, cv_name_weight, cv_weight, orthogonal_weight
    # the views the evaluation reads (code/MultiKE_model.py:263-277: embedding_lookup of the NORMALISED relation-view table)
    out["view_rv_ent"] = m.rv_ent_embeds.detach().numpy()
    out["view_attr"] = m.attr_embeds.detach().numpy()


def ae_graph_fixture(out):
    """The literal auto-encoder's graph EXECUTED: `AutoEncoderModel._init_graph` and `_loss_optimizer` with its `encoder` /
    `decoder` (code/literal_encoder.py:41-91) on an instance made with `object.__new__`, eagerly (`_eager_tf`).  Every op of
    this graph forwards exactly (matmul, add, sigmoid / tanh, l2_normalize without an axis, pow, reduce_mean): nothing is restated
    but the optimizer.  Loss and float64 autograd gradients of every weight and bias, for the shipped activation string (matches
    neither branch: a linear model), tanh, sigmoid, with and without the batch-wide normalisation."""
    from unittest import mock
    queue = {"params": [], "feeds": [], "vars": {}, "trainable": []}
    _eager_tf(queue)
    ref_le = importlib.import_module("literal_encoder")
    dims = [30, 16, 8, 5]
    for ci, (active, normalize) in enumerate((("thah", True), ("tanh", True), ("sigmoid", False), ("tanh", False))):
        rng = np.random.default_rng(900 + ci)
        n = len(dims) - 1
        p = {}
        for i in range(n):
            p[f"encoder_h{i}"] = 0.3 * rng.standard_normal((dims[i], dims[i + 1]))
            p[f"encoder_b{i}"] = 0.3 * rng.standard_normal(dims[i + 1])
        for i in range(n):
            j = n - i
            p[f"decoder_h{i}"] = 0.3 * rng.standard_normal((dims[j], dims[j - 1]))
            p[f"decoder_b{i}"] = 0.3 * rng.standard_normal(dims[j - 1])
        x = rng.standard_normal((11, dims[0]))
        T_ = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
        queue["vars"], queue["trainable"] = T_, []
        queue["feeds"] = [torch.tensor(x, dtype=torch.float64)] * n          # _init_graph re-creates the placeholder inside its decoder loop
        m = object.__new__(ref_le.AutoEncoderModel)
        m.args = argparse.Namespace(dim=dims[-1], encoder_normalize=normalize, encoder_active=active, learning_rate=0.01, optimizer="Adagrad")
        m.weights, m.biases = {}, {}
        m.input_dimension, m.hidden_dimensions = dims[0], list(dims[1:])
        m.layer_num = n
        rec = []
        with mock.patch.object(ref_le, "generate_optimizer", lambda loss, lr, var_list=None, opt="SGD": rec.append((loss, lr, var_list, opt))):
            m._init_graph()
            m._loss_optimizer()
        assert not queue["feeds"] and len(rec) == 1 and rec[0][0] is m.loss and rec[0][2] is None
        assert sorted(m.weights) == sorted(k for k in p if "_h" in k) and sorted(m.biases) == sorted(k for k in p if "_b" in k)
        m.loss.backward()
        pre = f"ae{ci}_"
        out[pre + "meta"] = np.array([int(normalize)] + dims)
        out[pre + "active"] = np.array(active)
        out[pre + "x"], out[pre + "loss"] = x, np.float64(m.loss.item())
        for k, v in p.items():
            out[pre + "p_" + k], out[pre + "g_" + k] = v, T_[k].grad.numpy()


def eval_fixture(ref_alignment, out):
    """The reference's own greedy_alignment (code/base/alignment.py:8-79), single worker, on random embeddings."""
    import contextlib, io
    for ci, (n1, n2, d) in enumerate(((200, 300, 75), (64, 64, 20))):
        rng = np.random.default_rng(900 + ci)
        e2 = rng.standard_normal((n2, d)).astype(np.float32)
        e1 = (0.6 * e2[:n1] + rng.standard_normal((n1, d))).astype(np.float32)   # gold = same index, noisy
        top_k = [1, 5, 10, 50]
        with contextlib.redirect_stdout(io.StringIO()):
            rest_a, h1_a, mr_a, mrr_a = ref_alignment.greedy_alignment(e1, e2, top_k, 1, "inner", True, 0, True)
            mr_q, mrr_q, hits_q, _ = ref_alignment.calculate_rank(list(range(n1)), ref_alignment.sim(e1, e2, normalize=True),
                                                                   top_k, False, n1)
            mr_x, mrr_x, hits_x, _ = ref_alignment.calculate_rank(list(range(n1)), ref_alignment.sim(e1, e2, normalize=True),
                                                                   top_k, True, n1)
        pre = f"e{ci}_"
        out[pre + "e1"], out[pre + "e2"] = e1, e2
        out[pre + "top_k"] = np.array(top_k)
        out[pre + "hits_accurate"] = np.round(np.array(hits_x) / n1 * 100, 3)
        out[pre + "hits_quick"] = np.round(np.array(hits_q) / n1 * 100, 3)
        out[pre + "hits1"] = np.float64(h1_a)
        out[pre + "mr"], out[pre + "mrr"] = np.float64(mr_a), np.float64(mrr_a)
        out[pre + "rest"] = np.array(sorted(rest_a), dtype=np.int64)


def _py_ratio(a, b):
    """Independent statement of python-Levenshtein's `ratio` (edit distance with substitution cost 2, normalised by
    the summed lengths) -- plain O(len^2) DP, used only as the stand-in the reference module calls here."""
    la, lb = len(a), len(b)
    if la + lb == 0:
        return 1.0
    prev = list(range(lb + 1))
    for i in range(1, la + 1):
        cur = [i] + [0] * lb
        for j in range(1, lb + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (0 if a[i - 1] == b[j - 1] else 2))
        prev = cur
    return (la + lb - prev[lb]) / (la + lb)


def data_fixture(ref_kgs, ref_utils, ref_pa, out_json):
    """Runs the reference's readers / id assignment / KG containers / literal clean-up / predicate alignment on the
    folder `multike_amd.synthetic.write_dataset_folder` writes (deterministic) and stores what they return."""
    import contextlib
    import io
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from multike_amd.synthetic import write_dataset_folder
    folder = tempfile.mkdtemp() + "/"
    write_dataset_folder(folder)
    sys.modules["Levenshtein"].ratio = _py_ratio
    res = {"writer": {"seed": 11, "n_pairs": 60}}
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        for mode in ("swapping", "mapping", "sharing"):
            k = ref_kgs.read_kgs_from_folder(folder, "631/", mode, True)
            e = {"ent_ids1": k.kg1.entities_id_dict, "ent_ids2": k.kg2.entities_id_dict,
                 "rel_ids1": k.kg1.relations_id_dict, "rel_ids2": k.kg2.relations_id_dict,
                 "attr_ids1": k.kg1.attributes_id_dict, "attr_ids2": k.kg2.attributes_id_dict,
                 "train_links": k.train_links, "valid_links": k.valid_links, "test_links": k.test_links,
                 "entities_num": k.entities_num, "relations_num": k.relations_num, "attributes_num": k.attributes_num}
            for i, kg in ((1, k.kg1), (2, k.kg2)):
                e[f"local_rel{i}"] = sorted(kg.local_relation_triples_list)
                e[f"local_set_size{i}"] = len(kg.local_relation_triples_set)
                e[f"rel_num{i}"] = [kg.relation_triples_num, kg.local_relation_triples_num,
                                    kg.attribute_triples_num, kg.local_attribute_triples_num]
                e[f"sup_rel{i}"] = sorted(kg.sup_relation_triples_list or [])
                e[f"sup_attr{i}"] = sorted(kg.sup_attribute_triples_list or [])
                e[f"local_attr{i}"] = sorted(kg.local_attribute_triples_list)
            res[mode] = e
        # unordered: only layout facts are reproducible (hash order)
        k0 = ref_kgs.read_kgs_from_folder(folder, "631/", "swapping", False)
        res["unordered"] = {"n1": k0.kg1.entities_num, "n2": k0.kg2.entities_num,
                            "ids1_range": [min(k0.kg1.entities_id_dict.values()), max(k0.kg1.entities_id_dict.values())],
                            "ids2_range": [min(k0.kg2.entities_id_dict.values()), max(k0.kg2.entities_id_dict.values())],
                            "sup_rel": [len(k0.kg1.sup_relation_triples_list), len(k0.kg2.sup_relation_triples_list)]}
        k = ref_kgs.read_kgs_from_folder(folder, "631/", "swapping", True)
        cl = {}
        for i, kg in ((1, k.kg1), (2, k.kg2)):
            t, num, st = ref_utils.clear_attribute_triples(kg.local_attribute_triples_list)
            cl[f"triples{i}"] = sorted(t)
            cl[f"numbers{i}"] = sorted(num)
            cl[f"strings{i}"] = sorted(st)
        res["clear_attribute_triples"] = cl
        res["local_names"] = ref_utils.read_local_name(folder, set(k.kg1.entities_id_dict), set(k.kg2.entities_id_dict))
        res["is_number"] = {s: bool(ref_utils.is_number(s)) for s in ("12", "1e5", "-3.5", "abc", "\u00bd", "", "12a", "nan")}
        w2v = ref_utils.read_word2vec(folder + "wiki-news-300d-tiny.vec")
        res["word2vec"] = {"n": len(w2v), "amber_head": [float(x) for x in w2v["amber"][:4]]}
        args = types.SimpleNamespace(training_data=folder, predicate_init_sim=0.9, predicate_soft_sim=0.85)
        pam = ref_pa.PredicateAlignModel(k, args)
        emb_rng = np.random.default_rng(5)

        def snap(p):
            return {"relation_alignment_set": sorted(p.relation_alignment_set),
                    "attribute_alignment_set": sorted(p.attribute_alignment_set),
                    "relation_latent": sorted([a, b, s] for (a, b), s in p.relation_latent_match_pairs_similarity_dict_init.items()),
                    "attribute_latent": sorted([a, b, s] for (a, b), s in p.attribute_latent_match_pairs_similarity_dict_init.items()),
                    "sup_rel1": sorted(p.sup_relation_alignment_triples1), "sup_rel2": sorted(p.sup_relation_alignment_triples2),
                    "sup_attr1": sorted(p.sup_attribute_alignment_triples1), "sup_attr2": sorted(p.sup_attribute_alignment_triples2),
                    "rel_w1": sorted(p.relation_triples_w_weights1), "attr_w2": sorted(p.attribute_triples_w_weights2),
                    "train_relations1": sorted(p.train_relations1), "train_attributes2": sorted(p.train_attributes2)}
        pa = {"init": snap(pam)}
        rel_embed = emb_rng.standard_normal((k.relations_num, 8))
        attr_embed = emb_rng.standard_normal((k.attributes_num, 8))
        # make matched predicates close so that the refresh keeps some and drops others
        for (p1, p2, _) in sorted(pam.relation_alignment_set_init)[:3]:
            rel_embed[k.kg2.relations_id_dict[p2]] = rel_embed[k.kg1.relations_id_dict[p1]] + 0.05
        for (p1, p2, _) in sorted(pam.attribute_alignment_set_init)[:2]:
            attr_embed[k.kg2.attributes_id_dict[p2]] = attr_embed[k.kg1.attributes_id_dict[p1]] + 0.05
        pam.update_predicate_alignment(rel_embed)
        pam.update_predicate_alignment(attr_embed, predicate_type="attribute")
        pa["rel_embed"] = rel_embed.tolist()
        pa["attr_embed"] = attr_embed.tolist()
        pa["refreshed"] = snap(pam)
        pa["ratios"] = [[a, b, ref_pa.Levenshtein.ratio(a, b)] for a, b in
                        (("Hello world!", "Holly grail!"), ("kitten", "sitting"), ("", ""), ("abc", ""), ("amberOf", "amberOf"))]
        res["predicate_alignment"] = pa
    out_json.update(res)


def schedule_fixture(out_json):
    """The reference's epoch schedules, EXECUTED: `MultiKE_CV.run` (code/MultiKE_CSL.py:36-107) and `MultiKE_Late.run`
    (code/MultiKE_Late.py:201-280) are imported unmodified and run on a recording stand-in model (tests/schedule_mock.py:
    instances are created without __init__, every train / valid / test / save call appends to a trace).  TensorFlow is only
    touched at import time (default arguments such as `dtype=tf.float32`), so the forwarder module answers unknown
    attributes with a placeholder."""
    import contextlib
    import io
    from unittest import mock
    sys.path.insert(0, os.path.dirname(HERE))
    import schedule_mock as sm
    tf = sys.modules["tensorflow"]
    tf.__getattr__ = lambda name: mock.MagicMock(name="tf." + name)
    tf.nn.__getattr__ = lambda name: mock.MagicMock(name="tf.nn." + name)
    ref_csl = importlib.import_module("MultiKE_CSL")
    ref_late = importlib.import_module("MultiKE_Late")
    ref_bat = importlib.import_module("base.batch")

    class _Manager:                       # mp.Manager().Queue(): the queues are only passed through to the (mocked) loops
        def Queue(self):
            return None

    out_json["schedules"] = {}
    for method, mod, cls in (("ITC", ref_csl, ref_csl.MultiKE_CV), ("SSL", ref_late, ref_late.MultiKE_Late)):
        for name in sm.SCENARIOS:
            trace = []
            model = object.__new__(cls)
            fns = sm.instrument(model, name, trace)
            patches = [mock.patch.object(mod, "valid", fns["valid"]), mock.patch.object(mod, "test", fns["test"]),
                       mock.patch.object(ref_bat, "generate_neighbours", fns["neighbours"]),
                       mock.patch.object(mod.mp, "Manager", _Manager)]
            if mod is ref_late:
                patches += [mock.patch.object(mod, "valid_WVA", fns["valid_WVA"]), mock.patch.object(mod, "test_WVA", fns["test_WVA"])]
            with contextlib.ExitStack() as st, contextlib.redirect_stdout(io.StringIO()):
                for p_ in patches:
                    st.enter_context(p_)
                model.run()
            out_json["schedules"][f"{method}/{name}"] = trace


def pins_fixture(ref_batch, out):
    """Three importable pieces of the reference EXECUTED on small inputs (round-5 review, item 5):
      * `base.batch.generate_neighbours` / `find_neighbours` (code/base/batch.py:119-150) on a 700 x 20 matrix, k = 15 — the
        matrix is drawn so that every row's k-th and (k+1)-th similarities are at least 2e-5 apart (ten times the float32 error of a 20-term inner product of unit vectors) (no ties at the boundary:
        the expected neighbour SETS are exact whatever the arithmetic's last bits);
      * `MultiKE_Late.wva` / `_compute_weight` (code/MultiKE_Late.py:64-88) on three random views;
      * `AutoEncoderModel.encoder_multi_batches` (code/literal_encoder.py:114-144) on an instance made with `object.__new__`
        and NumPy weights injected behind `.eval(session=...)` (the method itself is NumPy), sigmoid and tanh, a row count that
        is and one that is not a multiple of the batch size."""
    import contextlib
    import io
    from unittest import mock
    # ---- k-NN refresh ------------------------------------------------------------------------------------------------
    n, d, k = 700, 20, 15
    seed = 0
    while True:
        rng = np.random.default_rng(4000 + seed)
        e = rng.standard_normal((n, d)).astype(np.float32)
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        sim = np.sort(e.astype(np.float64) @ e.astype(np.float64).T, axis=1)[:, ::-1]
        gap = sim[:, k - 1] - sim[:, k]
        bad = np.nonzero(gap <= 2e-5)[0]
        tries = 0
        while len(bad) and tries < 200:          # re-draw the rows whose boundary is too tight (a few of 700), re-check all
            e[bad] = rng.standard_normal((len(bad), d)).astype(np.float32)
            e[bad] /= np.linalg.norm(e[bad], axis=1, keepdims=True)
            sim = np.sort(e.astype(np.float64) @ e.astype(np.float64).T, axis=1)[:, ::-1]
            gap = sim[:, k - 1] - sim[:, k]
            bad = np.nonzero(gap <= 2e-5)[0]
            tries += 1
        if not len(bad):
            break
        seed += 1
        if seed > 20:
            raise SystemExit("no tie-free matrix found")
    ids = (np.arange(n) * 3 + 5).tolist()
    dic = ref_batch.generate_neighbours(e, ids, k, 2)
    out["knn_embeds"], out["knn_ids"], out["knn_k"] = e, np.asarray(ids, dtype=np.int64), np.int64(k)
    out["knn_table"] = np.asarray([sorted(dic[i]) for i in ids], dtype=np.int64)
    out["knn_min_gap"] = np.float64(gap.min())
    # ---- weighted view averaging ------------------------------------------------------------------------------------
    tf = sys.modules["tensorflow"]
    tf.__getattr__ = lambda name: mock.MagicMock(name="tf." + name)
    tf.nn.__getattr__ = lambda name: mock.MagicMock(name="tf.nn." + name)
    ref_late = importlib.import_module("MultiKE_Late")
    rng = np.random.default_rng(4100)
    views = [rng.standard_normal((300, 24)).astype(np.float32) for _ in range(3)]
    views[1] = (0.5 * views[0] + views[1]).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        w = ref_late.wva(*views)
    for i, v in enumerate(views):
        out[f"wva_view{i}"] = v
    out["wva_weights"] = np.asarray(w, dtype=np.float64)
    # ---- the NumPy final encode of the literal auto-encoder -----------------------------------------------------------
    ref_le = importlib.import_module("literal_encoder")

    class _Var:
        def __init__(self, a):
            self.a = a

        def eval(self, session=None):
            return self.a

    dims = [12, 8, 6, 4]
    rng = np.random.default_rng(4200)
    ws = [rng.standard_normal((dims[i], dims[i + 1])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) for i in range(3)]
    for i in range(3):
        out[f"enc_w{i}"], out[f"enc_b{i}"] = ws[i], bs[i]
    for rows in (25, 30):
        x = rng.standard_normal((rows, dims[0])).astype(np.float32)
        out[f"enc_x{rows}"] = x
        for act in ("sigmoid", "tanh"):
            m = object.__new__(ref_le.AutoEncoderModel)
            m.args = types.SimpleNamespace(dim=dims[-1], batch_size=10, encoder_active=act)
            m.input_dimension, m.layer_num, m.session = dims[0], 3, None
            m.weights = {f"encoder_h{i}": _Var(ws[i]) for i in range(3)}
            m.biases = {f"encoder_b{i}": _Var(bs[i]) for i in range(3)}
            with contextlib.redirect_stdout(io.StringIO()):
                out[f"enc_out{rows}_{act}"] = np.asarray(m.encoder_multi_batches(x), dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--only", default="", help="cnn / graphs: rewrite cnn_golden.npz / graphs_golden.npz only (the other fixtures keep their bytes)")
    a = ap.parse_args()
    code = os.path.join(a.reference, "code")
    if not os.path.isdir(code):
        sys.exit(f"reference not found at {code} (this script only runs in the build container)")
    sys.path.insert(0, code)
    tf = install_tf_forwarder()
    install_empty_standins()
    ref_losses = importlib.import_module("losses")
    ref_batch = importlib.import_module("base.batch")
    ref_attr_batch = importlib.import_module("attr_batch")
    ref_utils = importlib.import_module("utils")

    out = {}
    if a.only not in ("cnn", "graphs"):
        losses_fixture(ref_losses, tf, out)
        np.savez_compressed(os.path.join(HERE, "losses_golden.npz"), **out)
    cnn = {}
    if a.only != "graphs":
        cnn_fixture(cnn)
        cnn_reference_fixture(cnn)
        np.savez_compressed(os.path.join(HERE, "cnn_golden.npz"), **cnn)
    if a.only == "cnn":
        print("wrote cnn_golden.npz")
        return
    gr = {}
    graphs_fixture(gr)
    ae_graph_fixture(gr)
    np.savez_compressed(os.path.join(HERE, "graphs_golden.npz"), **gr)
    if a.only == "graphs":
        print("wrote graphs_golden.npz")
        return
    ev = {}
    eval_fixture(importlib.import_module("base.alignment"), ev)
    np.savez_compressed(os.path.join(HERE, "eval_golden.npz"), **ev)
    js = {}
    sampler_fixture(ref_batch, ref_attr_batch, js)
    host_fixture(ref_utils, js)
    with open(os.path.join(HERE, "sampler_golden.json"), "w") as f:
        json.dump(js, f, separators=(",", ":"))
    dj = {}
    data_fixture(importlib.import_module("base.kgs"), ref_utils, importlib.import_module("predicate_alignment"), dj)
    with open(os.path.join(HERE, "data_golden.json"), "w") as f:
        json.dump(dj, f, separators=(",", ":"), sort_keys=True)
    pins = {}
    pins_fixture(ref_batch, pins)
    np.savez_compressed(os.path.join(HERE, "pins_golden.npz"), **pins)
    sj = {}
    schedule_fixture(sj)
    with open(os.path.join(HERE, "schedule_golden.json"), "w") as f:
        json.dump(sj, f, separators=(",", ":"))
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
