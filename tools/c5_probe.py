#!/usr/bin/env python3
"""Does a cheap read-only probe predict which allocation of the C5 tables is a slow one (round-4 review, weak 5)?

`tools/c5_placement.py` showed that the C5 step moves 316-371 us between allocations whatever the three arrays' relative
layout.  Here every candidate array (2 GB, rows of 1 KB) is probed ALONE — `mke_gather_rows` of 2M random rows, read-only —
and then the relation step is timed on triples (table, accumulator, gradient) put together from the candidates: if the
step's time follows the sum of its three arrays' probe times, a constructor can allocate a few candidates, keep the fastest
and free the rest ("placement by trial").

    python tools/c5_probe.py [--cands 6] [--steps 120]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multike_amd import _lib
from multike_amd.runner import RelationViewRunner
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable


def probe(arr, idx, out, reps=5):
    """us per gather of len(idx) random rows of `arr` (median of reps)."""
    ts = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.gather_rows(arr, False, arr.shape[1], idx, out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[1:]))


def probe3(a, b, c, idx, out1, reps=5):
    """us per launch of mke_probe_rows: the same random rows of a (b, c) read together."""
    ts = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.probe_rows(a, b, c, idx, out1)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[1:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cands", type=int, default=6)
    ap.add_argument("--steps", type=int, default=120)
    a = ap.parse_args()
    n, d, n_rel, neg = 2_000_000, 256, 2000, 64
    kgs = SyntheticKGs(n_ent=n, n_rel=n_rel, seed=1234)
    sides = []
    for k in (0, 1):
        t = torch.as_tensor(kgs.triples[k], device="cuda")
        sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    idx = torch.randint(0, n, (2_000_000,), device="cuda", generator=g, dtype=torch.int32)
    out = torch.empty(idx.numel(), d, device="cuda")
    stride = _lib.stride_for(d)
    # 3 * cands candidate arrays, all alive at once (so that they are different physical pages)
    cands = [torch.zeros(n, stride, device="cuda") for _ in range(3 * a.cands)]
    pr = [probe(c, idx, out) for c in cands]
    print(json.dumps({"probe_us": [round(x, 1) for x in pr]}), flush=True)
    order = np.argsort(pr)
    init = torch.randn(n, d, device="cuda", generator=g) * float(np.sqrt(2.6 / (n + d)))

    def step_time(ids, label):
        E = EmbeddingTable(n, d, "e", trainable=False)
        E.trainable = True
        views = [cands[i] for i in ids]
        views[0].zero_(); views[0][:, :d] = init
        views[1].fill_(0.1)
        views[2].zero_()
        E.data, E.slots["relation"], E._grad_full, E._grad = views[0], views[1], views[2], views[2]
        R = EmbeddingTable(n_rel, d, "r", seed=2)
        bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, neg, seed=7)
        run = RelationViewRunner(E, R, bat, "relation", lr=0.001, hot_rows=False)
        run.run(0, 40)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run.run(40, 40 + a.steps); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / a.steps)
        out1 = out.view(-1)[:idx.numel()]
        p3 = probe3(cands[ids[0]], cands[ids[1]], cands[ids[2]], idx, out1)
        p2 = [probe3(cands[ids[x]], cands[ids[y]], None, idx, out1) for x, y in ((0, 1), (0, 2), (1, 2))]
        print(json.dumps({"label": label, "arrays": [int(i) for i in ids], "probe_us": [round(pr[i], 1) for i in ids],
                          "probe3_us": round(p3, 1), "probe_pairs_us": [round(x, 1) for x in p2],
                          "step_us": [round(x, 1) for x in ts]}), flush=True)

    k = a.cands
    step_time(order[:3], "three fastest probes")
    step_time(order[-3:], "three slowest probes")
    step_time(order[k:k + 3], "three median probes")
    step_time(order[:3][::-1], "three fastest, roles permuted")
    step_time([order[0], order[-1], order[-2]], "fast table, slow accumulator + gradient")
    step_time([order[-1], order[0], order[1]], "slow table, fast accumulator + gradient")
    step_time(order[:3], "three fastest probes again")
    rng = np.random.default_rng(0)
    for t in range(8):                                     # random triples: does probe3 rank them as the step does?
        step_time(rng.choice(len(cands), 3, replace=False), f"random triple {t}")


if __name__ == "__main__":
    main()
