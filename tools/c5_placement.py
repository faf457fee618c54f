#!/usr/bin/env python3
"""Why does the C5 launch time move 245-302 us between allocations of the same tables (round-4 review, weak 5)?

The step touches three arrays with the SAME row index at the same time: table[row], accumulator[row], gradient[row] (2 GB each at
C5).  Hypothesis tested here: what matters is not where the pages are but how the three bases sit RELATIVE to each other — when
their distance is a multiple of a large power of two, the three accesses of a row fall on the same memory channel / bank group.
The three arrays are carved out of ONE allocation at a chosen skew between them, the step loop is timed, and the same skews are
repeated on a second, different allocation (the first one kept alive): if the time follows the skew and not the allocation,
placement is the relative layout and a policy can pin it.

    python tools/c5_placement.py [--steps 120] [--config c5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multike_amd.runner import RelationViewRunner
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--config", default="c5")
    a = ap.parse_args()
    cfg = dict(n_ent=2_000_000, n_rel=2000, dim=256, neg=64) if a.config == "c5" else dict(n_ent=200_000, n_rel=550, dim=75, neg=25)
    kgs = SyntheticKGs(n_ent=cfg["n_ent"], n_rel=cfg["n_rel"], seed=1234)
    sides = []
    for k in (0, 1):
        t = torch.as_tensor(kgs.triples[k], device="cuda")
        sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
    d, n = cfg["dim"], cfg["n_ent"]
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    keep, rows = [], []

    def trial(skew_bytes, label):
        E = EmbeddingTable(n, d, "e", trainable=False)
        E.trainable = True
        stride, tab = E.stride, n * E.stride
        skew = skew_bytes // 4
        big = torch.zeros(3 * tab + 2 * skew + 1024, dtype=torch.float32, device="cuda")
        keep.append(big)                                    # the next trial gets other pages
        views = [big[k * (tab + skew):k * (tab + skew) + tab].view(n, stride) for k in range(3)]
        views[0][:, :d] = torch.randn(n, d, device="cuda", generator=g) * float(np.sqrt(2.6 / (n + d)))
        views[1].fill_(0.1)
        E.data, E.slots["relation"], E._grad_full, E._grad = views[0], views[1], views[2], views[2]
        R = EmbeddingTable(cfg["n_rel"], d, "r", seed=2)
        bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, cfg["neg"], seed=7)
        run = RelationViewRunner(E, R, bat, "relation", lr=0.001, hot_rows=False)
        run.run(0, 40)
        torch.cuda.synchronize()
        best = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run.run(40, 40 + a.steps)
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3 / a.steps)
        ptrs = [v.data_ptr() for v in views]
        rows.append({"label": label, "skew_bytes": skew_bytes, "us_per_step": [round(x, 1) for x in best],
                     "base_mod_2MiB": [p % (1 << 21) for p in ptrs], "base_mod_1GiB_MiB": [round((p % (1 << 30)) / 2 ** 20, 2) for p in ptrs],
                     "distance_MiB": [round((ptrs[k + 1] - ptrs[k]) / 2 ** 20, 4) for k in range(2)]})
        print(json.dumps(rows[-1]), flush=True)
        del run, bat, E, R

    skews = [0, 4096, 64 * 1024 + 1024, 1 << 20, (1 << 21) + 4096, 37 * 4096 + 1024]
    for alloc in ("A", "B"):
        for s in skews:
            trial(s, f"alloc {alloc}")
    # the product's own layout: three separate torch allocations, three times over
    for k in range(3):
        E = EmbeddingTable(n, d, "e", trainable=False)
        E.trainable = True
        E.data[:, :d] = torch.randn(n, d, device="cuda", generator=g) * float(np.sqrt(2.6 / (n + d)))
        R = EmbeddingTable(cfg["n_rel"], d, "r", seed=2)
        bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, cfg["neg"], seed=7)
        run = RelationViewRunner(E, R, bat, "relation", lr=0.001, hot_rows=False)
        run.run(0, 40)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run.run(40, 40 + a.steps); e1.record(); torch.cuda.synchronize()
        ptrs = [E.data.data_ptr(), E.slot("relation").data_ptr(), E.grad.data_ptr()]
        print(json.dumps({"label": f"separate allocations {k}", "us_per_step": round(e0.elapsed_time(e1) * 1e3 / a.steps, 1),
                          "distance_MiB": [round((ptrs[i + 1] - ptrs[i]) / 2 ** 20, 4) for i in range(2)],
                          "base_mod_1GiB_MiB": [round((p % (1 << 30)) / 2 ** 20, 2) for p in ptrs]}), flush=True)
        keep.append((E.data, E.slot("relation"), E.grad))


if __name__ == "__main__":
    main()
