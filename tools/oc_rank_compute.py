#!/usr/bin/env python3
"""What ONE rank of a G-rank owner-computes job computes per global step, measured on one GPU with nobody else there.

The kernels of rank 0 at world size G see G x the positives (HR / RT vectors read, gradient vectors written, codes scanned) and
1 / G of the entity rows; none of that depends on what the other ranks' vectors CONTAIN.  So the trainer is built as rank 0 of G
with a loop-back communicator — the all-gathered buffer is G copies of this rank's own block, the reduce-scatter keeps this
rank's slice, the all-reduce is the identity — and every phase between two collectives is bracketed with HIP events.  The
numbers are NOT a training result (the arithmetic is fed copies), they are the per-rank COMPUTE terms of DESIGN.md §5.2, which
round 3 had modelled.  Run it under rocprofv3 (tools/prof.sh) for the per-kernel table.

    python tools/oc_rank_compute.py [--world 8] [--config c2|c5] [--steps 60] [--chunks 1]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multike_amd import _lib
from multike_amd.distributed_oc import APPLY, BASES, COUNT, SCORE, UPDATE, OcComm, OwnerComputesTrainer
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import xavier_truncated_normal


class LoopbackComm(OcComm):
    def __init__(self, world, rank):
        super().__init__(None)
        self.world, self.rank = world, rank

    def all_gather(self, out, mine, async_op=False):
        out.view(self.world, -1).copy_(mine.reshape(1, -1).expand(self.world, -1))

    def reduce_scatter(self, out, inp, async_op=False):
        out.copy_(inp.view(self.world, -1)[self.rank].view_as(out))

    def all_reduce(self, t, op=None):
        pass

    def barrier(self, token):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--config", choices=["c2", "c5"], default="c2")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=VALUE", help="mke_set_option before the run (A/B of a kernel choice)")
    a = ap.parse_args()
    for kv in a.set:
        k, _, v = kv.partition("=")
        _lib.set_option(k, int(v))
    cfg = dict(n_ent=200_000, n_rel=550, dim=75, neg=25) if a.config == "c2" else dict(n_ent=2_000_000, n_rel=2000, dim=256, neg=64)
    G, B = a.world, 5000
    kgs = SyntheticKGs(n_ent=cfg["n_ent"], n_rel=cfg["n_rel"], seed=1234)
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    ent0 = (torch.randn(cfg["n_ent"], cfg["dim"], generator=g) * float(np.sqrt(2.6 / (cfg["n_ent"] + cfg["dim"])))).numpy()
    rel0 = xavier_truncated_normal(cfg["n_rel"], cfg["dim"], "cpu", seed=2).numpy()
    tr = OwnerComputesTrainer(kgs, ent0, rel0, B, cfg["neg"], 0, G, seed=1, chunks=a.chunks, comm=LoopbackComm(G, 0), prefetch=False)
    names = {BASES | COUNT: "bases+count", BASES: "bases", SCORE: "score", APPLY: "apply", UPDATE: "update", APPLY | UPDATE: "apply+update",
             BASES | COUNT | SCORE | APPLY | UPDATE: "whole step (one call)"}
    ev = []
    orig = tr.backend.run

    def timed_run(t, k, tag, phases, c, loss_slot):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(t, k, tag, phases, c, loss_slot)
        e1.record()
        ev.append((names.get(phases, str(phases)), e0, e1))

    n = min(a.steps, tr.steps - 1)
    for i in range(min(5, n)):
        tr.step(i)
    torch.cuda.synchronize()
    # per-epoch plan (sampler of ALL the epoch's negatives + code packing + mke_oc_plan), in line
    b = tr.bat
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p0.record()
    plan = tr._compute_plan((b.pos_h, b.pos_r, b.pos_t), b.rng_stream, 1)
    p1.record()
    torch.cuda.synchronize()
    plan_ms = p0.elapsed_time(p1)
    # untimed wall of the steps (collectives = device copies of the same byte counts: an in-HBM stand-in, not a link)
    t0 = time.perf_counter()
    for i in range(5, n):
        tr.step(i)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / max(1, n - 5)
    # instrumented pass
    tr.backend.run = timed_run
    torch.cuda._sleep(int(2.4e9 * 0.03))
    for i in range(n, min(tr.steps, n + 40)):
        tr.step(i)
    torch.cuda.synchronize()
    phases = {}
    for name, e0, e1 in ev:
        phases.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
    out = {"tool": "oc_rank_compute", "config": a.config, "options": a.set, "world": G, "rank": 0, "chunks": a.chunks, "global_batch": B * G,
           "scored_per_global_step": B * G * (1 + cfg["neg"]), "steps_per_epoch": tr.steps, "rows_owned": tr.n_local,
           "capacity_vectors": tr.C,
           "phase_us": {k: float(np.mean(v)) for k, v in phases.items()},
           "compute_us_per_step": float(sum(np.mean(v) * (len(v) / max(1, len(phases.get("score", v)))) for v in phases.values())),
           "epoch_plan_ms": plan_ms, "epoch_plan_us_per_step": plan_ms * 1e3 / tr.steps,
           "wall_us_per_step_loopback": wall * 1e6,
           "note": "rank 0's kernels at world-size shapes on one GPU; collectives are loop-back device copies (not measured links)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
