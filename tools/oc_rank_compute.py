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
from multike_amd.distributed_oc import APPLY, BASES, COUNT, PASS2, SCORE, UPDATE, OcComm, OwnerComputesTrainer
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import xavier_truncated_normal


class _Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class LoopbackComm(OcComm):
    """Stand-in for the three collectives on ONE GPU: the all-gathered buffer = `world` copies of this rank's block, the
    reduce-scatter keeps this rank's slice, the all-reduce is the identity — the REAL byte counts are copied in HBM.
    wire_gbps > 0: the collective also holds its stream for bytes-on-the-wire / wire_gbps (a device-side sleep), the
    modelled cost of the links (376 GB/s for an all-gather / reduce-scatter of equal blocks over 7 xGMI links),
    plus latency_us per call.  async_op=True (the trainer's chunked schedule): the copy and the wait run on the communicator's
    own stream, as torch.distributed's do, and the handle's wait() orders the caller's stream after them — what is measured
    is then how much of the modelled wire time the schedule hides behind this rank's kernels."""

    def __init__(self, world, rank, wire_gbps=0.0, latency_us=0.0):
        super().__init__(None)
        self.world, self.rank, self.wire_gbps, self.latency_us = world, rank, wire_gbps, latency_us
        self.stream = torch.cuda.Stream()
        self.clock_hz = 100e6          # torch.cuda._sleep counts cycles of the 100 MHz wall clock register on gfx9 (calibrated in main)
        self.calls = 0
        self.events, self.k = None, 0

    def for_plan(self):
        return self

    def _run(self, fn, wire_bytes, async_op):
        self.calls += 1
        cur = torch.cuda.current_stream()
        hold = (wire_bytes / (self.wire_gbps * 1e9) if self.wire_gbps > 0 else 0.0) + self.latency_us * 1e-6
        if not async_op:
            fn()
            if hold > 0:
                torch.cuda._sleep(int(hold * self.clock_hz))
            return None
        # as OcRcclComm._async: the communicator's own stream, two pooled events
        if self.events is None:
            self.events = [torch.cuda.Event() for _ in range(64)]
        e_in, e_out = self.events[self.k % 64], self.events[(self.k + 1) % 64]
        self.k += 2
        e_in.record(cur)
        self.stream.wait_event(e_in)
        with torch.cuda.stream(self.stream):
            fn()
            if hold > 0:
                torch.cuda._sleep(int(hold * self.clock_hz))
            e_out.record(self.stream)
        return _Work(e_out)

    def all_gather(self, out, mine, async_op=False):
        wire = (self.world - 1) * mine.numel() * mine.element_size()         # bytes this rank receives
        return self._run(lambda: out.view(self.world, -1).copy_(mine.reshape(1, -1).expand(self.world, -1)), wire, async_op)

    def reduce_scatter(self, out, inp, async_op=False):
        wire = (self.world - 1) * out.numel() * out.element_size()
        return self._run(lambda: out.copy_(inp.view(self.world, -1)[self.rank].view_as(out)), wire, async_op)

    def all_reduce(self, t, op=None):
        if self.latency_us > 0 or self.wire_gbps > 0:      # small replicated gradient: latency-bound
            self._run(lambda: None, 2 * (self.world - 1) / self.world * t.numel() * t.element_size(), False)

    def barrier(self, token):
        pass

    def native(self, tr=None):
        """mke_oc_comm of kind LOOPBACK: the same stand-in inside the library, for the native step loop (mke_oc_steps)."""
        cs = _lib.OcCommStruct()
        cs.kind, cs.world, cs.rank = _lib.OC_COMM_LOOPBACK, self.world, self.rank
        cs.wire_gbps, cs.latency_us = self.wire_gbps, self.latency_us
        return cs


def calibrate_sleep():
    """cycles of torch.cuda._sleep per second on this device"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(10_000_000)
    e1.record()
    torch.cuda.synchronize()
    return 10_000_000 / (e0.elapsed_time(e1) * 1e-3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--config", choices=["c2", "c5"], default="c2")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--wire-gbps", type=float, default=0.0, help="modelled link rate of an all-gather / reduce-scatter (0: HBM copies only)")
    ap.add_argument("--latency-us", type=float, default=0.0, help="modelled launch latency per collective call")
    ap.add_argument("--prefetch", action="store_true", help="the next epoch's plan on the side stream while the steps run (the product's default)")
    ap.add_argument("--zipf", type=float, default=0.0, help="head / tail entities of the synthetic triples ~ rank^-zipf (hub rows)")
    ap.add_argument("--rel-zipf", type=float, default=0.0, help="relation ids of the synthetic triples ~ rank^-rel_zipf")
    ap.add_argument("--hi-prio", type=int, default=0, help="1: the steps run on a high-priority stream (the epoch plan's side stream then yields to them)")
    ap.add_argument("--native", type=int, default=1, help="1 (default): the timed steps go through mke_oc_steps (one native call); 0: the Python step loop")
    ap.add_argument("--em", type=int, default=1, help="1 (default): entity-major second pass; 0: the atomics form of rounds 2-5")
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=VALUE", help="mke_set_option before the run (A/B of a kernel choice)")
    a = ap.parse_args()
    for kv in a.set:
        k, _, v = kv.partition("=")
        _lib.set_option(k, int(v))
    cfg = dict(n_ent=200_000, n_rel=550, dim=75, neg=25) if a.config == "c2" else dict(n_ent=2_000_000, n_rel=2000, dim=256, neg=64)
    G, B = a.world, 5000
    kgs = SyntheticKGs(n_ent=cfg["n_ent"], n_rel=cfg["n_rel"], seed=1234, zipf=a.zipf, rel_zipf=a.rel_zipf)
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    ent0 = (torch.randn(cfg["n_ent"], cfg["dim"], generator=g) * float(np.sqrt(2.6 / (cfg["n_ent"] + cfg["dim"])))).numpy()
    rel0 = xavier_truncated_normal(cfg["n_rel"], cfg["dim"], "cpu", seed=2).numpy()
    if a.hi_prio:
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    comm = LoopbackComm(G, 0, a.wire_gbps, a.latency_us)
    comm.clock_hz = calibrate_sleep()
    tr = OwnerComputesTrainer(kgs, ent0, rel0, B, cfg["neg"], 0, G, seed=1, chunks=a.chunks, comm=comm, prefetch=a.prefetch, entity_major=bool(a.em))
    names = {BASES | COUNT: "bases+count", BASES: "bases", SCORE: "score", APPLY: "apply", UPDATE: "update", APPLY | UPDATE: "apply+update",
             BASES | COUNT | SCORE | APPLY | UPDATE: "whole step (one call)", PASS2: "pass2", PASS2 | UPDATE: "pass2+update",
             BASES | SCORE | PASS2 | UPDATE: "whole step (one call)"}
    ev = []
    orig = tr.backend.run

    def timed_run(t, k, tag, phases, c, loss_slot):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(t, k, tag, phases, c, loss_slot)
        e1.record()
        ev.append((names.get(phases, str(phases)), e0, e1))

    n = a.steps if a.native else min(a.steps, tr.steps - 1)
    w0 = 0
    if a.native and a.prefetch:       # two whole epochs first: the second buffer set of the epoch plan is allocated on the way
        w0 = 2 * tr.steps
        tr.run(0, w0)
        n += w0
    for i in range(w0, w0 + min(5, n - w0)):
        tr.step(i)
    torch.cuda.synchronize()
    # untimed wall of the steps (collectives = device copies of the same byte counts: an in-HBM stand-in, not a link)
    t0 = time.perf_counter()
    if a.native:
        tr.run(w0 + 5, n - w0 - 5)                            # mke_oc_steps: runs of steps enqueued from C++ (epoch boundaries in Python)
    else:
        for i in range(5, n):
            tr.step(i)
    host = (time.perf_counter() - t0) / max(1, n - w0 - 5)    # the enqueue loop alone (nothing waits for the device)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / max(1, n - w0 - 5)
    # instrumented pass
    tr.backend.run = timed_run
    torch.cuda._sleep(int(2.4e9 * 0.03))
    for i in range(n, n + 40):
        if i % tr.steps == 0 and not a.native:
            break
        tr.step(i)
    torch.cuda.synchronize()
    # per-epoch plan (sampler of this rank's share of the epoch's negatives + code packing + mke_oc_plan + the entity-major
    # lists), in line, after everything else (it overwrites a buffer set a prefetched plan may be waiting in)
    b = tr.bat
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p0.record()
    tr._compute_plan((b.pos_h, b.pos_r, b.pos_t), b.rng_stream, 1)
    p1.record()
    torch.cuda.synchronize()
    plan_ms = p0.elapsed_time(p1)
    phases = {}
    for name, e0, e1 in ev:
        phases.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
    out = {"tool": "oc_rank_compute", "config": a.config, "zipf": a.zipf, "rel_zipf": a.rel_zipf, "options": a.set, "world": G, "rank": 0, "entity_major": tr.em, "native_loop": bool(a.native), "hi_prio_stream": bool(a.hi_prio), "em_refs_per_step": (tr._em["n_refs_host"] / max(1, tr.steps)) if tr.em else None,
           "em_rows_per_step": (int(tr._em["row0_host"][-1]) / max(1, tr.steps)) if tr.em else None,
           "em_long_rows_per_step": (int(tr._em["long0_host"][-1]) / max(1, tr.steps)) if tr.em else None, "chunks": a.chunks, "global_batch": B * G,
           "scored_per_global_step": B * G * (1 + cfg["neg"]), "steps_per_epoch": tr.steps, "rows_owned": tr.n_local,
           "capacity_vectors": tr.C,
           "phase_us": {k: float(np.mean(v)) for k, v in phases.items()},
           "compute_us_per_step": float(sum(np.mean(v) * (len(v) / max(1, len(phases.get("score", v)))) for v in phases.values())),
           "epoch_plan_ms": plan_ms, "epoch_plan_us_per_step": plan_ms * 1e3 / tr.steps,
           "wall_us_per_step_loopback": wall * 1e6, "host_us_per_step": host * 1e6,
           "wire_gbps_modelled": a.wire_gbps, "latency_us_modelled": a.latency_us, "prefetch": a.prefetch,
           "rank_share_of_epoch_sampled": 1.0 / G,
           "note": "rank 0's kernels at world-size shapes on one GPU; collectives are loop-back device copies (not measured links)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
