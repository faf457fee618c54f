#!/usr/bin/env python3
"""bench.py — scored (pos+neg) triples/s of the MultiKE relation-view train step on MI355X.

A "step" is one pass of the hot path over one batch: on-device negative sampling -> fused gather /
normalise / score / logistic loss / gradient scatter -> per-row Jacobian + Adagrad on both tables
(what one `session.run([relation_loss, relation_optimizer])` of the reference does,
code/MultiKE_model.py:304-310).  Workload at N=1: BASELINE.json configs[1] shape on synthetic triples
(|E|=200K, |R|=550, dim=75, neg=25, batch 5000 => 130K scored triples per step; SURVEY.md §8d "C2-synth").

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

    python bench.py --config c5        # BASELINE.json configs[4] per-GPU shape (|E|=2M, |R|=2K, dim=256, neg=64): the
                                       # HBM-resident size (2 GB table; C2's 192 MB working set sits in the Infinity Cache)

Prints ONE JSON line on rank 0 (see DESIGN.md §6 for every field).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # what a float4 streaming copy reaches on this part (same guide)

# SURVEY.md §8(d) shapes.  c2 = the shape BASELINE.json's metric is quoted on (configs[1], DBP-WD-like); c5 = configs[4]'s
# per-GPU shape.
CONFIGS = {
    "c2": dict(n_ent=200_000, n_rel=550, dim=75, neg=25, batch=5000, label="C2-synth"),
    "c5": dict(n_ent=2_000_000, n_rel=2000, dim=256, neg=64, batch=5000, label="C5-synth"),
}
KERNEL_SOURCES = ("mke_score.hip", "mke_update.hip", "mke_common.h")


def kernel_source_hash():
    """sha256 over the sources of the two kernels of the step: a committed PMC figure is only quoted for the build it
    was collected on (profiles/r0N_pmc_<config>.json carries the hash)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "multike_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=920)  # five synthetic epochs of 184 steps at c2
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--n-ent", type=int, default=None)
    ap.add_argument("--n-rel", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--neg", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-variants", action="store_true", help="skip the reference-default-shape side lines (N=10, truncated)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="time budget of each cpu_baseline leg")
    ap.add_argument("--sample-chunk", type=int, default=0, help="steps sampled per sampler launch (0 = whole epoch)")
    ap.add_argument("--rel-grad-copies", type=int, default=8, help="privatised copies of the relation gradient scratch (the model's default: MultiKE_model.REL_GRAD_COPIES)")
    ap.add_argument("--force-sharded", action="store_true", help="run the row-sharded multi-GPU path even at N=1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-epochs", type=int, default=1,
                    help="whole untimed epochs run BEFORE the --warmup steps (clocks, caches, first launches); the timed region "
                         "is unchanged: exactly --steps complete steps")
    ap.add_argument("--cpu-steps", type=int, default=360, help="upper bound of steps per cpu_baseline leg")
    ap.add_argument("--windows", type=int, default=None,
                    help="how many times the --steps window is timed (same bracket every time; `value` = the median window, the "
                         "first window is reported beside it).  Default: as many as make the timed work >= 60 ms, at least 50 "
                         "for windows of <= 100 steps, 5 otherwise")
    ap.add_argument("--zipf", type=float, default=0.0, help="Zipf exponent of the head / tail entities of the synthetic triples (0 = uniform)")
    ap.add_argument("--rel-zipf", type=float, default=0.0, help="Zipf exponent of the relation ids of the synthetic triples (0 = uniform)")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    for k in ("n_ent", "n_rel", "dim", "neg", "batch"):
        if getattr(a, k) is None:
            setattr(a, k, cfg[k])
        elif getattr(a, k) != cfg[k]:
            a.custom = True
    a.label = cfg["label"] if not getattr(a, "custom", False) else "custom"
    return a


def b_alg(dim):
    """Algorithmic bytes per scored triple: 3 ids + 3 gathered rows + 3 gradient rows."""
    return 12 + 24 * dim


def cpu_baseline(args, kgs, ent, rel):
    """The oracle's C restatement (oracle/mke_oracle.c: Philox sampler + mko_relation_step_mt_f32) timed on this box's
    host cores on a bounded sample of the same workload (at most one epoch of steps per leg, --cpu-seconds each): sampler +
    step with the touched-rows update, OpenMP over positives / triples / row ranges on the fastest thread count of a probe;
    the same on 1 thread; and, interleaved step by step with the first leg, the same arithmetic with the whole-table passes
    the reference's dense graph makes every step (whole-table normalise, Jacobian / Adagrad over every row)."""
    from oracle import c_oracle as co
    from oracle import multike_oracle as mo
    d, N, B = args.dim, args.neg, args.batch
    t1, t2 = kgs.triples
    b1, b2 = mo.kg_batch_split(len(t1), len(t2), B)
    sets = [co.TripleSet(t[:, 0], t[:, 1], t[:, 2]) for t in (t1, t2)]
    cores = host_threads()
    n_steps_epoch = int(np.ceil((len(t1) + len(t2)) / B))
    out = {}
    def one_step(orc, e, r, a, b, s):
        pos_parts, neg_parts = [], []
        for k, (t, bs) in enumerate(((t1, b1), (t2, b2))):
            p = t[s * bs:(s + 1) * bs]
            lo, hi = kgs.ent_range[k]
            neg_parts.append(co.neg_sample(p[:, 0], p[:, 1], p[:, 2], N, hi - lo, ent_lo=lo, known=sets[k],
                                           seed=(1, 0), stream_id=k, pos_offset=s * bs))
            pos_parts.append(p)
        pos = [np.concatenate([pos_parts[0][:, i], pos_parts[1][:, i]]) for i in range(3)]
        neg = [np.concatenate([neg_parts[0][i], neg_parts[1][i]]) for i in range(3)]
        orc.step(e, r, a, b, pos, neg, 0.001)
        return len(pos[0]) * (1 + N)

    # the thread count of the "all cores" legs: the fastest of {all, 1/2, 1/4, ... >= 8} on one probe step each (row-range
    # ownership and first-touch placement stop scaling well before 256 threads on a two-socket host)
    cands = sorted({c for c in (cores, cores // 2, cores // 4, cores // 8, 16, 8) if 1 <= c <= cores}, reverse=True)
    if len(cands) > 1:
        e, r = ent.copy(), rel.copy()
        a, b = np.full_like(e, 0.1), np.full_like(r, 0.1)
        best = None
        for c in cands:
            orc = co.RelationStepBaselineMT(e.shape[0], r.shape[0], d, dense=False, threads=c)
            co.set_threads(c)
            one_step(orc, e, r, a, b, 0)
            t0 = time.perf_counter()
            one_step(orc, e, r, a, b, 1)
            one_step(orc, e, r, a, b, 2)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, c)
            del orc
        threads_all = best[1]
        del e, a
    else:
        threads_all = cores
    def leg_state(dense, threads):
        e, r = ent.copy(), rel.copy()
        return [co.RelationStepBaselineMT(e.shape[0], r.shape[0], d, dense=dense, threads=threads), e, r,
                np.full_like(e, 0.1), np.full_like(r, 0.1), 0, 0, 0.0]     # ..., scored, steps, seconds

    def timed_step(st, s):
        t0 = time.perf_counter()
        n_sc = one_step(st[0], st[1], st[2], st[3], st[4], s)
        if s > 0:            # first step: page faults of the scratch, OpenMP thread start-up
            st[5] += n_sc; st[6] += 1; st[7] += time.perf_counter() - t0

    # the touched-rows leg and the whole-table leg run INTERLEAVED, step by step (round 2 ran them one after the other and
    # the second came out faster on two boxes: clock / placement drift between legs was part of that)
    co.set_threads(threads_all)
    sp, wt = leg_state(False, threads_all), leg_state(True, threads_all)
    for s in range(min(args.cpu_steps, n_steps_epoch) + 1):
        timed_step(sp, s)
        timed_step(wt, s)
        if sp[7] > args.cpu_seconds or wt[7] > args.cpu_seconds:
            break
    out["all"], out["whole"] = (sp[5] / sp[7], sp[6], sp[7]), (wt[5] / wt[7], wt[6], wt[7])
    del sp, wt
    co.set_threads(1)
    one = leg_state(False, 1)
    for s in range(min(args.cpu_steps, n_steps_epoch) + 1):
        timed_step(one, s)
        if one[7] > args.cpu_seconds:
            break
    out["one"] = (one[5] / one[7], one[6], one[7])
    del one
    v, steps, dt = out["all"]
    v1, s1, dt1 = out["one"]
    vd, dsteps, ddt = out["whole"]
    return {
        "value": v, "unit": "scored triples/s", "cores": threads_all, "kind": "port", "host_cores_available": cores,
        "sample": f"{steps} steps of the same workload ({dt:.1f}s): C restatement of sampler + relation-view step, "
                  f"touched-rows update, fp32, OpenMP on {threads_all} threads (fastest of {cands} on probe steps)",
        "one_thread_value": v1, "one_thread_sample": f"{s1} steps ({dt1:.1f}s), 1 thread",
        # NOT "the TF-CPU reference" and not slower by construction: the same per-triple arithmetic on rows of a table that
        # was normalised as a whole first, then Jacobian + Adagrad over EVERY row (the passes the reference's dense graph
        # makes every step).  On a host with a large last-level cache the whole-table pass streams the 60 MB table in ahead
        # of the step's random row gathers and can come out FASTER than the touched-rows form, which misses on ~half of them
        # which leg is which (round-4 review, weak 9): `whole_table_passes_value` is the reference-FAITHFUL cost model — TF1's dense
        # graph normalises the whole variable and applies Adagrad to every row at every step (code/base/initializers.py:26,
        # code/MultiKE_model.py:15-31) — while `value` is the same arithmetic restricted to the touched rows (what any sparse
        # implementation, this package included, does).  Neither is TensorFlow itself (BASELINE.md estimates that at 0.18 M/s).
        # (round 5 called the dense leg "reference_faithful_leg": it is a C / OpenMP cost model of the SHAPE of TF1's dense work, far
        # kinder to the reference than TF1 itself — the name oversold it: round-5 review, weak 7)
        "dense_cost_model_leg": "whole_table_passes_value",
        "whole_table_passes_value": vd,
        "whole_table_passes_sample": f"{dsteps} steps ({ddt:.1f}s) on {threads_all} threads, interleaved step by step with the "
                                     f"touched-rows leg: whole-table normalise + Jacobian/Adagrad over all {ent.shape[0]} rows",
    }


def reference_default_variants(args, kgs, ent0, rel0, sides):
    """Side lines (the headline stays the shape BASELINE.json's metric is quoted on): the reference's DEFAULT
    configuration (code/args.json:25-28) — neg_triple_num 10, and neg_sampling "truncated": candidates drawn from each
    entity's 2000 nearest neighbours (k = int(0.02 |E_kg|), code/MultiKE_CSL.py:89-102) instead of the whole KG.  Same
    tables, same KGs, fresh optimizer slots; the k-NN table is built by the product's own refresh (base/batch.neighbour_table)
    from the current relation-view rows and is NOT inside the timed region (the reference refreshes it every 20 epochs)."""
    from multike_amd.base.batch import neighbour_table
    from multike_amd.runner import RelationViewRunner
    from multike_amd.sampling import KGSide, RelationBatcher
    from multike_amd.tables import EmbeddingTable
    d, B, N = args.dim, args.batch, 10
    out = []
    for name, truncated in (("neg=10 uniform", False), ("neg=10 truncated k=2000 (reference default)", True)):
        E = EmbeddingTable(kgs.entities_num, d, "rv_ent_embeds", values=ent0)
        R = EmbeddingTable(kgs.relations_num, d, "rel_embeds", values=rel0)
        vs = [KGSide(kgs.entities(k), sides[k].known) for k in (0, 1)]
        knn_ms = None
        if truncated:
            knn_ms = []
            for _ in range(2):          # first call: cold (allocations, first launches of these kernels); second: warm
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in (0, 1):
                    ids = torch.as_tensor(kgs.entities(k), device="cuda")
                    kk = int((1 - 0.98) * len(ids))
                    tab, valid = neighbour_table(E.lookup(ids.to(torch.int32)), kgs.entities(k), kk, kgs.entities_num)
                    vs[k].set_neighbours(tab, valid)
                torch.cuda.synchronize()
                knn_ms.append((time.perf_counter() - t0) * 1e3)
        bat = RelationBatcher(kgs.triples[0], kgs.triples[1], vs[0], vs[1], B, N, seed=1234)
        runner = RelationViewRunner(E, R, bat, "relation", lr=0.001)
        runner.run()                                   # warm-up epoch
        torch.cuda.synchronize()
        n_ep = max(1, min(args.steps, 368) // bat.steps)
        t0 = time.perf_counter()
        for e in range(n_ep):
            bat.shuffle()
            runner.run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        scored = n_ep * int(bat.off[-1]) * (1 + N)
        row = {"name": name, "value": scored / dt, "unit": "triples/s", "steps": n_ep * bat.steps,
               "ms_per_step": dt / (n_ep * bat.steps) * 1e3, "scored_per_step": B * (1 + N)}
        if knn_ms is not None:
            row["knn_refresh_ms_untimed"] = {"cold_first_call": knn_ms[0], "warm_second_call": knn_ms[1],
                                             "what": "both KGs (2 x 100K entities, k = 2000), outside the timed region"}
        out.append(row)
        del runner, bat, E, R, vs
    out.append(attribute_step_variant(args))
    out.append(pytorch_port_variant(args, kgs, ent0, rel0, sides))
    return out


def pytorch_port_variant(args, kgs, ent0, rel0, sides):
    """Side line: the headline step as a straight PyTorch program on the SAME GPU — what a port of code/MultiKE_model.py:114-132
    costs when every op is an ATen kernel: tf.nn.l2_normalize of both tables, six embedding_lookups, the logistic loss,
    autograd, and Adagrad over the whole variables (the reference's dense semantics).  Negatives come from this package's
    device sampler and are NOT inside the timed region (the reference's sampler runs in CPU worker processes)."""
    from multike_amd.sampling import KGSide, RelationBatcher
    d, B, N = args.dim, args.batch, args.neg
    E = torch.nn.Parameter(torch.as_tensor(ent0, dtype=torch.float32, device="cuda").clone())
    R = torch.nn.Parameter(torch.as_tensor(rel0, dtype=torch.float32, device="cuda").clone())
    opt = torch.optim.Adagrad([E, R], lr=0.001, initial_accumulator_value=0.1, eps=0.0)
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), sides[0].known), KGSide(kgs.entities(1), sides[1].known),
                          B, N, seed=1234)
    n_steps = min(24, bat.steps)
    batches = []
    for s_ in range(n_steps):
        pos, neg = bat.batch(s_)
        batches.append(tuple(x.long() for x in pos) + tuple(x.long() for x in neg))

    def step(b):
        opt.zero_grad(set_to_none=True)
        En, Rn = torch.nn.functional.normalize(E, dim=1), torch.nn.functional.normalize(R, dim=1)
        ph, pr, pt, nh, nr, nt = b
        x = ((En[ph] + Rn[pr] - En[pt]) ** 2).sum(1)
        y = ((En[nh] + Rn[nr] - En[nt]) ** 2).sum(1)
        loss = torch.nn.functional.softplus(x).sum() + torch.nn.functional.softplus(-y).sum()
        loss.backward()
        opt.step()
    for b in batches[:4]:
        step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[4:]:
        step(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(1, n_steps - 4)
    return {"name": "the same step as a straight PyTorch program on this GPU (ATen kernels, autograd, dense Adagrad; sampler excluded)",
            "value": B * (1 + N) / dt, "unit": "triples/s", "steps": n_steps - 4, "ms_per_step": dt * 1e3, "scored_per_step": B * (1 + N)}


def attribute_step_variant(args):
    """Side line: one attribute-view step (code/MultiKE_model.py:134-151 + conv :34-63) at the same batch size — the other
    half of an ITC epoch's GPU time.  Bytes per triple (algorithmic): 3 row gathers (entity, attribute, literal) + 2 gradient
    rows (entity, attribute) + 4 ids/weight = 5 * 4 * dim + 16; flops per triple: 2 * (4 dim * dim dense + 7.2K conv MACs)
    forward, ~3x that with the backward.  The step is 4 dependent launches of 6-23 us at this size (6 until round 4): it is bound by
    kernel floors and latency, not by HBM or the matrix pipe — the roofline block says how far below both it sits."""
    from multike_amd.attr_cnn import AttrCNN
    from multike_amd.tables import EmbeddingTable, StepEngine
    d, B = args.dim, args.batch
    n_ent, n_attr, n_lit = args.n_ent, 600, 100_000
    E = EmbeddingTable(n_ent, d, "av_ent_embeds", seed=1)
    from multike_amd.MultiKE_model import ATTR_GRAD_COPIES
    A = EmbeddingTable(n_attr, d, "attr_embeds", normalize=False, seed=2, grad_copies=ATTR_GRAD_COPIES)   # as the model builds it
    lit = torch.nn.functional.normalize(torch.randn(n_lit, d, generator=torch.Generator().manual_seed(0)), dim=1).numpy()
    L = EmbeddingTable(n_lit, d, "literal_embeds", normalize=False, trainable=False, values=lit)
    cnn, eng = AttrCNN(d, seed=3), StepEngine()
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    n_steps = 120
    ih = torch.randint(0, n_ent, (n_steps * B,), device="cuda", generator=g, dtype=torch.int32)
    ia = torch.randint(0, n_attr, (n_steps * B,), device="cuda", generator=g, dtype=torch.int32)
    iv = torch.randint(0, n_lit, (n_steps * B,), device="cuda", generator=g, dtype=torch.int32)
    w = torch.rand(n_steps * B, device="cuda", generator=g)
    off = np.arange(n_steps + 1, dtype=np.int64) * B
    cnn.steps(eng, E, A, L, ih, ia, iv, w, off[:11])          # warm-up: 10 steps, one native call
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnn.steps(eng, E, A, L, ih, ia, iv, w, off)               # the epoch loop as ONE native call (mke_attr_steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    bytes_per = 5 * 4 * d + 16
    flops_per = 3 * 2 * (4 * d * d + 7200)
    return {"name": "attribute-view step (CNN scorer, fwd + bwd + updates)", "value": B / dt, "unit": "attribute triples/s",
            "steps": n_steps, "ms_per_step": dt * 1e3, "scored_per_step": B,
            "roofline": {"bound": "launch floors / latency (4 dependent launches per step)",
                         "alg_bytes_per_triple": bytes_per, "achieved_GBps": B * bytes_per / dt / 1e9,
                         "frac_hbm": B * bytes_per / dt / 1e9 / HBM_PEAK_GBS,
                         "flops_per_triple": flops_per, "achieved_TFLOPs": B * flops_per / dt / 1e12,
                         "frac_f32_matrix_peak": B * flops_per / dt / 1e12 / 157.3}}


def host_threads():
    """Threads the cpu_baseline legs may use: the cores this process may run on, capped by the cgroup's CPU quota (a
    container that sees 256 cores but is entitled to 32 of them runs a 256-thread OpenMP loop slower than one thread)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def pmc_traffic(config, custom=False):
    """HBM-side bytes per launch of k_triple_score from the committed rocprofv3 PMC passes (tools/pmc_passes.sh ->
    profiles/r0N_pmc_<config>.json), quoted only when the file was collected on THIS build of the kernels (source hash)
    and on this workload; None otherwise."""
    if custom:
        return None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):          # newest round first; a file is quoted only for the build it was collected on
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_{config}.json")) as f:
                pmc = json.load(f)
            if pmc.get("kernel_source_sha") == kernel_source_hash():
                return int(pmc["traffic_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            continue
    return None


def _windows(w, first, steps, n_win):
    """n_win windows of `steps` steps each (every one inside an epoch), the bracket of FusedWorkload.timed"""
    dts, scored = [], []
    i = first
    for _ in range(n_win):
        i = w.window_start(i, steps)
        dts.append(w.timed(i, steps))
        scored.append(sum(w.triples_of(k) for k in range(i, i + steps)))
        i += steps
    dts, scored = np.array(dts), np.array(scored)
    mid = int(np.argsort(dts)[len(dts) // 2])
    stats = {"n": int(n_win), "steps_per_window": int(steps), "min": float(dts.min() * 1e3), "p10": float(np.percentile(dts, 10) * 1e3),
             "median": float(dts[mid] * 1e3), "p90": float(np.percentile(dts, 90) * 1e3), "max": float(dts.max() * 1e3)}
    return float(dts[mid]), int(scored[mid]), stats, i


def launch_histogram(ms, width_us=None):
    """Histogram of the per-launch durations of the instrumented pass (HIP events), in us."""
    us = np.asarray(ms) * 1e3
    if width_us is None:
        width_us = max(0.5, round(float(np.ptp(us)) / 12, 1)) if len(us) > 1 else 1.0
    lo = np.floor(us.min() / width_us) * width_us
    edges = lo + width_us * np.arange(int(np.ceil((us.max() - lo) / width_us)) + 2)
    h, _ = np.histogram(us, bins=edges)
    return {"bin_us": float(width_us), "first_edge_us": float(lo), "counts": [int(x) for x in h[:max(1, int(np.nonzero(h)[0].max()) + 1)]],
            "min_us": float(us.min()), "p10_us": float(np.percentile(us, 10)), "median_us": float(np.median(us)),
            "p90_us": float(np.percentile(us, 90)), "max_us": float(us.max())}


def hbm_resident_variant(args):
    """Side line at BASELINE.json configs[4]'s per-GPU shape (C5-synth: |E| 2M, |R| 2K, dim 256, neg 64, batch 5000): the 2 GB
    table + slot + gradient scratch do not fit the 256 MB Infinity Cache, so this — not the C2 headline, whose 192 MB working
    set is cache-resident — is the HBM measurement of the same kernels.  Same code path as the headline: native
    step loop, windows of `steps` steps after `warmup` (median window reported, spread beside it), then the instrumented pass
    for the kernel's own duration and its launch-time histogram."""
    cfg = {k: CONFIGS["c5"][k] for k in ("n_ent", "n_rel", "dim", "neg", "batch")}
    from multike_amd.tables import PLACEMENT_LOG
    n_placed = len(PLACEMENT_LOG)
    t_setup = time.perf_counter()
    w = FusedWorkload(cfg, device_init=True)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    warm, steps = 200, 100
    w.run_steps(0, warm)
    dt, scored, stats, nxt = _windows(w, warm, steps, 8)
    roof = w.instrumented(nxt, 200, pmc_traffic("c5"), dt / steps * 1e6)
    roof["launch_histogram"] = w.last_hist
    return {"name": "C5-synth: the HBM-resident shape (BASELINE configs[4] per GPU: |E|=2M |R|=2K dim=256 neg=64 batch=5000)",
            "value": scored / dt, "unit": "triples/s", "steps": steps, "warmup": warm, "ms_per_step": dt / steps * 1e3,
            "window_ms": stats, "scored_per_step": cfg["batch"] * (1 + cfg["neg"]), "setup_s_untimed": t_setup, "roofline": roof,
            # placement by trial of the three 2 GB arrays (multike_amd/tables.py placed_rows): per array, the probe time of every
            # candidate allocation tried and which one was kept — HBM allocations come in two classes ~9 % apart on this probe
            "placement": PLACEMENT_LOG[n_placed:]}


def zipf_variant(args, exponent=1.0):
    """Side line (SURVEY 8d "a Zipf(1.0) head/tail variant to expose atomic contention"): the headline shape with the head and
    tail entities of the synthetic triples drawn ~ rank^-1 inside each KG (hub rows: the positives' shared-row flushes of many
    groups of a step land on the same rows; the sampler's corruptions stay uniform, as code/base/batch.py:86-116 draws them).
    Same code path, own roofline object; `vs_uniform_launch` = this kernel's launch time / the uniform headline's is filled in
    by the caller."""
    cfg = {k: CONFIGS["c2"][k] for k in ("n_ent", "n_rel", "dim", "neg", "batch")}
    w = FusedWorkload(cfg, zipf=exponent)
    warm, steps = w.n_steps_epoch, 20
    w.run_steps(0, warm + 5)
    dt, scored, stats, nxt = _windows(w, warm + 5, steps, 50)
    roof = w.instrumented(nxt, 100, None, dt / steps * 1e6)
    roof["launch_histogram"] = w.last_hist
    deg = np.bincount(np.concatenate([np.concatenate([t[:, 0], t[:, 2]]) for t in w.kgs.triples]), minlength=cfg["n_ent"])
    return {"name": f"C2-synth Zipf({exponent:g}): head / tail entities of the triples ~ rank^-{exponent:g} (hub rows)",
            "value": scored / dt, "unit": "triples/s", "steps": steps, "warmup": warm + 5, "ms_per_step": dt / steps * 1e3,
            "window_ms": stats, "scored_per_step": cfg["batch"] * (1 + cfg["neg"]), "roofline": roof, "hub_rows": w.hub_rows,
            "degree": {"max": int(deg.max()), "p99.9": float(np.percentile(deg, 99.9)), "mean": float(deg.mean()),
                       "share_of_references_on_top_100_rows": float(np.sort(deg)[-100:].sum() / deg.sum())}}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, the way the driver's
    documented line does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`),
    on a free port of the loopback interface; the ranks re-enter main() with WORLD_SIZE set."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def device_report(dist, staged, world, rank, local_rank):
    """What the process group really is — read back from torch.distributed, not from the command line: backend, world size
    and the device of every rank (index, name, PCI bus id), gathered over the group itself."""
    p = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": rank, "device": local_rank, "name": p.name,
            "pci_bus_id": getattr(p, "pci_bus_id", None), "hbm_GB": round(p.total_memory / 2 ** 30, 1)}
    if dist is None or not dist.is_initialized():
        return {"world": 1, "backend": None, "devices": [mine], "note": "single process: no process group at N=1"}
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, mine)
    return {"world": dist.get_world_size(), "backend": dist.get_backend() + (" (host-staged dry run: ranks share GPUs)" if staged else
                                                                             " (= RCCL on ROCm)" if dist.get_backend() == "nccl" else ""),
            "devices": every, "distinct_devices": len({(d["device"], d["pci_bus_id"]) for d in every})}


def roofline_object(kernel, ms, triples, dim, traffic):
    """One self-consistent object: `achieved` and `frac` are on the ALGORITHMIC basis (12 + 24 dim bytes per scored
    triple x the triples of a launch / the launch's duration by HIP events; frac = achieved / peak).  The kernel loads a
    positive's rows once for its N negatives, so it moves FEWER bytes than that model counts: the bytes the memory side really
    moved (`traffic`, rocprofv3 PMC passes on file for this build of the kernels, per launch) give the counter-based pair
    `achieved_counter` / `frac_counter` beside it."""
    ms, triples = np.asarray(ms), np.asarray(triples)
    avg_ms = float(ms.mean())
    alg = float((triples * b_alg(dim)).sum() / (ms.sum() * 1e-3) / 1e9)
    ach_counter = None if traffic is None else traffic / (avg_ms * 1e-3) / 1e9
    # Headline pair (`achieved`, `frac`): the PHYSICAL one — bytes the memory side moved (PMC passes on file for this build of the
    # kernels) / launch duration / peak — whenever it is on file (round-5 review: the algorithmic model counts a positive's rows
    # once per negative, the kernel loads them once per group, so the model's rate is not a fraction of anything and exceeded 1
    # at the C5 shape).  The algorithmic pair stays beside it under its own names; without counters on file the headline falls
    # back to it, capped at 1 and labelled.
    if ach_counter is not None:
        achieved, basis = ach_counter, "counter: HBM-side bytes per launch (rocprofv3 PMC, this build) / launch duration / peak"
    else:
        achieved, basis = min(alg, HBM_PEAK_GBS), "algorithmic bytes / launch duration / peak (no counter pass on file for this build; capped at the peak)"
    return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "frac_basis": basis,
            "achieved_algorithmic": alg, "frac_algorithmic": alg / HBM_PEAK_GBS,
            "traffic": traffic, "achieved_counter": ach_counter,
            "frac_counter": None if ach_counter is None else ach_counter / HBM_PEAK_GBS,
            "avg_launch_us": avg_ms * 1e3, "median_launch_us": float(np.median(ms)) * 1e3, "launches_timed": int(len(ms)),
            "alg_bytes_per_triple": b_alg(dim), "triples_per_launch": float(triples.mean()),
            "streaming_copy_GBps": HBM_ACHIEVABLE_GBS,      # what a float4 copy reaches on this part (same guide)
            "kernel_source_sha": kernel_source_hash()}


class FusedWorkload:
    """The single-GPU form of the step on one synthetic shape: tables, batcher, native runner, and the two measurements —
    the timed region (native step loop, no Python between steps) and the instrumented pass (HIP events per launch)."""

    def __init__(self, cfg, sample_chunk=None, rel_grad_copies=8, device_init=False, zipf=0.0, rel_zipf=0.0):
        from multike_amd.runner import RelationViewRunner
        from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
        from multike_amd.synthetic import SyntheticKGs
        from multike_amd.tables import EmbeddingTable, StepEngine
        from multike_amd.tables import xavier_truncated_normal  # the product's initialiser (TF1 xavier)
        self.cfg = cfg
        d, N, B = cfg["dim"], cfg["neg"], cfg["batch"]
        self.d, self.N, self.B = d, N, B
        self.kgs = kgs = SyntheticKGs(n_ent=cfg["n_ent"], n_rel=cfg["n_rel"], seed=1234, zipf=zipf, rel_zipf=rel_zipf)
        if device_init:      # side lines of big shapes: the same distribution drawn on the device (9 s on the host at 2M x 256)
            g = torch.Generator(device="cuda"); g.manual_seed(1234)
            def init(n):
                x = torch.empty(n, d, dtype=torch.float32, device="cuda")
                torch.nn.init.trunc_normal_(x, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=g)
                return x * float(np.sqrt(2.6 / (n + d)))
            self.ent0 = self.rel0 = None
            self.E = EmbeddingTable(kgs.entities_num, d, "rv_ent_embeds", trainable=False)
            self.E.trainable = True
            self.E.data[:, :d] = init(kgs.entities_num)
            self.R = EmbeddingTable(kgs.relations_num, d, "rel_embeds", trainable=False, grad_copies=rel_grad_copies)
            self.R.trainable = True
            self.R.data[:, :d] = init(kgs.relations_num)
        else:
            self.ent0 = xavier_truncated_normal(kgs.entities_num, d, "cpu", seed=1234).numpy()
            self.rel0 = xavier_truncated_normal(kgs.relations_num, d, "cpu", seed=1235).numpy()
            self.E = EmbeddingTable(kgs.entities_num, d, "rv_ent_embeds", values=self.ent0)
            self.R = EmbeddingTable(kgs.relations_num, d, "rel_embeds", values=self.rel0, grad_copies=rel_grad_copies)
        self.sides = []
        for k in (0, 1):
            t = torch.as_tensor(kgs.triples[k], device="cuda")
            self.sides.append(KGSide(kgs.entities(k),
                                     KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
        self.bat = RelationBatcher(kgs.triples[0], kgs.triples[1], self.sides[0], self.sides[1], B, N, seed=1234)
        self.bat.shuffle()  # epoch-boundary code path (randperm + regather) exercised once before the timed region
        self.runner = RelationViewRunner(self.E, self.R, self.bat, "relation", lr=0.001, sample_chunk=sample_chunk or None,
                                         hot_rows=None if os.environ.get("MKE_BENCH_HOT", "1") != "0" else False)   # A/B knob
        self.hub_rows = {"n": self.E.n_hot, "copies": self.E.hot_copies}
        self.eng = StepEngine()
        self.n_steps_epoch = self.bat.steps
        self.ev = []
        self._epoch_ready = 0          # index of the epoch whose shuffle has been done

    def _begin_epoch(self, ep):
        """the epoch-boundary shuffle of epoch `ep` (code/MultiKE_model.py:314-315), once"""
        if ep > self._epoch_ready:
            self.bat.shuffle()
            self._epoch_ready = ep

    def run_steps(self, i0, i1):
        """global step indices [i0, i1): whole epochs go through runner.run_epochs; a partial epoch is ONE call into the
        native runner."""
        bat, runner, n = self.bat, self.runner, self.n_steps_epoch
        i = i0
        while i < i1:
            s = i % n
            if s == 0 and i1 - i >= n:
                n_ep = (i1 - i) // n
                self._begin_epoch(i // n)
                runner.run_epochs(n_ep)       # shuffles between its epochs itself
                self._epoch_ready = i // n + n_ep - 1
                i += n_ep * n
                continue
            e = min(n, s + (i1 - i))
            if s == 0:
                self._begin_epoch(i // n)
            runner.run(s, e)
            i += e - s

    def window_start(self, i, steps):
        """First global step index >= i at which a window of `steps` steps lies inside one epoch; the steps skipped are RUN
        (untimed) and the next epoch's shuffle is done here, outside the window's bracket.  Windows longer than an epoch are
        left where they are."""
        n = self.n_steps_epoch
        if steps >= n or (i % n) + steps <= n:
            return i
        nxt = (i // n + 1) * n
        self.run_steps(i, nxt)
        self._begin_epoch(nxt // n)
        return nxt

    def triples_of(self, i):
        s = i % self.n_steps_epoch
        return int(self.bat.off[s + 1] - self.bat.off[s]) * (1 + self.N)

    def timed(self, first, steps, barrier=lambda: None):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.run_steps(first, first + steps)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def _step_timed(self, i):
        """Python-driven step with HIP events at every launch boundary (same stream): count | score | update."""
        from multike_amd import _lib
        E, R, d, N = self.E, self.R, self.d, self.N
        s = i % self.n_steps_epoch
        pos, neg = self.bat.batch(s)
        tag, lp = self.eng._next()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        _lib.count_entity_refs(pos[0], pos[2], neg[0], neg[2], N, E.refcount)
        e[1].record()
        hot = E.hot_struct()           # hub rows declared by the runner (none on the uniform synthetic KGs)
        _lib.triple_score_fwd_bwd_x(E.data, True, R.data, True, d, pos, None, neg, None, N, 1.0, E.grad, R.grad,
                                    E.touched, R.touched, tag, E.refcount, E.slot("relation"), _lib.OPT_ADAGRAD, 0.001, lp, hot=hot)
        e[2].record()
        _lib.rows_update_multi([(R.data, R.slot("relation"), R.grad, R.touched, True),
                                (E.data, E.slot("relation"), E.grad, E.touched, True, E.refcount, hot)], tag, E.stride, d,
                               _lib.OPT_ADAGRAD, 0.001)
        e[3].record()
        self.ev.append((e, pos[0].numel() * (1 + N), tag))

    def instrumented(self, base, n_inst, traffic, step_wall_us):
        """HIP events bracket every launch of the step.  This pass is driven from Python (one ctypes call per launch), i.e. the
        host is slower than the GPU; to keep host latency out of the event pairs the stream is first blocked by a spin kernel
        long enough for every instrumented step to be queued behind it, so that the GPU then runs them back to back."""
        from multike_amd.sampling import sample_negatives
        E, R, bat, runner, N = self.E, self.R, self.bat, self.runner, self.N
        t_h = time.perf_counter()
        for i in range(base, base + 5):
            self._step_timed(i)
        torch.cuda.synchronize()
        host_per_step = (time.perf_counter() - t_h) / 5
        self.ev.clear()
        torch.cuda._sleep(int(2.4e9 * (host_per_step * n_inst * 1.5 + 0.02)))
        for i in range(base + 5, base + 5 + n_inst):
            self._step_timed(i)
        torch.cuda.synchronize()
        cnt = np.array([e[0].elapsed_time(e[1]) for e, _, _ in self.ev])
        ms = np.array([e[1].elapsed_time(e[2]) for e, _, _ in self.ev])
        ums = np.array([e[2].elapsed_time(e[3]) for e, _, _ in self.ev])
        whole = np.array([e[0].elapsed_time(e[3]) for e, _, _ in self.ev])
        tr = np.array([n for _, n, _ in self.ev])
        roofline = roofline_object("k_triple_score", ms, tr, self.d, traffic)
        self.last_hist = launch_histogram(ms)
        # second kernel of the step, reported beside it: rows left to it (referenced more than once in the step) x 6 row
        # streams (grad, w, acc read; 0, w, acc written); rows referenced once were updated inside k_triple_score
        last_tag = self.ev[-1][2]
        touched_rows = int((E.touched == last_tag).sum()) + int((R.touched == last_tag).sum())
        upd_bytes = touched_rows * 6 * E.stride * 4
        roofline["update_kernel"] = {"kernel": "k_rows_update_multi", "avg_launch_us": float(ums.mean()) * 1e3,
                                     "touched_rows_last_step": touched_rows, "bytes_per_launch": upd_bytes,
                                     "achieved": upd_bytes / (float(ums.mean()) * 1e-3) / 1e9, "unit": "GB/s"}
        # the epoch sampler (one launch per epoch in the native loop), timed here with events and amortised over its steps
        total_neg = int(bat.off[-1]) * N
        samp_us = None
        if N and runner.neg[0].numel() >= total_neg:
            se0, se1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            se0.record()
            sample_negatives((bat.pos_h, bat.pos_r, bat.pos_t), bat.side1, N, seed=bat.rng_seed, stream_id=bat.rng_stream,
                             pos_offset=0, out=tuple(x[:total_neg] for x in runner.neg), side1=bat.side2, pos_kg=bat.pos_kg)
            se1.record()
            torch.cuda.synchronize()
            samp_us = se0.elapsed_time(se1) * 1e3 / self.n_steps_epoch
        # Two different loops, kept apart: the NATIVE loop (what `value` times: no events, the next step's reference counts
        # ride inside the score launch) and the INSTRUMENTED loop (three launches per step, an event record at every boundary:
        # each bracket carries its boundary and ~1-2 us of event overhead).  Parts of the second do not add up to the first,
        # so no difference of the two is printed; what the native step spends outside its two kernels is bounded from the
        # rocprofv3 kernel times (profiles/r04_gap_table_c2.md), not from these events.
        roofline["step_breakdown_us"] = {
            "native_loop": {"step_wall": step_wall_us, "sampler_amortised": samp_us},
            "instrumented_loop": {"count_kernel_incl_boundary": float(cnt.mean()) * 1e3,
                                  "score_kernel": roofline["avg_launch_us"],
                                  "update_kernel_incl_its_boundary": float(ums.mean()) * 1e3,
                                  "step_first_to_last_event": float(whole.mean()) * 1e3},
            "step_wall": step_wall_us, "score_kernel": roofline["avg_launch_us"],
            "update_kernel_incl_its_boundary": float(ums.mean()) * 1e3, "sampler_amortised": samp_us}
        return roofline


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (multike_amd has no CPU path)"
    # MKE_BENCH_COMM=staged: dry run of the multi-rank flow on fewer GPUs than ranks (ranks share devices, collectives go
    # through gloo staged over the host).  Exercises launch / barrier / max-over-ranks / rank-0 output; its number is NOT a
    # benchmark result and is labelled as such.
    staged = os.environ.get("MKE_BENCH_COMM", "") == "staged"
    if staged:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 and staged:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    elif world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a wedged collective must end the run with an error, not hold the node until the driver's limit
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=240))

    from multike_amd.utils import touch_library_kernels
    touch_library_kernels(torch.device("cuda", local_rank))      # what the model's constructor does (code-object loads of the library families)
    d, N, B = args.dim, args.neg, args.batch
    cfg = dict(n_ent=args.n_ent, n_rel=args.n_rel, dim=d, neg=N, batch=B)
    sharded = world > 1 or args.force_sharded
    rccl = None
    fused = None

    if sharded:
        from multike_amd.synthetic import SyntheticKGs
        from multike_amd.tables import xavier_truncated_normal  # the product's initialiser (TF1 xavier)
        kgs = SyntheticKGs(n_ent=args.n_ent, n_rel=args.n_rel, seed=1234)
        ent0 = xavier_truncated_normal(kgs.entities_num, d, "cpu", seed=1234).numpy()
        rel0 = xavier_truncated_normal(kgs.relations_num, d, "cpu", seed=1235).numpy()
        if dist is None:  # exercise the sharded path on one GPU (1-rank RCCL group)
            import torch.distributed as dist
            import tempfile
            dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="mke_rdv_"), rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))   # a rendezvous file: no TCP port to collide on
        # MKE_SHARD_MODE=oc (default): owner-computes step (multike_amd/distributed_oc.py: the negatives go to the rows, ~1
        # vector per positive crosses the links); =rowfetch: round 1's row-exchange step (multike_amd/distributed.py)
        shard_mode = os.environ.get("MKE_SHARD_MODE", "oc")
        if shard_mode == "rowfetch":
            from multike_amd.distributed import HostStagedComm, ShardedRelationTrainer
            trainer = ShardedRelationTrainer(kgs, ent0, rel0, B, N, rank, world, seed=1234, comm=HostStagedComm() if staged else None)
            trainer.bat.shuffle()
        else:
            from multike_amd.distributed_oc import OcHostStagedComm, OwnerComputesTrainer
            # split-batch pipelining pays when a collective's wire time is well above what an asynchronous collective
            # costs on its own (tools/rccl_latency.py at world 1: ~30 us of device time per async call + wait against
            # ~15 us in line; 27 us against 13 us on the host): two parts when a rank's all-gather moves >= 32 MB
            # (the c5 shape at 8 ranks: 40 MB now that one vector per positive travels), one otherwise (c2: 12.4 MB)
            # Round 6: with the step loop native (mke_oc_steps) the host can feed two streams, but the split still loses on the
            # DEVICE timeline — four stream hops per part at ~13 us each and two half-size score launches — at both shapes
            # (tools/oc_rank_compute.py, profiles/r06_oc_native.log): one part per step unless MKE_SHARD_CHUNKS says otherwise
            chunks = int(os.environ.get("MKE_SHARD_CHUNKS", "1"))
            # MKE_SHARD_PEER=1: peer-mapped blocks read / written directly by the score kernel instead of the all-gather /
            # reduce-scatter (opt-in: exercised with two ranks on one GPU only)
            trainer = OwnerComputesTrainer(kgs, ent0, rel0, B, N, rank, world, seed=1234, chunks=chunks,
                                           comm=OcHostStagedComm() if staged else None,
                                           peer_direct=os.environ.get("MKE_SHARD_PEER", "0") == "1")
        run_step = trainer.step
        n_steps_epoch = trainer.steps
        triples_of = trainer.global_scored

        def run_steps(i0, i1):
            if hasattr(trainer, "run"):              # owner-computes: one native call per run of steps inside an epoch (mke_oc_steps)
                trainer.run(i0, i1 - i0)
                return
            for i in range(i0, i1):
                run_step(i)
    else:
        fused = FusedWorkload(cfg, sample_chunk=args.sample_chunk, rel_grad_copies=args.rel_grad_copies, zipf=args.zipf, rel_zipf=args.rel_zipf)
        kgs, ent0, rel0 = fused.kgs, fused.ent0, fused.rel0
        n_steps_epoch = fused.n_steps_epoch
        run_steps, triples_of = fused.run_steps, fused.triples_of
    rccl = device_report(dist, staged, world, rank, local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()

    # pre-warm: the driver's invocation is the first command on a fresh box (cold clocks, cold caches, first launch of every
    # kernel).  Whole untimed epochs of the same work first, then the --warmup steps, then the timed region.
    pre = max(0, args.prewarm_epochs) * n_steps_epoch
    if pre:
        run_steps(0, pre)
        torch.cuda.synchronize()
    args_w0 = pre                                   # global step index of the first --warmup step
    run_steps(args_w0, args_w0 + args.warmup)

    def time_window(first):
        """exactly --steps steps between barrier + synchronize pairs; max over ranks"""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(first, first + args.steps)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt_], dtype=torch.float64, device="cpu" if staged else "cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax)
        return dt_

    # The contract's timed region (W warm-up steps, then exactly K steps) is the FIRST window.  It is then repeated, same
    # bracket, same K, so that the line carries a spread: `value` / `ms_per_step` are the MEDIAN window (first window beside it).
    first = args_w0 + args.warmup
    if fused is not None:
        first = fused.window_start(first, args.steps)
    dt_first = time_window(first)
    n_win = args.windows
    if n_win is None:
        n_win = max(50 if args.steps <= 100 else 5, int(np.ceil(0.060 / max(dt_first, 1e-6))))
        n_win = min(n_win, 400)
    win_dt, win_scored = [dt_first], [sum(triples_of(i) for i in range(first, first + args.steps))]
    nxt = first + args.steps
    # the window count is planned from the first window; if that one was an outlier (cold caches, a host hiccup) windows are added
    # until the timed region holds 60 ms (never with an explicit --windows, never beyond 400)
    while len(win_dt) < max(1, n_win) or (args.windows is None and sum(win_dt) < 0.060 and len(win_dt) < 400):
        if fused is not None:
            nxt = fused.window_start(nxt, args.steps)
        win_dt.append(time_window(nxt))
        win_scored.append(sum(triples_of(i) for i in range(nxt, nxt + args.steps)))
        nxt += args.steps
    win_dt, win_scored = np.array(win_dt), np.array(win_scored)
    rate = win_scored / win_dt
    mid = int(np.argsort(win_dt)[len(win_dt) // 2])            # the median window (an actual window, not an interpolation)
    # `value` / `ms_per_step` are the contract's ONE timed window — W warm-up steps, then exactly K steps — i.e. the FIRST window,
    # as in rounds 1-4 (round 5 printed the median of the repeated windows there: round-5 advice); the repeats give the spread
    # and the median beside it
    dt, scored = float(win_dt[0]), int(win_scored[0])
    dt_med, scored_med = float(win_dt[mid]), int(win_scored[mid])
    value = scored / dt
    window_ms = {"n": int(len(win_dt)), "steps_per_window": args.steps, "min": float(win_dt.min() * 1e3),
                 "p10": float(np.percentile(win_dt, 10) * 1e3), "median": float(dt_med * 1e3), "median_value": float(scored_med / dt_med),
                 "median_ms_per_step": float(dt_med / args.steps * 1e3),
                 "p90": float(np.percentile(win_dt, 90) * 1e3), "max": float(win_dt.max() * 1e3),
                 "first_window": float(dt_first * 1e3), "first_window_value": float(win_scored[0] / dt_first),
                 "timed_total_ms": float(win_dt.sum() * 1e3),
                 "value_min": float(rate.min()), "value_max": float(rate.max()),
                 "what": "every window = --steps consecutive steps inside one epoch, bracketed by barrier + synchronize; "
                         "`value` and `ms_per_step` are the FIRST window's (the contract's timed region); the median window beside them"}
    args_end = nxt                                   # first global step index after the timed windows

    roofline = None
    if not sharded:
        base = args_end
        n_inst = max(min(args.steps, 300), 100)      # >= 100 launches whatever --steps is (the driver passes 20)
        # PMC bytes per launch of this kernel (separate rocprofv3 --pmc passes), or None
        roofline = fused.instrumented(base, n_inst, pmc_traffic(args.config, getattr(args, "custom", False) or bool(args.zipf) or bool(args.rel_zipf)), dt / args.steps * 1e6)
        roofline["launch_histogram"] = fused.last_hist

    variants = None
    if not sharded and not args.no_variants and args.config == "c2" and not getattr(args, "custom", False) and not args.zipf and not args.rel_zipf:
        variants = reference_default_variants(args, kgs, ent0, rel0, fused.sides)
        del fused
        torch.cuda.empty_cache()
        zv = zipf_variant(args, 1.0)
        zv["roofline"]["vs_uniform_launch"] = zv["roofline"]["avg_launch_us"] / roofline["avg_launch_us"]
        zv["vs_uniform_step"] = zv["ms_per_step"] / (dt / args.steps * 1e3)
        variants.insert(0, zv)
        torch.cuda.empty_cache()
        variants.insert(0, hbm_resident_variant(args))

    if sharded:
        # same instrumentation on the sharded path: events around this rank's score-kernel launches (extra steps)
        trainer.score_events = []
        base = args_end
        run_steps(base, base + min(args.steps, 50))
        torch.cuda.synchronize()
        ms = np.array([a.elapsed_time(b) for a, b, _ in trainer.score_events])
        tr = np.array([n for _, _, n in trainer.score_events])
        trainer.score_events = None
        # host cost of the step path: the enqueue loop timed WITHOUT waiting for the device (Python + ctypes + torch.distributed
        # calls per global step); above the device time per step the host, not the GPU, paces the job
        n_host = min(40, n_steps_epoch - 1)
        torch.cuda.synchronize()
        th = time.perf_counter()
        base_h = base + min(args.steps, 50)            # steps are issued in order
        run_steps(base_h, base_h + n_host)
        th = time.perf_counter() - th
        torch.cuda.synchronize()
        host_us = th / max(1, n_host) * 1e6
        shard_info = trainer.check() if hasattr(trainer, "check") else {}   # raises when a row set overflowed (results invalid)
        # the replicated relation table must be bit-identical on every rank (its gradient is all-reduced, the update identical);
        # a transport that delivered different sums to different ranks shows up here, and the epoch's loss must be finite
        rel_t = getattr(trainer, "rel", None)
        if rel_t is not None and world > 1:
            ref_t = rel_t.clone()
            if staged:
                c = ref_t.cpu(); dist.broadcast(c, 0); ref_t.copy_(c)
            else:
                dist.broadcast(ref_t, 0)
            same = torch.tensor([1 if torch.equal(ref_t, rel_t) else 0], dtype=torch.int32, device="cpu" if staged else "cuda")
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if int(same) != 1:
                raise SystemExit("bench.py: the replicated relation table differs between ranks: collectives are not trustworthy "
                                 "(run tools/multi_gpu_selftest.py)")
            shard_info = dict(shard_info or {}, replicas_bit_identical=True)
        ep_loss = trainer.epoch_loss()
        if not np.isfinite(ep_loss):
            raise SystemExit(f"bench.py: non-finite loss {ep_loss} on the sharded path")
        roofline = roofline_object("k_oc_score" if shard_mode != "rowfetch" else "k_triple_score", ms, tr, d, None)
        roofline.update({"scope": "per GPU (rank 0): the triples whose corrupt entity this rank owns", "exchange": shard_info,
                         "host_us_per_step": host_us, "host_note": "enqueue loop of the step path without waiting for the device"
                                             + ("; MKE_OC_FORCE_COLLECTIVES=1: the G > 1 path with its collectives on the one-rank group" if getattr(trainer, "force_collectives", False) else ""),
                         "basis_note": "the algorithmic model counts 3 rows read + 3 written per scored triple; the owner-computes kernel "
                                       "reads ~1 vector per POSITIVE and one row per negative, so this fraction overstates its traffic "
                                       "(it can exceed 1) — a rate in the model's bytes, not a measurement of the memory system"})

    if rank == 0:
        out = {
            "metric": "scored triples/sec (pos+neg)", "value": value, "unit": "triples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "window_ms": window_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not staged else "synthetic; DRY RUN: ranks share GPUs, collectives staged through the host (not a result)",
            "config": {"workload": f"relation-view train step, {args.label}{f' Zipf({args.zipf:g})' if args.zipf else ''}{f' relation-Zipf({args.rel_zipf:g})' if args.rel_zipf else ''} |E|={args.n_ent} |R|={args.n_rel} dim={d} neg={N} "
                                   f"batch={B}" + (f"/GPU, rows sharded id%{world}" if sharded else ""),
                       "step": "on-device negative sampling + fused gather/score/loss/gradient + Jacobian/Adagrad row update",
                       "n_ent": args.n_ent, "n_rel": args.n_rel, "dim": d, "neg": N, "batch": B,
                       "scored_per_step": B * (1 + N) * world, "steps_per_epoch": n_steps_epoch},
            "roofline": roofline,
            "rccl": rccl,
        }
        if variants is not None:
            out["variants"] = variants
        from multike_amd.tables import PLACEMENT_LOG
        if PLACEMENT_LOG and variants is None:       # arrays of >= 1 GB placed by trial (the C5 variant carries its own log)
            out["placement"] = PLACEMENT_LOG
        if not sharded and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, kgs, ent0, rel0)
        line = json.dumps(out)
    else:
        line = None
    if dist is not None:
        dist.destroy_process_group()
    # RCCL writes its version banner through C stdio: into a pipe that is flushed at exit, i.e. AFTER everything Python
    # printed.  Flush it now, so that the JSON line is the last line of the stream whatever the parser takes.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
