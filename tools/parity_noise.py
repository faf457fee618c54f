#!/usr/bin/env python3
"""Where the whole-schedule parity tolerance comes from (tests/test_schedule_trace_gpu.py): the ITC / SSL schedule of the test
is run on the HIP path (atomic scatter, and again with mke_set_option("deterministic", 1)), the recorded batches are replayed
by the float64 oracle AND by the same oracle in float32 (NumPy's summation order: another correct fp32 implementation), and
for every trainable table the elements outside the tight band (rtol 1e-3, atol 2e-5 of the float64 truth) are listed with the
smallest norm their raw row went through during the schedule — the Jacobian of the normalised view divides by it
(code/base/initializers.py:26).

    python tools/parity_noise.py [ITC|SSL]      (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def replay(oracle, recs, track, snapshots=None, label=""):
    """phase by phase; after every phase the smallest row norm seen so far of every trainable table, and (snapshots = the HIP
    tables after each phase) where the two implementations part: the worst element error after each phase"""
    mins = {k: np.full(oracle.t[k].shape[0], np.inf) for k in track}
    losses = []
    epoch_of = {}
    prev_worst = 0.0
    for n, (phase, rec) in enumerate(recs):
        before = None
        if snapshots is not None and phase == "common":
            import copy
            before = (copy.deepcopy(oracle.t), copy.deepcopy(oracle.acc))
        losses.append(oracle.replay(phase, rec))
        for k in track:
            mins[k] = np.minimum(mins[k], np.linalg.norm(oracle.t[k].astype(np.float64), axis=1))
        if snapshots is not None:
            epoch_of[phase] = epoch_of.get(phase, 0) + 1
            errs = {k: np.abs(snapshots[n][k].astype(np.float64) - oracle.t[k]) for k in track}
            worst = max(track, key=lambda k: errs[k].max())
            r = int(errs[worst].max(1).argmax())
            print(f"    {label} after {phase:10s} #{epoch_of[phase]:2d}: " + "  ".join(f"{k} {errs[k].max():.1e}" for k in track)
                  + f"   worst: {worst} row {r}")
            now = max(e.max() for e in errs.values())
            if before is not None and now > 10 * max(prev_worst, 1e-7):
                explain_common(oracle, before, rec, r)
                for k in ("ent", "rv_ent", "av_ent"):      # the error vector of that row: along the row (immaterial to the
                    w64 = oracle.t[k][r]                   # normalised view every lookup reads) or across it?
                    e = snapshots[n][k][r].astype(np.float64) - w64
                    what = w64 / np.linalg.norm(w64)
                    rad = float(e @ what)
                    tan = float(np.linalg.norm(e - rad * what))
                    line = f"        {k} row {r}: |err| {np.linalg.norm(e):.2e} = radial {rad:+.2e} / tangential {tan:.2e}"
                    ak = "acc_cross_name_" + k
                    if ak in snapshots[n]:
                        a64 = oracle.acc[("cross_name", k)][r]
                        ea = snapshots[n][ak][r].astype(np.float64) - a64
                        line += f";  Adagrad slot: max |err| {np.abs(ea).max():.2e} (slot values {a64.min():.3f} .. {a64.max():.3f})"
                    print(line)
                    print("          err:", np.array2string(e, precision=1, max_line_width=200))
                    print("          w  :", np.array2string(w64, precision=3, max_line_width=200))
            prev_worst = now
    return losses, mins


def explain_common(oracle, before, rec, row):
    """The common-space phase after which the error jumped: replay it step by step from the state before it and print, for the
    row that came out worst, what the step saw — the raw rows' norms (the Jacobian of the normalised view divides by them)."""
    import copy
    from oracle import multike_oracle as mo
    t, acc = copy.deepcopy(before[0]), copy.deepcopy(before[1])
    idx, off = rec["idx"], rec["off"]
    for s_ in range(len(off) - 1):
        ids = np.asarray(idx[int(off[s_]):int(off[s_ + 1])], dtype=np.int64)
        hit = int(np.sum(ids == row))
        nb = {k: float(np.linalg.norm(t[k][row])) for k in ("ent", "rv_ent", "av_ent")}
        old = {k: t[k][row].copy() for k in ("ent", "rv_ent", "av_ent")}
        mo.common_space_step_dense(t["ent"], t["name"], t["rv_ent"], t["av_ent"], acc[("cross_name", "ent")], acc[("cross_name", "rv_ent")],
                                   acc[("cross_name", "av_ent")], ids, oracle.itc_lr, oracle.cv_name_weight, oracle.cv_weight)
        if hit:
            print(f"        step {s_}: row {row} sampled {hit}x; ||w|| before the step: " + "  ".join(f"{k} {v:.3e}" for k, v in nb.items())
                  + ";  after: " + "  ".join(f"{k} {np.linalg.norm(t[k][row]):.3e}" for k in nb)
                  + ";  |step|: " + "  ".join(f"{k} {np.linalg.norm(t[k][row] - old[k]):.3e}" for k in nb))
    small = {k: int(np.sum(np.linalg.norm(before[0][k], axis=1) < 1e-2)) for k in ("ent", "rv_ent", "av_ent")}
    print(f"        rows with ||w|| < 1e-2 before this phase: {small}")
    # How much does this phase AMPLIFY a perturbation of its input?  float64 finite differences: nudge one element of the row by
    # 1e-9 before the phase, replay the phase, look at the same row afterwards.  (An Adagrad step on a row read through
    # l2_normalize has d w_new / d w_old = 1 - lr * W / ||w||^2 * acc / (acc + g^2)^1.5 per element, W = the summed loss
    # weights: with a small accumulator and a near-zero gradient element that is lr W / (||w||^2 sqrt(acc)) >> 1.)
    def run_phase(state):
        t2, a2 = copy.deepcopy(state[0]), copy.deepcopy(state[1])
        for s_ in range(len(off) - 1):
            ids = np.asarray(idx[int(off[s_]):int(off[s_ + 1])], dtype=np.int64)
            mo.common_space_step_dense(t2["ent"], t2["name"], t2["rv_ent"], t2["av_ent"], a2[("cross_name", "ent")], a2[("cross_name", "rv_ent")],
                                       a2[("cross_name", "av_ent")], ids, oracle.itc_lr, oracle.cv_name_weight, oracle.cv_weight)
        return t2
    base = run_phase(before)
    rng = np.random.default_rng(0)
    others = [int(x) for x in rng.choice(before[0]["ent"].shape[0], 200, replace=False)]
    amps = {}
    for r_ in [row] + others:
        worst = 0.0
        for j_ in range(before[0]["ent"].shape[1]) if r_ == row else (int(rng.integers(before[0]["ent"].shape[1])),):
            st = (copy.deepcopy(before[0]), before[1])
            st[0]["ent"][r_, j_] += 1e-9
            out = run_phase(st)
            worst = max(worst, max(float(np.abs(out[k][r_] - base[k][r_]).max()) for k in ("ent", "rv_ent", "av_ent")) / 1e-9)
        amps[r_] = worst
    rest = np.array([amps[r_] for r_ in others])
    print(f"        float64 amplification of a 1e-9 nudge of one ent element over this phase: row {row} (worst element) x{amps[row]:.0f}; "
          f"200 random rows (one random element each): median x{np.median(rest):.2f}, 90th percentile x{np.quantile(rest, 0.9):.1f}, max x{rest.max():.0f}")


def main():
    method = sys.argv[1] if len(sys.argv) > 1 else "ITC"
    import test_schedule_trace_gpu as T
    from multike_amd import _lib
    from oracle.model_oracle import OracleMultiKE
    track = ("rv_ent", "av_ent", "ent", "rel", "attr")
    out = {}
    for mode in ("atomic", "deterministic"):
        _lib.set_option("deterministic", 1 if mode == "deterministic" else 0)
        try:
            snaps = []
            model, oracle, recs, losses, results, data, args = T._run(method, snapshots=snaps)
        finally:
            _lib.set_option("deterministic", 0)
        o32 = OracleMultiKE({k: v for k, v in oracle.t.items()}, oracle.cnn, oracle.M0, learning_rate=args.learning_rate,
                            itc_learning_rate=args.ITC_learning_rate, cv_name_weight=args.cv_name_weight, cv_weight=args.cv_weight,
                            orthogonal_weight=args.orthogonal_weight, dtype=np.float32)
        # how often every entity is referenced by ONE relation-view step (mean over the last epoch's steps): a hub row's
        # gradient is a sum of that many fp32 terms
        deg = np.zeros(oracle.t["rv_ent"].shape[0])
        last = [r for ph, r in recs if ph == "relation"][-1]
        for a in (last["pos"][0], last["pos"][2], last["neg"][0], last["neg"][2]):
            deg += np.bincount(np.asarray(a), minlength=len(deg))
        deg /= max(1, len(last["off"]) - 1)
        l64, mins = replay(oracle, recs, track, snaps if os.environ.get("MKE_NOISE_TRACE") else None, "HIP vs f64")
        l32, _ = replay(o32, recs, track)
        worst = max(abs(g - e) / abs(e) for (_, _, g), e in zip(losses, l64))
        worst32 = max(abs(float(g) - e) / abs(e) for g, e in zip(l32, l64))
        print(f"== {method} / {mode}: worst phase-loss error HIP {worst:.2e}, float32 oracle {worst32:.2e}")
        pairs = {"rv_ent": model.rv_ent_embeds, "av_ent": model.av_ent_embeds, "ent": model.ent_embeds, "rel": model.rel_embeds,
                 "attr": model.attr_embeds}
        for k, tab in pairs.items():
            ref = oracle.t[k]
            got = tab.raw().cpu().numpy().astype(np.float64)
            g32 = o32.t[k].astype(np.float64)
            for name, x in (("HIP", got), ("f32 oracle", g32)):
                err = np.abs(x - ref)
                bad = ~np.isclose(x, ref, rtol=1e-3, atol=2e-5)
                rows = np.unique(np.nonzero(bad)[0])
                print(f"  {k:7s} {name:10s} max |err| {err.max():.2e}  mean {err.mean():.2e}  outside band {bad.mean():.2e} "
                      f"({bad.sum()} elements in {len(rows)} rows)")
                if name == "HIP" and len(rows):
                    order = rows[np.argsort(-err[rows].max(1))][:8]
                    for r in order:
                        print(f"      row {r:5d}: max |err| {err[r].max():.2e}  min ||w|| over the schedule {mins[k][r]:.2e}  "
                              f"final ||w|| {np.linalg.norm(ref[r]):.2e}  f32-oracle |err| on this row {np.abs(g32[r] - ref[r]).max():.2e}"
                              + (f"  references per relation step {deg[r]:.0f} (median row {np.median(deg):.0f})" if len(deg) == ref.shape[0] else ""))
            # errors against the smallest norm the row went through: the amplification the explanation predicts
            e_row = np.abs(got - ref).max(1)
            q = np.quantile(mins[k], [0.0, 0.01, 0.1, 0.5])
            small = mins[k] <= q[1]
            print(f"          rows with min ||w|| in the lowest 1 % (<= {q[1]:.2e}): mean row error {e_row[small].mean():.2e}; the other rows "
                  f"{e_row[~small].mean():.2e}")
        out[mode] = {k: tab.raw().cpu().numpy() for k, tab in pairs.items()}
    for k in track:
        d = np.abs(out["atomic"][k].astype(np.float64) - out["deterministic"][k])
        print(f"atomic vs deterministic run, {k}: max {d.max():.2e} mean {d.mean():.2e}")


if __name__ == "__main__":
    main()
