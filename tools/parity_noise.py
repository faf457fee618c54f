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


def replay(oracle, recs, track):
    """phase by phase; after every phase the smallest row norm seen so far of every trainable table"""
    mins = {k: np.full(oracle.t[k].shape[0], np.inf) for k in track}
    losses = []
    for phase, rec in recs:
        losses.append(oracle.replay(phase, rec))
        for k in track:
            mins[k] = np.minimum(mins[k], np.linalg.norm(oracle.t[k].astype(np.float64), axis=1))
    return losses, mins


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
            model, oracle, recs, losses, results, data, args = T._run(method)
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
        l64, mins = replay(oracle, recs, track)
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
