"""Randomised whole-schedule sweep: random synthetic datasets and hyper-parameters (row width, the three batch sizes — smaller
and LARGER than the data —, negatives per positive, learning rates, view weights, gates, uniform / truncated sampling, ITC /
SSL) run through the product's drivers; the float64 whole-model oracle replays the recorded batches and every phase's epoch
loss must agree to 1e-4 (north_star's tolerance).  python tools/fuzz_model.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_schedule_trace_gpu as T

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    method = "ITC" if rng.random() < 0.6 else "SSL"
    dim = int(rng.choice([8, 20, 24, 33, 50, 64, 75, 88]))
    data_kw = dict(n_ent=2 * int(rng.integers(150, 1500)), n_rel=int(rng.integers(3, 60)), n_attr=int(rng.integers(3, 40)),
                   n_values=int(rng.integers(20, 800)), dim=dim, seed=int(rng.integers(0, 1000)))
    ep = int(rng.integers(3, 7))
    args_kw = dict(dim=dim, batch_size=int(rng.choice([64, 300, 900, 5000, 100000])), attribute_batch_size=int(rng.choice([50, 400, 1500, 100000])),
                   entity_batch_size=int(rng.choice([30, 500, 100000])), neg_triple_num=int(rng.choice([1, 3, 6, 10, 25])),
                   learning_rate=float(rng.choice([0.001, 0.01, 0.03])), ITC_learning_rate=float(rng.choice([0.004, 0.05])),
                   cv_name_weight=float(rng.uniform(0.3, 1.5)), cv_weight=float(rng.uniform(0.5, 2.0)), orthogonal_weight=float(rng.choice([1, 2])),
                   max_epoch=ep, shared_learning_max_epoch=int(rng.integers(1, 5)), start_valid=int(rng.integers(1, ep)),
                   eval_freq=int(rng.integers(1, 4)), start_predicate_soft_alignment=int(rng.integers(0, ep)),
                   truncated_freq=int(rng.integers(1, 4)), truncated_epsilon=float(rng.choice([0.9, 0.98])),
                   neg_sampling=str(rng.choice(["uniform", "truncated"])), seed=int(rng.integers(0, 1000)))
    if args_kw["neg_sampling"] == "truncated" and int((1 - args_kw["truncated_epsilon"]) * (data_kw["n_ent"] // 2)) < args_kw["neg_triple_num"]:
        args_kw["truncated_epsilon"] = 0.9 if int(0.1 * (data_kw["n_ent"] // 2)) >= args_kw["neg_triple_num"] else 0.5   # fewer neighbours than negatives
        # is the reference's random.sample ValueError and the product's (tests/test_native_loops_edge_gpu.py): not a case for this sweep
    desc = f"{method} data={data_kw} args={ {k: v for k, v in args_kw.items() if k not in ('dim',)} }"
    msg = ""
    try:
        model, oracle, recs, losses, results, data, args = T._run(method, data_kw, args_kw)
        worst = 0.0
        if len(recs) != len(losses):
            msg = f"{len(recs)} recorded phases vs {len(losses)} epoch calls"
        for (phase, rec), (p2, epoch, got) in zip(recs, losses):
            exp = oracle.replay(phase, rec)
            if phase != p2:
                msg = f"phase order {phase} vs {p2}"; break
            if exp == 0.0 and got == 0.0:
                continue
            err = abs(got - exp) / max(abs(exp), 1e-12)
            worst = max(worst, err)
            if not np.isfinite(got) or err > 1e-4:
                msg = f"epoch {epoch} phase {phase}: product {got!r} vs oracle {exp!r} (rel {err:.2e})"; break
        if not msg and not all(np.isfinite(v) for v in results.values() if isinstance(v, float)):
            msg = f"results {results}"
    except Exception as ex:  # noqa: BLE001
        import traceback
        msg = f"{type(ex).__name__}: {str(ex)[:300]} @ {traceback.format_exc().strip().splitlines()[-3][:160]}"
    if msg:
        bad += 1
        print(f"MODEL case {c}: {desc}: {msg}", flush=True)
    else:
        print(f"MODEL case {c}: {method} dim={dim} n_ent={data_kw['n_ent']} {len(losses)} phase epochs, worst loss error {worst:.1e}: ok", flush=True)
print(f"whole schedule: {cases - bad} / {cases} runs track the float64 oracle to 1e-4 per phase and epoch")
sys.exit(1 if bad else 0)
