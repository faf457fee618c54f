"""Randomised sweep of the multi-GPU drivers: `ShardedMultiKE_CV / ShardedMultiKE_Late` at 2..4 ranks sharing the one GPU
(collectives staged through gloo) on random datasets and hyper-parameters against the SAME driver at one rank: every printed
epoch loss of every phase (4 decimals) and the closing metrics must agree — sharding changes the order of fp32 sums and nothing
else (same batches: every draw is a function of (seed, epoch); same candidate tables from the sharded k-NN refresh; same metrics
from the sharded evaluator).  python tools/fuzz_sharded.py [cases] [seed]"""
import contextlib, io, os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def build(cfg):
    from multike_amd.synthetic import SyntheticData, synthetic_args
    dk, ak = cfg["data"], cfg["args"]
    data = SyntheticData(shared_structure=0.8, **dk)
    n1 = data.kgs.entities_num // 2
    rng = np.random.default_rng(dk["seed"])
    base = rng.standard_normal((n1, dk["dim"])).astype(np.float32)
    nm = np.concatenate([base, base + 0.8 * rng.standard_normal((n1, dk["dim"])).astype(np.float32)])
    data.local_name_vectors = nm / np.linalg.norm(nm, axis=1, keepdims=True)
    return data, synthetic_args(**ak)


def run(cfg, rank, world, comm_oc=None, comm_v=None):
    from multike_amd.distributed_run import ShardedMultiKE_CV, ShardedMultiKE_Late
    data, args = build(cfg)
    cls = ShardedMultiKE_CV if cfg["method"] == "ITC" else ShardedMultiKE_Late
    model = cls(data, args, data.predicate_align_model, rank, world, comm_oc, comm_v)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        res = model.run()
    torch.cuda.synchronize()
    return res, out.getvalue()


def worker(rank, world, rdv, ret, cfg):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{rdv}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_oc import OcHostStagedComm
        from multike_amd.distributed_views import HostStagedViewComm
        torch.cuda.set_device(0)
        res, log = run(cfg, rank, world, OcHostStagedComm(), HostStagedViewComm())
        if rank == 0:
            ret.put((res, log))
    finally:
        dist.destroy_process_group()


def losses(log):
    return [(l.split(",")[0][:60], float(m.group(1))) for l in log.splitlines() if (m := re.search(r"avg\. loss: ([-0-9.eE+naninf]+)", l))]


if __name__ == "__main__":
    import torch.multiprocessing as mp
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = mp.get_context("spawn")
    bad = 0
    for c in range(cases):
        dim = int(rng.choice([8, 16, 24, 33, 50, 75]))
        n_ent = 2 * int(rng.integers(200, 1200))
        ep = int(rng.integers(2, 5))
        N = int(rng.choice([1, 4, 6, 10]))
        cfg = dict(method="ITC" if rng.random() < 0.5 else "SSL",
                   data=dict(n_ent=n_ent, n_rel=int(rng.integers(3, 40)), n_attr=int(rng.integers(3, 30)), n_values=int(rng.integers(20, 500)),
                             dim=dim, seed=int(rng.integers(0, 1000))),
                   args=dict(dim=dim, batch_size=int(rng.choice([301, 900, 50000])), attribute_batch_size=int(rng.choice([203, 700, 50000])),
                             entity_batch_size=int(rng.choice([97, 500, 50000])), neg_triple_num=N, learning_rate=float(rng.choice([0.003, 0.02])),
                             ITC_learning_rate=float(rng.choice([0.004, 0.03])), max_epoch=ep, shared_learning_max_epoch=int(rng.integers(1, 3)),
                             start_valid=int(rng.integers(1, ep)), eval_freq=int(rng.integers(1, 3)), start_predicate_soft_alignment=int(rng.integers(0, ep)),
                             truncated_freq=int(rng.integers(1, 3)), truncated_epsilon=0.9 if int(0.1 * (n_ent // 2)) >= N else 0.5,
                             neg_sampling=str(rng.choice(["uniform", "truncated"])), seed=int(rng.integers(0, 1000)),
                             output=f"/tmp/multike_out_fuzz_{c}/"))
        world = int(rng.choice([2, 3, 4]))
        desc = f"world={world} {cfg}"
        msg = ""
        procs = []
        try:
            r1, log1 = run(cfg, 0, 1)
            rdv = tempfile.mktemp(prefix="mke_rdv_")
            ret = ctx.Queue()
            procs = [ctx.Process(target=worker, args=(r, world, rdv, ret, cfg)) for r in range(world)]
            for p in procs:
                p.start()
            rN, logN = ret.get(timeout=600)
            for p in procs:
                p.join(120)
                assert p.exitcode == 0, f"rank exit code {p.exitcode}"
            l1, lN = losses(log1), losses(logN)
            if [a for a, _ in l1] != [a for a, _ in lN] or not l1:
                msg = f"loss lines differ: {len(l1)} vs {len(lN)}"
            else:
                worst = max(abs(a - b) / max(abs(a), 1e-3) for (_, a), (_, b) in zip(l1, lN))
                if not np.isfinite(worst) or worst > 2e-3:         # 4 printed decimals; fp32 sum order through the epochs
                    k = int(np.argmax([abs(a - b) / max(abs(a), 1e-3) for (_, a), (_, b) in zip(l1, lN)]))
                    msg = f"loss line {k} ({l1[k][0]!r}): {l1[k][1]} at one rank vs {lN[k][1]}"
            for k in r1:
                if not msg and abs(rN[k] - r1[k]) > 5e-3:
                    msg = f"result {k}: {r1[k]} vs {rN[k]}"
        except Exception as ex:  # noqa: BLE001
            import traceback
            msg = f"{type(ex).__name__}: {str(ex)[:300]} @ {traceback.format_exc().strip().splitlines()[-3][:160]}"
            for p in procs:
                if p.is_alive():
                    p.terminate()
        if msg:
            bad += 1
            print(f"SHARDED case {c}: {desc}: {msg}", flush=True)
        else:
            print(f"SHARDED case {c}: {cfg['method']} world={world} dim={dim} n_ent={n_ent} {len(l1)} loss lines, worst relative difference {worst:.1e}: ok", flush=True)
    print(f"sharded drivers: {cases - bad} / {cases} runs agree with the one-rank run")
    sys.exit(1 if bad else 0)
