"""Randomised sweep of the owner-computes sharded trainer against the float64 dense oracle: random world sizes (1..8 ranks
sharing the one GPU, collectives staged through gloo or peer-direct), entity counts that do not divide by the world size (from 60:
re-draw rounds then make a fifth of the positives need BOTH vectors), Zipf head / tail entities with a low hub-row threshold, row
widths, negatives per positive (0..64), chunk counts, exclusive-row path on / off, entity-major second pass on / off, native step
loop (mke_oc_steps) or the Python loop.  python tools/fuzz_oc.py [cases] [seed]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def worker(rank, world, rdv, ret, kw, steps):
    import torch.distributed as dist
    import test_distributed_oc_gpu as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{rdv}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_oc import OcHostStagedComm
        torch.cuda.set_device(0)
        native = kw.pop("native", False)
        tr = T._make(rank, world, comm=OcHostStagedComm() if world > 1 else None, **kw)
        if native and tr._native_loop()[0]:          # mke_oc_steps (collectives by callback at world > 1)
            tr.run(0, steps)
        else:
            for i in range(steps):
                tr.step(i)
        full = tr.gather_entity_table().cpu().numpy()
        ok = tr.scratch_clean()
        loss = tr.epoch_loss()
        if rank == 0:
            ret.put((full, tr.rel[:, :kw["dim"]].cpu().numpy().copy(), loss, ok))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    import test_distributed_oc_gpu as T
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = mp.get_context("spawn")
    bad = 0
    only = {int(x) for x in os.environ.get("MKE_FUZZ_ONLY", "").split(",") if x}   # rerun these cases of the same draw sequence
    for c in range(cases):
        world = int(rng.choice([1, 2, 2, 3, 4, 5, 6, 7, 8, 8]))
        kw = dict(n_ent=int(rng.choice([int(rng.integers(60, 400)), int(rng.integers(400, 5000))])), dim=int(rng.choice([7, 16, 20, 33, 75, 100, 128, 200, 256, 300])),
                  neg=int(rng.choice([0, 1, 3, 8, 25, 33, 64])), b=int(rng.integers(20, 400)), chunks=int(rng.integers(1, 4)),
                  excl=bool(rng.random() < 0.7), peer=bool(world > 1 and rng.random() < 0.3))
        if kw["dim"] >= 200:
            kw["n_ent"] = min(kw["n_ent"], 1500)
        kw["neg"] = min(kw["neg"], kw["n_ent"] // 2 - 6)          # a KG's candidate population must hold the sample
        kw["zipf"] = float(rng.choice([0.0, 0.0, 1.0, 1.3]))
        kw["hot_min"] = [None, 3.0][int(rng.integers(0, 2))] if kw["zipf"] else None
        kw["em"] = bool(rng.random() < 0.7)                       # entity-major second pass (not with peer-direct: the trainer falls back)
        kw["native"] = bool(rng.random() < 0.6)                   # the native step loop
        ref_kw = dict(n_ent=kw["n_ent"], dim=kw["dim"], neg=kw["neg"], b=kw["b"], zipf=kw["zipf"])
        _, _, _, spe = T._reference(world, 1, **ref_kw)
        steps = int(min(spe, rng.integers(1, 7)))
        desc = f"world={world} steps={steps} {kw}"
        if only and c not in only:
            continue
        try:
            rdv = tempfile.mktemp(prefix="mke_rdv_")
            ret = ctx.Queue()
            procs = [ctx.Process(target=worker, args=(r, world, rdv, ret, kw, steps)) for r in range(world)]
            for p in procs:
                p.start()
            full, rel, loss, ok = ret.get(timeout=300)
            for p in procs:
                p.join(120)
                assert p.exitcode == 0, f"rank exit code {p.exitcode}"
            e, r, losses, _ = T._reference(world, steps, **ref_kw)
            msg = ""
            if not ok:
                msg += " scratch not consumed;"
            if abs(loss - sum(losses)) > 3e-6 * abs(sum(losses)):
                msg += f" loss {loss} vs {sum(losses)};"
            if not np.allclose(full, e, rtol=3e-4, atol=2e-6 + 3e-5 * np.abs(e).max()):
                out = ~np.isclose(full, e, rtol=3e-4, atol=2e-6 + 3e-5 * np.abs(e).max())
                rows = np.nonzero(out.any(axis=1))[0]
                msg += (f" ent max diff {np.abs(full - e).max():.2e} ({int(out.sum())} elements of {len(rows)} rows outside the band; "
                        f"their norms {np.round(np.linalg.norm(e[rows[:6]], axis=1), 4).tolist()}, median row norm "
                        f"{np.median(np.linalg.norm(e, axis=1)):.4f});")
            if not np.allclose(rel, r, rtol=3e-4, atol=2e-6 + 3e-5 * np.abs(r).max()):
                msg += f" rel max diff {np.abs(rel - r).max():.2e};"
        except Exception as ex:  # noqa: BLE001
            msg = f"{type(ex).__name__}: {str(ex)[:300]}"
            for p in procs:
                if p.is_alive():
                    p.terminate()
        if msg:
            bad += 1
            print(f"OC case {c}: {desc}: {msg}", flush=True)
        else:
            print(f"OC case {c}: {desc}: ok", flush=True)
    print(f"owner-computes trainer: {cases - bad} / {cases} cases agree with the float64 dense oracle")
    sys.exit(1 if bad else 0)
