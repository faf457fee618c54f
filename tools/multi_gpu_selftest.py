#!/usr/bin/env python3
"""First contact with a multi-GPU node: run this BEFORE `bench.py --gpus N`.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/multi_gpu_selftest.py
    MKE_BENCH_COMM=staged python -m torch.distributed.run ... --nproc-per-node 2 tools/multi_gpu_selftest.py   # dry run: ranks share GPUs, gloo

Every collective the sharded trainers issue is exercised once on small known data and checked
element for element, then one epoch fragment of every sharded loop is compared with the same global steps on ONE rank's
full tables.  Any mismatch raises on every rank (exit code != 0): a wrong transport must not produce a benchmark number.

  1. all_gather_into_tensor / reduce_scatter_tensor (relation view: HR / RT vectors and their gradients), blocking and async
  2. all_reduce of float32 buffers (relation / CNN / mapping gradients), of float64 scalars (batch-wide sums, losses), MAX
  3. all_gather of padded shards (checkpoint / evaluator / k-NN refresh), all_gather_object
  4. owner-computes relation steps at world N == the same global batches at world 1 (loss and the full entity table)
  5. sharded attribute view, common space, space mapping: two steps each == world 1
  6. (--peer, informational) peer-direct IPC mapping of the other ranks' blocks: reported, not fatal (opt-in path, MKE_SHARD_PEER=1)

Rank 0 prints one JSON line {"selftest": "ok", ...} at the end."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def fail(msg):
    raise SystemExit(f"[multi_gpu_selftest] rank {dist.get_rank()}: FAILED: {msg}")


def check(cond, msg):
    """All ranks agree on the verdict (a rank that alone saw a mismatch must still stop the others)."""
    bad = torch.tensor([0 if cond else 1], dtype=torch.int32, device=DEV if not STAGED else "cpu")
    dist.all_reduce(bad)
    if int(bad):
        fail(msg if not cond else f"another rank reported: {msg}")


def main():
    global DEV, STAGED
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    STAGED = os.environ.get("MKE_BENCH_COMM", "") == "staged"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    assert torch.cuda.is_available(), "needs GPUs"
    if STAGED:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    DEV = torch.device("cuda", local)
    import datetime
    if STAGED:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    else:
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=DEV, timeout=datetime.timedelta(seconds=300))
    from multike_amd.distributed_oc import OcComm, OcHostStagedComm, OwnerComputesTrainer
    from multike_amd.distributed_views import HostStagedViewComm, ViewComm
    oc = OcHostStagedComm() if STAGED else OcComm()
    vc = HostStagedViewComm() if STAGED else ViewComm()
    report = {"world": world, "backend": "gloo (staged dry run)" if STAGED else "nccl (RCCL)"}
    t0 = time.time()

    # ---- 1. all-gather / reduce-scatter of equal blocks ------------------------------------------------------------
    n = 4096
    mine = (torch.arange(n, dtype=torch.float32, device=DEV) + 1000.0 * rank)
    out = torch.empty(world * n, dtype=torch.float32, device=DEV)
    oc.all_gather(out, mine)
    exp = torch.cat([torch.arange(n, dtype=torch.float32, device=DEV) + 1000.0 * r for r in range(world)])
    check(torch.equal(out, exp), "all_gather_into_tensor returned wrong data")
    inp = torch.arange(world * n, dtype=torch.float32, device=DEV) * (rank + 1)
    rs = torch.empty(n, dtype=torch.float32, device=DEV)
    oc.reduce_scatter(rs, inp)
    tot = world * (world + 1) / 2
    exp = torch.arange(rank * n, (rank + 1) * n, dtype=torch.float32, device=DEV) * tot
    check(torch.allclose(rs, exp, rtol=1e-6), "reduce_scatter_tensor returned wrong sums")
    if not STAGED:
        out.zero_()
        w = oc.all_gather(out, mine, async_op=True)
        w.wait()
        check(bool((out.view(world, n)[rank] == mine).all()), "async all_gather + wait() returned wrong data")
    # ---- 2. all-reduce flavours ----------------------------------------------------------------------------------------
    g = torch.full((100_000,), float(rank + 1), dtype=torch.float32, device=DEV)
    oc.all_reduce(g)
    check(bool((g == tot).all()), "float32 all_reduce")
    s = torch.tensor([1.0 + rank * 2.0 ** -40], dtype=torch.float64, device=DEV)
    vc.all_reduce(s)
    check(abs(float(s) - (world + 2.0 ** -40 * world * (world - 1) / 2)) < 1e-15, "float64 all_reduce lost precision")
    m = torch.tensor([rank], dtype=torch.int64, device=DEV)
    oc.all_reduce(m, op=dist.ReduceOp.MAX)
    check(int(m) == world - 1, "int64 MAX all_reduce")
    # ---- 3. list all-gather + object all-gather --------------------------------------------------------------------
    parts = [torch.empty(7, 5, dtype=torch.float32, device=DEV) for _ in range(world)]
    oc.all_gather_list(parts, torch.full((7, 5), float(rank), dtype=torch.float32, device=DEV))
    check(all(bool((p == r).all()) for r, p in enumerate(parts)), "all_gather (list) returned wrong data")
    objs = oc.all_gather_object({"rank": rank})
    check([o["rank"] for o in objs] == list(range(world)), "all_gather_object")
    report["collectives_s"] = round(time.time() - t0, 2)

    # ---- 4. relation view: world N == world 1 on the same global batches -----------------------------------------------
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import xavier_truncated_normal as xtn     # the product's initialiser (TF1 xavier)
    n_ent, n_rel, d, N, P, steps = 20_000, 40, 75, 10, 256, 5
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, seed=3)
    ent0, rel0 = xtn(n_ent, d, "cpu", seed=3).numpy(), xtn(n_rel, d, "cpu", seed=4).numpy()
    for chunks in (1, 2):
        tr = OwnerComputesTrainer(kgs, ent0, rel0, P, N, rank, world, seed=7, lr=0.01, comm=oc, chunks=chunks)
        for i in range(steps):
            tr.step(i)
        loss = tr.epoch_loss()
        full = tr.gather_entity_table()
        one = OwnerComputesTrainer(kgs, ent0, rel0, P * world, N, 0, 1, seed=7, lr=0.01)     # the same GLOBAL steps on one rank
        for i in range(steps):
            one.step(i)
        l1 = one.epoch_loss()
        ref = one.gather_entity_table()
        check(abs(loss - l1) <= 2e-5 * abs(l1), f"relation view (chunks={chunks}): loss {loss} vs {l1} on one rank")
        err = float((full - ref).abs().max())
        check(err < 2e-5, f"relation view (chunks={chunks}): entity table differs from the one-rank run by {err}")
        check(bool(torch.allclose(tr.rel, one.rel, rtol=1e-4, atol=1e-6)), "relation table differs from the one-rank run")
        report[f"relation_chunks{chunks}_max_abs_diff"] = err
        del tr, one
    # ---- 5. the other sharded loops -----------------------------------------------------------------------------------
    from multike_amd.distributed_model import ShardedITC
    from multike_amd.attr_cnn import AttrCNN

    def build(r, w, c_oc, c_v):
        rg = np.random.default_rng(11)
        n_e, n_a, n_l, dd = 2400, 30, 300, 32
        k2 = SyntheticKGs(n_ent=n_e, n_rel=12, seed=9)
        t = lambda k: xtn(k, dd, "cpu", seed=int(rg.integers(1 << 30))).numpy()
        unit = lambda k: (lambda x: (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32))(rg.standard_normal((k, dd)))
        tables = {"rv_ent": t(n_e), "av_ent": t(n_e), "ent": t(n_e), "name": unit(n_e), "rel": t(12), "attr": t(n_a), "lit": unit(n_l)}
        cnn = [AttrCNN(dd, seed=100 + k).numpy_params() for k in range(3)]
        ri = lambda hi, k: rg.integers(0, hi, k)
        lists = {"attr": [(int(h), int(a), int(v), float(x)) for h, a, v, x in zip(ri(n_e, 900), ri(n_a, 900), ri(n_l, 900), rg.uniform(0.3, 1, 900))],
                 "ckge_rel": [(int(h), int(q), int(z)) for h, q, z in zip(ri(n_e, 500), ri(12, 500), ri(n_e, 500))],
                 "ckgp_rel": [(int(h), int(q), int(z), 0.7) for h, q, z in zip(ri(n_e, 300), ri(12, 300), ri(n_e, 300))],
                 "ckge_attr": [(int(h), int(a), int(v)) for h, a, v in zip(ri(n_e, 450), ri(n_a, 450), ri(n_l, 450))],
                 "ckga_attr": [(int(h), int(a), int(v), 0.5) for h, a, v in zip(ri(n_e, 200), ri(n_a, 200), ri(n_l, 200))],
                 "entities": [int(x) for x in rg.choice(n_e, 700, replace=False)]}
        mats = [np.linalg.qr(rg.standard_normal((dd, dd)))[0].astype(np.float32) for _ in range(3)]
        return ShardedITC(k2, tables, cnn, lists, r, w, batch_size=400, attribute_batch_size=300, entity_batch_size=250,
                          neg_triple_num=4, learning_rate=0.01, itc_learning_rate=0.02, seed=9, comm_oc=c_oc, comm_views=c_v,
                          mapping_matrices=mats)

    m = build(rank, world, oc, vc)
    la = [m.epoch(1), m.epoch_ssl(2)]
    got = m.gather()
    m1 = build(0, 1, None, None)
    lb = [m1.epoch(1), m1.epoch_ssl(2)]
    ref = m1.gather()
    for ea, eb in zip(la, lb):
        for k in eb:
            check(abs(ea[k] - eb[k]) <= 5e-5 * max(abs(eb[k]), 1e-12), f"sharded epoch: loss of phase {k}: {ea[k]} vs {eb[k]} on one rank")
    for k in ("ent", "rv", "av", "rel", "attr", "matrices"):
        err = float(np.abs(np.asarray(got[k]) - np.asarray(ref[k])).max())
        check(err < 5e-4, f"sharded epoch: table {k} differs from the one-rank run by {err}")
        report[f"epoch_{k}_max_abs_diff"] = err
    # ---- 6. peer-direct mapping (informational) -----------------------------------------------------------------------
    if world > 1 and "--peer" in sys.argv:       # opt-in: a rank that fails alone here could leave the others in a collective
        try:
            tr = OwnerComputesTrainer(kgs, ent0, rel0, P, N, rank, world, seed=7, lr=0.01, comm=oc, peer_direct=True)
            for i in range(2):
                tr.step(i)
            lp = tr.epoch_loss()
            report["peer_direct"] = "ok" if np.isfinite(lp) else "non-finite loss"
        except Exception as e:  # noqa: BLE001 — reported, not fatal: the path is opt-in
            report["peer_direct"] = f"unavailable: {type(e).__name__}: {str(e)[:120]}"
        oks = oc.all_gather_object(report["peer_direct"])
        report["peer_direct"] = oks if len(set(oks)) > 1 else oks[0]
    torch.cuda.synchronize()
    report["seconds"] = round(time.time() - t0, 1)
    report["selftest"] = "ok"
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(report), flush=True)


if __name__ == "__main__":
    main()
