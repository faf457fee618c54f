#!/usr/bin/env python3
"""Two processes, ONE GPU: does the ncclUniqueId made on rank 0 reach rank 1 intact (multike_amd/rccl.py hands it through
torch.distributed)?  RCCL refuses two ranks on one device — but only AFTER its bootstrap: both ranks have to find each other through
the id's socket address first.  So the expected outcome on a one-GPU box is a prompt `ncclCommInitRank` error on both ranks that names
the duplicate device (or a working communicator where the build allows it) — not a hang and not a connection error.
    python tools/rccl_two_ranks_one_gpu.py        -> one JSON line per rank, exit code 0 when both ranks got past the bootstrap"""
import json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rank_main(rank, rdv):
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + rdv, rank=rank, world_size=2)
    from multike_amd import rccl
    if os.environ.get("MKE_RCCL_TRUNCATED_ID") == "1":      # what rounds 5-6 shipped until this probe: the id cut at its first NUL byte
        full = rccl._uid_to_bytes
        rccl._uid_to_bytes = lambda uid: (full(uid).split(b"\0")[0] + bytes(128))[:128]
    t0 = time.time()
    out = {"rank": rank}
    try:
        c = rccl.Communicator()
        out["outcome"] = "communicator"
        try:
            c.self_check()
            out["self_check"] = "ok"
        except Exception as e:      # noqa: BLE001
            out["self_check"] = repr(e)[:300]
    except Exception as e:          # noqa: BLE001
        out["outcome"], out["error"] = "init_error", repr(e)[:300]
    out["seconds"] = round(time.time() - t0, 2)
    print(json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 2:
        return rank_main(int(sys.argv[1]), sys.argv[2])
    rdv = tempfile.mktemp(prefix="mke_rdv2_")
    env = dict(os.environ, NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), rdv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
          for r in range(2)]
    ok = True
    for r, p in enumerate(ps):
        try:
            so, se = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            print(json.dumps({"rank": r, "outcome": "HANG (killed after 180 s)", "stderr_tail": se[-600:]}))
            ok = False
            continue
        lines = [l for l in so.splitlines() if l.startswith("{")]
        print(lines[-1] if lines else json.dumps({"rank": r, "outcome": "no output", "rc": p.returncode, "stderr_tail": se[-600:]}))
        dup = [l for l in se.splitlines() if "uplicate" in l or "WARN" in l]
        if dup:
            print(json.dumps({"rank": r, "rccl_says": dup[:3]}))
        if lines:
            d = json.loads(lines[-1])
            # past the bootstrap = a communicator, or the refusal of the second rank on the same device (invalid usage)
            ok = ok and (d["outcome"] == "communicator" or "invalid usage" in d.get("error", "").lower() or "uplicate" in se)
        else:
            ok = False
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
