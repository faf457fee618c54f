"""Command-line entry points with the shape of the reference's run_ITC.py / run_SSL.py (code/run_ITC.py:1-21,
code/run_SSL.py:1-21): load args.json, point it at a dataset folder, build DataModel + PredicateAlignModel, run.

    python -m multike_amd.run --method ITC --training_data /data/BootEA_DBP_WD_100K/ [--args my_args.json] [--set k=v ...]

Hyper-parameters start from `utils.default_args()` (the values the reference ships in code/args.json); `--args` names a
JSON file in the reference's args.json format whose entries override them, `--set` overrides single entries."""
from __future__ import annotations

import argparse
import json


def main(argv=None):
    ap = argparse.ArgumentParser(description="MultiKE on MI355X: ITC (MultiKE_CV) or SSL (MultiKE_Late)")
    ap.add_argument("--method", choices=["ITC", "SSL"], default="ITC")
    ap.add_argument("--training_data", type=str, default="synthetic/", help="dataset folder (trailing slash optional)")
    ap.add_argument("--args", type=str, default=None, help="JSON file with args.json-style overrides")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override one args.json entry (VALUE parsed as JSON, falling back to a string)")
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node to train on: > 1 runs the row-sharded drivers (multike_amd/distributed_run.py), one "
                         "process per GPU over RCCL; started without torch.distributed.run, this command launches it itself")
    ap.add_argument("--synthetic", type=str, default=None, metavar="JSON",
                    help="train on multike_amd.synthetic.SyntheticData(**JSON) instead of a dataset folder (smoke runs, tests)")
    a = ap.parse_args(argv)
    import os
    import sys
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), "-m", "multike_amd.run"] + list(sys.argv[1:] if argv is None else argv)
        raise SystemExit(subprocess.call(cmd))

    from .data_model import DataModel
    from .MultiKE_CSL import MultiKE_CV
    from .MultiKE_Late import MultiKE_Late
    from .predicate_alignment import PredicateAlignModel
    from .utils import default_args, load_args

    args = default_args()
    if a.args:
        for k, v in vars(load_args(a.args)).items():
            setattr(args, k, v)
    args.training_data = a.training_data if a.training_data.endswith("/") else a.training_data + "/"
    for kv in a.set:
        k, _, v = kv.partition("=")
        try:
            v = json.loads(v)
        except json.JSONDecodeError:
            pass
        setattr(args, k, v)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, comm_oc, comm_v = 0, None, None
    if a.synthetic is None and a.training_data == "synthetic/" and not os.path.isdir(args.training_data):
        raise SystemExit("multike_amd.run: give --training_data <dataset folder> (or --synthetic '{...}' for a generated dataset)")
    if world > 1:
        # the process group and this rank's device come FIRST: DataModel trains the literal auto-encoder on the device, and
        # before set_device every rank would do that on GPU 0
        import torch
        import torch.distributed as dist
        from .distributed_run import ShardedMultiKE_CV, ShardedMultiKE_Late, init_process_group_from_env
        if os.environ.get("MKE_BENCH_COMM", "") == "staged":     # dry run: ranks share GPUs, collectives staged through gloo
            from .distributed_oc import OcHostStagedComm
            from .distributed_views import HostStagedViewComm
            rank = int(os.environ["RANK"])
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            comm_oc, comm_v = OcHostStagedComm(), HostStagedViewComm()
        else:
            rank, world = init_process_group_from_env()

    def load_data():
        if a.synthetic is not None:
            from .synthetic import SyntheticData
            data = SyntheticData(**dict({"dim": args.dim}, **json.loads(a.synthetic)))
            return data, data.predicate_align_model
        data = DataModel(args)
        return data, PredicateAlignModel(data.kgs, args)

    def _checksum(data):
        import hashlib
        import numpy as np
        h = hashlib.sha256()
        for name in ("value_vectors", "local_name_vectors"):
            v = getattr(data, name, None)
            if v is not None:
                h.update(np.ascontiguousarray(np.asarray(v, dtype=np.float32)).tobytes())
        return h.hexdigest()

    if world > 1:
        import contextlib
        import io
        # rank 0 prepares the dataset first (it trains the literal auto-encoder ONCE and writes the literal cache into the
        # folder); the others wait and then read that cache, so the replicated constants (literal / name vectors) are the
        # same bytes on every rank instead of N independently trained copies
        quiet = contextlib.redirect_stdout(io.StringIO()) if rank else contextlib.nullcontext()    # the readers print per rank
        with quiet:
            # rank 0's outcome travels to the others (an object broadcast, which the waiting ranks sit in with the process
            # group's own timeout): a failed preparation ends every rank with the error instead of leaving them in a barrier,
            # and the checksum of the literal vectors is compared afterwards — ranks that do not see the same dataset folder
            # (several nodes, a read-only mount) would otherwise train on different constants without a word
            status = [None]
            if rank == 0:
                try:
                    data, predicate_align_model = load_data()
                    status[0] = ("ok", _checksum(data))
                except Exception as e:          # noqa: BLE001 — reported on every rank, re-raised below
                    status[0] = ("failed", f"{type(e).__name__}: {e}")
            dist.broadcast_object_list(status, src=0)
            if status[0][0] != "ok":
                dist.destroy_process_group()
                raise SystemExit(f"multike_amd.run: rank 0 could not prepare the dataset: {status[0][1]}")
            if rank != 0:
                retrain, args.retrain_literal_embeds = getattr(args, "retrain_literal_embeds", False), False
                data, predicate_align_model = load_data()
                args.retrain_literal_embeds = retrain
                if _checksum(data) != status[0][1]:
                    raise SystemExit(f"multike_amd.run: rank {rank} loaded other literal / name vectors than rank 0 — every rank must "
                                     "read the same dataset folder (with rank 0's literal cache in it)")
            cls = ShardedMultiKE_CV if a.method == "ITC" else ShardedMultiKE_Late
            model = cls(data, args, predicate_align_model, rank, world, comm_oc, comm_v)
            res = model.run()               # the schedule prints once (rank 0), not once per rank
        if rank == 0:
            print("results:", json.dumps({k: float(v) for k, v in res.items()}))
        dist.destroy_process_group()
        return res
    data, predicate_align_model = load_data()
    model = (MultiKE_CV if a.method == "ITC" else MultiKE_Late)(data, args, predicate_align_model)
    return model.run()


if __name__ == "__main__":
    main()
