"""Command-line entry points with the shape of the reference's run_ITC.py / run_SSL.py (code/run_ITC.py:1-21,
code/run_SSL.py:1-21): load args.json, point it at a dataset folder, build DataModel + PredicateAlignModel, run.

    python -m multike_amd.run --method ITC --training_data /data/BootEA_DBP_WD_100K/ [--args args.json] [--set k=v ...]

`--args` defaults to the args.json next to this file (the reference's hyper-parameters, code/args.json, with
machine-specific paths removed)."""
from __future__ import annotations

import argparse
import json
import os


def main(argv=None):
    ap = argparse.ArgumentParser(description="MultiKE on MI355X: ITC (MultiKE_CV) or SSL (MultiKE_Late)")
    ap.add_argument("--method", choices=["ITC", "SSL"], default="ITC")
    ap.add_argument("--training_data", type=str, required=True, help="dataset folder (trailing slash optional)")
    ap.add_argument("--args", type=str, default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "args.json"))
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override one args.json entry (VALUE parsed as JSON, falling back to a string)")
    a = ap.parse_args(argv)

    from .data_model import DataModel
    from .MultiKE_CSL import MultiKE_CV
    from .MultiKE_Late import MultiKE_Late
    from .predicate_alignment import PredicateAlignModel
    from .utils import load_args

    args = load_args(a.args)
    args.training_data = a.training_data if a.training_data.endswith("/") else a.training_data + "/"
    for kv in a.set:
        k, _, v = kv.partition("=")
        try:
            v = json.loads(v)
        except json.JSONDecodeError:
            pass
        setattr(args, k, v)
    data = DataModel(args)
    predicate_align_model = PredicateAlignModel(data.kgs, args)
    model = (MultiKE_CV if a.method == "ITC" else MultiKE_Late)(data, args, predicate_align_model)
    return model.run()


if __name__ == "__main__":
    main()
