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
    ap.add_argument("--training_data", type=str, required=True, help="dataset folder (trailing slash optional)")
    ap.add_argument("--args", type=str, default=None, help="JSON file with args.json-style overrides")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override one args.json entry (VALUE parsed as JSON, falling back to a string)")
    a = ap.parse_args(argv)

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
    data = DataModel(args)
    predicate_align_model = PredicateAlignModel(data.kgs, args)
    model = (MultiKE_CV if a.method == "ITC" else MultiKE_Late)(data, args, predicate_align_model)
    return model.run()


if __name__ == "__main__":
    main()
