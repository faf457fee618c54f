#!/usr/bin/env python3
"""Map of HBM by allocation: N arrays of 2 GB (rows of 1 KB) allocated one after the other and all kept alive, each probed alone
with a read-only gather of 2M random rows (`mke_probe_rows`).  `tools/c5_probe.py` found two classes of allocation on one box
(792 us / 825 us per probe) and the relation step 16 % slower on the slow class; this prints where the classes lie.

    python tools/hbm_map.py [--n 100] [--gb 2]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multike_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--rows", type=int, default=2_000_000)
    a = ap.parse_args()
    n = a.rows
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    idx = torch.randint(0, n, (2_000_000,), device="cuda", generator=g, dtype=torch.int32)
    out = torch.empty(idx.numel(), device="cuda")
    arrs, rows = [], []
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(a.n):
        if torch.cuda.mem_get_info()[0] < 6 * 2 ** 30:
            break
        x = torch.zeros(n, 256, device="cuda")
        arrs.append(x)
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _lib.probe_rows(x, None, None, idx, out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        rows.append((k, x.data_ptr(), float(np.median(ts[1:]))))
    lo = min(r[2] for r in rows)
    print(json.dumps({"free_GB_at_start": round(free0 / 2 ** 30, 1), "arrays": len(rows), "fastest_us": round(lo, 1)}))
    for k, p, t in rows:
        print(f"alloc {k:3d}  va 0x{p:x}  probe {t:7.1f} us  {'#' * int(round((t / lo - 1) * 200))}")
    # second pass over the same arrays: is the class a property of the allocation (stable) or of the moment?
    again = []
    for k, x in enumerate(arrs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.probe_rows(x, None, None, idx, out); e1.record(); torch.cuda.synchronize()
        again.append(e0.elapsed_time(e1) * 1e3)
    print("second pass, us:", " ".join(f"{t:.0f}" for t in again))


if __name__ == "__main__":
    main()
