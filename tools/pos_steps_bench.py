#!/usr/bin/env python3
"""Positives-only relation steps (the cross-KG inference loops, code/MultiKE_model.py:349-369: B positives per step, no
negatives) on uniform / Zipf head-tail entities, with and without the hub rows declared on the entity table.
    python tools/pos_steps_bench.py [zipf] [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd.runner import run_positive_steps
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable

zipf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, d = 5000, 75
kgs = SyntheticKGs(n_ent=200_000, n_rel=550, seed=1234, zipf=zipf)
tr = np.concatenate(kgs.triples)
rng = np.random.default_rng(0)
tr = tr[rng.permutation(len(tr))]
steps = len(tr) // B
cols = tuple(torch.as_tensor(np.ascontiguousarray(tr[:steps * B, k]), device="cuda") for k in range(3))
off = np.arange(steps + 1, dtype=np.int64) * B
for hubs in (False, True):
    E = EmbeddingTable(200_000, d, "e", seed=1)
    R = EmbeddingTable(550, d, "r", seed=2, grad_copies=8)
    if hubs:
        deg = np.bincount(tr[:, [0, 2]].reshape(-1), minlength=200_000) / steps
        hot = np.nonzero(deg >= 20)[0]
        if len(hot):
            E.set_hot_rows(hot, 8)
    tag = 1
    for ep in range(epochs + 1):
        if ep == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        run_positive_steps(E, R, "ckge", cols, None, off, tag, 0.001, scale=2.0)
        tag += steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (epochs * steps)
    print(f"zipf {zipf}: positives-only step of {B} triples, hub rows {'declared (%d)' % E.n_hot if hubs else 'not declared'}: {dt * 1e6:.1f} us per step")
