#!/usr/bin/env python3
"""The attribute step (B = 5000) over widths, fused launches on and off: `python tools/attr_widths.py [steps]` -> one markdown table
(profiles/r06_attr_widths.md).  Times are host wall time over `steps` back-to-back steps (the queue stays full)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd import _lib
from multike_amd.attr_cnn import AttrCNN
from multike_amd.tables import EmbeddingTable, StepEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 5000
g = torch.Generator(device="cuda"); g.manual_seed(0)
print("| dim | fused launches (us/step) | `attr_fused_bwd` = 0 (us/step) |\n|---|---|---|")
for d in (32, 64, 75, 80, 96, 100, 112, 128, 130):
    row = []
    for fused in (1, 0):
        _lib.set_option("attr_fused_bwd", fused)
        E = EmbeddingTable(200_000, d, "av", seed=1)
        A = EmbeddingTable(600, d, "attr", normalize=False, seed=2, grad_copies=4)
        lit = np.random.default_rng(0).standard_normal((100_000, d)).astype(np.float32); lit /= np.linalg.norm(lit, axis=1, keepdims=True)
        L = EmbeddingTable(100_000, d, "lit", normalize=False, trainable=False, values=lit)
        cnn = AttrCNN(d, seed=3); eng = StepEngine()
        bs = [(torch.randint(0, 200_000, (B,), device="cuda", generator=g, dtype=torch.int32),
               torch.randint(0, 600, (B,), device="cuda", generator=g, dtype=torch.int32),
               torch.randint(0, 100_000, (B,), device="cuda", generator=g, dtype=torch.int32), torch.rand(B, device="cuda", generator=g)) for _ in range(8)]
        for i in range(20): cnn.step(eng, E, A, L, *bs[i % 8])
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(steps): cnn.step(eng, E, A, L, *bs[i % 8])
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps)
        row.append(best * 1e6)
    _lib.set_option("attr_fused_bwd", 1)
    print(f"| {d} | {row[0]:.1f} | {row[1]:.1f} |", flush=True)
