"""Host and device cost of one RCCL collective call at world_size 1 (the only size a 1-GPU box offers): a lower bound of
the per-collective overhead the sharded step pays on top of the wire time.  python tools/rccl_latency.py"""
import os, time
import torch
import torch.distributed as dist

import tempfile
dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="mke_rdv_"), rank=0, world_size=1)
torch.cuda.set_device(0)
for nbytes in (4, 176_000, 3_400_000, 23_500_000):
    n = max(1, nbytes // 4)
    a = torch.zeros(n, device="cuda"); b = torch.zeros(n, device="cuda")
    for name, fn in (("all_reduce", lambda: dist.all_reduce(a)),
                     ("all_gather", lambda: dist.all_gather_into_tensor(b, a)),
                     ("reduce_scatter", lambda: dist.reduce_scatter_tensor(b, a)),
                     ("all_gather async+wait", lambda: dist.all_gather_into_tensor(b, a, async_op=True).wait())):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(200): fn()
        e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{nbytes:>10} B {name:24s} host {1e6 * (t1 - t0) / 200:7.1f} us/call   device {1e3 * e0.elapsed_time(e1) / 200:7.1f} us/call")
dist.destroy_process_group()
