#!/usr/bin/env python3
"""The memory visits of one C2 `k_triple_score` launch, issued by a kernel with nothing else in it: 125K corrupt rows (80 floats) —
65 % referenced once in the step: row + accumulator read, both written in place; the rest: row read, atomic row add into the
scratch — and 15K positive rows (read + atomic row add), every visit an independent quarter-wave (no groups, no ids beyond the row
id, no arithmetic).  The product's launch: 34.3 us (rocprofv3) for the same visits plus the group structure, normalisation, scoring,
Jacobian and Adagrad.    python tools/score_visit_probe_c2.py [in-place share]"""
import ctypes as C
import json
import os
import subprocess
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void k_visits(float* w, float* acc, float* grad, const int32_t* rows, const uint8_t* inplace, int64_t nv) {
  const int j = threadIdx.x & 15;
  const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  if (v >= nv) return;
  const int64_t off = (int64_t)rows[v] * (16 * FPL) + j;
  const bool ip = inplace[v] != 0;
  float x[FPL], y[FPL];
#pragma unroll
  for (int k = 0; k < FPL; ++k) { x[k] = w[off + 16 * k]; y[k] = ip ? acc[off + 16 * k] : 0.f; }
  if (ip) {
#pragma unroll
    for (int k = 0; k < FPL; ++k) { acc[off + 16 * k] = y[k] + x[k] * x[k]; w[off + 16 * k] = x[k] * 0.999f; }
  } else {
#pragma unroll
    for (int k = 0; k < FPL; ++k) unsafeAtomicAdd(grad + off + 16 * k, x[k] * 1e-3f);
  }
}
extern "C" int launch(float* w, float* acc, float* grad, const int32_t* rows, const uint8_t* inplace, int64_t nv, void* st) {
  hipLaunchKernelGGL(k_visits, dim3((unsigned)((nv * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)st, w, acc, grad, rows, inplace, nv);
  return (int)hipGetLastError();
}
'''


def main():
    d = tempfile.mkdtemp(prefix="mke_probe_")
    src, so = os.path.join(d, "p.hip"), os.path.join(d, "p.so")
    open(src, "w").write(SRC)
    fpl = int(os.environ.get("PROBE_FPL", "5"))          # floats per lane: 5 = 80-float rows (C2), 16 = 256-float rows (C5)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", f"-DFPL={fpl}", "-shared", "-fPIC", src, "-o", so])
    lib = C.CDLL(so)
    lib.launch.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    import sys
    n, n_neg, n_posrows = 200_000, int(os.environ.get("PROBE_VISITS", "125000")), 15_000
    if len(sys.argv) > 2:       # [in-place share] [rows] [positive rows]: e.g. 0 25000 0 = what rank 0 of 8 scatters in `k_oc_score` at C2
        n = int(sys.argv[2])    # (125K corrupt rows of its 25K-row shard, every one an atomic row add)
        n_posrows = int(sys.argv[3]) if len(sys.argv) > 3 else n_posrows
    w, acc, grad = (torch.zeros(n, 16 * fpl, device="cuda") for _ in range(3))
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    st = torch.cuda.current_stream().cuda_stream
    share = float(sys.argv[1]) if len(sys.argv) > 1 else None      # force the in-place share (the product reports 65 % at C2): flags by coin
    ts = []
    for rep in range(80):
        # corrupt entities are drawn with replacement (as the sampler's are across positives); a row drawn once is finished in place
        e = torch.randint(0, n, (n_neg,), device="cuda", generator=g)
        cnt = torch.bincount(torch.cat([e, torch.randint(0, n, (n_posrows,), device="cuda", generator=g)]), minlength=n)
        pos_rows = torch.randint(0, n, (max(n_posrows, 1),), device="cuda", generator=g)[:n_posrows]
        rows = torch.cat([e, pos_rows]).to(torch.int32)
        ip = (cnt[e] == 1) if share is None else (torch.rand(n_neg, device="cuda", generator=g) < share)
        inplace = torch.cat([ip, torch.zeros(n_posrows, dtype=torch.bool, device="cuda")]).to(torch.uint8)
        for a_ in (w, acc, grad):
            a_.add_(0.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.launch(w.data_ptr(), acc.data_ptr(), grad.data_ptr(), rows.data_ptr(), inplace.data_ptr(), rows.numel(), st) == 0
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
        frac = float(inplace[:n_neg].float().mean())
    ts = sorted(ts[5:])
    rb = 64 * fpl
    nb = int(n_neg * frac) * 4 * rb + (n_neg - int(n_neg * frac)) * 2 * rb + n_posrows * 2 * rb
    print(json.dumps({"visits": int(rows.numel()), "in_place_share_of_corrupt_rows": round(frac, 3), "MB": round(nb / 1e6, 1),
                      "median_us": round(ts[len(ts) // 2], 2), "min_us": round(ts[0], 2), "p90_us": round(ts[int(len(ts) * 0.9)], 2),
                      "GBps_at_median": round(nb / (ts[len(ts) // 2] * 1e-6) / 1e9)}))


if __name__ == "__main__":
    main()
