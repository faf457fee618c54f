#!/usr/bin/env python3
"""Does it matter WHERE the three arrays of a row (weights, Adagrad accumulator, gradient scratch) live?  (round 5, after the
rows-in-flight experiment: more loads in flight made the C5 row update slower, which points at address translation, not at
concurrency.)  Random row visits at the C5 shape (2M rows x 256 floats), one wavefront per visit, 16 bytes per lane:

  separate : three arrays [n][256]           (today's layout: a visit touches three places 2 GB apart)
  together : one array   [n][768]            (row i = weights | accumulator | scratch, 3 KB contiguous)

for an update-like visit (3 reads + 3 writes) and a score-like visit (2 reads).  Builds its kernel with hipcc into /tmp.

    python tools/row_layout_probe.py [--rows 2000000] [--visits 34500] [--reps 50]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void k_visit(float* a0, float* a1, float* a2, const int32_t* rows, int64_t n_visits,
                                                          int64_t pitch, int mode) {
  const int lane = threadIdx.x & 63;
  const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (v >= n_visits) return;
  const int64_t off = (int64_t)rows[v] * pitch + lane * 4;
  float4 x = *reinterpret_cast<const float4*>(a0 + off);
  float4 y = *reinterpret_cast<const float4*>(a1 + off);
  if (mode == 0) {          // update-like: three reads, three writes
    float4 z = *reinterpret_cast<const float4*>(a2 + off);
    x.x += 1e-6f * z.x; y.x += z.y * z.y;
    *reinterpret_cast<float4*>(a2 + off) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(a1 + off) = y;
    *reinterpret_cast<float4*>(a0 + off) = x;
  } else {                  // score-like: two reads (a value that depends on both keeps them alive)
    if (x.x + y.x == 123.456f) a2[off] = 1.f;
  }
}
extern "C" int launch(float* a0, float* a1, float* a2, const int32_t* rows, int64_t n_visits, int64_t pitch, int mode, void* st) {
  const int64_t blocks = (n_visits * 64 + 255) / 256;
  hipLaunchKernelGGL(k_visit, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, a0, a1, a2, rows, n_visits, pitch, mode);
  return (int)hipGetLastError();
}
'''


def build():
    d = tempfile.mkdtemp(prefix="mke_probe_")
    src, so = os.path.join(d, "probe.hip"), os.path.join(d, "probe.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
    return C.CDLL(so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--visits", type=int, nargs="+", default=[34_500, 325_000])
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    lib = build()
    lib.launch.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    n, w = a.rows, a.width
    sep = [torch.zeros(n, w, device="cuda") for _ in range(3)]
    tog = torch.zeros(n, 3 * w, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    st = torch.cuda.current_stream().cuda_stream
    out = {"rows": n, "width": w, "results": []}
    for visits in a.visits:
        for mode, name in ((0, "update-like (3 reads + 3 writes)"), (1, "score-like (2 reads)")):
            res = {}
            for layout in ("separate", "together", "separate", "together"):
                ts = []
                for rep in range(a.reps):
                    rows = torch.randperm(n, device="cuda", generator=g)[:visits].to(torch.int32)
                    if layout == "separate":
                        p = [t.data_ptr() for t in sep]; pitch = w
                    else:
                        p = [tog.data_ptr() + 4 * w * k for k in range(3)]; pitch = 3 * w
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = lib.launch(p[0], p[1], p[2], rows.data_ptr(), visits, pitch, mode, st)
                    e1.record()
                    assert rc == 0
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                ts = sorted(ts[3:])
                res.setdefault(layout, []).append(ts[len(ts) // 2])
            nbytes = visits * w * 4 * (6 if mode == 0 else 2)
            row = {"visits": visits, "visit": name, "MB": nbytes / 1e6}
            for k, v in res.items():
                row[k + "_us"] = [round(x, 2) for x in v]
                row[k + "_GBps"] = [round(nbytes / (x * 1e-6) / 1e9) for x in v]
            out["results"].append(row)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
