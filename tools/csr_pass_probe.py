#!/usr/bin/env python3
"""What an entity-major second pass would cost (EXPERIMENTS R5.22 / R5.23), as an empty kernel: a quarter-wave per touched row walks
the row's references through a CSR (offsets -> reference -> a 320-byte base vector + a scalar coefficient), sums coefficient x vector
in registers, then reads and writes the row and its accumulator.  No atomics, no gradient scratch.

    one GPU, C2      : 35K touched rows of 200K, 59K references (1.7 per row), vectors out of 10K (2 per positive)
    rank 0 of 8, C2  : 25K rows (the whole shard), 125K references (5 per row), vectors out of 40K (one per global positive)

against today's terms for the same work — one GPU: 4.7 us of atomics inside the score launch + the 14.3 us update launch; rank 0 of
8: a 36 us atomics floor under k_oc_score + apply 7.4 us + update 11.9 us.    python tools/csr_pass_probe.py"""
import ctypes as C
import json
import os
import subprocess
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void k_pass2(float* w, float* acc, const float* vec, const float* coef, const int32_t* rows,
                                                          const int32_t* off, const int32_t* ref_vec, int64_t n_rows_touched) {
  const int j = threadIdx.x & 15;
  const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  if (v >= n_rows_touched) return;
  const int64_t ro = (int64_t)rows[v] * (16 * FPL) + j;
  const int lo = off[v], hi = off[v + 1];
  float x[FPL], y[FPL], g[FPL];
#pragma unroll
  for (int k = 0; k < FPL; ++k) g[k] = 0.f;
#pragma unroll
  for (int k = 0; k < FPL; ++k) { x[k] = w[ro + 16 * k]; y[k] = acc[ro + 16 * k]; }
  for (int r = lo; r < hi; r += 2) {                      // two references in flight
    const bool two = r + 1 < hi;
    const int64_t v0 = (int64_t)ref_vec[r] * (16 * FPL) + j, v1 = (int64_t)ref_vec[two ? r + 1 : r] * (16 * FPL) + j;
    const float c0 = coef[r], c1 = two ? coef[r + 1] : 0.f;
    float a[FPL], b[FPL];
#pragma unroll
    for (int k = 0; k < FPL; ++k) { a[k] = vec[v0 + 16 * k]; b[k] = vec[v1 + 16 * k]; }
#pragma unroll
    for (int k = 0; k < FPL; ++k) g[k] = fmaf(c1, b[k] - x[k], fmaf(c0, a[k] - x[k], g[k]));
  }
#pragma unroll
  for (int k = 0; k < FPL; ++k) { acc[ro + 16 * k] = fmaf(g[k], g[k], y[k]); w[ro + 16 * k] = x[k] - 1e-3f * g[k]; }
}
extern "C" int launch(float* w, float* acc, const float* vec, const float* coef, const int32_t* rows, const int32_t* off,
                      const int32_t* ref_vec, int64_t n, void* st) {
  hipLaunchKernelGGL(k_pass2, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)st, w, acc, vec, coef, rows, off, ref_vec, n);
  return (int)hipGetLastError();
}
'''


def main():
    d = tempfile.mkdtemp(prefix="mke_probe_")
    src, so = os.path.join(d, "p.hip"), os.path.join(d, "p.so")
    open(src, "w").write(SRC)
    fpl = int(os.environ.get("PROBE_FPL", "5"))          # floats per lane: 5 = 80-float rows (C2), 16 = 256-float rows (C5)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", f"-DFPL={fpl}", "-shared", "-fPIC", src, "-o", so])
    lib = C.CDLL(so)
    lib.launch.argtypes = [C.c_void_p] * 7 + [C.c_int64, C.c_void_p]
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    st = torch.cuda.current_stream().cuda_stream
    cases = (("one GPU, C2", 200_000, 35_000, 59_000, 10_000), ("rank 0 of 8, C2", 25_000, 25_000, 125_000, 40_000))
    if fpl == 16:       # C5: 2M rows x 256 floats, 64 negatives; a rank of eight owns 250K rows, ~104K of them touched per global step
        cases = (("rank 0 of 8, C5", 250_000, 104_000, 325_000, 40_000),)
    for name, n_table, n_touched, n_refs, n_vec in cases:
        w, acc = torch.zeros(n_table, 16 * fpl, device="cuda"), torch.zeros(n_table, 16 * fpl, device="cuda")
        vec = torch.zeros(n_vec, 16 * fpl, device="cuda")
        ts = []
        for rep in range(60):
            rows = torch.randperm(n_table, device="cuda", generator=g)[:n_touched].sort().values.to(torch.int32)
            # every touched row has one reference, the rest are dealt at random (a Poisson-like tail)
            owner = torch.cat([torch.arange(n_touched, device="cuda"),
                               torch.randint(0, n_touched, (n_refs - n_touched,), device="cuda", generator=g)])
            cnt = torch.bincount(owner, minlength=n_touched)
            off = torch.zeros(n_touched + 1, dtype=torch.int32, device="cuda")
            off[1:] = cnt.cumsum(0).to(torch.int32)
            ref_vec = torch.randint(0, n_vec, (n_refs,), device="cuda", generator=g).to(torch.int32)
            coef = torch.rand(n_refs, device="cuda", generator=g)
            for a_ in (w, acc, vec):
                a_.add_(0.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.launch(w.data_ptr(), acc.data_ptr(), vec.data_ptr(), coef.data_ptr(), rows.data_ptr(), off.data_ptr(),
                              ref_vec.data_ptr(), n_touched, st) == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[5:])
        nb = n_touched * 4 * 64 * fpl + n_refs * (64 * fpl + 8)
        print(json.dumps({"case": name, "touched_rows": n_touched, "references": n_refs, "MB": round(nb / 1e6, 1),
                          "median_us": round(ts[len(ts) // 2], 2), "min_us": round(ts[0], 2), "p90_us": round(ts[int(len(ts) * 0.9)], 2)}), flush=True)


if __name__ == "__main__":
    main()
