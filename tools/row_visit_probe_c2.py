#!/usr/bin/env python3
"""The emptiest kernels one can write for the C2 row update's visits (35K random distinct rows of 80 floats out of 200K; weights,
accumulator, scratch: 3 reads + 3 writes = 67 MB, all of it resident in the Infinity Cache), in two lane mappings:

  q16 : a quarter-wave per row, lane j holds floats j, j + 16, ... (five dword accesses per array) — the product's mapping
        (mke_common.h: the row sums are 16-lane DPP reductions)
  h32 : half a wavefront per row, lanes 0..19 hold 16 bytes each (one dwordx4 access per array; 12 lanes idle)

against `k_rows_update_multi`'s 14.3 us (rocprofv3) for the same visits — which also scans 200K flags and carries the next step's
reference counting on rider blocks.    python tools/row_visit_probe_c2.py"""
import ctypes as C
import json
import os
import subprocess
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void k_q16(float* a0, float* a1, float* a2, const int32_t* rows, int64_t nv) {
  const int j = threadIdx.x & 15;
  const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  if (v >= nv) return;
  const int64_t off = (int64_t)rows[v] * 80 + j;
  float x[5], y[5], z[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) { x[k] = a0[off + 16 * k]; y[k] = a1[off + 16 * k]; z[k] = a2[off + 16 * k]; }
#pragma unroll
  for (int k = 0; k < 5; ++k) { a2[off + 16 * k] = 0.f; a1[off + 16 * k] = y[k] + z[k] * z[k]; a0[off + 16 * k] = x[k] + 1e-6f * z[k]; }
}
extern "C" __global__ __launch_bounds__(256) void k_h32(float* a0, float* a1, float* a2, const int32_t* rows, int64_t nv) {
  const int l = threadIdx.x & 31;
  const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  if (v >= nv || l >= 20) return;
  const int64_t off = (int64_t)rows[v] * 80 + 4 * l;
  float4 x = *reinterpret_cast<const float4*>(a0 + off), y = *reinterpret_cast<const float4*>(a1 + off), z = *reinterpret_cast<const float4*>(a2 + off);
  x.x += 1e-6f * z.x; y.x += z.y * z.y;
  *reinterpret_cast<float4*>(a2 + off) = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(a1 + off) = y;
  *reinterpret_cast<float4*>(a0 + off) = x;
}
extern "C" int launch(int which, float* a0, float* a1, float* a2, const int32_t* rows, int64_t nv, void* st) {
  if (which == 0) hipLaunchKernelGGL(k_q16, dim3((unsigned)((nv * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)st, a0, a1, a2, rows, nv);
  else hipLaunchKernelGGL(k_h32, dim3((unsigned)((nv * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)st, a0, a1, a2, rows, nv);
  return (int)hipGetLastError();
}
'''


def main():
    d = tempfile.mkdtemp(prefix="mke_probe_")
    src, so = os.path.join(d, "p.hip"), os.path.join(d, "p.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
    lib = C.CDLL(so)
    lib.launch.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p]
    n, visits = 200_000, 35_000
    arr = [torch.zeros(n, 80, device="cuda") for _ in range(3)]
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    st = torch.cuda.current_stream().cuda_stream
    for sorted_rows in (False, True):
        for which, name in ((0, "q16"), (1, "h32"), (0, "q16"), (1, "h32")):
            ts = []
            for rep in range(60):
                rows = torch.randperm(n, device="cuda", generator=g)[:visits].to(torch.int32)
                if sorted_rows:
                    rows = rows.sort().values          # the product visits the set flags in row order
                for a_ in arr:
                    a_.add_(0.0)                      # the three tables pass through the caches between two launches, as in a step
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert lib.launch(which, arr[0].data_ptr(), arr[1].data_ptr(), arr[2].data_ptr(), rows.data_ptr(), visits, st) == 0
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[5:])
            nbytes = visits * 320 * 6
            print(json.dumps({"mapping": name, "rows_sorted": sorted_rows, "median_us": round(ts[len(ts) // 2], 2), "min_us": round(ts[0], 2),
                              "GBps_at_median": round(nbytes / (ts[len(ts) // 2] * 1e-6) / 1e9)}), flush=True)


if __name__ == "__main__":
    main()
