"""Does the attribute step's dflat = dz W^T product gain from a transposed copy of W (contiguous B operand)?  Times
mke_gemm_f32 on [5000 x 75] x [75 x 300] with B = W^T read in place (strided) vs from a contiguous copy."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multike_amd import _lib
n, d = 5000, 75
dz = torch.randn(n, d, device="cuda")
W = torch.randn(4 * d, d, device="cuda")
WT = W.t().contiguous()
out = torch.empty(n, 4 * d, device="cuda")
def t(fn, it=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
a = t(lambda: _lib.gemm_f32(dz, W, out, transpose_b=True))
ref = out.clone()
b = t(lambda: _lib.gemm_f32(dz, WT, out))
print(f"dflat with W read transposed in place: {a:.2f} us; with a contiguous W^T: {b:.2f} us; max diff {float((out-ref).abs().max()):.2e}")
c = t(lambda: torch.matmul(dz, WT, out=out))
print(f"library (torch.matmul): {c:.2f} us")
dzp = torch.zeros(n, 80, device="cuda"); dzp[:, :d] = dz
e = t(lambda: _lib.gemm_f32(dzp[:, :d], WT, out))
print(f"16-byte-load kernel (dz rows padded to 80 floats, contiguous W^T): {e:.2f} us; max diff {float((out-ref).abs().max()):.2e}")
# the weight-gradient product [301 x 5000] x [5000 x 75] for comparison: flat^T dz, split-K
flat = torch.randn(n, 4 * d + 4, device="cuda")
gW = torch.zeros(4 * d + 1, d, device="cuda")
f = t(lambda: _lib.gemm_f32(flat[:, :4 * d + 1], dz, gW, transpose_a=True, splits=32, accumulate=True))
print(f"dW = flat^T dz, 32 K splits (generic kernel): {f:.2f} us")
gWp = torch.zeros(4 * d + 1, 80, device="cuda")
g = t(lambda: _lib.gemm_f32(flat[:, :4 * d + 1], dzp[:, :d], gWp[:, :d], transpose_a=True, splits=32, accumulate=True))
print(f"dW with 16-byte loads (padded dz, padded gW): {g:.2f} us")
