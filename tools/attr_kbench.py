"""Per-kernel timings of the attribute-view step pieces (B = 5000, dim 75): `python tools/attr_kbench.py`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd import _lib

d, B = 75, 5000
dev = "cuda"
st = _lib.stride_for(d)
attr = torch.zeros(600, st, device=dev); attr[:, :d] = torch.randn(600, d, device=dev) * 0.1
lit = torch.zeros(100_000, st, device=dev); lit[:, :d] = torch.nn.functional.normalize(torch.randn(100_000, d, device=dev), dim=1)
ia = torch.randint(0, 600, (B,), device=dev, dtype=torch.int32); iv = torch.randint(0, 100_000, (B,), device=dev, dtype=torch.int32)
npar = 2 * d + 52 + 4 * d * d + d
params = torch.randn(npar, device=dev) * 0.1; gpar = torch.zeros(npar, device=dev)
flat = torch.empty(B, 4 * d, device=dev); dflat = torch.randn(B, 4 * d, device=dev) * 0.01
gattr = torch.zeros_like(attr); tattr = torch.zeros(600, dtype=torch.int32, device=dev)
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
import ctypes as C
P = lambda t: C.c_void_p(t.data_ptr())

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def fwd(): _lib.attr_conv_fwd(attr, False, lit, d, ia, iv, params, flat)
ws = torch.zeros(_lib.cnn_workspace_floats(d), device=dev)
def bwd(): _lib.attr_conv_bwd(attr, False, lit, d, ia, iv, params, dflat, gpar, gattr, tattr, 1, ws)
def bwd0(): _lib.attr_conv_bwd(attr, False, lit, d, ia, iv, params, dflat, gpar, gattr, tattr, 1, None)
print(f"conv_fwd {timeit(fwd):.1f} us")
print(f"conv_bwd (workspace) {timeit(bwd):.1f} us; direct {timeit(bwd0):.1f} us")
gpar.zero_(); bwd(); torch.cuda.synchronize(); a = gpar[:2 * d + 52].clone(); gpar.zero_(); bwd0(); torch.cuda.synchronize()
print("workspace vs direct param grads max rel diff", float(((a - gpar[:2 * d + 52]).abs() / (gpar[:2 * d + 52].abs() + 1e-6)).max()), "ws left zero:", float(ws.abs().max()) == 0.0)
W = params[2 * d + 52: 2 * d + 52 + 4 * d * d].view(4 * d, d)
z = torch.empty(B, d, device=dev); gout = torch.randn(B, d, device=dev); gW = torch.zeros(4 * d, d, device=dev)
def g1(): _lib.gemm_f32(flat, W, z)
def g2(): _lib.gemm_f32(flat.t(), gout, gW, splits=32, accumulate=True)
def g3(): _lib.gemm_f32(gout, W.t(), dflat)
for name, fn in (("z = flat W [5000x300x75]", g1), ("dW = flat^T g [300x5000x75] split 32", g2), ("dflat = g W^T [5000x75x300]", g3)):
    print(f"gemm {name}: {timeit(fn, 50):.1f} us (host-inclusive)")
