"""Dense f32 GEMM rates at the literal auto-encoder's shapes (M = 5000 rows per batch; code/literal_encoder.py:63-91):
the hand-written MFMA kernels (mke_gemm_f32 / mke_dense_* ) beside the library's (torch.matmul -> hipBLASLt / rocBLAS),
HIP events around 20 back-to-back launches each.  Peak: 157.3 TFLOP/s f32 on the matrix cores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd import _lib

M = 5000
SHAPES = [("fwd  enc0  X W", M, 1024, 1500, False, False), ("fwd  enc1", M, 512, 1024, False, False), ("fwd  enc2", M, 75, 512, False, False),
          ("fwd  dec0", M, 512, 75, False, False), ("fwd  dec2", M, 1500, 1024, False, False),
          ("dW   enc0  X^T dZ", 1500, 1024, M, True, False), ("dW   enc1", 1024, 512, M, True, False), ("dW   enc2", 512, 75, M, True, False),
          ("dA   enc1  dZ W^T", M, 1024, 512, False, True), ("dA   dec2", M, 1024, 1500, False, True), ("dA   enc2", M, 512, 75, False, True)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda"); g.manual_seed(0)
print("| product | M | N | K | ours us | ours TF/s | % of 157 | library us | library TF/s |")
print("|---|---|---|---|---|---|---|---|---|")
for name, m, n, k, ta, tb in SHAPES:
    # rows padded to 16 bytes, as the auto-encoder stores its weights and activations (mke_ae_plan)
    p4 = lambda v: (v + 3) // 4 * 4
    mk = lambda r, c: torch.randn(r, p4(c), device="cuda", generator=g)[:, :c]
    a = mk(k, m) if ta else mk(m, k)
    b = mk(n, k) if tb else mk(k, n)
    out = torch.zeros(m, p4(n), device="cuda")[:, :n]
    A, B = (a.t() if ta else a), (b.t() if tb else b)
    fl = 2.0 * m * n * k
    splits = int(os.environ.get("SPLITS", "1")) if not ta else max(1, min(64, (256 * 4) // (((m + 63) // 64) * ((n + 63) // 64))))
    t_ours = timeit(lambda: _lib.gemm_f32(a, b, out, transpose_a=ta, transpose_b=tb, splits=splits, accumulate=splits > 1))
    Ac, Bc, outc = A.contiguous(), B.contiguous(), torch.zeros(m, n, device="cuda")
    t_lib = timeit(lambda: torch.matmul(Ac, Bc, out=outc))
    ref = A.double() @ B.double()
    out.zero_(); _lib.gemm_f32(a, b, out, transpose_a=ta, transpose_b=tb, splits=splits, accumulate=splits > 1)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print(f"| {name} (splits {splits}, err {err:.1e}) | {m} | {n} | {k} | {t_ours:.1f} | {fl / t_ours / 1e6:.1f} | {100 * fl / t_ours / 1e6 / 157.3:.0f} | {t_lib:.1f} | {fl / t_lib / 1e6:.1f} |")
