"""Attribute-view step only (B = 5000, dim 75), for rocprofv3 --kernel-trace: `python tools/attr_prof.py [steps] [B] [dim]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd.attr_cnn import AttrCNN
from multike_amd.tables import EmbeddingTable, StepEngine

from multike_amd import _lib
for kv in filter(None, os.environ.get("MKE_SET", "").split(",")):      # MKE_SET=option=value,...: A/B of a kernel choice
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d, B = (int(sys.argv[3]) if len(sys.argv) > 3 else 75), (int(sys.argv[2]) if len(sys.argv) > 2 else 5000)
E = EmbeddingTable(200_000, d, "av", seed=1); A = EmbeddingTable(600, d, "attr", normalize=False, seed=2, grad_copies=int(os.environ.get("ATTR_COPIES", "4")))
lit = np.random.default_rng(0).standard_normal((100_000, d)).astype(np.float32); lit /= np.linalg.norm(lit, axis=1, keepdims=True)
L = EmbeddingTable(100_000, d, "lit", normalize=False, trainable=False, values=lit)
cnn = AttrCNN(d, seed=3); eng = StepEngine()
g = torch.Generator(device="cuda"); g.manual_seed(0)
ZIPF = float(os.environ.get("ATTR_ZIPF", "0"))       # attribute ids ~ rank^-ZIPF (real attribute frequencies are heavy-tailed: a label-like attribute is in a large share of the triples)
def batch():
    if ZIPF > 0:
        pr = torch.arange(1, 601, device="cuda", dtype=torch.float64) ** (-ZIPF)
        ia = torch.multinomial((pr / pr.sum()).float(), B, replacement=True, generator=g).to(torch.int32)
    else:
        ia = torch.randint(0, 600, (B,), device="cuda", generator=g, dtype=torch.int32)
    return (torch.randint(0, 200_000, (B,), device="cuda", generator=g, dtype=torch.int32), ia,
            torch.randint(0, 100_000, (B,), device="cuda", generator=g, dtype=torch.int32), torch.rand(B, device="cuda", generator=g))
bs = [batch() for _ in range(8)]
for i in range(10): cnn.step(eng, E, A, L, *bs[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps): cnn.step(eng, E, A, L, *bs[i % 8])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"attribute-view CNN step (B={B}, dim={d}): {dt*1e6:.1f} us/step -> {B/dt/1e6:.1f} M triples/s")

# ---- the same step as a straight PyTorch program (gathers, F.conv2d, matmul, autograd, sparse Adagrad by index) ----------------
# What code/MultiKE_model.py:34-63,134-151 costs on this part when every op is a library / ATen kernel: the step is a chain of
# ~100 small launches.  Same shapes and arithmetic (BN affine, two SAME 2x4 convolutions with tanh, width normalisation, dense
# layer, batch-wide normalisation, weighted softplus loss, Adagrad on the CNN parameters and on the touched rows).
if os.environ.get("ATTR_LIBRARY", "1") == "1":
    import torch.nn.functional as F
    dev = torch.device("cuda")
    Et = torch.nn.Parameter(E.raw().clone()); At = torch.nn.Parameter(A.raw().clone()); Lt = L.raw().clone()
    P = {k: torch.nn.Parameter(torch.as_tensor(v, dtype=torch.float32, device=dev)) for k, v in cnn.numpy_params().items()}
    acc = {id(p): torch.full_like(p, 0.1) for p in [Et, At] + list(P.values())}

    def tstep(ih, ia, iv, w):
        ih, ia, iv = ih.long(), ia.long(), iv.long()
        th = F.normalize(Et[ih], dim=1); ta = At[ia]; tv = Lt[iv]
        x = torch.stack([ta, tv], 1) * (P["gamma"] / np.sqrt(1.0 + 1e-3)) + P["beta"]
        x = x[:, None]
        for K, b in ((P["K1"], P["b1"]), (P["K2"], P["b2"])):
            x = torch.tanh(F.conv2d(F.pad(x, (1, 2, 0, 1)), K.permute(3, 2, 0, 1), b))
        x = x.permute(0, 2, 3, 1)
        x = x * torch.rsqrt(torch.clamp_min((x * x).sum(2, keepdim=True), 1e-12))
        z = torch.tanh(x.reshape(x.shape[0], -1) @ P["W"] + P["bias"])
        o = z * torch.rsqrt(torch.clamp_min((z * z).sum(), 1e-12))
        loss = (F.softplus(((th - o) ** 2).sum(1)) * w).sum()
        params = [Et, At] + list(P.values())
        grads = torch.autograd.grad(loss, params)
        with torch.no_grad():
            for p_, g_ in zip(params, grads):          # dense Adagrad (the reference's semantics); rows with zero gradient do not move
                a_ = acc[id(p_)]
                a_.addcmul_(g_, g_)
                p_.addcdiv_(g_, a_.sqrt(), value=-0.001)
        return loss
    for i in range(3): tstep(*bs[i % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for i in range(n): tstep(*bs[i % 8])
    torch.cuda.synchronize(); dl = (time.perf_counter() - t0) / n
    print(f"the same step as a straight PyTorch program (ATen / library kernels, autograd, dense Adagrad): {dl*1e6:.0f} us/step; native / PyTorch = {dt/dl:.3f}")
