"""Literal auto-encoder training at the reference's shape (code/literal_encoder.py:26-31: 1500 -> 1024 -> 512 -> 75 and back,
batches of 5000): ms per batch of the native epoch call (mke_ae_train_steps: hand-written MFMA GEMMs with fused
epilogues, hand-derived backward, HIP Adagrad) and the TFLOP/s it amounts to (6 forward + 6 dW + 5 dA products)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multike_amd.literal_encoder import AutoEncoderModel
from multike_amd.synthetic import synthetic_args

L = int(os.environ.get("AE_ROWS", "50000"))
act = os.environ.get("AE_ACT", "thah")
g = torch.Generator(device="cpu"); g.manual_seed(0)
x = torch.randn(L, 1500, generator=g).numpy()
args = synthetic_args(dim=75, batch_size=5000, learning_rate=0.001, encoder_active=act, encoder_normalize=True, encoder_epoch=1)
m = AutoEncoderModel(x.reshape(L, 5, 300), args, seed=1)
with torch.no_grad():
    m.params.mul_(0.02)
d = [1500, 1024, 512, 75]
pairs = [(d[i], d[i + 1]) for i in range(3)] * 2
flop_fwd = sum(2.0 * 5000 * a * b for a, b in pairs)
flop = flop_fwd * 3 - 2.0 * 5000 * 1500 * 1024       # dW for all six, dA for all but the first encoder layer
with contextlib.redirect_stdout(io.StringIO()):
    m.train_one_epoch(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_ep = 5
    for e in range(n_ep):
        m.train_one_epoch(e + 1)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (n_ep * (L // 5000))
print(f"activation {act!r}: {dt * 1e3:.3f} ms per 5000-row training step = {flop / dt / 1e12:.1f} TFLOP/s "
      f"({100 * flop / dt / 157.3e12:.0f} % of the 157.3 TFLOP/s f32 matrix peak); {flop / 1e9:.1f} GFLOP per step")

# ---- the same step through the library: torch.nn.functional.linear (hipBLASLt) + autograd + torch.optim.Adagrad ----------------
# (what a straight PyTorch port of code/literal_encoder.py:63-107 costs on this part: library GEMMs, separate elementwise /
# reduction kernels, autograd bookkeeping; same shapes, same activation, row-normalised code, mean-squared reconstruction loss)
if os.environ.get("AE_LIBRARY", "1") == "1":
    dev = torch.device("cuda")
    dims = [1500, 1024, 512, 75, 512, 1024, 1500]
    Ws = [torch.nn.Parameter(0.02 * torch.randn(dims[i], dims[i + 1], device=dev)) for i in range(6)]
    bs = [torch.nn.Parameter(torch.zeros(dims[i + 1], device=dev)) for i in range(6)]
    opt = torch.optim.Adagrad(Ws + bs, lr=0.001, initial_accumulator_value=0.1, eps=0.0)
    xb = torch.randn(5000, 1500, device=dev)
    xb = xb / xb.norm(dim=1, keepdim=True)

    def step():
        opt.zero_grad(set_to_none=True)
        h = xb
        for i in range(6):
            h = torch.tanh(torch.addmm(bs[i], h, Ws[i]))
            if i == 2:
                h = torch.nn.functional.normalize(h, dim=1)
        loss = ((h - xb) ** 2).sum()
        loss.backward()
        opt.step()
        return loss
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dl = (time.perf_counter() - t0) / n
    print(f"the same step on torch autograd + the library's GEMMs + torch.optim.Adagrad: {dl * 1e3:.3f} ms per step "
          f"({flop / dl / 1e12:.1f} TFLOP/s); native / library = {dt / dl:.2f}")
