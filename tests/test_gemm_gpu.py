"""The hand-written f32 MFMA GEMM (mke_gemm_f32) against float64 matmul: every operand orientation the attribute step
uses, ragged sizes, split-K accumulation."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,ta,tb,splits", [(5000, 75, 300, False, False, 1), (300, 75, 5000, True, False, 32),
                                                (5000, 300, 75, False, True, 1), (1, 1, 1, False, False, 1),
                                                (65, 33, 17, True, True, 1), (128, 64, 64, False, False, 4),
                                                (37, 300, 1024, True, False, 7),
                                                # 16-byte-load kernel (k_gemm_vec), the four operand layouts, ragged M / N / K edges
                                                (5000, 1024, 1500, False, False, 1), (1500, 1024, 5000, True, False, 4),
                                                (5000, 1024, 512, False, True, 1), (516, 260, 1028, True, True, 3),
                                                (68, 72, 36, False, False, 1), (4, 4, 4, True, True, 1)])
def test_matches_float64(M, N, K, ta, tb, splits):
    from multike_amd import _lib
    g = torch.Generator(device="cuda"); g.manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g)
    b = torch.randn((N, K) if tb else (K, N), device="cuda", generator=g)
    out = torch.zeros(M, N, device="cuda")
    _lib.gemm_f32(a, b, out, transpose_a=ta, transpose_b=tb, splits=splits, accumulate=splits > 1)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    scale = float(ref.abs().max()) + 1e-30
    assert float((out.double() - ref).abs().max()) / scale < 5e-6
    # accumulate on top of an existing C
    base = torch.randn(M, N, device="cuda", generator=g)
    out2 = base.clone()
    _lib.gemm_f32(a, b, out2, transpose_a=ta, transpose_b=tb, splits=splits, accumulate=True)
    assert float((out2.double() - (base.double() + ref)).abs().max()) / scale < 5e-6


def test_asymmetric_operand_catches_transposition():
    """A = I check with an ASYMMETRIC B (guide: a symmetric B would hide a row/col swap in the C write)."""
    from multike_amd import _lib
    n = 96
    b = torch.arange(n * n, dtype=torch.float32, device="cuda").reshape(n, n) / 1000.0
    out = torch.empty(n, n, device="cuda")
    _lib.gemm_f32(torch.eye(n, device="cuda"), b, out)
    assert torch.equal(out, b)
