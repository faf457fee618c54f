import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd import _lib
d, B = 75, 5000
flat = torch.randn(B, 4 * d + 4, device="cuda"); W = torch.randn(4 * d + 1, d, device="cuda"); z = torch.empty(B, d, device="cuda")
for dbg in (0, 1, 2, 4, 8, 16, 3, 7, 15, 31):
    _lib.set_option("cnn_debug", dbg)
    for _ in range(20 + dbg): _lib.gemm_f32(flat[:, :4 * d + 1], W, z)   # call count encodes the variant
    torch.cuda.synchronize()
