"""Next epoch's negatives sampled on a side stream that may only use a FEW compute units (hipExtStreamCreateWithCUMask) while the
current epoch trains: does the sampler then trickle along under the training kernels instead of starving them (prefetch on an
unrestricted side stream measured slower than in-line sampling)?   python tools/prefetch_masked.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd.runner import RelationViewRunner
from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
from multike_amd.synthetic import SyntheticKGs
from multike_amd.tables import EmbeddingTable

hip = C.CDLL("libamdhip64.so")
kgs = SyntheticKGs(n_ent=200_000, n_rel=550, seed=1234)
sides = []
for k in (0, 1):
    t = torch.as_tensor(kgs.triples[k], device="cuda")
    sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(8), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def run(mask_bits, n_epochs=6):
    E = EmbeddingTable(kgs.entities_num, 75, "e", seed=1); R = EmbeddingTable(kgs.relations_num, 75, "r", seed=2)
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], 5000, 25, seed=1)
    r = RelationViewRunner(E, R, bat, "relation", lr=0.001)
    pf = mask_bits is not None
    r.run_epochs(2, prefetch=pf)                      # creates the side stream machinery
    if pf and mask_bits != "all":
        r._side = masked_stream(mask_bits)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.run_epochs(n_epochs, prefetch=pf)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n_epochs * r.steps) * 1e6


every = lambda k: sum(1 << i for i in range(0, 256, k))
cases = [("in-line sampler (default)", None), ("prefetch, unrestricted side stream", "all"),
         ("prefetch, 8 CUs (every 32nd)", every(32)), ("prefetch, 16 CUs (every 16th)", every(16)), ("prefetch, 32 CUs (every 8th)", every(8)),
         ("prefetch, 64 CUs (every 4th)", every(4)), ("prefetch, CUs 0-31", (1 << 32) - 1), ("in-line sampler (default)", None)]
for name, m in cases:
    print(f"{name:40s}: {run(m):6.2f} us/step", flush=True)
