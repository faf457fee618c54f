"""Experiment: the two-stream ITC epoch with the RELATION group's stream restricted to a subset of the compute units
(hipExtStreamCreateWithCUMask), the attribute group (the epoch's critical path: latency chains on <= 5,000 wavefronts) unrestricted.
python tools/epoch_masked.py [n_ent]   -> ms per epoch for every mask tried"""
import os, sys, time, ctypes, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd.MultiKE_CSL import MultiKE_CV
from multike_amd.synthetic import SyntheticData, synthetic_args

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(8), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


n_ent = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
data = SyntheticData(n_ent=n_ent, n_rel=550, n_attr=600, n_values=100_000, dim=75, link_share=0.3, seed=5)
args = synthetic_args(dim=75, max_epoch=3, start_valid=10 ** 6, neg_sampling="uniform", start_predicate_soft_alignment=0)
m = MultiKE_CV(data, args, data.predicate_align_model)
m._prepare()
m.overlap_views = True


def epochs(stream, n=6):
    out = []
    for i in range(1, n + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            if stream is None:
                m._train_views(i)
            else:
                stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stream):
                    m._train_views(i)
                torch.cuda.current_stream().wait_stream(stream)
            m.train_common_space_learning_1epo(i, m._entity_list)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
    return out


print("unrestricted:", " ".join(f"{x:.1f}" for x in epochs(None)))
for n in (224, 192, 160, 128, 96):
    low = set(range(n))                                             # the low n bits
    per = {w * 32 + b for w in range(8) for b in range(n // 8)}     # the first n / 8 bits of every 32-bit word
    for name, bits in (("low bits", low), ("n/8 per word", per)):
        print(f"relation stream on {n} CUs ({name}):", " ".join(f"{x:.1f}" for x in epochs(masked_stream(bits))))
print("unrestricted:", " ".join(f"{x:.1f}" for x in epochs(None)))
