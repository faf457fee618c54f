"""The two phase groups of an ITC epoch (relation group / attribute group, disjoint state) on two streams that each own a share of
the compute units (hipExtStreamCreateWithCUMask) instead of sharing all of them: does isolation buy back what the two latency
chains lose to each other (C2-synth: 8.6 + 11.2 ms alone, 16.9 ms together, 12.3 if they overlapped perfectly)?
python tools/overlap_masked.py"""
import contextlib, ctypes as C, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd.MultiKE_CSL import MultiKE_CV
from multike_amd.synthetic import SyntheticData, synthetic_args

hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(8), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


data = SyntheticData(n_ent=200_000, n_rel=550, n_attr=600, n_values=100_000, dim=75, link_share=0.3, seed=5)
neg = int(sys.argv[1]) if len(sys.argv) > 1 else 25
args = synthetic_args(dim=75, max_epoch=3, start_valid=10 ** 6, neg_sampling="uniform", start_predicate_soft_alignment=0, neg_triple_num=neg)
m = MultiKE_CV(data, args, data.predicate_align_model)
m._prepare()


def epoch_ms(main=None, side=None, n=4):
    m.overlap_views = True
    m._side_stream = side
    ts = []
    for i in range(1, n + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            with (torch.cuda.stream(main) if main is not None else contextlib.nullcontext()):
                m._train_views(i)
            if main is not None:
                torch.cuda.current_stream().wait_stream(main)
            m.train_common_space_learning_1epo(i, m._entity_list)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[1:])


# bit i of the mask = compute unit i of the device's enumeration; `every` spreads a share evenly over that enumeration
def share(num, den, phase=0):      # num of every den compute units
    return sum(1 << i for i in range(256) if (i + phase) % den < num)

full = (1 << 256) - 1
print(f"neg_triple_num {neg}; two unrestricted streams (the product's default): {epoch_ms():.2f} ms")
m.overlap_views = False
torch.cuda.synchronize(); t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    m._train_views(9); m.train_common_space_learning_1epo(9, m._entity_list)
torch.cuda.synchronize(); print(f"one stream: {(time.perf_counter() - t0) * 1e3:.2f} ms")
for name, rel_bits, attr_bits in (
        ("both streams created with a FULL mask", full, full),
        ("relation 128 / attribute 128 (alternating units)", share(1, 2), share(1, 2, 1)),
        ("relation 192 / attribute 64 (3 of 4 / 1 of 4)", share(3, 4), share(1, 4, 1)),
        ("relation 160 / attribute 96 (5 of 8 / 3 of 8)", share(5, 8), share(3, 8, 3)),
        ("relation 128 / attribute 128 (units 0-127 / 128-255)", (1 << 128) - 1, full ^ ((1 << 128) - 1)),
        ("relation all / attribute 128 (alternating)", full, share(1, 2, 1)),
        ("relation 192 / attribute all", share(3, 4), full)):
    a, b = masked_stream(rel_bits), masked_stream(attr_bits)
    print(f"{name:60s}: {epoch_ms(a, b):.2f} ms", flush=True)
print(f"two unrestricted streams again: {epoch_ms():.2f} ms")
