"""Do the two phase groups of an epoch sit on different hardware queues?  _train_views with the side stream taken from a pool of
streams (HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues round-robin: a side stream that lands on the default stream's
queue serialises the groups), with stream priorities, and on one stream.  python tools/overlap_queues.py"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from multike_amd.MultiKE_CSL import MultiKE_CV
from multike_amd.synthetic import SyntheticData, synthetic_args
data = SyntheticData(n_ent=200_000, n_rel=550, n_attr=600, n_values=100_000, dim=75, link_share=0.3, seed=5)
args = synthetic_args(dim=75, max_epoch=3, start_valid=10 ** 6, neg_sampling="uniform", start_predicate_soft_alignment=0)
m = MultiKE_CV(data, args, data.predicate_align_model)
m._prepare()
def epoch_ms(main=None, side=None, n=5):
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
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[1:])
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
print(f"default stream + new side stream: {epoch_ms():.2f} ms")
streams = [torch.cuda.Stream() for _ in range(8)]
for i in (0, 1, 2, 3, 5, 7):
    print(f"default stream + side = stream #{i}: {epoch_ms(None, streams[i]):.2f} ms")
for a, b in ((0, 1), (0, 2), (1, 3), (2, 5)):
    print(f"main = stream #{a}, side = stream #{b}: {epoch_ms(streams[a], streams[b]):.2f} ms")
hi = torch.cuda.Stream(priority=-1); lo = torch.cuda.Stream(priority=0)
print(f"main = low priority, side = high: {epoch_ms(lo, hi):.2f} ms;  main = high, side = low: {epoch_ms(hi, lo):.2f} ms")
m.overlap_views = False
ts = []
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        m._train_views(9)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"one stream: {min(ts):.2f} ms")
