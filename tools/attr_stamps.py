"""Where the forward convolution + dense launch spends its time: s_memtime stamps of every wavefront of a scratch build
(tools/ab/libstamp.so: mke_attr_cnn.hip with STAMP(i) lines, not in the tree) after a few warm steps.  Timing only."""
import ctypes, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, "multike_amd", "libmultike_hip.so"), "/tmp/keep.so")
shutil.copy(os.path.join(ROOT, "tools", "ab", "libstamp.so"), os.path.join(ROOT, "multike_amd", "libmultike_hip.so"))
try:
    import numpy as np, torch
    os.environ["ATTR_LIBRARY"] = "0"
    sys.argv = [sys.argv[0], "50"] + sys.argv[1:]
    exec(open(os.path.join(ROOT, "tools", "attr_prof.py")).read())
    from multike_amd import _lib
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 16))()
    _lib.lib().mke_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    rc = _lib.lib().mke_debug_stamps(buf)
    st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
    nw = min(1024, ((B + 15) // 16) * 4)
    st = st[:nw]
    t0 = st[:, 0].min()
    bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd" or os.environ.get("STAMPS") == "bwd"
    names = ["enter", "W fragment loads issued", "conv params / gamma / beta loaded", "ids arrived", "rows arrived (+ attr norm)", "x strips staged",
             "conv1 done", "conv2 + width norms done", "flat stored (LDS + global issue)", "block barrier passed", "MFMA done", "s_acc exchanged (barrier)",
             "tanh + z stored", "block sum done"]
    if bwd:
        names = ["enter", "conv params / gamma / beta loaded, s_part zeroed", "ids arrived", "rows arrived (+ attr norm)", "x strips staged", "conv1 (recomputed)",
                 "conv2 + width norms (recomputed)", "width-norm backward, conv2 parameter gradients", "conv2 transposed, conv1 parameter gradients",
                 "conv1 transposed, dx, attribute-row scatter", "block barrier passed", "gamma / beta through LDS (2 barriers)", "parameter atomics issued"]
        nw = min(1024, ((B + 3) // 4) * 2)
        st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)[:nw]
        t0 = st[:, 0].min()
    print(f"rc {rc}; {nw} wavefronts")
    print("stage | median cycles since the wavefront entered | median delta | first wavefront to reach it (since first enter) | last")
    prev = None
    for i, n in enumerate(names):
        rel = st[:, i] - st[:, 0]
        dlt = "" if prev is None else f"{int(np.median(st[:, i] - st[:, prev]))}"
        print(f"{i:2d} {n:45s} {int(np.median(rel)):7d} {dlt:>7s} {int(st[:, i].min() - t0):8d} {int(st[:, i].max() - t0):8d}")
        prev = i
    print("wavefront entry times since the first (cycles): p10 / p50 / p90 / max", [int(x) for x in np.percentile(st[:, 0] - t0, [10, 50, 90, 100])])
finally:
    shutil.copy("/tmp/keep.so", os.path.join(ROOT, "multike_amd", "libmultike_hip.so"))
