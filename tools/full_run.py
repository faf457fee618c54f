"""The reference's run_ITC.py / run_SSL.py flow at DBP-WD-100K scale on a synthetic dataset folder, timed phase by phase.
python tools/full_run.py [n_pairs] [max_epoch] [ITC|SSL]"""
import os, sys, tempfile, time, contextlib, io, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multike_amd.data_model import DataModel
from multike_amd.MultiKE_CSL import MultiKE_CV
from multike_amd.MultiKE_Late import MultiKE_Late
from multike_amd.predicate_alignment import PredicateAlignModel
from multike_amd.synthetic import synthetic_args, write_dataset_folder
import torch

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
method = sys.argv[3] if len(sys.argv) > 3 else "ITC"
folder = tempfile.mkdtemp() + "/"
t = time.time()
wf = write_dataset_folder(folder, n_pairs=n_pairs, n_extra=n_pairs // 20, n_rel=300, n_attr=300, triples_per_entity=4.4, shared_structure=0.8)
print(f"folder written in {time.time() - t:.1f}s")
args = synthetic_args(training_data=folder, output=folder + "out/", word2vec_path=wf, dataset_division="631/", encoder_epoch=100,
                      encoder_active="tanh", encoder_normalize=True, retrain_literal_embeds=False, literal_normalize=True, is_save=True,
                      max_epoch=epochs, shared_learning_max_epoch=epochs, start_valid=100, eval_freq=10)
quiet = contextlib.redirect_stdout(io.StringIO())
prof = None
if os.environ.get("FULL_RUN_PROFILE") == "1":       # cProfile of the host side: where DataModel and run() spend their wall time
    import cProfile, pstats
    prof = cProfile.Profile()
t = time.time()
with quiet:
    if prof: prof.enable()
    data = DataModel(args)
    if prof: prof.disable()
if prof:
    pstats.Stats(prof).sort_stats("cumulative").print_stats(28)
    prof = cProfile.Profile()
torch.cuda.synchronize(); t_data = time.time() - t
t = time.time()
with quiet:
    pam = PredicateAlignModel(data.kgs, args)
t_pam = time.time() - t
k = data.kgs
print(f"DataModel {t_data:.1f}s (entities {k.entities_num}, relation triples {k.kg1.relation_triples_num}+{k.kg2.relation_triples_num}, attribute triples "
      f"{k.kg1.attribute_triples_num}+{k.kg2.attribute_triples_num}, literals {len(data.literal_list)}); PredicateAlignModel {t_pam:.1f}s")
t = time.time()
buf = io.StringIO()
wall = collections.OrderedDict()
with contextlib.redirect_stdout(buf):
    cprof = None
    if os.environ.get("FULL_RUN_TIMELINE") == "1":
        import cProfile, pstats
        cprof = cProfile.Profile(); cprof.enable()
    model = (MultiKE_CV if method == "ITC" else MultiKE_Late)(data, args, pam)
    if os.environ.get("FULL_RUN_HOST_REFRESH") == "1":     # A/B: the predicate refresh's per-triple work on the host (as before round 4)
        pam.device = None
    torch.cuda.synchronize(); t_ctor = time.time() - t
    if cprof:
        cprof.disable()
    if os.environ.get("FULL_RUN_TIMELINE") == "1":   # host wall time inside each call of run()'s loop (no synchronisation added)
        def timed(name):
            f = getattr(model, name)
            def g(*a_, **k_):
                t0 = time.time()
                try:
                    return f(*a_, **k_)
                finally:
                    e = wall.setdefault(name, [0, 0.0, 0.0]); dt = time.time() - t0
                    e[0] += 1; e[1] += dt; e[2] = max(e[2], dt)
            setattr(model, name, g)
        for nm in ("_prepare", "_test", "_train_views", "train_common_space_learning_1epo", "_valid", "_update_predicate_alignment",
                   "_finish_predicate_update", "_refresh_neighbours", "save", "_join_save", "train_relation_view_1epo",
                   "train_attribute_view_1epo", "train_cross_kg_entity_inference_relation_view_1epo",
                   "train_cross_kg_entity_inference_attribute_view_1epo", "train_cross_kg_relation_inference_1epo",
                   "train_cross_kg_attribute_inference_1epo"):
            if hasattr(model, nm):
                timed(nm)
    if prof: prof.enable()
    res = model.run()
    if prof: prof.disable()
torch.cuda.synchronize()
if prof:
    pstats.Stats(prof).sort_stats("tottime").print_stats(40)
print(f"{type(model).__name__}: constructor {t_ctor:.2f}s + run() {time.time() - t - t_ctor:.2f}s = {time.time() - t:.1f}s for {epochs} epochs (validation from epoch 100 every 10, k-NN refresh every 20, predicate refresh every 10, final save + 4 tests)")
print("test Hits@1:", {k_: round(float(v), 3) for k_, v in res.items()})
log = buf.getvalue().splitlines()
import re
if cprof:
    print("the constructor:"); pstats.Stats(cprof).sort_stats("cumulative").print_stats(22)
if wall:
    print("host wall time by call of run()'s loop: " + " | ".join(f"{k} {v[0]} x, {v[1]:.3f} s (max {v[2] * 1e3:.0f} ms)" for k, v in wall.items()))
agg = collections.OrderedDict()
for l in log:
    m = re.match(r"epoch \d+ of (.*?), avg\. loss: .*?time: ([0-9.]+)s", l)
    if m:
        agg[m.group(1)] = agg.get(m.group(1), 0.0) + float(m.group(2))
    m = re.match(r"generating neighbors of .* costs ([0-9.]+) s", l)
    if m:
        agg["k-NN refresh"] = agg.get("k-NN refresh", 0.0) + float(m.group(1))
    m = re.search(r"quick results: .*time = ([0-9.]+) s", l)
    if m:
        agg["valid/test ranking"] = agg.get("valid/test ranking", 0.0) + float(m.group(1))
print("ranking times (s):", [float(re.search(r"time = ([0-9.]+) s", l).group(1)) for l in log if "results: hits@" in l])
print("time by phase (s):", {k_: round(v, 2) for k_, v in agg.items()}, "| sum", round(sum(agg.values()), 2))
print("epochs executed:", sum(1 for l in log if l.startswith("epoch ") and l.rstrip().endswith(":")), "| k-NN refreshes:", sum(1 for l in log if "neighbors of" in l))
print(" | ".join([l.split("costs ")[1] for l in log if "neighbors of" in l]))
