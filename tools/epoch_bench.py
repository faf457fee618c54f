"""Whole ITC epochs (MultiKE_CV schedule) at C2-synth scale: where does an epoch's wall time go?
python tools/epoch_bench.py [n_ent] [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multike_amd.MultiKE_CSL import MultiKE_CV
from multike_amd.synthetic import SyntheticData, synthetic_args

n_ent = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.time()
data = SyntheticData(n_ent=n_ent, n_rel=550, n_attr=600, n_values=100_000, dim=75, link_share=0.3, seed=5)
print(f"synthetic data built in {time.time() - t0:.1f}s: rel triples {data.kgs.kg1.local_relation_triples_num + data.kgs.kg2.local_relation_triples_num}, "
      f"attr triples {data.kgs.kg1.local_attribute_triples_num + data.kgs.kg2.local_attribute_triples_num}")
args = synthetic_args(dim=75, max_epoch=epochs, start_valid=10 ** 6, neg_sampling="uniform", start_predicate_soft_alignment=0)
m = MultiKE_CV(data, args, data.predicate_align_model)
m._prepare()
phases = [("relation view", lambda i: m.train_relation_view_1epo(i, m._rel_steps, m._rel_tasks, None, None, None)),
          ("ckge relation", lambda i: m.train_cross_kg_entity_inference_relation_view_1epo(i, m._ckge_rel_triples)),
          ("ckgp relation", lambda i: m.train_cross_kg_relation_inference_1epo(i, m._ckgp_rel_triples)),
          ("attribute view", lambda i: m.train_attribute_view_1epo(i, m._attr_steps, m._attr_tasks, None, None, None)),
          ("ckge attribute", lambda i: m.train_cross_kg_entity_inference_attribute_view_1epo(i, m._ckge_attr_triples)),
          ("ckga attribute", lambda i: m.train_cross_kg_attribute_inference_1epo(i, m._ckga_attr_triples)),
          ("common space", lambda i: m.train_common_space_learning_1epo(i, m._entity_list))]
import contextlib, io
for ov in (False, True):
    m.overlap_views = ov
    for i in range(1, 4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            m._train_views(i); m.train_common_space_learning_1epo(i, m._entity_list)
        torch.cuda.synchronize(); print(f"driver epoch (overlap_views={ov}): {(time.perf_counter() - t0) * 1e3:.1f} ms")
for i in range(1, epochs + 1):
    tot = 0.0
    line = []
    for name, fn in phases:
        torch.cuda.synchronize(); t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            fn(i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        tot += dt; line.append(f"{name} {dt * 1e3:.1f}")
    print(f"epoch {i}: {tot * 1e3:.1f} ms  | " + " | ".join(line) + " (ms)")
