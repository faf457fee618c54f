"""Differential sweep of the on-disk inputs — BUILD CONTAINER ONLY (it imports the reference from
/root/reference/code, under the same stand-in modules tests/golden/make_golden.py installs; nothing of the reference is stored):
random dataset folders from `multike_amd.synthetic.write_dataset_folder` (sizes, predicate counts, density, shared structure,
seeds) are read by the reference's readers / id assignment / KG containers / literal clean-up / local names / predicate alignment
(initial + refreshed on random embeddings) and by this package's; every structure must be equal (floats to 1e-12).
python tools/fuzz_readers.py [cases] [seed]"""
import contextlib, importlib, io, os, shutil, sys, tempfile, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference/code"
if not os.path.isdir(REF):
    sys.exit("reference not found (this script only runs in the build container)")
import numpy as np
import make_golden as mg

sys.path.insert(0, REF)
mg.install_tf_forwarder(); mg.install_empty_standins()
ref_kgs, ref_utils, ref_pa = (importlib.import_module(m) for m in ("base.kgs", "utils", "predicate_alignment"))
sys.modules["Levenshtein"].ratio = mg._py_ratio
from multike_amd import predicate_alignment as pa
from multike_amd import utils as ut
from multike_amd.base import kgs as our_kgs
from multike_amd.synthetic import write_dataset_folder


def close(a, b, path=""):
    if isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), f"{path}: keys differ"
        for k in a:
            close(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), f"{path}: length {len(a)} vs {len(b)}"
        for i, (x, y) in enumerate(zip(a, b)):
            close(x, y, f"{path}[{i}]")
    elif isinstance(a, float) or isinstance(b, float):
        assert abs(a - b) <= 1e-12, f"{path}: {a} vs {b}"
    else:
        assert a == b, f"{path}: {a!r} vs {b!r}"


def kg_snapshot(k):
    e = {"ent_ids1": k.kg1.entities_id_dict, "ent_ids2": k.kg2.entities_id_dict, "rel_ids1": k.kg1.relations_id_dict,
         "rel_ids2": k.kg2.relations_id_dict, "attr_ids1": k.kg1.attributes_id_dict, "attr_ids2": k.kg2.attributes_id_dict,
         "train_links": [list(x) for x in k.train_links], "valid_links": [list(x) for x in k.valid_links],
         "test_links": [list(x) for x in k.test_links], "nums": [k.entities_num, k.relations_num, k.attributes_num],
         "useful1": list(k.useful_entities_list1), "useful2": list(k.useful_entities_list2),
         "train_entities1": list(k.train_entities1), "test_entities2": list(k.test_entities2), "valid_entities1": list(k.valid_entities1)}
    for i, kg in ((1, k.kg1), (2, k.kg2)):
        e[f"local_rel{i}"] = sorted(list(x) for x in kg.local_relation_triples_list)
        e[f"local_set{i}"] = sorted(list(x) for x in kg.local_relation_triples_set)
        e[f"nums{i}"] = [kg.relation_triples_num, kg.local_relation_triples_num, kg.attribute_triples_num, kg.local_attribute_triples_num,
                         kg.entities_num, kg.relations_num, kg.attributes_num]
        e[f"sup_rel{i}"] = sorted(list(x) for x in (kg.sup_relation_triples_list or []))
        e[f"sup_attr{i}"] = sorted(list(x) for x in (kg.sup_attribute_triples_list or []))
        e[f"local_attr{i}"] = sorted(list(x) for x in kg.local_attribute_triples_list)
        e[f"entities_list{i}"] = list(kg.entities_list)
    return e


def pam_snapshot(p):
    return {"relation_alignment_set": sorted(p.relation_alignment_set), "attribute_alignment_set": sorted(p.attribute_alignment_set),
            "relation_latent": sorted([a, b, s] for (a, b), s in p.relation_latent_match_pairs_similarity_dict_init.items()),
            "attribute_latent": sorted([a, b, s] for (a, b), s in p.attribute_latent_match_pairs_similarity_dict_init.items()),
            "sup_rel1": sorted(p.sup_relation_alignment_triples1), "sup_rel2": sorted(p.sup_relation_alignment_triples2),
            "sup_attr1": sorted(p.sup_attribute_alignment_triples1), "sup_attr2": sorted(p.sup_attribute_alignment_triples2),
            "rel_w1": sorted(p.relation_triples_w_weights1), "rel_w2": sorted(p.relation_triples_w_weights2),
            "attr_w1": sorted(p.attribute_triples_w_weights1), "attr_w2": sorted(p.attribute_triples_w_weights2),
            "train_relations1": sorted(p.train_relations1), "train_relations2": sorted(p.train_relations2),
            "train_attributes1": sorted(p.train_attributes1), "train_attributes2": sorted(p.train_attributes2)}


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    kw = dict(n_pairs=int(rng.integers(5, 160)), n_extra=int(rng.integers(0, 20)), n_rel=int(rng.integers(1, 14)), n_attr=int(rng.integers(2, 14)),
              seed=int(rng.integers(0, 10**6)), triples_per_entity=float(rng.uniform(0.5, 5.0)), shared_structure=float(rng.choice([0.0, 0.5, 0.9])))
    folder = tempfile.mkdtemp(prefix="mke_fuzz_") + "/"
    msg = ""
    try:
        write_dataset_folder(folder, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            for mode in ("swapping", "mapping", "sharing"):
                close(kg_snapshot(ref_kgs.read_kgs_from_folder(folder, "631/", mode, True)),
                      kg_snapshot(our_kgs.read_kgs_from_folder(folder, "631/", mode, True)), mode)
            kr = ref_kgs.read_kgs_from_folder(folder, "631/", "swapping", True)
            ko = our_kgs.read_kgs_from_folder(folder, "631/", "swapping", True)
            for i, (a, b) in enumerate(((kr.kg1, ko.kg1), (kr.kg2, ko.kg2))):
                ta, na, sa = ref_utils.clear_attribute_triples(a.local_attribute_triples_list)
                tb, nb, sb = ut.clear_attribute_triples(b.local_attribute_triples_list)
                close([sorted(ta), sorted(na), sorted(sa)], [sorted(list(x) if isinstance(x, tuple) else x for x in tb), sorted(nb), sorted(sb)]
                      if False else [sorted(tb), sorted(nb), sorted(sb)], f"clear{i}")
            close(ref_utils.read_local_name(folder, set(kr.kg1.entities_id_dict), set(kr.kg2.entities_id_dict)),
                  ut.read_local_name(folder, set(ko.kg1.entities_id_dict), set(ko.kg2.entities_id_dict)), "local_names")
            sims = (float(rng.choice([0.6, 0.8, 0.9, 0.95])), float(rng.choice([0.5, 0.7, 0.85])))
            args = types.SimpleNamespace(training_data=folder, predicate_init_sim=sims[0], predicate_soft_sim=sims[1])
            pr, po = ref_pa.PredicateAlignModel(kr, args), pa.PredicateAlignModel(ko, args)
            close(pam_snapshot(pr), pam_snapshot(po), "pam.init")
            er = np.random.default_rng(kw["seed"])
            rel_embed, attr_embed = er.standard_normal((kr.relations_num, 8)), er.standard_normal((kr.attributes_num, 8))
            for (p1, p2, _) in sorted(pr.relation_alignment_set_init)[:3]:
                rel_embed[kr.kg2.relations_id_dict[p2]] = rel_embed[kr.kg1.relations_id_dict[p1]] + 0.05
            for (p1, p2, _) in sorted(pr.attribute_alignment_set_init)[:2]:
                attr_embed[kr.kg2.attributes_id_dict[p2]] = attr_embed[kr.kg1.attributes_id_dict[p1]] + 0.05
            raised = []
            for p in (pr, po):
                try:
                    p.update_predicate_alignment(rel_embed)
                    p.update_predicate_alignment(attr_embed, predicate_type="attribute")
                    raised.append(None)
                except Exception as ex:  # noqa: BLE001  (degenerate folders: the reference's own KeyError for a predicate without triples)
                    raised.append((type(ex).__name__, str(ex)))
            assert raised[0] == raised[1], f"pam.refresh: reference raised {raised[0]}, this package {raised[1]}"
            if raised[0] is None:
                close(pam_snapshot(pr), pam_snapshot(po), "pam.refreshed")
    except AssertionError as ex:
        msg = f"DIFFERS at {str(ex)[:300]}"
    except Exception as ex:  # noqa: BLE001
        import traceback
        msg = f"{type(ex).__name__}: {str(ex)[:200]} @ {traceback.format_exc().strip().splitlines()[-3][:200]}"
    finally:
        shutil.rmtree(folder, ignore_errors=True)
    if msg:
        bad += 1
    print(f"READERS case {c}: {kw}: {msg or 'ok'}", flush=True)
print(f"readers / containers / literal clean-up / predicate alignment: {cases - bad} / {cases} folders identical to the reference's")

# ---- host arithmetic of the batch surface (code/base/batch.py:36-54, code/attr_batch.py:4-50, code/utils.py:35-49) --------------
ref_batch, ref_attr_batch = importlib.import_module("base.batch"), importlib.import_module("attr_batch")
from multike_amd import attr_batch as our_ab
from multike_amd.base import batch as our_bat
bad2 = 0
for c in range(cases * 4):
    n1, n2 = int(rng.integers(0, 400)), int(rng.integers(1, 400))
    bs = int(rng.integers(1, 2 * (n1 + n2) + 2))
    t1 = [(int(rng.integers(0, 50)), int(rng.integers(0, 5)), int(rng.integers(0, 50))) for _ in range(n1)]
    t2 = [(int(rng.integers(50, 100)), int(rng.integers(5, 9)), int(rng.integers(50, 100))) for _ in range(n2)]
    a1 = [t + (float(rng.random()),) for t in t1]; a2 = [t + (float(rng.random()),) for t in t2]
    e1, e2 = list(range(50)), list(range(50, 100))
    steps = int(np.ceil((n1 + n2) / bs)) + 1
    msg = ""
    try:
        for step in range(steps):
            close(list(ref_batch.generate_relation_triple_batch(t1, t2, set(t1), set(t2), e1, e2, bs, step, None, None, 0)),
                  list(our_bat.generate_relation_triple_batch(t1, t2, set(t1), set(t2), e1, e2, bs, step, None, None, 0)), f"rel step {step}")
            close(list(ref_attr_batch.generate_attribute_triple_batch(a1, a2, set(a1), set(a2), e1, e2, bs, step, None, None, 0)),
                  list(our_ab.generate_attribute_triple_batch(a1, a2, set(a1), set(a2), e1, e2, bs, step, None, None, 0)), f"attr step {step}")
            for fixed in (False, True):
                if n1:
                    close(ref_batch.generate_pos_triples(t1, bs, step, is_fixed_size=fixed), our_bat.generate_pos_triples(t1, bs, step, is_fixed_size=fixed),
                          f"pos step {step} fixed {fixed}")
        idx = list(range(int(rng.integers(0, 300)))); n = int(rng.integers(1, 12))
        close(ref_utils.task_divide(idx, n), ut.task_divide(idx, n), "task_divide")
    except AssertionError as ex:
        msg = f"DIFFERS at {str(ex)[:300]}"
    except Exception as ex:  # noqa: BLE001
        import traceback
        msg = f"{type(ex).__name__}: {str(ex)[:200]} @ {traceback.format_exc().strip().splitlines()[-3][:200]}"
    if msg:
        bad2 += 1
        print(f"HOST case {c}: n1={n1} n2={n2} batch={bs}: {msg}", flush=True)
print(f"batch-surface host arithmetic: {cases * 4 - bad2} / {cases * 4} cases identical to the reference's")
sys.exit(1 if bad + bad2 else 0)
