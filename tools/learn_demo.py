"""End-to-end sanity run: a synthetic dataset FOLDER whose two KGs share structure -> DataModel -> ITC training ->
Hits@k per view over the epochs.  python tools/learn_demo.py [n_pairs] [epochs] [shared]"""
import os, sys, tempfile, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multike_amd.data_model import DataModel
from multike_amd.MultiKE_CSL import MultiKE_CV
from multike_amd.MultiKE_Late import test
from multike_amd.predicate_alignment import PredicateAlignModel
from multike_amd.synthetic import synthetic_args, write_dataset_folder

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 60
shared = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
folder = tempfile.mkdtemp() + "/"
wf = write_dataset_folder(folder, n_pairs=n_pairs, n_extra=n_pairs // 10, n_rel=40, n_attr=30, triples_per_entity=5.0, shared_structure=shared)
args = synthetic_args(training_data=folder, output=folder + "out/", word2vec_path=wf, dataset_division="631/", encoder_epoch=5,
                      encoder_active="tanh", encoder_normalize=True, retrain_literal_embeds=False, literal_normalize=True, dim=75,
                      batch_size=2000, attribute_batch_size=2000, entity_batch_size=2000, neg_triple_num=10, learning_rate=0.01,
                      ITC_learning_rate=0.01, max_epoch=epochs, start_valid=10 ** 6, eval_freq=10, start_predicate_soft_alignment=10,
                      truncated_freq=10, truncated_epsilon=0.98, is_save=False)
with contextlib.redirect_stdout(io.StringIO()):
    data = DataModel(args)
    pam = PredicateAlignModel(data.kgs, args)
    m = MultiKE_CV(data, args, pam)
    m._prepare()
k = data.kgs
print(f"entities {k.entities_num}, relation triples {k.kg1.relation_triples_num}+{k.kg2.relation_triples_num}, attribute triples "
      f"{k.kg1.attribute_triples_num}+{k.kg2.attribute_triples_num}, train/valid/test links {len(k.train_links)}/{len(k.valid_links)}/{len(k.test_links)}, "
      f"matched relations {len(pam.relation_alignment_set)}, matched attributes {len(pam.attribute_alignment_set)}")
def hits():
    out = {}
    for c in ("nv", "rv", "av", "final"):
        with contextlib.redirect_stdout(io.StringIO()):
            out[c] = round(float(test(m, embed_choice=c)), 2)
    return out
print("epoch 0", hits())
for i in range(1, epochs + 1):
    with contextlib.redirect_stdout(io.StringIO()):
        m._train_views(i)
        m.train_common_space_learning_1epo(i, m._entity_list)
        if i >= args.start_predicate_soft_alignment and i % 10 == 0:
            m._update_predicate_alignment()
        m._refresh_neighbours(i)
    if i % 10 == 0 or i == epochs:
        print("epoch", i, hits())
