"""The ITC / SSL drivers on several GPUs: `python -m multike_amd.run --gpus N` (one process per GPU, torch.distributed over RCCL).

`ShardedMultiKE_CV` / `ShardedMultiKE_Late` ARE the single-GPU drivers (`MultiKE_CSL.MultiKE_CV.run`, `MultiKE_Late.MultiKE_Late.run`
— the loops pinned to the reference's own `run()` by tests/test_schedule_golden.py, code/MultiKE_CSL.py:36-107,
code/MultiKE_Late.py:201-280): same schedule code, same gates.  What is replaced is what the schedule calls:

  train_*_1epo            one phase of `distributed_model.ShardedITC` on the row-sharded tables (entity tables id % world, small
                          tables and CNN sets replicated)
  valid / test / WVA      rank-sharded evaluation: the view's rows of the evaluated entities are assembled on every rank (one
                          all-reduce of a [n, dim] buffer each rank fills at the positions it owns), every rank ranks ITS block of
                          the KG1 rows against all KG2 rows on the MFMA evaluator (`k_align_rank`), the Hits / MR / MRR sums
                          are all-reduced
  truncated-sampling refresh   every rank assembles the relation view's rows of a KG's useful entities the same way, computes the
                          k nearest neighbours of ITS slice of them (`neighbour_table(part=...)`), the slices are all-gathered
                          into the candidate table every rank's sampler reads (each rank draws the negatives of all positives)
  soft predicate alignment     host-side on the REPLICATED relation / attribute tables: every rank computes the same lists
  save                    rank 0, from gathered tables

New design: the reference is single-device."""
from __future__ import annotations

import math
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .attr_cnn import AttrCNN
from .base.alignment import alignment_counts, tie_aware_metrics
from .base.batch import neighbour_table
from .distributed_model import ShardedITC
from .MultiKE_CSL import MultiKE_CV
from .MultiKE_Late import MultiKE_Late, _compute_weight
from .tables import xavier_truncated_normal
from .utils import generate_out_folder, save_embeddings


class _RelationKGs:
    """What `OwnerComputesTrainer` reads of the two KGs: relation triples, entity lists, the sampler's known-triple sets."""

    def __init__(self, kgs):
        arr = lambda x: np.asarray(list(x), dtype=np.int32).reshape(-1, 3)
        self.triples = [arr(kgs.kg1.local_relation_triples_list), arr(kgs.kg2.local_relation_triples_list)]
        self.known = [arr(kgs.kg1.local_relation_triples_set), arr(kgs.kg2.local_relation_triples_set)]
        self._ents = [np.asarray(kgs.kg1.entities_list, dtype=np.int32), np.asarray(kgs.kg2.entities_list, dtype=np.int32)]

    def entities(self, k):
        return self._ents[k]


class _View:
    """`model.rv_ent_embeds` for code written against the single-GPU model: `.eval()` = the full normalised table, gathered."""

    def __init__(self, owner, name):
        self.owner, self.name = owner, name

    def eval(self, session=None):
        return self.owner.rows(self.name, np.arange(self.owner.m.n_ent, dtype=np.int64)).cpu().numpy()


class _ShardedMixin:
    def _init_sharded(self, data, args, predicate_align_model, rank, world, comm_oc=None, comm_views=None, with_mapping=False):
        self.predicate_align_model, self.args, self.data = predicate_align_model, args, data
        assert args.alignment_module == 'swapping'
        if args.optimizer != "Adagrad":
            raise _lib.MultiKEHipError("the multi-GPU drivers are built for the reference's default optimizer (Adagrad)")
        self.kgs = kgs = data.kgs
        self.kg1, self.kg2 = kgs.kg1, kgs.kg2
        self.rank, self.world = rank, world
        self.session, self.device = None, torch.device("cuda")
        self.flag1, self.flag2, self.early_stop = -1, -1, False
        self.overlap_views = False
        self.out_folder = generate_out_folder(args.output, args.training_data, '', self.__class__.__name__) if rank == 0 else None
        d, seed = args.dim, int(getattr(args, "seed", 0))
        # the same initial state as the single-GPU model (multike_amd/MultiKE_model.py `_define_variables`, same seeds)
        t = lambda n, k: xavier_truncated_normal(n, d, "cpu", seed=seed + k).numpy()
        tables = {"rv_ent": t(kgs.entities_num, 1), "rel": t(kgs.relations_num, 2), "av_ent": t(kgs.entities_num, 3),
                  "attr": t(kgs.attributes_num, 4), "ent": t(kgs.entities_num, 5),
                  "name": np.asarray(data.local_name_vectors, dtype=np.float32), "lit": np.asarray(data.value_vectors, dtype=np.float32)}
        mats = None
        if with_mapping:
            g = torch.Generator(device="cpu")
            g.manual_seed(seed + 6)
            mats = []
            for _ in range(3):      # tf.initializers.orthogonal(): QR of a normal matrix, sign-fixed
                q, r = torch.linalg.qr(torch.randn(d, d, generator=g))
                mats.append((q * torch.sign(torch.diagonal(r))).numpy())
        cnn = [AttrCNN(d, self.device, seed=seed + 11 + k).numpy_params() for k in range(3)]
        pam = predicate_align_model
        self._installed = {"ckge_rel": kgs.kg1.sup_relation_triples_list + kgs.kg2.sup_relation_triples_list,
                           "ckge_attr": kgs.kg1.sup_attribute_triples_list + kgs.kg2.sup_attribute_triples_list,
                           "ckgp_rel": pam.sup_relation_alignment_triples1 + pam.sup_relation_alignment_triples2,
                           "ckga_attr": pam.sup_attribute_alignment_triples1 + pam.sup_attribute_alignment_triples2}
        lists = dict(self._installed, attr=(pam.attribute_triples_w_weights1, pam.attribute_triples_w_weights2),
                     entities=kgs.kg1.entities_list + kgs.kg2.entities_list)
        attr_n = kgs.kg1.local_attribute_triples_num + kgs.kg2.local_attribute_triples_num
        self.m = ShardedITC(_RelationKGs(kgs), tables, cnn, lists, rank, world, batch_size=args.batch_size,
                            attribute_batch_size=args.attribute_batch_size, entity_batch_size=args.entity_batch_size,
                            neg_triple_num=args.neg_triple_num, learning_rate=args.learning_rate,
                            itc_learning_rate=args.ITC_learning_rate, cv_name_weight=args.cv_name_weight, cv_weight=args.cv_weight,
                            seed=seed, comm_oc=comm_oc, comm_views=comm_views, mapping_matrices=mats,
                            mapping_learning_rate=args.learning_rate, orthogonal_weight=args.orthogonal_weight,
                            attr_steps=int(math.ceil(attr_n / args.batch_size)))     # the reference divides by batch_size (:40)
        self._vc = self.m.common.comm
        self._oc = self.m.relation.comm
        self._held_nb = (None, None)
        for name in ("name_embeds", "rv_ent_embeds", "av_ent_embeds", "ent_embeds"):
            setattr(self, name, _View(self, {"name_embeds": "nv", "rv_ent_embeds": "rv", "av_ent_embeds": "av", "ent_embeds": "final"}[name]))
        self.rel_embeds, self.attr_embeds = self.m.rel, self.m.attr        # replicated: plain tables

    def _prepare(self):
        """The schedule's own preparation (step counts, supervision lists), then: the lists it will hand to the phases ARE the
        ones the sharded trainers were built on (same concatenations) — recorded by identity so that only a soft-alignment
        refresh (new list objects) rebuilds a trainer."""
        super()._prepare()
        self._installed = {"ckge_rel": self._ckge_rel_triples, "ckge_attr": self._ckge_attr_triples,
                           "ckgp_rel": self._ckgp_rel_triples, "ckga_attr": self._ckga_attr_triples}

    # --- rows of a view, assembled on every rank ----------------------------------------------------------------------
    def rows(self, choice, ids, w=(1, 1, 1)) -> torch.Tensor:
        """[len(ids), dim] float32 on the device: the rows `choice` denotes (code/MultiKE_Late.py:15-28) of the GLOBAL entity
        ids, identical on every rank.  Each rank fills the positions it owns; one all-reduce (adding zeros is exact)."""
        m, G, r = self.m, self.world, self.rank
        ids = torch.as_tensor(np.asarray(ids, dtype=np.int64), device=self.device)
        mine = torch.nonzero(ids % G == r).reshape(-1)
        loc = (ids[mine] // G).to(torch.int32)
        tab = {"nv": m.name, "rv": m.rv_ent, "av": m.av_ent, "final": m.ent}
        if choice == "avg":
            part = w[0] * m.name.lookup(loc) + w[1] * m.rv_ent.lookup(loc) + w[2] * m.av_ent.lookup(loc)
        else:
            part = tab[choice].lookup(loc)
        out = torch.zeros(ids.numel(), m.ent.dim, dtype=torch.float32, device=self.device)
        out[mine] = part
        self._vc.all_reduce(out)
        return out

    def _rank_block(self, e1: torch.Tensor, e2: torch.Tensor, top_k):
        """Hits@k / MR / MRR of e1 rows against e2 rows (gold column = row index), the rows split over the ranks: this rank ranks
        rows [lo, hi); the columns are rotated so that its block's gold columns come first (a permutation of the columns does
        not change a rank)."""
        n1 = e1.shape[0]
        lo, hi = n1 * self.rank // self.world, n1 * (self.rank + 1) // self.world
        acc = torch.zeros(len(top_k) + 2, dtype=torch.float64, device=self.device)
        if hi > lo:
            cols = torch.cat([e2[lo:hi], e2[:lo], e2[hi:]], 0)
            greater, ties, _ = alignment_counts(e1[lo:hi], cols, normalize=True, device=self.device)
            hits, mr, mrr = tie_aware_metrics(greater, ties, top_k)
            acc += torch.tensor(list(hits) + [mr * (hi - lo), mrr * (hi - lo)], dtype=torch.float64, device=self.device)
        self._vc.all_reduce(acc)
        acc = acc.cpu().numpy() / n1
        return np.round(acc[:len(top_k)] * 100, 3), float(acc[-2]), float(acc[-1])

    def _evaluate(self, e1, e2, label, accurate):
        t = time.time()
        hits, mr, mrr = self._rank_block(e1, e2, self.args.top_k)
        if self.rank == 0:
            print(label)
            if accurate:
                print("accurate results: hits@{} = {}%, mr = {:.3f}, mrr = {:.6f}, time = {:.3f} s ".format(self.args.top_k, hits, mr, mrr, time.time() - t))
            else:
                print("quick results: hits@{} = {}%, time = {:.3f} s ".format(self.args.top_k, hits, time.time() - t))
        self.last_hits = hits
        return mrr

    def _valid(self, embed_choice, w=(1, 1, 1)):
        k = self.kgs
        return self._evaluate(self.rows(embed_choice, k.valid_entities1, w), self.rows(embed_choice, k.valid_entities2 + k.test_entities2, w),
                              f"{embed_choice} valid results:", False)

    def _test(self, embed_choice, w=(1, 1, 1)):
        k = self.kgs
        return self._evaluate(self.rows(embed_choice, k.test_entities1, w), self.rows(embed_choice, k.test_entities2, w),
                              f"{embed_choice} test results:", True)

    def _wva(self, ents1, ents2, label, accurate):
        """code/MultiKE_Late.py:99-173: the three views weighted by their mean cosine to the views' average."""
        v1 = [self.rows(c, ents1).cpu().numpy() for c in ("nv", "rv", "av")]
        v2 = [self.rows(c, ents2).cpu().numpy() for c in ("nv", "rv", "av")]
        import contextlib
        import io
        with (contextlib.redirect_stdout(io.StringIO()) if self.rank else contextlib.nullcontext()):   # rank 0 prints the weights
            ws = [np.array([_compute_weight(v[0], v[1], v[2]), _compute_weight(v[1], v[0], v[2]), _compute_weight(v[2], v[0], v[1])])
                  for v in (v1, v2)]
        wsum = ws[0] + ws[1]
        wsum = wsum / wsum.sum()
        dev = lambda a: torch.as_tensor(a, dtype=torch.float32, device=self.device)
        e1 = dev(sum(x * v for x, v in zip(wsum, v1)))
        e2 = dev(sum(x * v for x, v in zip(wsum, v2)))
        return self._evaluate(e1, e2, label, accurate)

    def _valid_WVA(self):
        k = self.kgs
        return self._wva(k.valid_entities1, k.valid_entities2 + k.test_entities2, 'wvag valid results:', False)

    def _test_WVA(self):
        return self._wva(self.kgs.test_entities1, self.kgs.test_entities2, 'wvag test results:', True)

    # --- the training phases ---------------------------------------------------------------------------------------
    def _phase(self, name, text, epoch, scale=1.0):
        start = time.time()
        loss, denom = self.m.run_phase(name, epoch)
        avg = loss * scale / max(denom, 1)
        if self.rank == 0:
            print('epoch {} of {}, avg. loss: {:.4f}, time: {:.4f}s'.format(epoch, text, avg, time.time() - start))
        return avg

    def _list_phase(self, name, text, epoch, triples):
        if triples is not self._installed.get(name):       # the predicate lists are re-created by every soft-alignment update
            self.m.set_lists(**{name: triples})
            self._installed[name] = triples
        return self._phase(name, text, epoch) if len(triples) else None

    def train_relation_view_1epo(self, epoch, triple_steps, steps_tasks, batch_queue, neighbors1, neighbors2):
        if neighbors1 is not self._held_nb[0] or neighbors2 is not self._held_nb[1]:
            self.m.relation.set_neighbours((neighbors1, neighbors2))
            self._held_nb = (neighbors1, neighbors2)
        return self._phase("relation", 'rel. view', epoch)

    def train_attribute_view_1epo(self, epoch, triple_steps, steps_tasks, batch_queue, neighbors1, neighbors2):
        return self._phase("attribute", 'att. view', epoch)

    def train_cross_kg_entity_inference_relation_view_1epo(self, epoch, sup_triples):
        return self._list_phase("ckge_rel", 'cross-kg entity inference in rel. view', epoch, sup_triples)

    def train_cross_kg_entity_inference_attribute_view_1epo(self, epoch, sup_triples):
        return self._list_phase("ckge_attr", 'cross-kg entity inference in attr. view', epoch, sup_triples)

    def train_cross_kg_relation_inference_1epo(self, epoch, sup_triples):
        return self._list_phase("ckgp_rel", 'cross-kg relation inference in rel. view', epoch, sup_triples)

    def train_cross_kg_attribute_inference_1epo(self, epoch, sup_triples):
        return self._list_phase("ckga_attr", 'cross-kg attribute inference in attr. view', epoch, sup_triples)

    def train_common_space_learning_1epo(self, epoch, entities):
        cvw = float(self.args.cv_weight)
        return self._phase("common", 'common space learning', epoch, scale=1.0 / cvw if cvw else 0.0)

    def train_shared_space_mapping_1epo(self, epoch, entities):
        return self._phase("mapping", 'shared space learning', epoch)

    # --- between epochs ------------------------------------------------------------------------------------------------
    def _refresh_neighbours(self, i):
        """code/MultiKE_CSL.py:89-102 on G ranks: each computes the neighbours of its slice of a KG's useful entities."""
        a, kgs, G = self.args, self.kgs, self.world
        if a.neg_sampling != 'truncated' or i % a.truncated_freq != 0:
            return
        t1 = time.time()
        out = []
        for kg, useful in ((kgs.kg1, kgs.useful_entities_list1), (kgs.kg2, kgs.useful_entities_list2)):
            k = int((1 - a.truncated_epsilon) * kg.entities_num)
            if k < a.neg_triple_num or k > len(useful):
                raise ValueError(f"truncated sampling: {k} neighbours per entity must be >= neg_triple_num ({a.neg_triple_num}) "
                                 f"and <= the {len(useful)} useful entities of the KG")
            emb = self.rows("rv", useful)
            if G == 1:
                out.append(neighbour_table(emb, useful, k, kgs.entities_num, device=self.device))
                continue
            table, valid, ids = neighbour_table(emb, useful, k, kgs.entities_num, device=self.device, part=(self.rank, G))
            pad = -(-len(useful) // G)
            rows = torch.zeros(pad, k, dtype=torch.int32, device=self.device)
            idp = torch.full((pad,), -1, dtype=torch.int64, device=self.device)
            rows[:ids.numel()] = table[ids]
            idp[:ids.numel()] = ids
            parts_r = [torch.empty_like(rows) for _ in range(G)]
            parts_i = [torch.empty_like(idp) for _ in range(G)]
            self._oc.all_gather_list(parts_r, rows)
            self._oc.all_gather_list(parts_i, idp)
            for pr, pi in zip(parts_r, parts_i):
                ok = pi >= 0
                table[pi[ok]] = pr[ok]
                valid[pi[ok]] = 1
            out.append((table, valid))
        self._neighbors = tuple(out)
        if self.rank == 0:
            print("generating neighbors of {} entities costs {:.3f} s.".format(len(self._entity_list), time.time() - t1))

    def save(self):
        full = self.m.gather()                       # collective: every rank takes part, rank 0 writes
        if self.rank != 0:
            return
        nrm = lambda x: x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-6)
        save_embeddings(self.out_folder, self.kgs, nrm(full["ent"]), np.asarray(self.data.local_name_vectors), nrm(full["rv"]),
                        nrm(full["av"]), self.m.rel.eval(), self.m.attr.eval())


class ShardedMultiKE_CV(_ShardedMixin, MultiKE_CV):
    """run_ITC.py's model on `world` GPUs."""

    def __init__(self, data, args, predicate_align_model, rank, world, comm_oc=None, comm_views=None):
        self._init_sharded(data, args, predicate_align_model, rank, world, comm_oc, comm_views, with_mapping=False)


class ShardedMultiKE_Late(_ShardedMixin, MultiKE_Late):
    """run_SSL.py's model on `world` GPUs."""

    def __init__(self, data, args, predicate_align_model, rank, world, comm_oc=None, comm_views=None):
        self._init_sharded(data, args, predicate_align_model, rank, world, comm_oc, comm_views, with_mapping=True)


def init_process_group_from_env():
    """(rank, world) of a `python -m torch.distributed.run` launch; backend "nccl" (= RCCL), one GPU per process."""
    import datetime
    import os
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=600))
    return rank, world
