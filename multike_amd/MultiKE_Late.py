"""MultiKE_Late.py surface of the reference (code/MultiKE_Late.py): `valid` / `test` / weighted view averaging
(`wva`, `valid_WVA`, `test_WVA`) and the SSL driver `MultiKE_Late` (late combination: view training, then
`shared_learning_max_epoch` epochs of shared-space mapping).  The callers of the hot path — §8f-1 "next" row: same
phase order, gates (`i > start_predicate_soft_alignment`, `i % eval_freq`, `i % truncated_freq`), step counts and loss
normalisation as the reference; the work itself runs on the GPU through `MultiKE_model.MultiKE`."""
from __future__ import annotations

import math
import time

import numpy as np

from .base import evaluation as eva
from .base.batch import neighbour_table
from .MultiKE_model import MultiKE
from .utils import task_divide


def _view_embeddings(model, embed_choice, w):
    """code/MultiKE_Late.py:15-28 / 40-53: which [|E|, dim] matrix an `embed_choice` denotes (normalised views)."""
    pick = {"nv": model.name_embeds, "rv": model.rv_ent_embeds, "av": model.av_ent_embeds, "final": model.ent_embeds}
    if embed_choice in pick:
        return pick[embed_choice].eval(session=model.session)
    if embed_choice == "avg":
        return (w[0] * model.name_embeds.eval(session=model.session) + w[1] * model.rv_ent_embeds.eval(session=model.session)
                + w[2] * model.av_ent_embeds.eval(session=model.session))
    return model.ent_embeds.eval(session=model.session)


def _device_ids(model, key, entities):
    """int32 device tensor of an entity list of `model.kgs`, made once per model (the lists never change)."""
    import torch
    cache = model.__dict__.setdefault("_eval_ids", {})
    if key not in cache:
        cache[key] = torch.as_tensor(np.asarray(entities(), dtype=np.int32), device=model.device)
    return cache[key]


def _view_rows(model, embed_choice, w, ids):
    """Rows `ids` of the matrix `_view_embeddings` denotes, gathered ON THE DEVICE (HIP gather of the normalised view): the
    evaluator then never sees a host copy — the reference's `.eval(session)` of a whole [|E|, dim] table + fancy indexing is
    a 60 MB read-back and 20 ms of host work per call at 200K entities.  None when the model's tables are not device tables."""
    pick = {"nv": model.name_embeds, "rv": model.rv_ent_embeds, "av": model.av_ent_embeds, "final": model.ent_embeds}
    tabs = [pick[embed_choice]] if embed_choice in pick else \
        ([model.name_embeds, model.rv_ent_embeds, model.av_ent_embeds] if embed_choice == "avg" else [model.ent_embeds])
    if not all(hasattr(t, "lookup") for t in tabs) or getattr(model, "device", None) is None:
        return None
    if embed_choice == "avg":
        return w[0] * tabs[0].lookup(ids) + w[1] * tabs[1].lookup(ids) + w[2] * tabs[2].lookup(ids)
    return tabs[0].lookup(ids)


def _eval_pair(model, embed_choice, w, key1, ents1, key2, ents2):
    ids1 = ids2 = None
    if getattr(model, "device", None) is not None and hasattr(model.ent_embeds, "lookup"):
        ids1, ids2 = _device_ids(model, key1, ents1), _device_ids(model, key2, ents2)
        e1 = _view_rows(model, embed_choice, w, ids1)
        if e1 is not None:
            return e1, _view_rows(model, embed_choice, w, ids2)
    ent_embeds = _view_embeddings(model, embed_choice, w)
    return ent_embeds[ents1(), ], ent_embeds[ents2(), ]


def valid(model, embed_choice='avg', w=(1, 1, 1)):
    """code/MultiKE_Late.py:14-36: valid entities of KG1 against valid+test entities of KG2."""
    k = model.kgs
    embeds1, embeds2 = _eval_pair(model, embed_choice, w, "valid1", lambda: k.valid_entities1,
                                  "valid2+test2", lambda: k.valid_entities2 + k.test_entities2)
    print(embed_choice, 'valid results:')
    _, mrr_12 = eva.valid(embeds1, embeds2, None, model.args.top_k, model.args.test_threads_num, normalize=True)
    return mrr_12


def test(model, embed_choice='avg', w=(1, 1, 1)):
    """code/MultiKE_Late.py:39-61."""
    k = model.kgs
    embeds1, embeds2 = _eval_pair(model, embed_choice, w, "test1", lambda: k.test_entities1, "test2", lambda: k.test_entities2)
    print(embed_choice, 'test results:')
    _, mrr_12 = eva.valid(embeds1, embeds2, None, model.args.top_k, model.args.test_threads_num, normalize=True)
    return mrr_12


def _unit_rows(x):
    import torch
    if isinstance(x, torch.Tensor):          # device rows
        n = torch.linalg.norm(x, dim=1, keepdim=True)
        return x / torch.where(n == 0, torch.ones_like(n), n)
    n = np.linalg.norm(x, axis=1, keepdims=True)
    return x / np.where(n == 0, 1.0, n)


def _compute_weight(embeds1, embeds2, embeds3):
    """code/MultiKE_Late.py:64-81: mean cosine between a view and the average of the three views."""
    other = _unit_rows((embeds1 + embeds2 + embeds3) / 3)
    weights = (_unit_rows(embeds1) * other).sum(1)  # the diagonal of the similarity matrix, without the matrix
    mean = np.mean(weights) if isinstance(weights, np.ndarray) else np.float32(float(weights.mean()))
    print(tuple(weights.shape), mean)
    return mean


def wva(embeds1, embeds2, embeds3):
    """code/MultiKE_Late.py:84-88 (the function returns after the three weights; the rest of its body is dead)."""
    return (_compute_weight(embeds1, embeds2, embeds3), _compute_weight(embeds2, embeds1, embeds3),
            _compute_weight(embeds3, embeds1, embeds2))


def _wva_eval(model, ents1, ents2, label, keys=None):
    tabs = (model.name_embeds, model.rv_ent_embeds, model.av_ent_embeds)
    if keys is not None and getattr(model, "device", None) is not None and all(hasattr(t, "lookup") for t in tabs):
        ids1, ids2 = _device_ids(model, keys[0], lambda: ents1), _device_ids(model, keys[1], lambda: ents2)
        v1 = [t.lookup(ids1) for t in tabs]          # rows gathered on the device: no whole-table read-back
        v2 = [t.lookup(ids2) for t in tabs]
    else:
        views = tuple(t.eval() for t in tabs)
        v1 = [v[ents1, ] for v in views]
        v2 = [v[ents2, ] for v in views]
    wsum = np.array(wva(*v1)) + np.array(wva(*v2))
    wsum = wsum / wsum.sum()
    print('weights', *wsum)
    embeds1 = sum(float(w) * v for w, v in zip(wsum, v1))
    embeds2 = sum(float(w) * v for w, v in zip(wsum, v2))
    print(label)
    _, mrr_12 = eva.valid(embeds1, embeds2, None, model.args.top_k, model.args.test_threads_num, normalize=True)
    return mrr_12


def valid_WVA(model):
    """code/MultiKE_Late.py:99-135."""
    return _wva_eval(model, model.kgs.valid_entities1, model.kgs.valid_entities2 + model.kgs.test_entities2,
                     'wvag valid results:', keys=("valid1", "valid2+test2"))


def test_WVA(model):
    """code/MultiKE_Late.py:138-173."""
    return _wva_eval(model, model.kgs.test_entities1, model.kgs.test_entities2, 'wvag test results:', keys=("test1", "test2"))


class _ScheduledMultiKE(MultiKE):
    """What `MultiKE_CV.run` and `MultiKE_Late.run` share (code/MultiKE_CSL.py:36-56, code/MultiKE_Late.py:201-223):
    step counts, supervision lists, the per-epoch view training block and the periodic refreshes."""

    # The evaluation calls of the schedules go through these hooks (module-level functions by default, looked up when called);
    # the multi-GPU drivers (multike_amd/distributed_run.py) override them with rank-sharded evaluation.
    def _valid(self, embed_choice):
        return valid(self, embed_choice=embed_choice)

    def _test(self, embed_choice):
        return test(self, embed_choice=embed_choice)

    def _valid_WVA(self):
        return valid_WVA(self)

    def _test_WVA(self):
        return test_WVA(self)

    defer_predicate_update = False      # set True by the model's constructor on a GPU (instances made without it run in line)

    def _prepare(self):
        kgs, pam, a = self.kgs, self.predicate_align_model, self.args
        rel_n = kgs.kg1.local_relation_triples_num + kgs.kg2.local_relation_triples_num
        attr_n = kgs.kg1.local_attribute_triples_num + kgs.kg2.local_attribute_triples_num
        self._rel_steps = int(math.ceil(rel_n / a.batch_size))
        self._attr_steps = int(math.ceil(attr_n / a.batch_size))  # the reference divides by batch_size here too
        self._rel_tasks = task_divide(list(range(self._rel_steps)), a.batch_threads_num)
        self._attr_tasks = task_divide(list(range(self._attr_steps)), a.batch_threads_num)
        self._ckge_rel_triples = kgs.kg1.sup_relation_triples_list + kgs.kg2.sup_relation_triples_list
        self._ckge_attr_triples = kgs.kg1.sup_attribute_triples_list + kgs.kg2.sup_attribute_triples_list
        self._refresh_predicate_lists()
        self._neighbors = (None, None)
        self._entity_list = kgs.kg1.entities_list + kgs.kg2.entities_list

    def _refresh_predicate_lists(self):
        pam = self.predicate_align_model
        self._ckgp_rel_triples = pam.sup_relation_alignment_triples1 + pam.sup_relation_alignment_triples2
        self._ckga_attr_triples = pam.sup_attribute_alignment_triples1 + pam.sup_attribute_alignment_triples2

    def _train_views(self, i):
        """One epoch of the six view / cross-KG phases in the reference's order.

        The relation group (relation view -> ckge-rel -> ckgp-rel: rv_ent_embeds, rel_embeds) and the attribute group
        (attribute view -> ckge-attr -> ckga-attr: av_ent_embeds, attr_embeds, the CNNs) touch disjoint state, so they
        commute; with `overlap_views` the two groups are enqueued on two HIP streams and meet again before common-space
        learning.  Each group is a chain of small, launch-floor-dominated kernels that leaves most of the chip idle, and
        nothing synchronises inside the epoch (the loss read-backs are deferred to the end), unlike the per-step overlap
        schemes that were measured and rejected for the relation view alone."""
        a = self.args
        n1, n2 = self._neighbors
        soft = i > a.start_predicate_soft_alignment

        def relation_group_a():
            self.train_relation_view_1epo(i, self._rel_steps, self._rel_tasks, None, n1, n2)
            self.train_cross_kg_entity_inference_relation_view_1epo(i, self._ckge_rel_triples)

        def relation_group_b():
            if soft:
                self.train_cross_kg_relation_inference_1epo(i, self._ckgp_rel_triples)

        def attribute_group_a():
            self.train_attribute_view_1epo(i, self._attr_steps, self._attr_tasks, None, n1, n2)
            self.train_cross_kg_entity_inference_attribute_view_1epo(i, self._ckge_attr_triples)

        def attribute_group_b():
            if soft:
                self.train_cross_kg_attribute_inference_1epo(i, self._ckga_attr_triples)

        if not getattr(self, "overlap_views", True) or a.optimizer not in ("Adagrad", "SGD"):
            relation_group_a()
            self._finish_predicate_update()
            relation_group_b()
            attribute_group_a()
            attribute_group_b()
            return
        import torch
        main = torch.cuda.current_stream()
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream
        self._defer_losses, self._pending = True, []
        try:
            # two streams: 18.0 ms per epoch against 22.0 on one (C2 shape).  Measured and not kept: a higher stream priority
            # for either group (no change), enqueueing the two groups from two host threads (18.9 ms: the device, not the
            # host, is what the two chains share).
            # A predicate refresh deferred from the end of the previous epoch (`_update_predicate_alignment`) replaces the lists
            # three loops read: the relation- and attribute-inference loops, and — through
            # `attribute_triples_w_weights1/2` (code/predicate_alignment.py:170-174, read at code/MultiKE_model.py:325-328) —
            # the attribute VIEW itself.  Only the relation view and its entity-inference loop are list-independent: they are
            # enqueued first, the refresh is finished while the device works through them, and everything else follows it —
            # the order in which the reference's epoch i + 1 sees the lists of the refresh at the end of epoch i
            # (code/MultiKE_CSL.py:80-87).  (Round 4 enqueued the attribute view before the refresh: one epoch on stale weights
            # after every refresh.)
            start = torch.cuda.Event()
            start.record(main)                      # what the side stream must see: everything before this epoch
            relation_group_a()
            rel_a, self._pending = self._pending, []
            self._finish_predicate_update()
            side.wait_event(start)
            with torch.cuda.stream(side):
                attribute_group_a()
                attribute_group_b()
            attr_ab, self._pending = self._pending, []
            attr_a, attr_b = attr_ab, []
            relation_group_b()
            self._pending = rel_a + self._pending + attr_a + attr_b       # the reference prints the relation group's losses first
            main.wait_stream(side)
        finally:
            self._defer_losses = False
        for p in self._pending:
            p.finish()
        self._pending = []

    def _update_predicate_alignment(self):
        """code/MultiKE_CSL.py:80-87 / code/MultiKE_Late.py:244-251 (host-side soft predicate alignment)."""
        pam = self.predicate_align_model
        if getattr(self, "defer_predicate_update", False) and hasattr(pam, "update_predicate_alignment"):
            # The refresh reads rel_embeds / attr_embeds as they are NOW and its lists are first read by the next epoch's
            # relation- / attribute-inference loops.  Snapshot the two (small) tables to pinned host memory without waiting
            # and do the host work when those loops are about to be enqueued (`_finish_predicate_update`, called by
            # `_train_views` after the phases that do not need the lists are on the device): same inputs, same lists, and the
            # device is not idle meanwhile.
            import torch
            snaps = []
            for tab in (self.rel_embeds, self.attr_embeds):
                dev = tab.lookup(None)
                host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
                host.copy_(dev, non_blocking=True)
                snaps.append((dev, host))                    # the device tensor is held until the copy has completed
            ev = torch.cuda.Event()
            ev.record()
            self._pending_predicate = (ev, snaps)
            return
        if hasattr(pam, "update_predicate_alignment"):
            pam.update_predicate_alignment(self.rel_embeds.eval(session=self.session))
            pam.update_predicate_alignment(self.attr_embeds.eval(session=self.session), predicate_type='attribute')
        self._refresh_predicate_lists()

    def _finish_predicate_update(self):
        """The host side of a deferred `_update_predicate_alignment` (no-op when none is pending)."""
        pend = getattr(self, "_pending_predicate", None)
        if pend is None:
            return
        self._pending_predicate = None
        ev, snaps = pend
        ev.synchronize()
        pam = self.predicate_align_model
        import torch
        if getattr(pam, "device", None) is None:
            pam.update_predicate_alignment(snaps[0][1].numpy())
            pam.update_predicate_alignment(snaps[1][1].numpy(), predicate_type='attribute')
            self._refresh_predicate_lists()
            return
        # the lists are built in HBM from static inputs (the KGs' triples) and two small tables: on a stream of their own, so that
        # neither the table uploads nor the gathers queue behind the phases already enqueued on the training streams
        if getattr(self, "_refresh_stream", None) is None:
            self._refresh_stream = torch.cuda.Stream(device=self.device)
        rs = self._refresh_stream
        with torch.cuda.stream(rs):
            pam.update_predicate_alignment(snaps[0][1].numpy())
            pam.update_predicate_alignment(snaps[1][1].numpy(), predicate_type='attribute')
            self._refresh_predicate_lists()
        done = torch.cuda.Event()
        done.record(rs)
        for st in (torch.cuda.current_stream(), getattr(self, "_side_stream", None)):
            if st is not None:
                st.wait_event(done)

    def _refresh_neighbours(self, i):
        """Truncated negative sampling: k-NN candidate lists every `truncated_freq` epochs
        (code/MultiKE_CSL.py:89-102).  Stays on the device: (candidate table, valid flags) per KG."""
        a, kgs = self.args, self.kgs
        if a.neg_sampling != 'truncated' or i % a.truncated_freq != 0:
            return
        t1 = time.time()
        assert 0.0 < a.truncated_epsilon < 1.0
        out = []
        for kg, useful in ((kgs.kg1, kgs.useful_entities_list1), (kgs.kg2, kgs.useful_entities_list2)):
            k = int((1 - a.truncated_epsilon) * kg.entities_num)
            # the reference's random.sample(candidates, neg_triple_num) raises ValueError on a shorter list
            # (code/base/batch.py:96-99); say so before the kernels see it
            if k < a.neg_triple_num or k > len(useful):
                raise ValueError(f"truncated sampling: {k} neighbours per entity (int((1 - truncated_epsilon) * "
                                 f"{kg.entities_num})) must be >= neg_triple_num ({a.neg_triple_num}) and <= the "
                                 f"{len(useful)} useful entities of the KG")
            emb = self.rv_ent_embeds.lookup(self._ids(useful))
            out.append(neighbour_table(emb, useful, k, kgs.entities_num, device=self.device))
        self._neighbors = tuple(out)
        print("generating neighbors of {} entities costs {:.3f} s.".format(len(self._entity_list), time.time() - t1))

    def _ids(self, lst):
        import torch
        return torch.as_tensor(np.asarray(lst, dtype=np.int32), device=self.device)


class MultiKE_Late(_ScheduledMultiKE):
    """code/MultiKE_Late.py:176-280 — run_SSL.py's model."""

    def __init__(self, data, args, attr_align_model):
        super().__init__(data, args, attr_align_model)
        self.flag1, self.flag2, self.early_stop = -1, -1, False
        self.defer_predicate_update = True       # the soft predicate-alignment refresh's host work under the next epoch's kernels
        if hasattr(self.predicate_align_model, "_refresh_device"):
            self.predicate_align_model.device = self.device    # ... and its per-triple work in HBM (no list upload)
        self._define_variables()
        self._define_name_view_graph()
        self._define_relation_view_graph()
        self._define_attribute_view_graph()
        self._define_cross_kg_entity_reference_relation_view_graph()
        self._define_cross_kg_entity_reference_attribute_view_graph()
        self._define_cross_kg_relation_reference_graph()
        self._define_cross_kg_attribute_reference_graph()
        self._define_common_space_learning_graph()
        self._define_space_mapping_graph()

    def run(self):
        a = self.args
        self._prepare()
        self._valid('nv')
        self._valid('avg')
        for i in range(1, a.max_epoch + 1):
            print('epoch {}:'.format(i))
            self._train_views(i)
            if i >= a.start_valid and i % a.eval_freq == 0:
                self._valid('rv')
                self._valid('av')
                self._valid('avg')
                self._valid_WVA()
                if i >= a.start_predicate_soft_alignment:
                    self._update_predicate_alignment()
            if self.early_stop or i == a.max_epoch:
                break
            self._refresh_neighbours(i)
        for i in range(1, a.shared_learning_max_epoch + 1):
            self.train_shared_space_mapping_1epo(i, self._entity_list)
            if i >= a.start_valid and i % a.eval_freq == 0:
                self._valid('final')
        self._finish_predicate_update()
        self._save_async = True
        self.save()
        results = {k: self._test(k) for k in ('nv', 'rv', 'av', 'avg')}
        results['wva'] = self._test_WVA()
        results['final'] = self._test('final')
        self._join_save()
        return results
