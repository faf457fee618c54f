"""MultiKE_model.py — the reference's model / op surface (code/MultiKE_model.py) on the HIP hot path.

Same public names, same argument orders, same epoch-loop semantics: `get_optimizer`, `generate_optimizer`, `conv`,
class `MultiKE` with `_define_variables`, the eight `_define_*_graph` builders, the nine `train_*_1epo` loops,
`eval_kg{1,2}[_useful]_ent_embeddings`, `save`.  What differs is how a "graph" runs: instead of a TF session
executing dense whole-table ops fed from Python lists, every `train_*_1epo` enqueues fused HIP kernels over index
streams that already live in HBM.  Each `_define_*_graph` therefore creates a small "graph" record: which
tables, which optimizer slot (one Adagrad accumulator per graph per variable, as each `generate_optimizer` call
creates in TF — SURVEY.md §9.3-4), which learning rate.

`data`, `args`, `attr_align_model` are the reference's objects (DataModel / ARGs / PredicateAlignModel) or anything
with the same attributes; only the attributes this file reads are required (see `synthetic.SyntheticData`).
"""
from __future__ import annotations

import itertools
import math
import time

import numpy as np
import torch

from . import _lib
from . import losses
from .attr_cnn import AttrCNN
from .runner import RelationViewRunner
from .sampling import KGSide, KnownTripleSet, RelationBatcher, int_triples
from .tables import ADAGRAD_INIT_ACC, EmbeddingTable, StepEngine
from .utils import generate_out_folder, save_embeddings, touch_library_kernels

_HIP_OPTS = ("Adagrad", "SGD")            # touched-rows rules: fused / native multi-step paths
ATTR_GRAD_COPIES = 4
REL_GRAD_COPIES = 8
_DENSE_OPTS = tuple(_lib.DENSE_OPTS)        # Adam, Adadelta: whole-variable kernels, step-wise loops


class Optimizer:
    """What `get_optimizer` returns: the kind + learning rate; state lives in per-graph slots."""

    def __init__(self, kind: str, learning_rate: float):
        self.kind, self.learning_rate = kind, float(learning_rate)


def get_optimizer(opt, learning_rate):
    """code/MultiKE_model.py:15-25.  Adagrad (the reference's default, code/args.json:18) and SGD run on the fused
    touched-rows paths; Adadelta / Adam move zero-gradient weights too and run on whole-variable kernels."""
    if opt in ("Adagrad", "Adadelta", "Adam"):
        return Optimizer(opt, learning_rate)
    return Optimizer("SGD", learning_rate)


class DenseOptimizerState:
    """generate_optimizer for a loss over DENSE torch parameters (the 75x75 mapping matrices): autograd gradient +
    TF1 update rule (ApplyAdagrad: acc0 = 0.1, no epsilon)."""

    def __init__(self, var_list, learning_rate, opt="SGD"):
        self.vars = list(var_list)
        self.opt = get_optimizer(opt, learning_rate)
        self.acc = [torch.full_like(v, ADAGRAD_INIT_ACC) for v in self.vars]
        self.s1 = [torch.zeros_like(v) for v in self.vars] if self.opt.kind in _DENSE_OPTS else None
        self.s2 = [torch.zeros_like(v) for v in self.vars] if self.opt.kind in _DENSE_OPTS else None
        self.step = 0

    @torch.no_grad()
    def apply(self, grads):
        self.step += 1
        lr = self.opt.learning_rate
        for i, (v, a, g) in enumerate(zip(self.vars, self.acc, grads)):
            if g is None:
                continue
            if self.opt.kind == "Adam":        # TF1 ApplyAdam, defaults
                m, vv = self.s1[i], self.s2[i]
                m.mul_(0.9).add_(g, alpha=0.1)
                vv.mul_(0.999).add_(g * g, alpha=0.001)
                lr_t = lr * math.sqrt(1.0 - 0.999 ** self.step) / (1.0 - 0.9 ** self.step)
                v.sub_(lr_t * m / (torch.sqrt(vv) + 1e-8))
            elif self.opt.kind == "Adadelta":  # TF1 ApplyAdadelta, defaults
                acc, upd = self.s1[i], self.s2[i]
                acc.mul_(0.95).add_(g * g, alpha=0.05)
                u = torch.sqrt(upd + 1e-8) / torch.sqrt(acc + 1e-8) * g
                v.sub_(lr * u)
                upd.mul_(0.95).add_(u * u, alpha=0.05)
            elif self.opt.kind == "Adagrad":
                a.add_(g * g)
                v.sub_(self.opt.learning_rate * g / torch.sqrt(a))
            else:
                v.sub_(self.opt.learning_rate * g)


def generate_optimizer(loss, learning_rate, var_list=None, opt="SGD"):
    """code/MultiKE_model.py:28-31 for a torch scalar `loss` over dense leaf tensors `var_list`: computes the
    gradients and applies one update.  Returns the optimizer state (keep it and call `.apply` for further steps)."""
    if var_list is None:
        raise _lib.MultiKEHipError("generate_optimizer needs var_list on this backend (table variables are updated "
                                   "by the fused HIP steps, not through autograd)")
    st = DenseOptimizerState(var_list, learning_rate, opt)
    st.apply(torch.autograd.grad(loss, st.vars, allow_unused=True))
    return st


def conv(attr_hs, attr_as, attr_vs, dim, feature_map_size=2, kernel_size=(2, 4), activation="tanh", layer_num=2, *,
         cnn: AttrCNN | None = None):
    """code/MultiKE_model.py:34-63 on gathered DEVICE rows [B, dim]: returns the score vector [B], differentiable w.r.t.
    `attr_hs`, `attr_as` and the CNN's parameters (`cnn.params`, one packed leaf tensor — set `requires_grad_()` on it to
    receive its gradient) through a hand-written backward (multike_amd/attr_cnn.py `_ConvScore`: HIP conv-stack kernels, the
    dense layer on the library's own MFMA GEMM; no host copy, no library GEMM).  Same positional signature as the reference.
    `cnn` owns the parameters; None = a fresh parameter set with tf.layers' default initialisers, as every un-scoped
    `tf.layers` call creates in TF1 — it is returned on the result as `score.cnn`.  The training graphs do not go through
    this op: they use `AttrCNN.step(s)`, which fuses forward, backward and update."""
    if feature_map_size != 2 or tuple(kernel_size) != (2, 4) or layer_num != 2 or (activation != "tanh" and activation is not torch.tanh):
        raise _lib.MultiKEHipError("conv: only the reference's configuration (2 filters, 2x4, 2 layers, tanh) is built")
    from .attr_cnn import conv_score
    if cnn is None:
        cnn = AttrCNN(int(dim), attr_hs.device)
        cnn.params.requires_grad_(True)
    score = conv_score(attr_hs, attr_as, attr_vs, dim, cnn)
    score.cnn = cnn
    return score


def _dev_i32(x, device):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.int32), device=device)


class _TripleList:
    """A Python list of (h, r, t[, w]) tuples mirrored in HBM as int32 columns (+ float32 weights)."""

    def __init__(self, triples, device):
        self.n = len(triples)
        self.device = device
        if self.n == 0:
            self.cols, self.w = None, None
            return
        if getattr(triples, "dev", None) is not None:      # base.kgs.TripleArray made in HBM (the device-side predicate refresh)
            self.cols, self.w = triples.dev
            return
        if hasattr(triples, "cols"):          # base.kgs.TripleArray: one [n, 3] upload, the columns split on the device
            t = torch.as_tensor(triples.cols.astype(np.int32), device=device)
            self.cols = tuple(t[:, k].contiguous() for k in range(3))
            self.w = None if triples.w is None else torch.as_tensor(triples.w.astype(np.float32), device=device)
            return
        width = len(triples[0])
        if width == 3 and all(isinstance(x, (int, np.integer)) for x in triples[0]):
            flat = np.fromiter(itertools.chain.from_iterable(triples), dtype=np.int64, count=3 * self.n)   # 3x faster than asarray(list)
            t = torch.as_tensor(flat.reshape(self.n, 3).astype(np.int32), device=device)
            self.cols = tuple(t[:, k].contiguous() for k in range(3))
            self.w = None
            return
        arr = np.asarray([t[:3] for t in triples], dtype=np.int32)
        self.cols = tuple(torch.as_tensor(np.ascontiguousarray(arr[:, k]), device=device) for k in range(3))
        self.w = (torch.as_tensor(np.asarray([t[3] for t in triples], dtype=np.float32), device=device)
                  if width > 3 else None)

    def sample_epoch(self, batch_size, steps, seed, stream_id):
        """`steps` x random.sample(list, batch_size) (code/MultiKE_model.py:358: distinct inside a step, steps
        independent) in one sampler launch + one gather per column.  Returns (cols, w, idx) in epoch order."""
        idx = _lib.sample_distinct(self.n, batch_size, steps, seed, stream_id, device=self.device).reshape(-1)
        il = idx.long()
        cols = tuple(c[il] for c in self.cols)
        return cols, (None if self.w is None else self.w[il]), idx


class _EpochLoss:
    """The closing line of a `train_*_1epo` call, separated from the enqueueing of its kernels: reading the loss is the only
    host synchronisation of an epoch, so the drivers can enqueue several independent phases before reading any of them."""

    def __init__(self, text, epoch, value, denom, start, scale=1.0):
        self.text, self.epoch, self.value, self.denom, self.start, self.scale = text, epoch, value, denom, start, scale

    def finish(self):
        loss = (float(self.value) if self.value is not None else 0.0) * self.scale / max(self.denom, 1)
        print('epoch {} of {}, avg. loss: {:.4f}, time: {:.4f}s'.format(self.epoch, self.text, loss, time.time() - self.start))
        return loss


class MultiKE:

    def __check_args(self):
        assert self.args.alignment_module == 'swapping'  # for cross-KG inference (code/MultiKE_model.py:68-69)

    def __init__(self, data, args, attr_align_model):
        self.predicate_align_model = attr_align_model
        self.args = args
        self.__check_args()
        self.data = data
        self.kgs = kgs = data.kgs
        self.kg1 = kgs.kg1
        self.kg2 = kgs.kg2
        self.out_folder = generate_out_folder(self.args.output, self.args.training_data, '', self.__class__.__name__)
        self.session = None  # there is no TF session; kept so that `x.eval(session=model.session)` call sites work
        self.device = torch.device("cuda")
        self.engine = StepEngine(self.device)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(getattr(args, "seed", 0)))
        self._lists: dict = {}
        touch_library_kernels(self.device)      # code-object loads of the library kernel families: here, not inside the first epochs
        self._defer_losses = False      # drivers set it while they enqueue independent phases on two streams
        # test / debugging hook: `recorder(phase, **index_streams)` is called by every native train_*_1epo with the exact
        # device index streams the epoch consumed (positives, sampled negatives, weights, step offsets) — what a float64
        # oracle needs to replay the schedule on identical batches (tests/test_schedule_trace_gpu.py)
        self._recorder = None
        self._pending: list = []
        if self.args.optimizer not in _HIP_OPTS + _DENSE_OPTS:
            raise _lib.MultiKEHipError(f"optimizer {self.args.optimizer!r}: Adagrad, SGD, Adam or Adadelta")

    # ------------------------------------------------------------------------------------------------
    def _define_variables(self):
        """code/MultiKE_model.py:86-107."""
        d, dev = self.args.dim, self.device
        seed = int(getattr(self.args, "seed", 0))
        self.literal_embeds = EmbeddingTable(len(self.data.value_vectors), d, "literal_embeds", normalize=False,
                                             trainable=False, values=self.data.value_vectors, device=dev)
        self.name_embeds = EmbeddingTable(self.kgs.entities_num, d, "name_embeds", normalize=False, trainable=False,
                                          values=self.data.local_name_vectors, device=dev)
        self.rv_ent_embeds = EmbeddingTable(self.kgs.entities_num, d, "rv_ent_embeds", True, device=dev, seed=seed + 1)
        # relation rows: a few hundred rows take one gradient flush per positive, and real relation frequencies are heavy-tailed — the
        # scratch is privatised 8 ways for the native optimizers (group g flushes into copy g % 8; relation ids ~ Zipf(1.0): score
        # launch 46.1 -> 37.3 us, Zipf(1.5): 71.8 -> 39.6, uniform unchanged; EXPERIMENTS R5.17)
        self.rel_embeds = EmbeddingTable(self.kgs.relations_num, d, "rel_embeds", True, device=dev, seed=seed + 2,
                                         grad_copies=1 if self.args.optimizer in _DENSE_OPTS else REL_GRAD_COPIES)
        self.av_ent_embeds = EmbeddingTable(self.kgs.entities_num, d, "av_ent_embeds", True, device=dev, seed=seed + 3)
        # "False important!" (code/MultiKE_model.py:96-97): attribute embeddings are NOT read through l2_normalize
        # A few hundred attribute rows take every step's 5,000 triples and real attribute frequencies are heavy-tailed: the gradient
        # scratch of this table is privatised 4 ways (triple t adds to copy t % 4, the update sums them): with Zipf(1.0) attribute
        # ids the step went 52.8 -> 76.5 us without and stays at the uniform figure with them (EXPERIMENTS R5.16).  The dense
        # optimizers' step-wise path reads the scratch as one [rows][stride] array: one copy there.
        copies = 1 if self.args.optimizer in _DENSE_OPTS else ATTR_GRAD_COPIES
        self.attr_embeds = EmbeddingTable(self.kgs.attributes_num, d, "attr_embeds", False, device=dev, seed=seed + 4, grad_copies=copies)
        self.ent_embeds = EmbeddingTable(self.kgs.entities_num, d, "ent_embeds", True, device=dev, seed=seed + 5)
        g = torch.Generator(device="cpu")
        g.manual_seed(seed + 6)

        def orthogonal():  # tf.initializers.orthogonal(): QR of a normal matrix, sign-fixed
            q, r = torch.linalg.qr(torch.randn(d, d, generator=g))
            return (q * torch.sign(torch.diagonal(r))).to(dev).requires_grad_(True)

        self.nv_mapping, self.rv_mapping, self.av_mapping = orthogonal(), orthogonal(), orthogonal()
        self.eye_mat = torch.eye(d, device=dev)

    # --- view-specific embedding models ---------------------------------------------------------------
    def _define_name_view_graph(self):
        pass

    def _define_relation_view_graph(self):
        """code/MultiKE_model.py:114-132.  The known-triple filter is `local_relation_triples_set`, which in the
        reference aliases the set that also holds the swapped triples."""
        dev = self.device
        sides = []
        for kg in (self.kg1, self.kg2):
            # membership only: the order in which the hash set is filled does not matter, so no sort
            known = int_triples(kg.local_relation_triples_set)
            t = torch.as_tensor(known, device=dev)
            sides.append(KGSide(kg.entities_list,
                                KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous()), device=dev))
        self._rel_batcher = RelationBatcher(self.kg1.local_relation_triples_list, self.kg2.local_relation_triples_list,
                                            sides[0], sides[1], self.args.batch_size, self.args.neg_triple_num, device=dev,
                                            seed=int(getattr(self.args, "seed", 0)))
        self._rel_runner = (RelationViewRunner(self.rv_ent_embeds, self.rel_embeds, self._rel_batcher, "relation",
                                               lr=self.args.learning_rate, optimizer=self.args.optimizer)
                            if self.args.optimizer not in _DENSE_OPTS else None)
        self._neighbor_ids = (None, None)

    def _define_attribute_view_graph(self):
        """code/MultiKE_model.py:134-151 (variable_scope 'cnn')."""
        self._attr_cnn = AttrCNN(self.args.dim, self.device, seed=int(getattr(self.args, "seed", 0)) + 11)

    # --- cross-kg identity inference --------------------------------------------------------------------
    def _define_cross_kg_name_view_graph(self):
        pass

    def _define_cross_kg_entity_reference_relation_view_graph(self):
        """code/MultiKE_model.py:158-169: 2 * relation_logistic_loss_wo_negs, optimizer slot 'ckge_rel'."""
        self._ckge_rel = dict(opt="ckge_rel", scale=2.0)

    def _define_cross_kg_entity_reference_attribute_view_graph(self):
        """code/MultiKE_model.py:171-185: its own CNN parameter set (name_scope only => new tf.layers variables)."""
        self._ckge_attr_cnn = AttrCNN(self.args.dim, self.device, seed=int(getattr(self.args, "seed", 0)) + 12)

    def _define_cross_kg_relation_reference_graph(self):
        """code/MultiKE_model.py:187-201: 2 * logistic_loss_wo_negs (weighted), slot 'ckgp_rel'."""
        self._ckgp_rel = dict(opt="ckgp_rel", scale=2.0)

    def _define_cross_kg_attribute_reference_graph(self):
        """code/MultiKE_model.py:203-221: weighted, NOT doubled (:218), third CNN parameter set."""
        self._ckga_attr_cnn = AttrCNN(self.args.dim, self.device, seed=int(getattr(self.args, "seed", 0)) + 13)

    # --- intermediate combination -----------------------------------------------------------------------
    def _define_common_space_learning_graph(self):
        """code/MultiKE_model.py:225-239: optimizer on cv_weight * loss with ITC_learning_rate, slot 'cross_name'."""
        self._cross_name = dict(opt="cross_name", lr=self.args.ITC_learning_rate)

    def _define_space_mapping_graph(self):
        """code/MultiKE_model.py:241-261: only variables whose name starts with 'shared' are optimised:
        ent_embeds ('shared'+'embeddings') and the three mappings ('shared'+'combination')."""
        self._shared_comb = DenseOptimizerState([self.nv_mapping, self.rv_mapping, self.av_mapping], self.args.learning_rate,
                                                self.args.optimizer)

    # --- read paths -----------------------------------------------------------------------------------------
    def eval_kg1_ent_embeddings(self):
        return self.rv_ent_embeds.lookup(_dev_i32(self.kgs.kg1.entities_list, self.device)).cpu().numpy()

    def eval_kg2_ent_embeddings(self):
        return self.rv_ent_embeds.lookup(_dev_i32(self.kgs.kg2.entities_list, self.device)).cpu().numpy()

    def eval_kg1_useful_ent_embeddings(self):
        return self.rv_ent_embeds.lookup(_dev_i32(self.kgs.useful_entities_list1, self.device)).cpu().numpy()

    def eval_kg2_useful_ent_embeddings(self):
        return self.rv_ent_embeds.lookup(_dev_i32(self.kgs.useful_entities_list2, self.device)).cpu().numpy()

    def save(self):
        """code/MultiKE_model.py:279-287: same six .npy files and id TSVs."""
        arrays = (self.ent_embeds.eval(), self.name_embeds.eval(), self.rv_ent_embeds.eval(), self.av_ent_embeds.eval(),
                  self.rel_embeds.eval(), self.attr_embeds.eval())
        if getattr(self, "_save_async", False):
            # the drivers' closing save: the ~250 MB of .npy files are written by a host thread while the closing tests rank on
            # the device (the arrays are host copies already); `_join_save` at the end of run() waits for the files
            import threading
            self._save_thread = threading.Thread(target=save_embeddings, args=(self.out_folder, self.kgs) + arrays)
            self._save_thread.start()
            return
        save_embeddings(self.out_folder, self.kgs, *arrays)

    def _join_save(self):
        th = getattr(self, "_save_thread", None)
        if th is not None:
            th.join()
            self._save_thread = None

    # --- helpers ----------------------------------------------------------------------------------------------
    _CACHE_MAX = 24

    def _cached(self, key_obj, build):
        """Device mirror of a host list, cached per list OBJECT.  The entry keeps a reference to the list, so its id()
        cannot be recycled for a different list while the entry lives (the predicate lists are re-created every ten
        epochs); the cache is bounded, oldest entry out."""
        key = id(key_obj)
        hit = self._lists.pop(key, None)
        if hit is None or hit[0] is not key_obj or hit[1] != len(key_obj):
            hit = (key_obj, len(key_obj), build(key_obj))
        self._lists[key] = hit              # (re-)inserted last: the entry evicted below is the least recently USED one — the
        while len(self._lists) > self._CACHE_MAX:      # supervision lists of every epoch must outlive the predicate lists that
            self._lists.pop(next(iter(self._lists)))   # are re-created every ten epochs (rebuilding one costs 50-90 ms of host time)
        return hit[2]

    def _list(self, triples) -> _TripleList:
        return self._cached(triples, lambda t: _TripleList(t, self.device))

    def _set_neighbours(self, neighbors1, neighbors2):
        """Truncated-sampling dicts {entity: [k neighbours]} (code/base/batch.py:119-150) -> device candidate tables."""
        held = self._neighbor_ids            # the objects last installed (held, so that their ids cannot be recycled)
        if held[0] is neighbors1 and held[1] is neighbors2 and getattr(self, "_neighbours_installed", False):
            return
        self._neighbor_ids, self._neighbours_installed = (neighbors1, neighbors2), True
        for side, nb in ((self._rel_batcher.side1, neighbors1), (self._rel_batcher.side2, neighbors2)):
            if nb is None or (isinstance(nb, dict) and not nb):
                side.set_neighbours(None, None)
                continue
            if isinstance(nb, tuple):  # already a device (candidate table, valid flags) pair: base.batch.neighbour_table
                side.set_neighbours(nb[0], nb[1])
                continue
            k = min(len(v) for v in nb.values())
            table = np.zeros((self.kgs.entities_num, k), dtype=np.int32)
            valid = np.zeros(self.kgs.entities_num, dtype=np.uint8)
            for e, lst in nb.items():
                table[e] = np.asarray(lst[:k], dtype=np.int32)
                valid[e] = 1
            side.set_neighbours(torch.as_tensor(table, device=self.device), torch.as_tensor(valid, device=self.device))

    def _done(self, pending: _EpochLoss):
        if self._defer_losses:
            self._pending.append(pending)
            return pending
        return pending.finish()

    # --- training for multi-view embeddings ---------------------------------------------------------------
    def train_relation_view_1epo(self, epoch, triple_steps, steps_tasks, batch_queue, neighbors1, neighbors2):
        """code/MultiKE_model.py:291-317.  `steps_tasks` / `batch_queue` are accepted for driver compatibility; batches
        are produced and consumed on the device by the native step runner."""
        start = time.time()
        self._set_neighbours(neighbors1, neighbors2)
        run, b = self._rel_runner, self._rel_batcher
        # the epoch buffers are re-created by every shuffle; they may have been allocated on another stream than the one
        # this epoch runs on (the drivers run this phase on a side stream): tell the allocator, or the old buffers could
        # be handed out again while this stream still reads them
        cur = torch.cuda.current_stream()
        for t_ in (b.pos_h, b.pos_r, b.pos_t, b.pos_kg):
            t_.record_stream(cur)
        steps = min(triple_steps, b.steps)
        trained = int(b.off[steps])
        if run is not None:
            run.run(0, steps)
            total = run.loss[:steps].sum()
            if self._recorder is not None:
                if run.sample_chunk < run.steps or run.overlap:
                    raise _lib.MultiKEHipError("recorder: the relation runner must sample whole epochs (the default)")
                N = b.neg_per_pos
                self._recorder("relation", pos=(b.pos_h[:trained], b.pos_r[:trained], b.pos_t[:trained]),
                               neg=tuple(x[:trained * N] for x in run.neg), off=b.off[:steps + 1].copy(), neg_per_pos=N)
        else:   # Adam / Adadelta: step-wise (sampler launch -> fused score/gradient -> whole-table updates)
            from .sampling import sample_negatives
            N, total = b.neg_per_pos, None
            for s_ in range(steps):
                lo, hi = int(b.off[s_]), int(b.off[s_ + 1])
                pos = (b.pos_h[lo:hi], b.pos_r[lo:hi], b.pos_t[lo:hi])
                neg = sample_negatives(pos, b.side1, N, seed=b.rng_seed, stream_id=b.rng_stream, pos_offset=lo, side1=b.side2,
                                       pos_kg=b.pos_kg[lo:hi]) if N else None
                lp = self.engine.relation_step(self.rv_ent_embeds, self.rel_embeds, "relation", pos, neg, N,
                                               lr=self.args.learning_rate, optimizer=self.args.optimizer).sum()
                total = lp if total is None else total + lp
        self._rel_batcher.shuffle()  # random.shuffle of both positive lists (:314-315)
        return self._done(_EpochLoss('rel. view', epoch, total, trained, start))

    def _attr_lists(self):
        pam = self.predicate_align_model
        w1, w2 = pam.attribute_triples_w_weights1, pam.attribute_triples_w_weights2
        held = getattr(self, "_attr_key", None)          # the list objects themselves: ids alone could be recycled
        if held is None or held[0] is not w1 or held[1] is not w2:
            self._attr_key = (w1, w2)
            self._attr1 = _TripleList(w1, self.device)
            self._attr2 = _TripleList(w2, self.device)
            self._attr_perm = [None, None]
        return self._attr1, self._attr2

    def _attr_epoch_layout(self, n1, n2, b1, b2, steps):
        """Epoch order of the attribute view: step s = [KG1 positions s*b1 .. , KG2 positions s*b2 ..] (code/base/batch.py
        slicing, code/attr_batch.py).  -> (step_off host int64 [steps+1], dest1, dest2 device int64: where position p of
        each KG's (shuffled) list lands in the concatenated epoch arrays)."""
        key = (n1, n2, b1, b2, steps)
        if getattr(self, "_attr_layout_key", None) != key:
            c1 = np.clip(n1 - np.arange(steps) * b1, 0, b1) if b1 > 0 else np.zeros(steps, np.int64)
            c2 = np.clip(n2 - np.arange(steps) * b2, 0, b2) if b2 > 0 else np.zeros(steps, np.int64)
            off = np.zeros(steps + 1, dtype=np.int64)
            off[1:] = np.cumsum(c1 + c2)
            p1, p2 = np.arange(int(c1.sum())), np.arange(int(c2.sum()))
            d1 = off[p1 // max(b1, 1)] + p1 % max(b1, 1) if len(p1) else p1
            d2 = off[p2 // max(b2, 1)] + c1[p2 // max(b2, 1)] + p2 % max(b2, 1) if len(p2) else p2
            self._attr_layout = (off, torch.as_tensor(d1, dtype=torch.int64, device=self.device),
                                 torch.as_tensor(d2, dtype=torch.int64, device=self.device))
            self._attr_layout_key = key
        return self._attr_layout

    def train_attribute_view_1epo(self, epoch, triple_steps, steps_tasks, batch_queue, neighbors1, neighbors2):
        """code/MultiKE_model.py:319-345: weighted positives only (neg_triples_num = 0, :331), CNN scorer.  The epoch's
        batches are laid out once on the device and all steps run inside one native call (`mke_attr_steps`)."""
        from .sampling import kg_batch_split
        start = time.time()
        l1, l2 = self._attr_lists()
        B = self.args.attribute_batch_size
        b1, b2 = kg_batch_split(l1.n, l2.n, B)
        off, d1, d2 = self._attr_epoch_layout(l1.n, l2.n, b1, b2, triple_steps)
        total = int(off[-1])
        value = None
        if total > 0:
            i32, f32 = torch.int32, torch.float32
            cols = [torch.empty(total, dtype=i32, device=self.device) for _ in range(3)] + [torch.empty(total, dtype=f32, device=self.device)]
            for li, (lst, dest) in enumerate(((l1, d1), (l2, d2))):
                m = dest.numel()
                if m == 0:
                    continue
                perm = self._attr_perm[li]
                for k, src in enumerate(lst.cols + (lst.w,)):
                    cols[k][dest] = src[:m] if perm is None else src[perm[:m]]
            value = self._run_attr_steps(self._attr_cnn, cols[:3], cols[3], off, 1.0, "attribute").sum()
            if self._recorder is not None:
                self._recorder("attribute", cols=tuple(cols[:3]), w=cols[3], off=np.asarray(off).copy())
        # random.shuffle of both weighted lists (:342-343): a device permutation applied when the epoch is laid out
        self._attr_perm = [torch.randperm(l.n, generator=self._gen, device=self.device) if l.n else None for l in (l1, l2)]
        return self._done(_EpochLoss('att. view', epoch, value, total, start))

    # --- training for cross-kg identity inference ----------------------------------------------------------
    def _next_sample_stream(self, lane=0):
        """(seed, stream id) of the next `mke_sample_distinct` launch: one stream id per call.  Three independent
        sequences (lane 0: relation group, 1: attribute group, 2: everything else), so that the draws of a phase do not
        depend on the order in which the drivers enqueue the two groups of an epoch."""
        calls = self.__dict__.setdefault("_sample_calls", [0, 0, 0])
        calls[lane] += 1
        return (int(getattr(self.args, "seed", 0)) & 0xFFFFFFFF, 0x4D4B45), calls[lane] * 4 + lane

    def _positives_epoch(self, epoch, sup_triples, batch_size, run_fn, label, lane, key=None):
        """Shared loop shape of code/MultiKE_model.py:349-437: steps = ceil(len / B); each step is
        random.sample(sup_triples, B) (B = len if one step).  All steps of the epoch are sampled by one launch and run
        inside one native call; `run_fn(cols, w, step_off)` returns the loss partials [steps, LOSS_PARTIALS]."""
        if len(sup_triples) == 0:
            return None
        start = time.time()
        lst = self._list(sup_triples)
        steps = int(math.ceil(lst.n / batch_size))
        bs = batch_size if steps > 1 else lst.n
        seed, stream = self._next_sample_stream(lane)
        cols, w, idx = lst.sample_epoch(bs, steps, seed, stream)
        self._last_sample = (seed, stream, lst.n, bs, steps)   # tests replay it with oracle.sampler_oracle.distinct_sample
        ring = run_fn(cols, w, np.arange(steps + 1, dtype=np.int64) * bs)
        if self._recorder is not None:
            self._recorder(key, cols=cols, w=w, off=np.arange(steps + 1, dtype=np.int64) * bs)
        return self._done(_EpochLoss(label, epoch, ring.sum(), steps * bs, start))

    def _run_attr_steps(self, cnn, cols, w, off, scale, opt_name):
        """All steps of an attribute-type epoch: one native call (Adagrad / SGD) or a step-wise loop (Adam / Adadelta)."""
        a = self.args
        tabs = (self.engine, self.av_ent_embeds, self.attr_embeds, self.literal_embeds)
        if a.optimizer not in _DENSE_OPTS:
            return cnn.steps(*tabs, cols[0], cols[1], cols[2], w, off, scale=scale, opt_name=opt_name, lr=a.learning_rate,
                             optimizer=a.optimizer)
        parts = []
        for s_ in range(len(off) - 1):
            lo, hi = int(off[s_]), int(off[s_ + 1])
            parts.append(cnn.step(*tabs, cols[0][lo:hi], cols[1][lo:hi], cols[2][lo:hi], None if w is None else w[lo:hi],
                                  scale=scale, opt_name=opt_name, lr=a.learning_rate, optimizer=a.optimizer).sum())
        return torch.stack(parts) if parts else torch.zeros(0, dtype=torch.float64, device=self.device)

    def _relation_positive_steps(self, g, cols, w, off):
        if self.args.optimizer in _DENSE_OPTS:
            parts = []
            for s_ in range(len(off) - 1):
                lo, hi = int(off[s_]), int(off[s_ + 1])
                parts.append(self.engine.relation_step(self.rv_ent_embeds, self.rel_embeds, g["opt"], tuple(c[lo:hi] for c in cols),
                                                       None, lr=self.args.learning_rate, pos_w=None if w is None else w[lo:hi],
                                                       scale=g["scale"], optimizer=self.args.optimizer).sum())
            return torch.stack(parts) if parts else torch.zeros(0, dtype=torch.float64, device=self.device)
        from .runner import run_positive_steps
        tag_base = self.engine.tag + 1
        self.engine.tag += len(off) - 1
        return run_positive_steps(self.rv_ent_embeds, self.rel_embeds, g["opt"], cols, w, off, tag_base,
                                  lr=self.args.learning_rate, scale=g["scale"], optimizer=self.args.optimizer)

    def train_cross_kg_entity_inference_relation_view_1epo(self, epoch, sup_triples):
        """code/MultiKE_model.py:349-369."""
        return self._positives_epoch(epoch, sup_triples, self.args.batch_size,
                                     lambda cols, w, off: self._relation_positive_steps(self._ckge_rel, cols, None, off),
                                     'cross-kg entity inference in rel. view', 0, 'ckge_rel')

    def train_cross_kg_entity_inference_attribute_view_1epo(self, epoch, sup_triples):
        """code/MultiKE_model.py:371-391: 2 * sum log(1+exp(-conv))."""
        return self._positives_epoch(
            epoch, sup_triples, self.args.attribute_batch_size,
            lambda cols, w, off: self._run_attr_steps(self._ckge_attr_cnn, cols, None, off, 2.0, "ckge_attr"),
            'cross-kg entity inference in attr. view', 1, 'ckge_attr')

    def train_cross_kg_relation_inference_1epo(self, epoch, sup_triples):
        """code/MultiKE_model.py:393-414: weighted 4-tuples, x2."""
        return self._positives_epoch(epoch, sup_triples, self.args.batch_size,
                                     lambda cols, w, off: self._relation_positive_steps(self._ckgp_rel, cols, w, off),
                                     'cross-kg relation inference in rel. view', 0, 'ckgp_rel')

    def train_cross_kg_attribute_inference_1epo(self, epoch, sup_triples):
        """code/MultiKE_model.py:416-437: weighted, not doubled."""
        return self._positives_epoch(
            epoch, sup_triples, self.args.attribute_batch_size,
            lambda cols, w, off: self._run_attr_steps(self._ckga_attr_cnn, cols, w, off, 1.0, "ckga_attr"),
            'cross-kg attribute inference in attr. view', 1, 'ckga_attr')

    # --- shared / common space ------------------------------------------------------------------------------
    def _entity_tensor(self, entities):
        return self._cached(entities, lambda e: _dev_i32(e, self.device))

    def _entity_batches(self, entities, batch_size):
        t = self._entity_tensor(entities)
        steps = int(math.ceil(len(entities) / batch_size))
        bs = batch_size if steps > 1 else len(entities)
        seed, stream = self._next_sample_stream(2)
        self._last_sample = (seed, stream, len(entities), bs, steps)
        picked = t[_lib.sample_distinct(len(entities), bs, steps, seed, stream, device=self.device).long()]   # [steps, bs]
        for s in range(steps):
            yield picked[s], bs

    def train_shared_space_mapping_1epo(self, epoch, entities):
        """code/MultiKE_model.py:439-454 + graph :241-261 (SSL).  Three space_mapping_loss terms.  Adagrad / SGD: one native
        call per epoch.  Adam / Adadelta (or dim > 88): step-wise through the differentiable `losses.space_mapping_loss`."""
        start = time.time()
        st = self._shared_comb
        total, trained = None, 0
        if self.args.optimizer not in _DENSE_OPTS and self.args.dim <= 88 and len(entities):
            # native path: every step of the epoch inside one call (`mke_mapping_steps`): MFMA GEMMs for V M and V^T dP, the
            # batch-wide normalisation as two partial-sum passes, one launch for the row update + the three matrices
            from .runner import SpaceMappingState, run_space_mapping_steps
            if getattr(self, "_map_state", None) is None:
                self._map_state = SpaceMappingState(st.vars, self.device)
                for k, name in enumerate(("nv_mapping", "rv_mapping", "av_mapping")):   # the matrices now live in the pack
                    setattr(self, name, self._map_state.M[k])
                st.vars = [self._map_state.M[k] for k in range(3)]
            t = self._entity_tensor(entities)
            B = self.args.entity_batch_size
            steps = int(math.ceil(len(entities) / B))
            bs = B if steps > 1 else len(entities)
            seed, stream = self._next_sample_stream(2)
            self._last_sample = (seed, stream, len(entities), bs, steps)
            idx = t[_lib.sample_distinct(len(entities), bs, steps, seed, stream, device=self.device).reshape(-1).long()]
            tag_base = self.engine.tag + 1
            self.engine.tag += steps
            ring = run_space_mapping_steps(self._map_state, self.ent_embeds, [self.name_embeds, self.rv_ent_embeds, self.av_ent_embeds],
                                           idx, np.arange(steps + 1, dtype=np.int64) * bs, "shared_comb", tag_base,
                                           self.args.learning_rate, self.args.orthogonal_weight, optimizer=self.args.optimizer)
            if self._recorder is not None:
                self._recorder("mapping", idx=idx, off=np.arange(steps + 1, dtype=np.int64) * bs)
            epoch_loss = float(ring.sum()) / (steps * bs)
            print('epoch {} of shared space learning, avg. loss: {:.4f}, time: {:.4f}s'.format(epoch, epoch_loss,
                                                                                               time.time() - start))
            return epoch_loss
        for idx, bs in self._entity_batches(entities, self.args.entity_batch_size):
            final = self.ent_embeds.lookup(idx).requires_grad_(True)
            views = (self.name_embeds.lookup(idx), self.rv_ent_embeds.lookup(idx), self.av_ent_embeds.lookup(idx))
            loss = sum(losses.space_mapping_loss(v, final, m, self.eye_mat, self.args.orthogonal_weight)
                       for v, m in zip(views, st.vars))
            grads = torch.autograd.grad(loss, [final] + st.vars)
            st.apply(grads[1:])
            tag, _ = self.engine._next()
            self.ent_embeds.grad[:, :self.args.dim].index_add_(0, idx.long(), grads[0])
            self.ent_embeds.touched[idx.long()] = tag
            self.engine._apply(self.ent_embeds, "shared_comb", self.args.optimizer, self.args.learning_rate, tag)
            total = loss.detach() if total is None else total + loss.detach()
            trained += bs
        epoch_loss = float(total) / max(trained, 1)
        print('epoch {} of shared space learning, avg. loss: {:.4f}, time: {:.4f}s'.format(epoch, epoch_loss,
                                                                                           time.time() - start))
        return epoch_loss

    def train_common_space_learning_1epo(self, epoch, entities):
        """code/MultiKE_model.py:458-473 + graph :225-239 (ITC): cv_name_weight*align(ent,name) + align(ent,rv) +
        align(ent,av); the optimizer minimises cv_weight * loss with ITC_learning_rate; the reported loss is unscaled."""
        start = time.time()
        g = self._cross_name
        cvw = float(self.args.cv_weight)
        total, trained = None, 0
        if self.args.optimizer not in _DENSE_OPTS and len(entities):
            # all steps of the epoch in one native call: one sampler launch, one gather of the entity ids
            from .runner import run_alignment_steps
            t = self._entity_tensor(entities)
            B = self.args.entity_batch_size
            steps = int(math.ceil(len(entities) / B))
            bs = B if steps > 1 else len(entities)
            seed, stream = self._next_sample_stream(2)
            self._last_sample = (seed, stream, len(entities), bs, steps)
            idx = t[_lib.sample_distinct(len(entities), bs, steps, seed, stream, device=self.device).reshape(-1).long()]
            tag_base = self.engine.tag + 1
            self.engine.tag += steps
            ring = run_alignment_steps([self.ent_embeds, self.name_embeds, self.rv_ent_embeds, self.av_ent_embeds],
                                       [(0, 1, cvw * float(self.args.cv_name_weight)), (0, 2, cvw), (0, 3, cvw)], idx, idx,
                                       np.arange(steps + 1, dtype=np.int64) * bs, g["opt"], tag_base, g["lr"],
                                       optimizer=self.args.optimizer)
            if self._recorder is not None:
                self._recorder("common", idx=idx, off=np.arange(steps + 1, dtype=np.int64) * bs)
            epoch_loss = float(ring.sum()) / cvw / (steps * bs) if cvw != 0 else 0.0
            print('epoch {} of common space learning, avg. loss: {:.4f}, time: {:.4f}s'.format(epoch, epoch_loss,
                                                                                               time.time() - start))
            return epoch_loss
        for idx, bs in self._entity_batches(entities, self.args.entity_batch_size):
            terms = [(self.ent_embeds, idx, self.name_embeds, idx, cvw * float(self.args.cv_name_weight)),
                     (self.ent_embeds, idx, self.rv_ent_embeds, idx, cvw),
                     (self.ent_embeds, idx, self.av_ent_embeds, idx, cvw)]
            s = self.engine.alignment_step(terms, g["opt"], g["lr"], optimizer=self.args.optimizer)
            total = s if total is None else total + s
            trained += bs
        epoch_loss = float(total) / cvw / max(trained, 1) if cvw != 0 else 0.0
        print('epoch {} of common space learning, avg. loss: {:.4f}, time: {:.4f}s'.format(epoch, epoch_loss,
                                                                                           time.time() - start))
        return epoch_loss
