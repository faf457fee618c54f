"""A recording stand-in for the MultiKE model, shared by `tests/golden/make_golden.py` (which runs the REFERENCE's
`MultiKE_CV.run` / `MultiKE_Late.run` on it, code/MultiKE_CSL.py:36-107, code/MultiKE_Late.py:201-280) and by
`tests/test_schedule_golden.py` (which runs the PRODUCT's drivers on it).  Both sides see the same objects and append to the
same kind of trace, so the two traces can be compared event for event: phase order, the `i > start_predicate_soft_alignment`
/ `i % 10` / `i % eval_freq` / `i % truncated_freq` gates, the lists handed to every phase (they change when the soft predicate
alignment is refreshed), which neighbour tables the relation view receives, the early `break`, the closing `save` + tests.

Nothing here trains anything: every `train_*_1epo`, `valid`, `test`, ... is replaced by a recorder.
"""
import types

# every scenario: overrides of the reference's args.json values + the validation round after which `early_stop` is raised
# (None: never — the reference itself never sets it; the `break` it guards is exercised here all the same)
SCENARIOS = {
    "default_gates_30_epochs": dict(max_epoch=30, shared_learning_max_epoch=4, start_valid=10, eval_freq=10,
                                    start_predicate_soft_alignment=10, truncated_freq=20, neg_sampling="truncated"),
    "small_gates": dict(max_epoch=12, shared_learning_max_epoch=5, start_valid=2, eval_freq=2, start_predicate_soft_alignment=3,
                        truncated_freq=3, neg_sampling="truncated"),
    "uniform_sampling": dict(max_epoch=11, shared_learning_max_epoch=3, start_valid=4, eval_freq=3, start_predicate_soft_alignment=0,
                             truncated_freq=2, neg_sampling="uniform"),
    "max_epoch_on_a_validation_epoch": dict(max_epoch=20, shared_learning_max_epoch=2, start_valid=5, eval_freq=5,
                                            start_predicate_soft_alignment=5, truncated_freq=10, neg_sampling="truncated"),
    "early_stop_after_second_validation": dict(max_epoch=40, shared_learning_max_epoch=6, start_valid=3, eval_freq=3,
                                               start_predicate_soft_alignment=2, truncated_freq=4, neg_sampling="truncated",
                                               _early_stop_round=2),
    "valid_never_reached": dict(max_epoch=5, shared_learning_max_epoch=0, start_valid=100, eval_freq=10,
                                start_predicate_soft_alignment=10, truncated_freq=20, neg_sampling="truncated"),
}

BASE_ARGS = dict(batch_size=5000, attribute_batch_size=5000, entity_batch_size=5000, batch_threads_num=4, test_threads_num=8,
                 neg_triple_num=10, truncated_epsilon=0.98, top_k=[1, 5, 10, 50], optimizer="Adagrad", learning_rate=0.001,
                 dim=75, alignment_module="swapping")


class _Embeds:
    """`model.rel_embeds.eval(session=...)` / `.lookup(ids)`: returns a token naming the variable."""

    def __init__(self, name):
        self.name = name

    def eval(self, session=None):
        return "eval:" + self.name

    def lookup(self, ids):
        return "lookup:" + self.name


class _Neighbours(dict):
    """What `generate_neighbours` / `neighbour_table` returns here: has a len() (the reference prints it) and a tag."""

    def __init__(self, tag):
        super().__init__(tag=tag)
        self.tag = tag


def tag_of(nb):
    if nb is None:
        return None
    if isinstance(nb, tuple):          # the product hands a (table, valid) pair per KG
        nb = nb[0]
    return getattr(nb, "tag", repr(type(nb)))


class _PredicateAlignModel:
    """sup_*_alignment_triples{1,2}: lists whose CONTENT carries the refresh count, so a trace shows which version a phase got."""

    def __init__(self, trace):
        self._trace, self.version = trace, {"relation": 0, "attribute": 0}
        self._fill()

    def _fill(self):
        r, a = self.version["relation"], self.version["attribute"]
        self.sup_relation_alignment_triples1 = [[1, 100 + r, 2, 0.9]] * (2 + r)
        self.sup_relation_alignment_triples2 = [[1001, 200 + r, 1002, 0.9]]
        self.sup_attribute_alignment_triples1 = [[3, 300 + a, 4, 0.9]] * (1 + a)
        self.sup_attribute_alignment_triples2 = [[1003, 400 + a, 1004, 0.9]] * 2

    def update_predicate_alignment(self, embeds, predicate_type='relation'):
        self._trace.append(["update_predicate_alignment", predicate_type, embeds])
        self.version[predicate_type] += 1
        self._fill()


def make_args(scenario: str):
    d = dict(BASE_ARGS)
    d.update({k: v for k, v in SCENARIOS[scenario].items() if not k.startswith("_")})
    return types.SimpleNamespace(**d)


def make_kgs():
    n = 1000                                                  # entities per KG: k = int(0.02 * 1000) = 20 neighbours
    kg = lambda lo, nr, na: types.SimpleNamespace(
        local_relation_triples_num=nr, local_attribute_triples_num=na, entities_num=n,
        entities_list=list(range(lo, lo + n)),
        sup_relation_triples_list=[[lo, 1, lo + 1]] * 3, sup_attribute_triples_list=[[lo, 2, 7]] * 2)
    kg1, kg2 = kg(0, 23_456, 31_111), kg(n, 17_001, 9_999)   # 40,457 relation / 41,110 attribute triples: 9 and 9 steps of 5000
    return types.SimpleNamespace(kg1=kg1, kg2=kg2, entities_num=2 * n, useful_entities_list1=list(range(0, n, 2)) * 2,
                                 useful_entities_list2=list(range(n, 2 * n)), valid_entities1=[1], valid_entities2=[n + 1],
                                 test_entities1=[2], test_entities2=[n + 2])


def instrument(model, scenario: str, trace: list):
    """Attach the recording attributes to `model` (an instance created WITHOUT running __init__).  Returns the callables the
    drivers' module-level names (`valid`, `test`, `valid_WVA`, `test_WVA`, neighbour builders) must be replaced with."""
    model.args = make_args(scenario)
    model.kgs = make_kgs()
    model.kg1, model.kg2 = model.kgs.kg1, model.kgs.kg2
    model.predicate_align_model = _PredicateAlignModel(trace)
    model.session = None
    model.flag1, model.flag2, model.early_stop = -1, -1, False
    model.device = "cpu"
    model.overlap_views = False                 # product only: one stream (the two-stream enqueue order is a GPU matter)
    for name in ("name_embeds", "rv_ent_embeds", "av_ent_embeds", "ent_embeds", "rel_embeds", "attr_embeds"):
        setattr(model, name, _Embeds(name))
    lst = lambda x: [list(t) for t in x]

    def rel_view(i, steps, tasks, queue, n1, n2):
        trace.append(["train_relation_view_1epo", i, steps, [list(t) for t in tasks], tag_of(n1), tag_of(n2)])

    def attr_view(i, steps, tasks, queue, n1, n2):
        trace.append(["train_attribute_view_1epo", i, steps, [list(t) for t in tasks], tag_of(n1), tag_of(n2)])

    model.train_relation_view_1epo = rel_view
    model.train_attribute_view_1epo = attr_view
    for name in ("train_cross_kg_entity_inference_relation_view_1epo", "train_cross_kg_relation_inference_1epo",
                 "train_cross_kg_entity_inference_attribute_view_1epo", "train_cross_kg_attribute_inference_1epo"):
        setattr(model, name, (lambda nm: lambda i, triples: trace.append([nm, i, lst(triples)]))(name))
    for name in ("train_common_space_learning_1epo", "train_shared_space_mapping_1epo"):
        setattr(model, name, (lambda nm: lambda i, ents: trace.append([nm, i, len(ents), ents[0], ents[-1]]))(name))
    model.eval_kg1_useful_ent_embeddings = lambda: "lookup:rv_ent_embeds"
    model.eval_kg2_useful_ent_embeddings = lambda: "lookup:rv_ent_embeds"
    model.save = lambda: trace.append(["save"])

    stop_round = SCENARIOS[scenario].get("_early_stop_round")
    rounds = {"n": 0}

    def valid(m, embed_choice='avg', w=(1, 1, 1)):
        trace.append(["valid", embed_choice])
        if embed_choice == 'rv':                # first validation call of a round in both drivers
            rounds["n"] += 1
            if stop_round is not None and rounds["n"] >= stop_round:
                m.early_stop = True
        return 0.5

    def test(m, embed_choice='avg', w=(1, 1, 1)):
        trace.append(["test", embed_choice])
        return 0.5

    def valid_wva(m):
        trace.append(["valid_WVA"])
        return 0.5

    def test_wva(m):
        trace.append(["test_WVA"])
        return 0.5

    calls = {"n": 0}

    def neighbours(embeds, entity_list, k, *rest, **kw):
        """bat.generate_neighbours(embeds, entities, k, threads) / neighbour_table(embeds, entities, k, n_total, device=)."""
        calls["n"] += 1
        trace.append(["generate_neighbours", embeds, len(entity_list), k])
        return _Neighbours(f"nb{calls['n']}")

    return dict(valid=valid, test=test, valid_WVA=valid_wva, test_WVA=test_wva, neighbours=neighbours)
