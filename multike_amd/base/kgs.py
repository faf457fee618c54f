"""On-disk inputs -> the id-space containers the hot path consumes.

Host-side data preparation only (no kernels): the TSV readers, URI -> id assignment, the `KG` / `KGs` containers and
the 'swapping' supervision triples of the reference (code/base/read.py:13-110,130-167,216-243,341-364,
code/base/kg.py:1-143, code/base/kgs.py:5-97), rebuilt around one idea: every container keeps the reference's
attribute names (lists / sets / dicts the drivers and batchers read) *and* a packed int32 array of its id triples
(`relation_triples_array`, `sup_relation_triples_array`) that is what actually gets uploaded to HBM.

Differences from the reference, all deliberate:
  * `ordered=False` id assignment and every `list(set)` there follow Python's per-process string-hash order, i.e.
    the reference is not reproducible run to run.  Here unordered means *first appearance in the input file*, and
    set -> list conversions are sorted, so the same folder always yields the same ids.  The id *layout* is the
    reference's: KG1 ids [0, n1), KG2 ids [n1, n1+n2) (code/base/read.py:75-84); `ordered=True` (frequency
    order, interleaved 2i / 2i+1, code/base/read.py:61-74) is reproduced exactly and pinned by
    tests/golden/data_golden.json.
  * malformed lines raise `ValueError` naming file and line instead of a bare `assert`.
"""
from __future__ import annotations

from collections import Counter

import numpy as np

__all__ = ["KG", "KGs", "TripleArray", "read_relation_triples", "read_attribute_triples", "read_links", "read_dict", "read_pair_ids",
           "pair2file", "dict2file", "line2file", "sort_elements", "generate_mapping_id", "generate_sharing_id",
           "uris_list_2ids", "uris_pair_2ids", "uris_relation_triple_2ids", "uris_attribute_triple_2ids",
           "generate_sup_relation_triples", "generate_sup_attribute_triples", "swap_relation_triples", "swap_attribute_triples",
           "read_kgs_from_folder",
           "read_kgs_from_files", "parse_triples"]


# ----------------------------------------------------------------------------------------------------------------
# a list of id triples kept as columns
# ----------------------------------------------------------------------------------------------------------------
class TripleArray:
    """A list of (h, p, t) or (h, p, t, w) id tuples stored as an int array [n, 3] (+ float64 weights): what the
    reference builds as Python lists of tuples with per-element loops (code/predicate_alignment.py:17-44), here produced
    and concatenated with array operations and uploaded to HBM without a per-tuple conversion.  Reads like the list it
    replaces: len(), iteration / indexing yield tuples, `a + b` concatenates."""

    __slots__ = ("_cols", "_w", "dev", "_host", "_n")

    def __init__(self, cols, w=None):
        self._cols = np.ascontiguousarray(cols, dtype=np.int64).reshape(-1, 3)
        self._w = None if w is None else np.ascontiguousarray(w, dtype=np.float64).reshape(-1)
        self.dev, self._host, self._n = None, None, len(self._cols)
        if self._w is not None and len(self._w) != len(self._cols):
            raise ValueError("weights and triples differ in length")

    @classmethod
    def on_device(cls, dev_cols, dev_w, host):
        """A list that was produced in HBM: `dev_cols` three int32 device columns, `dev_w` float32 device weights (or None) —
        what the training loops read, without an upload — and `host()` -> (int array [n, 3], float64 weights or None), the
        same list made on the host, called only if somebody reads the list as a list."""
        self = cls.__new__(cls)
        self._cols, self._w, self.dev, self._host, self._n = None, None, (tuple(dev_cols), dev_w), host, int(dev_cols[0].shape[0])
        return self

    def _materialise(self):
        if self._cols is None:
            cols, w = self._host()
            self._cols = np.ascontiguousarray(cols, dtype=np.int64).reshape(-1, 3)
            self._w = None if w is None else np.ascontiguousarray(w, dtype=np.float64).reshape(-1)
            if len(self._cols) != self._n:
                raise RuntimeError("the device-built triple list and its host form differ in length")

    @property
    def cols(self):
        self._materialise()
        return self._cols

    @property
    def w(self):
        self._materialise()
        return self._w

    @property
    def weighted(self):
        return (self.dev[1] is not None) if self._cols is None else (self._w is not None)

    def __len__(self):
        return self._n

    def _tuple(self, i):
        h, p, t = self.cols[i].tolist()
        return (h, p, t) if self.w is None else (h, p, t, float(self.w[i]))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return TripleArray(self.cols[i], None if self.w is None else self.w[i])
        return self._tuple(i)

    def __iter__(self):
        rows = self.cols.tolist()
        if self.w is None:
            return (tuple(r) for r in rows)
        return ((r[0], r[1], r[2], w) for r, w in zip(rows, self.w.tolist()))

    def __add__(self, other):
        if not isinstance(other, TripleArray):
            other = TripleArray.from_tuples(other)
        if self.dev is not None and other.dev is not None and self.weighted == other.weighted:   # both live in HBM: so does the sum
            import torch
            a, b = self, other

            def host():
                c = _HostView(a) + _HostView(b)
                return c.cols, c.w
            return TripleArray.on_device(tuple(torch.cat([x, y]) for x, y in zip(a.dev[0], b.dev[0])),
                                         None if a.dev[1] is None else torch.cat([a.dev[1], b.dev[1]]), host)
        if (self.w is None) != (other.w is None) and len(self) and len(other):
            raise ValueError("cannot concatenate weighted and unweighted triples")
        w = None if (self.w is None and other.w is None) else np.concatenate(
            [x.w if x.w is not None else np.zeros(0) for x in (self, other)])
        return TripleArray(np.concatenate([self.cols, other.cols]), w)

    __radd__ = lambda self, other: TripleArray.from_tuples(other) + self   # noqa: E731

    @classmethod
    def from_tuples(cls, triples):
        triples = list(triples)
        if not triples:
            return cls(np.zeros((0, 3), dtype=np.int64))
        return cls([t[:3] for t in triples], [t[3] for t in triples] if len(triples[0]) > 3 else None)


def _HostView(x):
    """The same list without its device form (so that `a + _HostView(b)` concatenates on the host)."""
    return TripleArray(x.cols, x.w)


# ----------------------------------------------------------------------------------------------------------------
# readers / writers (code/base/read.py:216-299,341-364)
# ----------------------------------------------------------------------------------------------------------------
class _OrderedSet(dict):
    """A set that remembers first-insertion order (what makes `ordered=False` deterministic here).  Behaves as a
    set for everything the callers do with the reference's return values (len, in, iteration, |, -)."""

    def add(self, x):
        self[x] = None

    def __or__(self, other):
        out = _OrderedSet(self)
        for x in other:
            out[x] = None
        return out

    def __sub__(self, other):
        return _OrderedSet((x, None) for x in self if x not in other)


def _fields(path):
    with open(path, "r", encoding="utf8") as f:
        for no, line in enumerate(f, 1):
            yield no, line.rstrip("\n").split("\t")


def read_relation_triples(file_path):
    """`h \\t r \\t t` per line -> (triples, entities, relations), fields stripped (code/base/read.py:216-232)."""
    triples, entities, relations = _OrderedSet(), _OrderedSet(), _OrderedSet()
    if file_path is None:
        return triples, entities, relations
    with open(file_path, "r", encoding="utf8") as f:
        for no, line in enumerate(f, 1):
            p = line.rstrip("\n").split("\t")
            if len(p) != 3:
                raise ValueError(f"{file_path}:{no}: expected 3 tab-separated fields, got {len(p)}")
            h, r, t = p[0].strip(), p[1].strip(), p[2].strip()
            triples[(h, r, t)] = None          # _OrderedSet is a dict: plain stores, no method call per line
            entities[h] = None
            entities[t] = None
            relations[r] = None
    return triples, entities, relations


def read_attribute_triples(file_path):
    """>= 3 tab fields (shorter lines skipped); fields past the third are appended to the value with single spaces;
    a trailing '.' (N-Triples terminator) is dropped (code/base/read.py:341-364)."""
    triples, entities, attributes = _OrderedSet(), _OrderedSet(), _OrderedSet()
    if file_path is None:
        return triples, entities, attributes
    with open(file_path, "r", encoding="utf8") as f:
        for line in f:
            p = line.strip().split("\t")
            if len(p) < 3:
                continue
            head, attr = p[0].strip(), p[1].strip()
            value = " ".join([p[2].strip()] + [x.strip() for x in p[3:]]) if len(p) > 3 else p[2].strip()
            value = value.strip().rstrip(".").strip()
            triples[(head, attr, value)] = None
            entities[head] = None
            attributes[attr] = None
    return triples, entities, attributes


def read_links(file_path):
    """`e1 \\t e2` per line -> list of pairs in file order (code/base/read.py:235-250)."""
    links = []
    for no, p in _fields(file_path):
        if len(p) != 2:
            raise ValueError(f"{file_path}:{no}: expected 2 tab-separated fields, got {len(p)}")
        links.append((p[0].strip(), p[1].strip()))
    return links


def read_dict(file_path):
    """`key \\t int` per line (the kg*_ids files `save_embeddings` writes; code/base/read.py:253-262)."""
    out = {}
    for no, p in _fields(file_path):
        if len(p) != 2:
            raise ValueError(f"{file_path}:{no}: expected 2 tab-separated fields")
        out[p[0]] = int(p[1])
    return out


def read_pair_ids(file_path):
    return [(int(a), int(b)) for _, (a, b) in _fields(file_path)]


def pair2file(file, pairs):
    if pairs is None:
        return
    with open(file, "w", encoding="utf8") as f:
        f.writelines(f"{i}\t{j}\n" for i, j in pairs)


def dict2file(file, dic):
    if dic is None:
        return
    pair2file(file, dic.items())


def line2file(file, lines):
    if lines is None:
        return
    with open(file, "w", encoding="utf8") as f:
        f.writelines(line + "\n" for line in lines)


# ----------------------------------------------------------------------------------------------------------------
# id assignment (code/base/read.py:13-84)
# ----------------------------------------------------------------------------------------------------------------
def sort_elements(triples, elements_set):
    """Elements by descending (occurrence count over all three triple positions, URI) (code/base/read.py:13-25)."""
    freq = Counter()
    for tri in triples:
        for x in tri:
            if x in elements_set:
                freq[x] += 1
    ranked = sorted(freq.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)
    return [k for k, _ in ranked], dict(freq)


def generate_mapping_id(kg1_triples, kg1_elements, kg2_triples, kg2_elements, ordered=True):
    """Disjoint id spaces.  ordered: i-th most frequent element of KG1 -> 2i, of KG2 -> 2i+1, the longer tail
    continues contiguously after 2*min(n1,n2) (code/base/read.py:59-74).  unordered: KG1 first, then KG2
    (code/base/read.py:75-84)."""
    if ordered:
        o1, _ = sort_elements(kg1_triples, kg1_elements)
        o2, _ = sort_elements(kg2_triples, kg2_elements)
        m = min(len(o1), len(o2))
        ids1 = {e: 2 * i if i < m else 2 * m + (i - m) for i, e in enumerate(o1)}
        ids2 = {e: 2 * i + 1 if i < m else 2 * m + (i - m) for i, e in enumerate(o2)}
    else:
        ids1 = {e: i for i, e in enumerate(dict.fromkeys(kg1_elements))}
        n1 = len(ids1)
        ids2 = {e: n1 + i for i, e in enumerate(dict.fromkeys(kg2_elements))}
    if len(ids1) != len(set(kg1_elements)) or len(ids2) != len(set(kg2_elements)):
        raise ValueError("id assignment lost elements (an element never occurs in its KG's triples)")
    return ids1, ids2


def generate_sharing_id(train_links, kg1_triples, kg1_elements, kg2_triples, kg2_elements, ordered=True):
    """Linked elements share one id (code/base/read.py:28-56)."""
    if ordered:
        partner = {y: x for x, y in train_links}
        linked2 = [y for _, y in train_links]
        ids1, ids2 = generate_mapping_id(kg1_triples, kg1_elements, kg2_triples,
                                         [e for e in dict.fromkeys(kg2_elements) if e not in partner], ordered=True)
        for y in linked2:
            ids2[y] = ids1[partner[y]]
    else:
        ids1, ids2 = {}, {}
        for e1, e2 in train_links:
            if e1 not in kg1_elements or e2 not in kg2_elements:
                raise ValueError(f"link ({e1}, {e2}) names an element outside the KGs")
            ids1[e1] = ids2[e2] = len(ids1)
        nxt = len(ids1)
        for ids, elements in ((ids1, kg1_elements), (ids2, kg2_elements)):
            for e in dict.fromkeys(elements):
                if e not in ids:
                    ids[e] = nxt
                    nxt += 1
    if len(ids1) != len(set(kg1_elements)) or len(ids2) != len(set(kg2_elements)):
        raise ValueError("id assignment lost elements")
    return ids1, ids2


def _need(key, table, what):
    try:
        return table[key]
    except KeyError:
        raise ValueError(f"{what} {key!r} has no id") from None


def uris_list_2ids(uris, ids):
    return [_need(u, ids, "element") for u in uris]


def uris_pair_2ids(uris, ids1, ids2):
    return [(_need(a, ids1, "KG1 entity"), _need(b, ids2, "KG2 entity")) for a, b in uris]


def uris_relation_triple_2ids(uris, ent_ids, rel_ids):
    try:
        return [(ent_ids[h], rel_ids[r], ent_ids[t]) for h, r, t in uris]
    except KeyError:      # slow pass only to name the offender
        return [(_need(h, ent_ids, "entity"), _need(r, rel_ids, "relation"), _need(t, ent_ids, "entity")) for h, r, t in uris]


def uris_attribute_triple_2ids(uris, ent_ids, attr_ids):
    """Heads must be entities of the relation graph; the value stays a string (code/base/read.py:120-127)."""
    try:
        return [(ent_ids[h], attr_ids[a], v) for h, a, v in uris]
    except KeyError:
        return [(_need(h, ent_ids, "entity (attribute-triple head)"), _need(a, attr_ids, "attribute"), v) for h, a, v in uris]


# ----------------------------------------------------------------------------------------------------------------
# 'swapping' supervision (code/base/read.py:130-167)
# ----------------------------------------------------------------------------------------------------------------
def generate_sup_relation_triples(sup_links, rt_dict1, hr_dict1, rt_dict2, hr_dict2):
    """Every triple touching a linked entity is re-stated with the counterpart in that position; the KG1-derived
    set joins KG1's triples, the KG2-derived set KG2's."""
    new1, new2 = set(), set()
    for e1, e2 in sup_links:
        new1.update((e2, r, t) for r, t in rt_dict1.get(e1, ()))
        new1.update((h, r, e2) for h, r in hr_dict1.get(e1, ()))
        new2.update((e1, r, t) for r, t in rt_dict2.get(e2, ()))
        new2.update((h, r, e1) for h, r in hr_dict2.get(e2, ()))
    return new1, new2


def generate_sup_attribute_triples(sup_links, av_dict1, av_dict2):
    new1, new2 = set(), set()
    for e1, e2 in sup_links:
        new1.update((e2, a, v) for a, v in av_dict1.get(e1, ()))
        new2.update((e1, a, v) for a, v in av_dict2.get(e2, ()))
    return new1, new2


# ----------------------------------------------------------------------------------------------------------------
# containers (code/base/kg.py, code/base/kgs.py)
# ----------------------------------------------------------------------------------------------------------------
def parse_triples(triples):
    s, p, o = set(), set(), set()
    for a, b, c in triples:
        s.add(a)
        p.add(b)
        o.add(c)
    return s, p, o


def _stable_list(items):
    """set -> list in a reproducible order (ints / strings / tuples thereof)."""
    try:
        return sorted(items)
    except TypeError:
        return sorted(items, key=repr)


def _group(pairs):
    out = {}
    for k, v in pairs:
        out.setdefault(k, set()).add(v)
    return out


def swap_relation_triples(triples, link_map):
    """'swapping' supervision of one KG in one pass over its triples (same set as `generate_sup_relation_triples` builds
    from rt_dict / hr_dict, code/base/read.py:130-146): every triple whose head (tail) is a linked entity, re-stated with
    the counterpart in that position."""
    new = {(link_map[h], r, t) for h, r, t in triples if h in link_map}
    new.update((h, r, link_map[t]) for h, r, t in triples if t in link_map)
    return new


def swap_attribute_triples(triples, link_map):
    """code/base/read.py:149-164 in one pass."""
    return {(link_map[h], a, v) for h, a, v in triples if h in link_map}


class KG:
    """One knowledge graph, either in URI space or in id space (code/base/kg.py:10-143).

    `relation_triples_set` and `local_relation_triples_set` are the SAME set object, as in the reference
    (code/base/kg.py:58-61), so `add_sup_relation_triples` also grows the "local" set -- that is the set the
    negative sampler filters against -- while `local_relation_triples_list` / `_num` keep the
    pre-supervision triples the relation view trains on.

    Everything derivable from the two triple sets (the sorted lists, rt_dict / hr_dict / av_dict, the per-entity predicate
    dicts) is computed on first access: the reference builds all of it eagerly for four KG objects per run (two of them the
    URI-space KGs, whose dicts nothing reads), which at 100K entities per KG was 12 s of an 18 s load."""

    _REL_DERIVED = ("relation_triples_list", "local_relation_triples_list", "entities_list", "relations_list", "rt_dict", "hr_dict",
                    "entity_relations_dict")
    _ATTR_DERIVED = ("attribute_triples_list", "local_attribute_triples_list", "attributes_list", "av_dict",
                     "entity_attributes_dict")

    def __init__(self, relation_triples, attribute_triples, verbose=False):
        self.entities_id_dict = self.relations_id_dict = self.attributes_id_dict = None
        self.sup_relation_triples_set, self.sup_relation_triples_list = None, None
        self.sup_attribute_triples_set, self.sup_attribute_triples_list = None, None
        self.set_relations(relation_triples)
        self.set_attributes(attribute_triples)
        if verbose:
            print(self.statistics())

    def __getattr__(self, name):          # only reached when `name` is not set: the lazily derived attributes
        if name == "local_relation_triples_list":
            v = _stable_list(self.local_relation_triples_set)     # add_sup_* snapshots this before the alias grows
        elif name == "relation_triples_list":
            v = self.local_relation_triples_list if self.sup_relation_triples_set is None else _stable_list(self.relation_triples_set)
        elif name == "local_attribute_triples_list":
            v = _stable_list(self.local_attribute_triples_set)
        elif name == "attribute_triples_list":
            v = self.local_attribute_triples_list if self.sup_attribute_triples_set is None else _stable_list(self.attribute_triples_set)
        elif name == "entities_list":
            v = _stable_list(self.entities_set)
        elif name == "relations_list":
            v = _stable_list(self.relations_set)
        elif name == "attributes_list":
            v = _stable_list(self.attributes_set)
        elif name == "rt_dict":
            v = _group((h, (r, t)) for h, r, t in self.local_relation_triples_list)
        elif name == "hr_dict":
            v = _group((t, (h, r)) for h, r, t in self.local_relation_triples_list)
        elif name == "entity_relations_dict":
            v = _group((h, r) for h, r, _ in self.local_relation_triples_list)
        elif name == "av_dict":
            v = _group((h, (a, v_)) for h, a, v_ in self.local_attribute_triples_list)
        elif name == "entity_attributes_dict":
            v = _group((h, a) for h, a, _ in self.local_attribute_triples_list)
        else:
            raise AttributeError(name)
        self.__dict__[name] = v
        return v

    def _forget(self, names):
        for n in names:
            self.__dict__.pop(n, None)

    def statistics(self) -> str:
        return (f"KG: {self.entities_num} entities, {self.relations_num} relations, {self.attributes_num} attributes, "
                f"{self.relation_triples_num} relation triples ({self.local_relation_triples_num} local), "
                f"{self.attribute_triples_num} attribute triples ({self.local_attribute_triples_num} local)")

    # -- relation side --
    def set_relations(self, relation_triples):
        self._forget(self._REL_DERIVED)
        self.sup_relation_triples_set, self.sup_relation_triples_list = None, None
        self.relation_triples_set = self.local_relation_triples_set = set(relation_triples)
        heads, self.relations_set, tails = parse_triples(self.relation_triples_set)
        self.entities_set = heads | tails
        self.entities_num, self.relations_num = len(self.entities_set), len(self.relations_set)
        self.relation_triples_num = self.local_relation_triples_num = len(self.relation_triples_set)

    def generate_relation_triple_dict(self):
        self._forget(("rt_dict", "hr_dict"))
        return self.rt_dict, self.hr_dict

    def parse_relations(self):
        self._forget(("entity_relations_dict",))
        return self.entity_relations_dict

    # -- attribute side --
    def set_attributes(self, attribute_triples):
        self._forget(self._ATTR_DERIVED)
        self.sup_attribute_triples_set, self.sup_attribute_triples_list = None, None
        self.attribute_triples_set = self.local_attribute_triples_set = set(attribute_triples)
        _, self.attributes_set, _ = parse_triples(self.attribute_triples_set)
        self.attributes_num = len(self.attributes_set)
        self.attribute_triples_num = self.local_attribute_triples_num = len(self.attribute_triples_set)

    def generate_attribute_triple_dict(self):
        self._forget(("av_dict",))
        return self.av_dict

    def parse_attributes(self):
        self._forget(("entity_attributes_dict",))
        return self.entity_attributes_dict

    def set_id_dict(self, entities_id_dict, relations_id_dict, attributes_id_dict):
        self.entities_id_dict = entities_id_dict
        self.relations_id_dict = relations_id_dict
        self.attributes_id_dict = attributes_id_dict

    # -- supervision --
    def add_sup_relation_triples(self, sup_triples):
        self.local_relation_triples_list                                   # snapshot of the pre-supervision triples
        self.sup_relation_triples_set = set(sup_triples)
        self.sup_relation_triples_list = _stable_list(self.sup_relation_triples_set)
        self.relation_triples_set |= self.sup_relation_triples_set          # in place: the alias grows too
        self._forget(("relation_triples_list",))
        self.relation_triples_num = len(self.relation_triples_set)

    def add_sup_attribute_triples(self, sup_triples):
        self.local_attribute_triples_list
        self.sup_attribute_triples_set = set(sup_triples)
        self.sup_attribute_triples_list = _stable_list(self.sup_attribute_triples_set)
        self.attribute_triples_set |= self.sup_attribute_triples_set
        self._forget(("attribute_triples_list",))
        self.attribute_triples_num = len(self.attribute_triples_set)

    # -- packed views for the device side (id-space KGs only) --
    @staticmethod
    def _pack(triples) -> np.ndarray:
        return np.asarray(triples, dtype=np.int32).reshape(-1, 3)

    @property
    def relation_triples_array(self) -> np.ndarray:
        """int32 [n,3] (h, r, t) of the local (pre-supervision) relation triples, list order."""
        return self._pack(self.local_relation_triples_list)

    @property
    def sup_relation_triples_array(self) -> np.ndarray:
        return self._pack(self.sup_relation_triples_list or [])

    @property
    def known_relation_triples_array(self) -> np.ndarray:
        """Everything the sampler must not emit as a negative: local + supervision triples."""
        return self._pack(_stable_list(self.local_relation_triples_set))


class KGs:
    """The aligned pair in id space plus link splits (code/base/kgs.py:5-76)."""

    def __init__(self, kg1: KG, kg2: KG, train_links, valid_links, test_links=None, mode="mapping", ordered=True):
        assign = generate_sharing_id if mode == "sharing" else generate_mapping_id
        pre = (lambda links: (links,)) if mode == "sharing" else (lambda links: ())
        ent_ids1, ent_ids2 = assign(*pre(train_links), kg1.relation_triples_set, kg1.entities_set,
                                    kg2.relation_triples_set, kg2.entities_set, ordered=ordered)
        rel_ids1, rel_ids2 = assign(*pre([]), kg1.relation_triples_set, kg1.relations_set,
                                    kg2.relation_triples_set, kg2.relations_set, ordered=ordered)
        attr_ids1, attr_ids2 = assign(*pre([]), kg1.attribute_triples_set, kg1.attributes_set,
                                      kg2.attribute_triples_set, kg2.attributes_set, ordered=ordered)
        self.uri_kg1, self.uri_kg2 = kg1, kg2
        id_kgs = []
        for kg, e, r, a in ((kg1, ent_ids1, rel_ids1, attr_ids1), (kg2, ent_ids2, rel_ids2, attr_ids2)):
            k = KG(uris_relation_triple_2ids(kg.relation_triples_set, e, r),
                   uris_attribute_triple_2ids(kg.attribute_triples_set, e, a))
            k.set_id_dict(e, r, a)
            id_kgs.append(k)
        self.kg1, self.kg2 = id_kgs

        self.uri_train_links, self.uri_valid_links = train_links, valid_links
        self.train_links = uris_pair_2ids(train_links, ent_ids1, ent_ids2)
        self.valid_links = uris_pair_2ids(valid_links, ent_ids1, ent_ids2)
        self.uri_test_links = test_links
        self.test_links = uris_pair_2ids(test_links, ent_ids1, ent_ids2) if test_links is not None else []
        for split in ("train", "valid", "test"):
            links = getattr(self, split + "_links")
            if len(set(links)) != len(links):
                raise ValueError(f"duplicate pairs in {split}_links")
            setattr(self, split + "_entities1", [a for a, _ in links])
            setattr(self, split + "_entities2", [b for _, b in links])

        if mode == "swapping":
            m12, m21 = dict(self.train_links), {b: a for a, b in self.train_links}
            self.kg1.add_sup_relation_triples(swap_relation_triples(self.kg1.local_relation_triples_set, m12))
            self.kg2.add_sup_relation_triples(swap_relation_triples(self.kg2.local_relation_triples_set, m21))
            self.kg1.add_sup_attribute_triples(swap_attribute_triples(self.kg1.local_attribute_triples_set, m12))
            self.kg2.add_sup_attribute_triples(swap_attribute_triples(self.kg2.local_attribute_triples_set, m21))

        self.useful_entities_list1 = self.train_entities1 + self.valid_entities1 + self.test_entities1
        self.useful_entities_list2 = self.train_entities2 + self.valid_entities2 + self.test_entities2
        self.entities_num = len(self.kg1.entities_set | self.kg2.entities_set)
        self.relations_num = len(self.kg1.relations_set | self.kg2.relations_set)
        self.attributes_num = len(self.kg1.attributes_set | self.kg2.attributes_set)


def read_kgs_from_folder(training_data_folder, division, mode, ordered):
    """The dataset layout of the reference's README (README.md:10-20; code/base/kgs.py:79-92)."""
    f = training_data_folder
    r1, _, _ = read_relation_triples(f + "rel_triples_1")
    r2, _, _ = read_relation_triples(f + "rel_triples_2")
    a1, _, _ = read_attribute_triples(f + "attr_triples_1")
    a2, _, _ = read_attribute_triples(f + "attr_triples_2")
    links = [read_links(f + division + name) for name in ("train_links", "valid_links", "test_links")]
    return _build(r1, r2, a1, a2, *links, mode=mode, ordered=ordered)


def read_kgs_from_files(kg1_relation_triples, kg2_relation_triples, kg1_attribute_triples, kg2_attribute_triples,
                        train_links, valid_links, test_links, mode):
    return _build(kg1_relation_triples, kg2_relation_triples, kg1_attribute_triples, kg2_attribute_triples, train_links,
                  valid_links, test_links, mode=mode, ordered=True)


def _build(r1, r2, a1, a2, train_links, valid_links, test_links, mode, ordered):
    uri1, uri2 = _UriKG(r1, a1), _UriKG(r2, a2)
    return KGs(uri1, uri2, train_links, valid_links, test_links=test_links, mode=mode, ordered=ordered)


class _UriKG(KG):
    """URI-space KG that keeps file order for its element sets so that unordered id assignment is deterministic.  One pass
    per triple list; none of the derived lists / dicts of `KG` is ever needed for it."""

    def set_relations(self, relation_triples):
        self._forget(self._REL_DERIVED)
        self.sup_relation_triples_set, self.sup_relation_triples_list = None, None
        self.relation_triples_set = self.local_relation_triples_set = (
            relation_triples if isinstance(relation_triples, _OrderedSet) else set(relation_triples))
        ents, rels = _OrderedSet(), _OrderedSet()
        for h, r, t in relation_triples:
            ents[h] = None
            ents[t] = None
            rels[r] = None
        self.entities_set, self.relations_set = ents, rels
        self.entities_num, self.relations_num = len(ents), len(rels)
        self.relation_triples_num = self.local_relation_triples_num = len(self.relation_triples_set)

    def set_attributes(self, attribute_triples):
        self._forget(self._ATTR_DERIVED)
        self.sup_attribute_triples_set, self.sup_attribute_triples_list = None, None
        self.attribute_triples_set = self.local_attribute_triples_set = (
            attribute_triples if isinstance(attribute_triples, _OrderedSet) else set(attribute_triples))
        attrs = _OrderedSet()
        for _, a, _ in attribute_triples:
            attrs[a] = None
        self.attributes_set = attrs
        self.attributes_num = len(attrs)
        self.attribute_triples_num = self.local_attribute_triples_num = len(self.attribute_triples_set)
