"""`DataModel`: a dataset folder -> everything `MultiKE_model.MultiKE` reads (reference
code/data_model.py:66-160).  Host-side preparation; the only device work is the literal auto-encoder
(`literal_encoder.LiteralEncoder`, HIP GEMM path) when `literal_vectors.npy` is not cached in the folder.

Produces, with the reference's attribute names:
  kgs                  id-space KG pair, 'swapping' supervision included (base/kgs.py)
  literal_list / literal_vectors_mat / literal_id_dic      cleaned attribute values + entity local names, encoded
  local_name_vectors   [entities_num, dim] name-view table, row i = literal vector of entity i's local name
  value_vectors        [n_values, dim] literal table of the attribute view; attribute triples are re-written to
                       (entity id, attribute id, value id) and the supervision triples regenerated from them

Set -> list conversions are sorted (the reference's are hash-ordered, i.e. differ from run to run), so literal ids
and value ids are reproducible.  With `literal_normalize` rows are l2-normalised like sklearn's
`preprocessing.normalize` (zero rows stay zero).
"""
from __future__ import annotations

import os

import numpy as np

from .base.kgs import read_kgs_from_folder, swap_attribute_triples
from .utils import CharHashEmbedder, clear_attribute_triples, read_local_name, read_word2vec

LITERAL_EMBEDDINGS_FILE = "literal_vectors.npy"
LITERAL_FILE = "literals.txt"


def save_literal_vectors(folder, literal_list, literal_vectors):
    """code/data_model.py:26-33: the cache pair; one literal per line, row-aligned with the .npy."""
    if len(literal_list) != len(literal_vectors):
        raise ValueError("literal list and vectors differ in length")
    np.save(folder + LITERAL_EMBEDDINGS_FILE, np.asarray(literal_vectors))
    with open(folder + LITERAL_FILE, "w", encoding="utf-8") as f:
        f.writelines(l + "\n" for l in literal_list)


def load_literal_vectors(folder):
    """code/data_model.py:36-45."""
    mat = np.load(folder + LITERAL_EMBEDDINGS_FILE)
    with open(folder + LITERAL_FILE, "r", encoding="utf-8") as f:
        literal_list = [line.rstrip("\n") for line in f]
    return literal_list, np.asarray(mat)


def generate_literal_id_dic(literal_list):
    dic = {l: i for i, l in enumerate(literal_list)}
    if len(dic) != len(literal_list):
        raise ValueError("duplicate literals in the literal list")
    return dic


def l2_normalize_rows(mat):
    mat = np.asarray(mat, dtype=np.float64)
    n = np.sqrt((mat * mat).sum(axis=1, keepdims=True))
    n[n == 0.0] = 1.0
    return mat / n


class DataModel:
    def __init__(self, args, device="cuda", char_embedder="hash"):
        self.args = args
        self.device = device
        self.kgs = read_kgs_from_folder(args.training_data, args.dataset_division, args.alignment_module, False)
        self.entities = self.kgs.kg1.entities_set | self.kgs.kg2.entities_set
        self.word2vec_path = getattr(args, "word2vec_path", None)
        self._char_embedder = CharHashEmbedder() if char_embedder == "hash" else char_embedder
        self.entity_local_name_dict = read_local_name(args.training_data, set(self.kgs.kg1.entities_id_dict),
                                                      set(self.kgs.kg2.entities_id_dict))
        cleaned1, _, _ = clear_attribute_triples(self.kgs.kg1.local_attribute_triples_list)
        cleaned2, _, _ = clear_attribute_triples(self.kgs.kg2.local_attribute_triples_list)
        self._cleaned = (cleaned1, cleaned2)
        self._generate_literal_vectors()
        self._generate_name_vectors_mat()
        self._generate_attribute_value_vectors()

    # code/data_model.py:78-95
    def _generate_literal_vectors(self):
        folder = self.args.training_data
        if not getattr(self.args, "retrain_literal_embeds", False) and os.path.exists(folder + LITERAL_EMBEDDINGS_FILE):
            self.literal_list, self.literal_vectors_mat = load_literal_vectors(folder)
        else:
            from .literal_encoder import LiteralEncoder
            values = [v for (_, _, v) in self._cleaned[0] + self._cleaned[1]]
            names = list(self.entity_local_name_dict.values())
            self.literal_list = sorted(set(values + names))
            word2vec = read_word2vec(self.word2vec_path)
            enc = LiteralEncoder(self.literal_list, word2vec, self.args, char_embedder=self._char_embedder,
                                 device=self.device)
            self.literal_vectors_mat = np.asarray(enc.encoded_literal_vector)
            save_literal_vectors(folder, self.literal_list, self.literal_vectors_mat)
        if self.literal_vectors_mat.shape[0] != len(self.literal_list):
            raise ValueError("literal_vectors.npy and literals.txt are out of step")
        self.literal_id_dic = generate_literal_id_dic(self.literal_list)

    # code/data_model.py:97-118
    def _generate_name_vectors_mat(self):
        n = len(self.entities)
        uri_of = {}
        for kg in (self.kgs.kg1, self.kgs.kg2):
            uri_of.update((i, u) for u, i in kg.entities_id_dict.items())
        if sorted(uri_of) != list(range(n)):
            raise ValueError("entity ids are not a permutation of range(entities_num)")
        rows = []
        for i in range(n):
            name = self.entity_local_name_dict[uri_of[i]]
            if name not in self.literal_id_dic:
                raise ValueError(f"local name {name!r} of {uri_of[i]} is not in the literal cache; "
                                 f"set retrain_literal_embeds or delete {LITERAL_EMBEDDINGS_FILE}")
            rows.append(self.literal_id_dic[name])
        mat = np.asarray(self.literal_vectors_mat)[rows]
        self.local_name_vectors = l2_normalize_rows(mat) if self.args.literal_normalize else mat

    # code/data_model.py:120-160
    def _generate_attribute_value_vectors(self):
        self.literal_set = set(self.literal_list)
        kept = [{(h, a, v) for (h, a, v) in cleaned if v in self.literal_set} for cleaned in self._cleaned]
        values_list = sorted({v for side in kept for (_, _, v) in side})
        value_id = {v: i for i, v in enumerate(values_list)}
        for kg, side in zip((self.kgs.kg1, self.kgs.kg2), kept):
            kg.set_attributes({(h, a, value_id[v]) for (h, a, v) in side})
        m12, m21 = dict(self.kgs.train_links), {b: a for a, b in self.kgs.train_links}
        self.kgs.kg1.add_sup_attribute_triples(swap_attribute_triples(self.kgs.kg1.local_attribute_triples_set, m12))
        self.kgs.kg2.add_sup_attribute_triples(swap_attribute_triples(self.kgs.kg2.local_attribute_triples_set, m21))
        self.values_list = values_list
        vecs = np.asarray(self.literal_vectors_mat)[[self.literal_id_dic[v] for v in values_list]]
        self.value_vectors = l2_normalize_rows(vecs) if self.args.literal_normalize else vecs
