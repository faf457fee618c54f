"""Import-path compatibility with the reference (code/base/read.py): readers, id assignment and supervision
triples live in base/kgs.py; `save_embeddings` in utils.py."""
from .kgs import (dict2file, generate_mapping_id, generate_sharing_id, generate_sup_attribute_triples,  # noqa: F401
                  generate_sup_relation_triples, line2file, pair2file, read_attribute_triples, read_dict, read_links,
                  read_pair_ids, read_relation_triples, sort_elements, uris_attribute_triple_2ids, uris_list_2ids,
                  uris_pair_2ids, uris_relation_triple_2ids)
