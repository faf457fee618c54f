"""Host-side helpers with the reference's names and semantics (code/utils.py) that the hot path's callers use.
Only what the path needs: args loading, step->producer division, output folder naming, embedding savers."""
from __future__ import annotations

import json
import os
import time

import numpy as np


class ARGs:
    """Attribute bag over the JSON dict (code/utils.py:19-22)."""

    def __init__(self, dic):
        for k, v in dic.items():
            setattr(self, k, v)


def load_args(file_path):
    """code/utils.py:10-16 — same key names as code/args.json."""
    with open(file_path, "r") as f:
        args_dict = json.load(f)
    print("load arguments:", args_dict)
    return ARGs(args_dict)


def task_divide(idx, n):
    """code/utils.py:35-49: n-1 chunks of total//n and a last chunk with the remainder; degenerate cases
    return a single task."""
    total = len(idx)
    if n <= 0 or total == 0 or n > total:
        return [idx]
    if n == total:
        return [[i] for i in idx]
    size = total // n
    tasks = [idx[k * size:(k + 1) * size] for k in range(n - 1)]
    tasks.append(idx[(n - 1) * size:])
    return tasks


def merge_dic(dic1, dic2):
    return {**dic1, **dic2}


def generate_out_folder(out_folder, training_data_path, div_path, method_name):
    """code/utils.py:52-57."""
    path = training_data_path.strip("/").split("/")[-1]
    folder = out_folder + method_name + "/" + path + "/" + div_path + str(time.strftime("%Y%m%d%H%M%S")) + "/"
    print("results output folder:", folder)
    return folder


def dict2file(file, dic):
    if dic is None:
        return
    with open(file, "w", encoding="utf8") as f:
        for i, j in dic.items():
            f.write(str(i) + "\t" + str(j) + "\n")
    print(file, "saved.")


def save_embeddings(folder, kgs, ent_embeds, nv_ent_embeds, rv_ent_embeds, av_ent_embeds, rel_embeds, attr_embeds):
    """Same file names / formats as code/utils.py:70-91 so downstream tooling keeps working."""
    os.makedirs(folder, exist_ok=True)
    if ent_embeds is not None:
        np.save(folder + "ent_embeds.npy", ent_embeds)
        np.save(folder + "nv_ent_embeds.npy", nv_ent_embeds)
        np.save(folder + "rv_ent_embeds.npy", rv_ent_embeds)
        np.save(folder + "av_ent_embeds.npy", av_ent_embeds)
    if rel_embeds is not None:
        np.save(folder + "rel_embeds.npy", rel_embeds)
    if attr_embeds is not None:
        np.save(folder + "attr_embeds.npy", attr_embeds)
    for name, kg, attr in (("kg1_ent_ids", kgs.kg1, "entities_id_dict"), ("kg2_ent_ids", kgs.kg2, "entities_id_dict"),
                           ("kg1_rel_ids", kgs.kg1, "relations_id_dict"), ("kg2_rel_ids", kgs.kg2, "relations_id_dict"),
                           ("kg1_attr_ids", kgs.kg1, "attributes_id_dict"), ("kg2_attr_ids", kgs.kg2, "attributes_id_dict")):
        dict2file(folder + name, getattr(kg, attr, None))
    print("Embeddings saved!")
