"""Shared helpers of the `-m gpu` parity tests (all call the product through the C-ABI binding)."""
import numpy as np
import torch

from multike_amd.tables import EmbeddingTable


def dev_i32(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device="cuda")


def dev_f32(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def make_tables(ent, rel, ent_norm=True, rel_norm=True, rel_grad_copies=1):
    E = EmbeddingTable(ent.shape[0], ent.shape[1], "ent", normalize=ent_norm, values=ent)
    R = EmbeddingTable(rel.shape[0], rel.shape[1], "rel", normalize=rel_norm, values=rel, grad_copies=rel_grad_copies)
    return E, R


def grouped_batch(rng, n_ent, n_rel, P, N, irregular=True):
    ph, pr, pt = rng.integers(0, n_ent, P), rng.integers(0, n_rel, P), rng.integers(0, n_ent, P)
    nh, nr, nt = np.repeat(ph, N), np.repeat(pr, N), np.repeat(pt, N)
    side = rng.integers(0, 2, P * N).astype(bool)
    c = rng.integers(0, n_ent, P * N)
    nh = np.where(side, c, nh)
    nt = np.where(side, nt, c)
    if irregular and P * N > 8:
        k = rng.integers(0, P * N, max(1, P * N // 50))
        nh[k] = rng.integers(0, n_ent, len(k))
        nt[k] = rng.integers(0, n_ent, len(k))
        k2 = rng.integers(0, P * N, max(1, P * N // 70))
        nr[k2] = rng.integers(0, n_rel, len(k2))
        k3 = rng.integers(0, P * N, 2)  # negatives equal to their positive
        nh[k3], nt[k3] = np.repeat(ph, N)[k3], np.repeat(pt, N)[k3]
    return tuple(a.astype(np.int32) for a in (ph, pr, pt)), tuple(a.astype(np.int32) for a in (nh, nr, nt))
