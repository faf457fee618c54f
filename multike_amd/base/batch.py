"""base/batch.py surface of the reference (code/base/batch.py) over Python lists, backed by the device sampler.

These functions keep the reference's names and argument orders so that code written against `bat.*` runs
unchanged; each call converts its list arguments to device tensors (cached per list object), runs
`mke_neg_sample` and converts back.  The training loops in `MultiKE_model.py` do NOT go through this list
interface (it would put Python back on the critical path) — they use `sampling.RelationBatcher` directly.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ..sampling import KGSide, KnownTripleSet, kg_batch_split, sample_negatives

_cache: dict = {}
_CACHE_MAX = 16
_call_counter = [0]


def _cached(objs, extra, build):
    """Device-side mirror of host containers, cached per container OBJECT: the entry holds the objects, so an id() cannot
    be recycled for another container while the entry lives; bounded, oldest entry out."""
    key = tuple(id(o) for o in objs) + tuple(extra)
    hit = _cache.get(key)
    if hit is None or any(a is not b for a, b in zip(hit[0], objs)):
        hit = (tuple(objs), build())
        _cache.pop(key, None)
        _cache[key] = hit
        while len(_cache) > _CACHE_MAX:
            _cache.pop(next(iter(_cache)))
    return hit[1]


def _known(triples_set, device):
    def build():
        arr = np.array(list(triples_set), dtype=np.int32).reshape(-1, 3)
        t = torch.as_tensor(arr, device=device)
        return KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())
    return _cached((triples_set,), ("known", len(triples_set)), build)


def _side(entities_list, triples_set, neighbor, device):
    def build():
        side = KGSide(entities_list, _known(triples_set, device) if triples_set is not None else None, device=device)
        if neighbor:
            n_total = max(max(entities_list), max(neighbor)) + 1
            k = min(len(v) for v in neighbor.values())
            table = np.zeros((n_total, k), dtype=np.int32)
            valid = np.zeros(n_total, dtype=np.uint8)
            for e, lst in neighbor.items():
                table[e] = np.asarray(lst[:k], dtype=np.int32)
                valid[e] = 1
            side.set_neighbours(torch.as_tensor(table, device=device), torch.as_tensor(valid, device=device))
        return side
    return _cached((entities_list, triples_set, neighbor), ("side", len(entities_list)), build)


def generate_pos_triples(triples, batch_size, step, is_fixed_size=False):
    """code/base/batch.py:45-54."""
    lo = step * batch_size
    chunk = triples[lo:min(lo + batch_size, len(triples))]
    if is_fixed_size and len(chunk) < batch_size:
        chunk = chunk + triples[:batch_size - len(chunk)]
    return chunk


def generate_neg_triples_fast(pos_batch, all_triples_set, entities_list, neg_triples_num, neighbor=None, max_try=10,
                              seed=None, device="cuda"):
    """code/base/batch.py:86-116 on the device: same distribution, Philox stream (`seed` defaults to a per-call
    counter, as the reference draws from an unseeded global RNG)."""
    if len(pos_batch) == 0 or neg_triples_num == 0:
        return []
    side = _side(entities_list, all_triples_set, neighbor, device)
    arr = torch.as_tensor(np.asarray(pos_batch, dtype=np.int32).reshape(-1, 3), device=device)
    pos = (arr[:, 0].contiguous(), arr[:, 1].contiguous(), arr[:, 2].contiguous())
    if seed is None:
        _call_counter[0] += 1
        seed = (0x5EED, _call_counter[0])
    nh, nr, nt = sample_negatives(pos, side, neg_triples_num, seed=seed, max_try=max_try)
    out = torch.stack([nh, nr, nt], 1).cpu().numpy()
    return [tuple(int(v) for v in row) for row in out]


def generate_relation_triple_batch(triple_list1, triple_list2, triple_set1, triple_set2, entity_list1, entity_list2,
                                   batch_size, step, neighbor1, neighbor2, neg_triples_num):
    """code/base/batch.py:33-42."""
    b1, b2 = kg_batch_split(len(triple_list1), len(triple_list2), batch_size)
    pos1 = generate_pos_triples(triple_list1, b1, step)
    pos2 = generate_pos_triples(triple_list2, b2, step)
    neg1 = generate_neg_triples_fast(pos1, triple_set1, entity_list1, neg_triples_num, neighbor=neighbor1)
    neg2 = generate_neg_triples_fast(pos2, triple_set2, entity_list2, neg_triples_num, neighbor=neighbor2)
    return pos1 + pos2, neg1 + neg2


def generate_relation_triple_batch_queue(triple_list1, triple_list2, triple_set1, triple_set2, entity_list1, entity_list2,
                                         batch_size, steps, out_queue, neighbor1, neighbor2, neg_triples_num):
    """code/base/batch.py:22-30 (without the producer process's exit(0): there is no forked producer here)."""
    for step in steps:
        out_queue.put(generate_relation_triple_batch(triple_list1, triple_list2, triple_set1, triple_set2, entity_list1,
                                                     entity_list2, batch_size, step, neighbor1, neighbor2,
                                                     neg_triples_num))


def neighbour_table(entity_embeds, entity_list, neighbors_num, n_ent_total, device="cuda", block_rows=4096,
                    rows_per_launch=None, part=None):
    """Truncated-sampling k-NN refresh on the device (code/base/batch.py:119-150): inner product of the (already
    row-normalised) relation-view rows of one KG's useful entities, top `neighbors_num` per row INCLUDING the entity
    itself, unordered.  Returns (cand_table [n_ent_total, k] int32, cand_valid [n_ent_total] uint8) for
    `KGSide.set_neighbours`.

    When k is a small share of a long row the n x n similarity matrix is never built: a per-row threshold a bit below
    the k-th largest value is estimated from a fixed column sample (`mke_sim_sample` + `mke_topk_rows`),
    `mke_sim_select` computes the similarities tile by tile on the matrix cores and keeps only the ~1.4 k columns above
    the threshold, `mke_topk_rows` takes the exact top k of that short list.  The few rows whose estimate came out too
    tight (fewer than k hits) or too loose (a segment overflowed) are redone at full width (`mke_sim_sample` + `mke_topk_long`,
    which is also the path of short rows).  The result is the exact top-k set.

    part = (r, G): compute only the r-th of G contiguous slices of the rows (in the function's own working order) — the
    multi-GPU refresh: every rank holds all rows, ranks the slice it is given against all columns and the slices are
    all-gathered.  Returns (table, valid, ids_of_the_slice) with only those rows filled."""
    from .. import _lib
    e = (entity_embeds.to(device).float() if isinstance(entity_embeds, torch.Tensor)
         else torch.as_tensor(np.asarray(entity_embeds), dtype=torch.float32, device=device))
    ids = torch.as_tensor(np.asarray(entity_list), dtype=torch.int64, device=device)
    n, d = e.shape
    k = int(neighbors_num)
    table = torch.zeros(n_ent_total, k, dtype=torch.int32, device=device)
    valid = torch.zeros(n_ent_total, dtype=torch.uint8, device=device)
    n_samp, cap = 4096, _pow2_at_least(int(1.4 * k) + 64)
    short = n >= 8 * n_samp and cap * 4 <= n and cap <= 4096
    if d > _lib.SIM_SELECT_KPADS[-1]:     # wider than the widest table the package supports (MKE_MAX_STRIDE): no second backend
        raise _lib.MultiKEHipError(f"neighbour_table: rows of {d} floats exceed the widest k_sim_select instantiation "
                                   f"({_lib.SIM_SELECT_KPADS[-1]} = MKE_MAX_STRIDE)")
    kpad = min(x for x in _lib.SIM_SELECT_KPADS if x >= d)

    def full_width(rows):
        """top k of whole similarity rows (short KGs; rows whose threshold estimate was off).  The similarities come from the
        same sweep as the main pass (`mke_sim_sample` against ALL columns: identical fma chains) — a library GEMM here was
        the refresh's only library call and cost its 170 ms initialisation at the first refresh of a run."""
        cols = ep if ep is not None else _padded(e)
        src = cols[rows].contiguous()
        sim = _lib.sim_sample(src, kpad, 0, int(src.shape[0]), cols)
        return _lib.topk_long(sim, k).long()          # exact top k of whole rows: radix select in the package's own kernel

    def _padded(x):
        out = torch.zeros(x.shape[0], kpad, dtype=torch.float32, device=device)
        out[:, :d] = x
        return out
    ep = None

    p_lo, p_hi = (0, n) if part is None else (n * part[0] // part[1], n * (part[0] + 1) // part[1])
    if not short:
        ep = _padded(e)
        for lo in range(p_lo, p_hi, block_rows):
            hi = min(p_hi, lo + block_rows)
            table[ids[lo:hi]] = ids[full_width(slice(lo, hi))].to(torch.int32)
        valid[ids[p_lo:p_hi]] = 1
        return (table, valid) if part is None else (table, valid, ids[p_lo:p_hi])

    # The working order and the column sample are keyed permutations computed on the device with INTEGER arithmetic only
    # (argsort of a splitmix64-style mix of i + seed — `>>` on int64 is an arithmetic shift here, so it is not the textbook
    # function, only a fixed one): a function of (n, seed) alone, identical on every rank, device architecture and torch
    # build.  Round 3 drew them from a device torch.Generator — fast (a CPU permutation of 100K ids is 9 ms the GPU waits for),
    # but in part mode the slices of the multi-GPU refresh are DEFINED in this order, and a generator stream that differed
    # between ranks would silently leave candidate rows unfilled.
    def keyed_perm(seed):
        x = torch.arange(n, dtype=torch.int64, device=device) + seed
        x = (x ^ (x >> 30)) * -4658895280553007687        # 0xBF58476D1CE4E5B9 as int64 (wrap-around multiply)
        x = (x ^ (x >> 27)) * -7723592293110705685        # 0x94D049BB133111EB
        x = x ^ (x >> 31)
        return torch.argsort(x, stable=True)
    # Work in a fixed random order of the entities: a row's hits are then spread evenly over the column segments whatever
    # the order of the ids (in id order similar entities sit together — URIs of one namespace, one generator block — and
    # most rows overflowed one of their segments on the DBP-WD-like folder: 54-75 % of the rows went to the full-width path).
    perm = keyed_perm(0x3C6EF372)
    e, ids = e[perm], ids[perm]
    ep = _padded(e)
    ids32 = ids.to(torch.int32)
    samp = keyed_perm(0x1B873593)[:n_samp]
    es = ep[samp].contiguous()
    m = min(n_samp, int(math.ceil(1.4 * k * n_samp / n)) + 8)
    chunk = (1 << 29) // cap - 128                        # rows per launch: 2^29 candidate slots (8 bytes each) at most
    if rows_per_launch:
        chunk = min(chunk, int(rows_per_launch))
    for lo in range(p_lo, p_hi, chunk):
        hi = min(p_hi, lo + chunk)
        # enough (row block, column segment) work items to fill the chip several times over, segments of >= 4096 columns
        n_seg = 1
        while n_seg < 8 and ((hi - lo + 127) // 128) * n_seg < 6144 and n // (2 * n_seg) >= 4096:
            n_seg *= 2
        tau = torch.empty(hi - lo, dtype=torch.float32, device=device)
        for a in range(lo, hi, 16384):                    # sample similarities, [16384, 4096] at a time
            b = min(hi, a + 16384)
            _, kth, _ = _lib.topk_rows(_lib.sim_sample(ep, kpad, a, b, es), m, want_idx=False, want_kth=True)
            tau[a - lo:b - lo] = kth
        cand, cnt = _lib.sim_select(ep, kpad, lo, hi, tau, n_seg, cap // n_seg)
        out, status = _lib.topk_candidates(cand, cnt, k, id_map=ids32)
        table[ids[lo:hi]] = out
        bad = torch.nonzero(status).reshape(-1)
        if bad.numel():
            rows = bad + lo
            for a in range(0, rows.numel(), block_rows):
                r = rows[a:a + block_rows]
                table[ids[r]] = ids[full_width(r)].to(torch.int32)
    valid[ids[p_lo:p_hi]] = 1
    return (table, valid) if part is None else (table, valid, ids[p_lo:p_hi])


def _pow2_at_least(x: int) -> int:
    p = 1
    while p < x:
        p *= 2
    return p


def generate_neighbours(entity_embeds, entity_list, neighbors_num, threads_num):
    """code/base/batch.py:119-140 — dict {entity: [k neighbours]} (the reference's return type)."""
    n_total = int(max(entity_list)) + 1
    table, _ = neighbour_table(entity_embeds, entity_list, neighbors_num, n_total)
    rows = table[torch.as_tensor(np.asarray(entity_list), dtype=torch.int64, device=table.device)].cpu().numpy()
    return {int(e): rows[i].tolist() for i, e in enumerate(entity_list)}
