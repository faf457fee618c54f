"""mke_oc_em_plan through the C-ABI, by hand: the entity-major reference lists of an epoch against a direct NumPy enumeration of the
header's definition (include/multike_hip.h: owned elements sorted by (step, local row), a row's references in element order —
negatives in code order, then own term / head vector / tail vector / relation head / relation tail per position), and the degenerate
shapes the trainer never produces: an epoch without positives, steps without positives, ranks that own nothing."""
import ctypes as C

import numpy as np
import pytest
import torch

from multike_amd import _lib

pytestmark = pytest.mark.gpu

GV, PLUS = 0x80000000, 1 << 24


def _plan(ph, pr, pt, codes, N, sh, st, step_lo, G, rank, n_local, n_rel, capacity, chunks=1):
    dev = "cuda"
    i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=dev) if len(a) else torch.zeros(1, dtype=torch.int32, device=dev)
    n_steps = len(step_lo) - 1
    n_all = len(ph)
    t = dict(ph=i32(ph), pr=i32(pr), pt=i32(pt), codes=i32(codes), sh=i32(sh), st=i32(st),
             step_lo=torch.as_tensor(np.asarray(step_lo, dtype=np.int64), device=dev))
    z32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
    z64 = lambda n: torch.zeros(n, dtype=torch.int64, device=dev)
    b = dict(keys=z64(capacity + 1), keys_alt=z64(capacity + 1), flags=z32(capacity + 1), scan=z32(capacity + 1), vals_alt=z32(capacity + 1),
             scratch8=z64(capacity + 1), waves=z32(2 * (_lib.OC_EM_WAVES + 1)), refs=z32(2 * capacity), rows=z32(capacity), off=z32(capacity + 1),
             row0=z64(n_steps + 1), item_row=z32(capacity + 1), item_off=z32(capacity + 1), item_part=z32(capacity + 1),
             long_row=z32(capacity // 32 + 2), long_part0=z32(capacity // 32 + 2), steps3=z64(3 * (n_steps + 1)), n_refs=z64(1),
             temp=torch.zeros(_lib.oc_em_plan_temp_bytes(capacity), dtype=torch.uint8, device=dev))
    a = _lib.OcEmPlanArgs()
    p = lambda x: x.data_ptr()
    a.pos_h, a.pos_r, a.pos_t, a.codes, a.neg_per_pos = p(t["ph"]), p(t["pr"]), p(t["pt"]), p(t["codes"]), N
    a.slot_h, a.slot_t, a.step_lo, a.n_steps, a.chunks = p(t["sh"]), p(t["st"]), p(t["step_lo"]), n_steps, chunks
    a.n_all = n_all
    a.max_step = int(max([step_lo[k + 1] - step_lo[k] for k in range(n_steps)], default=0))
    a.n_ranks, a.rank, a.n_local, a.n_rel = G, rank, n_local, n_rel
    a.keys, a.keys_alt, a.capacity = p(b["keys"]), p(b["keys_alt"]), capacity
    a.vals_alt, a.scratch8, a.wave_scratch = p(b["vals_alt"]), p(b["scratch8"]), p(b["waves"])
    a.refs, a.rows, a.off, a.flags, a.scan = p(b["refs"]), p(b["rows"]), p(b["off"]), p(b["flags"]), p(b["scan"])
    a.step_row0, a.n_refs = p(b["row0"]), p(b["n_refs"])
    a.item_row, a.item_off, a.item_part = p(b["item_row"]), p(b["item_off"]), p(b["item_part"])
    a.long_row, a.long_part0 = p(b["long_row"]), p(b["long_part0"])
    s3 = b["steps3"]
    a.step_item0, a.step_long0, a.step_part0 = p(s3), p(s3) + 8 * (n_steps + 1), p(s3) + 16 * (n_steps + 1)
    a.temp, a.temp_bytes = p(b["temp"]), b["temp"].numel()
    _lib.oc_em_plan(a)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in b.items() if k not in ("temp", "waves", "keys", "keys_alt", "flags", "scan", "vals_alt", "scratch8")}
    out["steps3"] = out["steps3"].reshape(3, n_steps + 1)
    return out


def _expected(ph, pr, pt, codes, N, sh, st, step_lo, G, rank, n_local):
    """{(step, row): [(locator, coefficient index), ...]} in element order, by the header's definition."""
    n_all = len(ph)
    step_of = np.searchsorted(np.asarray(step_lo), np.arange(n_all), side="right") - 1
    lists = {}

    def add(p, ent, row, loc, cidx):
        if ent < 0 or ent % G != rank:
            return
        lists.setdefault((int(step_of[p]), int(row)), []).append((loc & 0xFFFFFFFF, cidx))

    for p in range(n_all):                       # the negatives, in code order
        i = p - step_lo[step_of[p]]
        for n in range(N):
            c = int(codes[p * N + n])
            ent, rt = (c & 0x3FFFFFFF) >> 1, c & 1
            src = pt[p] if rt else ph[p]         # the vector that travels: RT from the tail's owner, HR from the head's
            slot = st[p] if rt else sh[p]
            add(p, ent, ent // G, ((src % G) << 24) | (rt << 23) | slot, i * (N + 1) + n)
    for p in range(n_all):                       # own term, head / tail gradient vectors, the relation row's two
        i = p - step_lo[step_of[p]]
        hr = sh[p] >= 0
        own = pt[p] if hr else ph[p]
        rt = 0 if hr else 1
        src = pt[p] if rt else ph[p]
        add(p, own, own // G, ((src % G) << 24) | (rt << 23) | (st[p] if rt else sh[p]), i * (N + 1) + N)
        if sh[p] >= 0:
            add(p, ph[p], ph[p] // G, GV | sh[p], 0)
        if st[p] >= 0:
            add(p, pt[p], pt[p] // G, GV | (1 << 23) | st[p], 0)
        if sh[p] >= 0:
            add(p, ph[p], n_local + pr[p], GV | PLUS | sh[p], 0)
        if st[p] >= 0:
            add(p, pt[p], n_local + pr[p], GV | PLUS | (1 << 23) | st[p], 0)
    return lists


def _case(seed, G, rank, n_ent, n_rel, sizes, N, hub=False):
    rng = np.random.default_rng(seed)
    n_all = int(sum(sizes))
    step_lo = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    draw = (lambda n: np.minimum(rng.zipf(1.3, n) - 1, n_ent - 1)) if hub else (lambda n: rng.integers(0, n_ent, n))
    ph, pt, pr = draw(n_all), draw(n_all), rng.integers(0, n_rel, n_all)
    side = rng.integers(0, 2, n_all)                                    # one coin per positive ...
    both = rng.random(n_all) < 0.1                                      # ... and a few that need both vectors
    codes = np.zeros(n_all * N, dtype=np.int64)
    for p in range(n_all):
        for n in range(N):
            s = side[p] if not both[p] else rng.integers(0, 2)
            codes[p * N + n] = (int(draw(1)[0]) << 1) | int(s)
    need_rt = np.array([any(codes[p * N:(p + 1) * N] & 1) for p in range(n_all)]) if N else np.zeros(n_all, bool)
    need_hr = ~need_rt | np.array([any((codes[p * N:(p + 1) * N] & 1) == 0) for p in range(n_all)]) if N else np.ones(n_all, bool)
    sh, st = np.full(n_all, -1), np.full(n_all, -1)
    for s in range(len(sizes)):                                         # slots: per step and owner, in position order
        cnt_h, cnt_t = np.zeros(G, int), np.zeros(G, int)
        for p in range(step_lo[s], step_lo[s + 1]):
            if need_hr[p]:
                sh[p] = cnt_h[ph[p] % G]; cnt_h[ph[p] % G] += 1
            if need_rt[p]:
                st[p] = cnt_t[pt[p] % G]; cnt_t[pt[p] % G] += 1
    return ph, pr, pt, codes, sh, st, step_lo


@pytest.mark.parametrize("G,rank,sizes,N,hub", [(1, 0, [7, 5, 9], 3, False), (4, 2, [40, 0, 33, 1], 5, False), (3, 1, [64, 64], 25, False),
                                                  (8, 7, [30] * 5, 8, True), (2, 0, [50, 50, 50], 0, False), (5, 4, [1, 1, 1], 1, False)])
def test_lists_equal_the_direct_enumeration(G, rank, sizes, N, hub):
    n_ent, n_rel = 97, 6
    n_local = (n_ent + G - 1) // G
    ph, pr, pt, codes, sh, st, step_lo = _case(G * 100 + rank, G, rank, n_ent, n_rel, sizes, N, hub)
    want = _expected(ph, pr, pt, codes, N, sh, st, step_lo, G, rank, n_local)
    n_refs = sum(len(v) for v in want.values())
    out = _plan(ph, pr, pt, codes, N, sh, st, step_lo, G, rank, n_local, n_rel, capacity=n_refs + 17)
    assert int(out["n_refs"][0]) == n_refs
    keys = sorted(want)
    row0 = out["row0"]
    assert int(row0[-1]) == len(keys)
    refs = out["refs"].view(np.uint32).reshape(-1, 2)
    for u, (s, row) in enumerate(keys):
        assert int(out["rows"][u]) == row and int(row0[s]) <= u < int(row0[s + 1])
        lo, hi = int(out["off"][u]), int(out["off"][u + 1])
        assert [(int(a), int(b)) for a, b in refs[lo:hi]] == want[(s, row)], (s, row)
    # the work items: every row's list in segments of at most 32 references, in order; long rows' partial slots consecutive
    item0 = out["steps3"][0]
    n_items = int(item0[-1])
    w = 0
    for u, key in enumerate(keys):
        lo, hi = int(out["off"][u]), int(out["off"][u + 1])
        nseg = -(-(hi - lo) // 32)
        for k in range(nseg):
            r = int(out["item_row"][w]) & 0xFFFFFFFF
            assert r & 0x3FFFFFFF == key[1] and bool(r & 0x80000000) == (nseg > 1)
            assert int(out["item_off"][w]) == lo + 32 * k and (int(out["item_part"][w]) >= 0) == (nseg > 1)
            seg = refs[lo + 32 * k:min(hi, lo + 32 * k + 32), 0]
            assert bool(r & 0x40000000) == (nseg > 1 or bool((seg & GV).any()))
            w += 1
    assert w == n_items and int(out["item_off"][n_items]) == n_refs


@pytest.mark.parametrize("sizes,N", [([], 3), ([0], 3), ([0, 0, 0], 0)])
def test_an_epoch_without_positives_gives_empty_lists(sizes, N):
    z = np.zeros(0, dtype=np.int64)
    step_lo = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    out = _plan(z, z, z, z, N, z, z, step_lo, 4, 1, 25, 6, capacity=64)
    assert int(out["n_refs"][0]) == 0 and not out["row0"].any() and not out["steps3"].any()


def test_overflow_reports_the_count_and_writes_inside_the_capacity():
    G, rank, N = 2, 1, 9
    n_local = 49
    ph, pr, pt, codes, sh, st, step_lo = _case(5, G, rank, 97, 6, [60, 60], N)
    want = _expected(ph, pr, pt, codes, N, sh, st, step_lo, G, rank, n_local)
    n_refs = sum(len(v) for v in want.values())
    out = _plan(ph, pr, pt, codes, N, sh, st, step_lo, G, rank, n_local, 6, capacity=n_refs // 3)
    assert int(out["n_refs"][0]) == n_refs            # the caller re-plans at this size (distributed_oc._finish_plan)
