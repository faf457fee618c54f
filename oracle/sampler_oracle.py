"""CPU ORACLE (test infrastructure, not product code) — the relation-view batch generator and negative sampler.

Two restatements live here:

1. `mt_*`: the reference's sampler as it runs on CPython — Mersenne-Twister `random` +
   `numpy.random.binomial` — restated from code/base/batch.py:22-54,86-116.  `tests/golden/make_golden.py`
   runs the reference's own `base/batch.py` under fixed seeds and `tests/test_oracle_golden.py` demands
   list-for-list equality with this restatement: that PINS it.

2. `philox_*`: the counter-based specification the HIP sampler implements (`include/multike_hip.h`
   mke_neg_sample; multike_amd/csrc/mke_sampler.hip).  Same distribution as (1): per
   positive, up to `max_try` rounds; one fair coin per round picks the corrupted side; `need` distinct
   candidates are drawn without replacement from that side's candidate list; known triples are dropped
   except in the last round.  The random stream is Philox4x32-10 indexed by (positive, round, slot,
   attempt) so it is order-independent; the device output must equal this restatement BIT FOR BIT
   (tests/test_sampler_gpu.py) and (1) and (2) are compared statistically (tests/test_sampler_oracle.py).
"""
from __future__ import annotations

import random

import numpy as np

from .multike_oracle import kg_batch_split

# ----------------------------------------------------------------------------------------------
# (1) Mersenne-Twister restatement (what the reference does)
# ----------------------------------------------------------------------------------------------


def mt_epoch_slice(triples, batch_size, step):
    """code/base/batch.py:45-54 with is_fixed_size=False: contiguous slice, short/empty at the end."""
    lo = step * batch_size
    return triples[lo:min(lo + batch_size, len(triples))]


def _mt_one_positive(h, r, t, known, everyone, want, near, max_try, stats=None):
    """Rounds for one positive — code/base/batch.py:91-112.  RNG call order is part of the contract:
    one np.random.binomial(1, 0.5), then one random.sample(candidates, still_needed), per round.
    `stats` (tests/test_sampler_oracle.py): a dict whose "rounds" list receives the rounds this positive used."""
    got = []
    pool_h = near.get(h, everyone)
    pool_t = near.get(t, everyone)
    missing = want
    for attempt in range(max_try):
        if np.random.binomial(1, 0.5):
            fresh = {(x, r, t) for x in random.sample(pool_h, missing)}
        else:
            fresh = {(h, r, x) for x in random.sample(pool_t, missing)}
        if attempt == max_try - 1:  # last round: unfiltered
            got.extend(fresh)
            break
        got.extend(fresh - known)
        if len(got) == want:
            break
        missing = want - len(got)
    assert len(got) == want
    if stats is not None:
        stats.setdefault("rounds", []).append(attempt + 1)
    return got


def mt_negatives(pos_batch, known, everyone, want, near=None, max_try=10, stats=None):
    """code/base/batch.py:86-116 generate_neg_triples_fast."""
    near = {} if near is None else near
    out = []
    for (h, r, t) in pos_batch:
        out.extend(_mt_one_positive(h, r, t, known, everyone, want, near, max_try, stats))
    assert len(out) == want * len(pos_batch)
    return out


def mt_relation_batch(triples1, triples2, known1, known2, ents1, ents2, batch_size, step, near1, near2, want):
    """code/base/batch.py:33-42 generate_relation_triple_batch."""
    b1, b2 = kg_batch_split(len(triples1), len(triples2), batch_size)
    p1 = mt_epoch_slice(triples1, b1, step)
    p2 = mt_epoch_slice(triples2, b2, step)
    n1 = mt_negatives(p1, known1, ents1, want, near=near1)
    n2 = mt_negatives(p2, known2, ents2, want, near=near2)
    return p1 + p2, n1 + n2


# ----------------------------------------------------------------------------------------------
# (2) Philox specification (what the device implements)
# ----------------------------------------------------------------------------------------------
_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw — SC'11), ten rounds, as in Random123."""
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK, p1 & _MASK, ((p0 >> 32) ^ c3 ^ k1) & _MASK, p0 & _MASK
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


class _Drawer:
    """Bounded draws for one (positive, round, slot): attempt a uses word a&3 of block a>>2;
    Lemire multiply-shift with exact rejection."""

    def __init__(self, gi, rnd, slot, sid, k0, k1, n):
        self.gi, self.rnd, self.slot, self.sid, self.k0, self.k1, self.n = gi, rnd, slot, sid, k0, k1, n
        self.attempt = 0

    def next(self):
        n = self.n
        while True:
            blk, word = self.attempt >> 2, self.attempt & 3
            x = philox4x32_10(self.gi, (self.rnd | (blk << 8)) & _MASK, self.slot, self.sid, self.k0, self.k1)[word]
            self.attempt += 1
            m = x * n
            low = m & _MASK
            if low < n:
                thresh = ((1 << 32) - n) % n
                if low < thresh:
                    continue
            return m >> 32


def triple_key(h, r, t):
    """Packed 64-bit key of the device hash set: h<<38 | t<<12 | r."""
    return (int(h) << 38) | (int(t) << 12) | int(r)


def philox_negatives(pos_h, pos_r, pos_t, want, n_all, ent_lo=0, ent_list=None, cand_table=None, cand_valid=None,
                     known=None, seed=(0, 0), stream_id=0, pos_offset=0, max_try=10, stats=None, _coin_per_slot=False):
    """Specification of mke_neg_sample.  `known` is a Python set of (h, r, t) or None.
    Returns int32 arrays (neg_h, neg_r, neg_t) of length len(pos_h) * want.
    `stats`: a dict whose "rounds" list receives the rounds each positive used.  `_coin_per_slot=True` is a deliberately
    WRONG variant (one coin per negative instead of one per round) that exists only so that
    tests/test_sampler_oracle.py can show its two-sample tests have the power to reject it."""
    k0, k1 = seed[0] & _MASK, seed[1] & _MASK
    P = len(pos_h)
    nh = np.zeros(P * want, dtype=np.int32)
    nr = np.zeros(P * want, dtype=np.int32)
    nt = np.zeros(P * want, dtype=np.int32)
    cand_k = 0 if cand_table is None else cand_table.shape[1]
    for i in range(P):
        h, r, t = int(pos_h[i]), int(pos_r[i]), int(pos_t[i])
        gi = (i + pos_offset) & _MASK
        got = used = 0
        for rnd in range(max_try):
            if got >= want:
                break
            used += 1
            need = want - got
            coin = philox4x32_10(gi, rnd, _MASK, stream_id, k0, k1)[0] >> 31
            x = h if coin else t
            use_tbl = cand_table is not None and (cand_valid is None or cand_valid[x] != 0)
            n = cand_k if use_tbl else n_all
            finals = []
            for slot in range(need):
                dr = _Drawer(gi, rnd, slot, stream_id, k0, k1, n)
                p = dr.next()
                while p in finals:  # without replacement: differ from every earlier slot's final draw
                    p = dr.next()
                finals.append(p)
            for slot, p in enumerate(finals):
                if _coin_per_slot:
                    coin = philox4x32_10(gi, rnd, (_MASK - 1 - slot) & _MASK, stream_id, k0, k1)[0] >> 31
                    x = h if coin else t
                if use_tbl:
                    e = int(cand_table[x, p])
                elif ent_list is not None:
                    e = int(ent_list[p])
                else:
                    e = ent_lo + p
                cand = (e, r, t) if coin else (h, r, e)
                if rnd < max_try - 1 and known is not None and cand in known:
                    continue
                o = i * want + got
                nh[o], nr[o], nt[o] = cand[0], cand[1], cand[2]
                got += 1
        assert got == want
        if stats is not None:
            stats.setdefault("rounds", []).append(used)
    return nh, nr, nt


# ----------------------------------------------------------------------------------------------------------------
# random.sample(list, batch) per step as a keyed permutation (mke_sample_distinct).  The reference draws with
# Python's `random.sample` (code/MultiKE_model.py:358,380,402,425,446); the property that matters -- `batch` DISTINCT
# positions per step, steps independent, uniform -- is kept, the bit stream is this build's own (Philox-keyed Feistel
# network with cycle walking), restated here for the bit-exact device test.
# ----------------------------------------------------------------------------------------------------------------
def distinct_sample(n, batch, n_steps, seed=(0, 0), stream_id=0):
    """-> int32 [n_steps, batch]: out[s, i] = pi_s(i)."""
    import numpy as np
    assert 0 <= batch <= n < 2 ** 31
    half = 1
    while (1 << (2 * half)) < n:
        half += 1
    mask = (1 << half) - 1
    out = np.zeros((n_steps, batch), dtype=np.int32)
    for s in range(n_steps):
        ka = philox4x32_10(s, 0, 0x5A4D504C, stream_id, seed[0], seed[1])
        kb = philox4x32_10(s, 1, 0x5A4D504C, stream_id, seed[0], seed[1])
        keys = [int(k) for k in (ka[0], ka[1], ka[2], ka[3], kb[0], kb[1])]
        x = np.arange(batch, dtype=np.uint64)
        todo = np.ones(batch, dtype=bool)
        while todo.any():
            l, r = x[todo] >> np.uint64(half), x[todo] & np.uint64(mask)
            for k in keys:
                f = r.copy()
                f = (f * np.uint64(0x9E3779B1) + np.uint64(k)) & np.uint64(0xFFFFFFFF)
                f ^= f >> np.uint64(15)
                f = (f * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
                f ^= f >> np.uint64(13)
                f = (f * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
                f ^= f >> np.uint64(16)
                l, r = r, l ^ (f & np.uint64(mask))
            x[todo] = (l << np.uint64(half)) | r
            todo = x >= np.uint64(n)
        out[s] = x.astype(np.int32)
    return out
