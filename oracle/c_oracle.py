"""CPU ORACLE (test infrastructure) — ctypes wrapper of oracle/libmke_oracle.so, the plain-C restatement
(oracle/mke_oracle.c) used as a fast checker at full batch sizes and as bench.py's `cpu_baseline` port."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmke_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(
            os.path.getmtime(os.path.join(_HERE, f)) for f in ("mke_oracle.c", "mke_oracle_impl.h")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        for suf in ("f32", "f64"):
            fn = getattr(_lib, "mko_relation_step_" + suf)
            fn.restype = C.c_double
            getattr(_lib, "mko_relation_step_mt_" + suf).restype = C.c_double
        _lib.mko_neg_sample.restype = C.c_int
        _lib.mko_set_threads.restype = None
        _lib.mko_set_insert.restype = None
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RelationStepOracle:
    """Holds the scratch buffers of mko_relation_step for fixed table shapes."""

    def __init__(self, n_ent, n_rel, dim, dtype=np.float64, dense=False):
        self.n_ent, self.n_rel, self.dim, self.dtype, self.dense = n_ent, n_rel, dim, np.dtype(dtype), dense
        self.g_ent = np.zeros((n_ent, dim), dtype)
        self.g_rel = np.zeros((n_rel, dim), dtype)
        self.mark_ent = np.zeros(n_ent, np.uint8)
        self.mark_rel = np.zeros(n_rel, np.uint8)
        self.list_ent = np.zeros(n_ent, np.int32)
        self.list_rel = np.zeros(n_rel, np.int32)
        self.norm_ent = np.zeros((n_ent, dim), dtype) if dense else None
        self.norm_rel = np.zeros((n_rel, dim), dtype) if dense else None
        self.fn = getattr(lib(), "mko_relation_step_" + ("f32" if self.dtype == np.float32 else "f64"))

    def step(self, ent, rel, acc_ent, acc_rel, pos, neg, lr, pos_w=None, neg_w=None, scale=1.0, ent_norm=True,
             rel_norm=True, update=True):
        for a in (ent, rel, acc_ent, acc_rel):
            assert a.dtype == self.dtype and a.flags.c_contiguous
        ph, pr, pt = (np.ascontiguousarray(a, np.int32) for a in pos)
        if neg is None:
            z = np.zeros(0, np.int32)
            neg = (z, z, z)
        nh, nr, nt = (np.ascontiguousarray(a, np.int32) for a in neg)
        pw = None if pos_w is None else np.ascontiguousarray(pos_w, self.dtype)
        nw = None if neg_w is None else np.ascontiguousarray(neg_w, self.dtype)
        return self.fn(_p(ent), _p(rel), _p(acc_ent), _p(acc_rel), C.c_int64(self.n_ent), C.c_int64(self.n_rel),
                       C.c_int(self.dim), _p(ph), _p(pr), _p(pt), _p(pw), C.c_int64(len(ph)), _p(nh), _p(nr), _p(nt),
                       _p(nw), C.c_int64(len(nh)), C.c_double(scale), C.c_double(lr), C.c_int(int(ent_norm)),
                       C.c_int(int(rel_norm)), C.c_int(int(self.dense)), C.c_int(int(update)), _p(self.g_ent),
                       _p(self.g_rel), _p(self.mark_ent), _p(self.mark_rel), _p(self.list_ent), _p(self.list_rel),
                       _p(self.norm_ent), _p(self.norm_rel))


def set_threads(n: int):
    """Threads of mko_neg_sample (the sampled negatives do not depend on it)."""
    lib().mko_set_threads(C.c_int(int(n)))


class RelationStepBaselineMT:
    """bench.py's `cpu_baseline` leg: the OpenMP step (mko_relation_step_mt_f32), touched-rows or dense cost model.
    Not a parity checker (a row's gradient is summed in thread-schedule order)."""

    def __init__(self, n_ent, n_rel, dim, dense=False, threads=1):
        f = np.float32
        self.n_ent, self.n_rel, self.dim, self.dense, self.threads = n_ent, n_rel, dim, dense, int(threads)
        self.g_ent, self.g_rel = np.zeros((n_ent, dim), f), np.zeros((n_rel, dim), f)
        self.mark_ent, self.mark_rel = np.zeros(n_ent, np.uint8), np.zeros(n_rel, np.uint8)
        self.inv_ent, self.inv_rel = np.ones(n_ent, f), np.ones(n_rel, f)
        self.norm_ent = np.zeros((n_ent, dim), f) if dense else None
        self.norm_rel = np.zeros((n_rel, dim), f) if dense else None
        self.gtrip = None
        self.fn = lib().mko_relation_step_mt_f32

    def step(self, ent, rel, acc_ent, acc_rel, pos, neg, lr):
        for a in (ent, rel, acc_ent, acc_rel):
            assert a.dtype == np.float32 and a.flags.c_contiguous
        ph, pr, pt = (np.ascontiguousarray(a, np.int32) for a in pos)
        nh, nr, nt = (np.ascontiguousarray(a, np.int32) for a in neg)
        if self.gtrip is None or self.gtrip.shape[0] < len(ph) + len(nh):
            self.gtrip = np.zeros((len(ph) + len(nh), self.dim), np.float32)
        return self.fn(_p(ent), _p(rel), _p(acc_ent), _p(acc_rel), C.c_int64(self.n_ent), C.c_int64(self.n_rel),
                       C.c_int(self.dim), _p(ph), _p(pr), _p(pt), C.c_int64(len(ph)), _p(nh), _p(nr), _p(nt),
                       C.c_int64(len(nh)), C.c_double(lr), C.c_int(int(self.dense)), _p(self.g_ent), _p(self.g_rel),
                       _p(self.mark_ent), _p(self.mark_rel), _p(self.inv_ent), _p(self.inv_rel), _p(self.norm_ent),
                       _p(self.norm_rel), _p(self.gtrip), C.c_int(self.threads))


class TripleSet:
    """Open-addressing set of packed (h, r, t) keys — same key packing as the device set."""

    def __init__(self, h, r, t):
        n = len(h)
        cap = 1
        while cap < 2 * n + 2:
            cap *= 2
        self.cap = cap
        self.keys = np.full(cap, 0xFFFFFFFFFFFFFFFF, np.uint64)
        lib().mko_set_insert(_p(np.ascontiguousarray(h, np.int32)), _p(np.ascontiguousarray(r, np.int32)),
                             _p(np.ascontiguousarray(t, np.int32)), C.c_int64(n), _p(self.keys), C.c_uint64(cap))


def neg_sample(pos_h, pos_r, pos_t, want, n_all, ent_lo=0, ent_list=None, cand_table=None, cand_valid=None,
               known: TripleSet | None = None, seed=(0, 0), stream_id=0, pos_offset=0, max_try=10):
    ph, pr, pt = (np.ascontiguousarray(a, np.int32) for a in (pos_h, pos_r, pos_t))
    n = len(ph)
    nh = np.zeros(n * want, np.int32)
    nr = np.zeros(n * want, np.int32)
    nt = np.zeros(n * want, np.int32)
    el = None if ent_list is None else np.ascontiguousarray(ent_list, np.int32)
    ct = None if cand_table is None else np.ascontiguousarray(cand_table, np.int32)
    cv = None if cand_valid is None else np.ascontiguousarray(cand_valid, np.uint8)
    rc = lib().mko_neg_sample(_p(ph), _p(pr), _p(pt), C.c_int64(n), C.c_int64(pos_offset), C.c_int(want),
                              C.c_int(max_try), _p(el), C.c_int32(ent_lo), C.c_int32(n_all), _p(ct), _p(cv),
                              C.c_int32(0 if ct is None else ct.shape[1]), _p(None if known is None else known.keys),
                              C.c_uint64(0 if known is None else known.cap), C.c_uint32(seed[0]), C.c_uint32(seed[1]),
                              C.c_uint32(stream_id), _p(nh), _p(nr), _p(nt))
    assert rc == 0
    return nh, nr, nt
