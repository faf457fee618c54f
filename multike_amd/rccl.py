"""RCCL called directly (ctypes on the librccl.so PyTorch ships and has already loaded) for the collectives of the sharded
steps, ON THE STREAM THE KERNELS RUN ON.

Why not torch.distributed for them: ProcessGroupNCCL runs every collective on a stream of its own and orders it against the
caller's stream with two events.  Measured on MI355X (`tools/stream_hop_probe.py`, EXPERIMENTS R5.3): a hop to another
stream and back costs ~26 us on the DEVICE timeline (inter-queue barrier packets) and ~30 us of host time, per collective —
three collectives per global step, on a step whose kernels take 76 us at the C2 shape.  A collective enqueued on the compute
stream itself needs no event at all: stream order is the dependency.  (`async_op=True`, the chunk-pipelined schedule, still
goes to a second stream — there the overlap is the point — with two pre-created events instead of four fresh ones.)

The reference has no multi-device code (new design); torch.distributed stays in charge of rendezvous — the
unique id travels through the existing process group — and of every non-hot-path exchange.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

_DT = {torch.float32: 7, torch.float64: 8, torch.int32: 2, torch.int64: 4, torch.uint8: 1, torch.float16: 6, torch.bfloat16: 9}
_SUM = 0
_lib = None


class RcclError(RuntimeError):
    pass


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]     # NOT c_char: reading a c_char array field stops at the first NUL byte


def _uid_to_bytes(uid: "_UniqueId") -> bytes:
    """All 128 bytes of an ncclUniqueId (it holds a socket address: NUL bytes from its second byte on)."""
    return C.string_at(C.addressof(uid), C.sizeof(uid))


def _uid_from_bytes(raw: bytes) -> "_UniqueId":
    if not isinstance(raw, (bytes, bytearray)) or len(raw) != C.sizeof(_UniqueId):
        raise RcclError(f"ncclUniqueId: {C.sizeof(_UniqueId)} bytes expected, got {len(raw) if hasattr(raw, '__len__') else type(raw)}")
    uid = _UniqueId()
    C.memmove(C.addressof(uid), bytes(raw), C.sizeof(uid))
    return uid


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = C.CDLL(path)                 # the copy torch itself uses (already mapped: no second RCCL in the process)
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclReduceScatter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _ck(rc, what):
    if rc != 0:
        raise RcclError(f"{what}: {lib().ncclGetErrorString(C.c_int(rc)).decode()} ({rc})")


class Communicator:
    """One RCCL communicator over the ranks of a torch.distributed group (default: the world).  The unique id is created on
    the group's first rank and broadcast through torch.distributed; every rank must construct it at the same point."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RcclError("torch.distributed must be initialised first (it carries the unique id)")
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        L = lib()
        uid = _UniqueId()
        box = [None]
        if self.rank == 0:
            _ck(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
            box[0] = _uid_to_bytes(uid)
        if self.world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            uid = _uid_from_bytes(box[0])
        self._comm = C.c_void_p()
        _ck(L.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def _stream(self, stream):
        return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)

    @staticmethod
    def _buf(t, name):
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DT:
            raise RcclError(f"{name}: contiguous CUDA tensor of a supported dtype expected")
        return C.c_void_p(t.data_ptr())

    def all_gather(self, out, mine, stream=None):
        if out.numel() != self.world * mine.numel() or out.dtype != mine.dtype:
            raise RcclError("all_gather: out must hold world x mine")
        _ck(lib().ncclAllGather(self._buf(mine, "mine"), self._buf(out, "out"), mine.numel(), _DT[mine.dtype], self._comm,
                                self._stream(stream)), "ncclAllGather")

    def reduce_scatter(self, out, inp, stream=None):
        if inp.numel() != self.world * out.numel() or out.dtype != inp.dtype:
            raise RcclError("reduce_scatter: inp must hold world x out")
        _ck(lib().ncclReduceScatter(self._buf(inp, "inp"), self._buf(out, "out"), out.numel(), _DT[out.dtype], _SUM, self._comm,
                                    self._stream(stream)), "ncclReduceScatter")

    def all_reduce(self, t, stream=None):
        _ck(lib().ncclAllReduce(self._buf(t, "t"), self._buf(t, "t"), t.numel(), _DT[t.dtype], _SUM, self._comm, self._stream(stream)),
            "ncclAllReduce")

    def self_check(self):
        """sum over ranks of (rank + 1) through all three collectives; raises when a result is wrong"""
        G, r = self.world, self.rank
        t = torch.full((4,), float(r + 1), device="cuda")
        self.all_reduce(t)
        every = torch.empty(4 * G, device="cuda")
        self.all_gather(every, torch.full((4,), float(r + 1), device="cuda"))
        part = torch.empty(4, device="cuda")
        self.reduce_scatter(part, torch.arange(4 * G, dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()
        want = G * (G + 1) / 2
        ok = bool((t == want).all()) and every.view(G, 4)[:, 0].tolist() == [float(k + 1) for k in range(G)] and \
            part.tolist() == [float(G * (4 * r + k)) for k in range(4)]
        if not ok:
            raise RcclError(f"RCCL self-check failed on rank {r}: all_reduce {t.tolist()}, all_gather {every.tolist()}, reduce_scatter {part.tolist()}")

    def destroy(self):
        if self._comm:
            lib().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()
