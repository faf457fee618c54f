"""multike_amd/rccl.py: the ncclUniqueId as it travels through torch.distributed.  It holds a socket address — NUL bytes from its
second byte on — and a ctypes `c_char` array field reads as a C string: until round 6 the id was cut at its first NUL byte, so a
communicator over more than one rank could never form (`ncclCommInitRank`: network error after 60 s on every rank, found with
tools/rccl_two_ranks_one_gpu.py; one-rank communicators, all a one-GPU box can run, never send the id anywhere)."""
import ctypes as C

import pytest


def test_all_128_bytes_survive_the_round_trip():
    from multike_amd import rccl
    raw = bytes([2, 0, 0x9c, 0x40, 127, 0, 0, 1] + [0] * 8 + list(range(1, 113)))      # a sockaddr_in, then anything
    assert len(raw) == 128
    uid = rccl._uid_from_bytes(raw)
    assert C.sizeof(uid) == 128 and rccl._uid_to_bytes(uid) == raw
    again = rccl._uid_from_bytes(rccl._uid_to_bytes(uid))
    assert bytes(again) == raw


@pytest.mark.parametrize("bad", [b"", b"\x02", bytes(127), bytes(129), None])
def test_a_short_or_missing_id_is_refused(bad):
    from multike_amd import rccl
    with pytest.raises(rccl.RcclError):
        rccl._uid_from_bytes(bad)
