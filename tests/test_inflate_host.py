"""The device inflater's decoder (platypus_amd/csrc/pg_inflate.hip.h), instantiated for the host with one
lane (tests/native/inflate_host.hip), against zlib: stored, fixed and dynamic blocks, all strategies and levels,
every alignment of the stream in its buffer; damaged streams end with an error or the right size, never
outside their buffers.  CPU only."""
import ctypes
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "inflate_host.hip")
LIB = os.path.join(HERE, "native", "libinflate_host.so")
HEADER = os.path.join(os.path.dirname(HERE), "platypus_amd", "csrc", "pg_inflate.hip.h")


@pytest.fixture(scope="module")
def infl():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HEADER)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-o", LIB, SRC])
    L = ctypes.CDLL(LIB)
    L.infl_host.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_char_p, ctypes.c_uint]
    return L


def _run(L, body, n_out, lead):
    raw = bytes(lead) + body + bytes(64)
    out = ctypes.create_string_buffer(n_out + 64)
    rc = L.infl_host(raw, lead, len(body), out, n_out)
    assert out.raw[n_out:] == bytes(64)                           # nothing written past the block's output
    return rc, out.raw[:n_out]


def test_host_instance_equals_zlib(infl):
    rng = np.random.default_rng(1)
    n = 0
    for trial in range(120):
        size = int(rng.integers(0, 65536))
        kind = trial % 6
        if kind == 0:
            data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
        elif kind == 1:
            data = rng.integers(0, 4, size, dtype=np.uint8).tobytes()
        elif kind == 2:
            data = (b"ACGTTGCA" * 9000)[:size]
        elif kind == 3:
            data = bytes([int(rng.integers(0, 256))]) * size
        elif kind == 4:
            data = b"".join(rng.integers(30, 42, 150, dtype=np.uint8).tobytes() + b"name%d\x00" % i for i in range(size // 160))
        else:
            data = open(HEADER, "rb").read()[:size]
        for level, strat in ((0, 0), (1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
            body = c.compress(data) + c.flush()
            rc, out = _run(infl, body, len(data), 18 + trial % 7)
            assert rc == 0 and out == data, (trial, level, strat)
            n += 1
    assert n == 840


def test_damaged_streams_stay_inside_their_buffers(infl):
    rng = np.random.default_rng(2)
    data = b"".join(rng.integers(30, 42, 150, dtype=np.uint8).tobytes() + b"r%d\x00" % i for i in range(300))
    body = zlib.compress(data, 6)[2:-4]
    seen = set()
    for trial in range(400):
        bad = bytearray(body)
        for _ in range(int(rng.integers(1, 6))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        if trial % 5 == 0:
            bad = bad[:int(rng.integers(1, len(bad)))]            # truncated
        rc, out = _run(infl, bytes(bad), len(data), 20)
        seen.add(rc)
        assert 0 <= rc <= 8
    assert len(seen) >= 4                                         # several different kinds of error were met
    rc, _ = _run(infl, body, len(data) - 1, 20)                   # ISIZE smaller than what the stream holds
    assert rc in (6, 8)
    rc, _ = _run(infl, body, len(data) + 1, 20)
    assert rc == 8
