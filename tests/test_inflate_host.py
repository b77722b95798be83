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


class _Bits(object):
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):                                   # low bit first (RFC 1951, 3.1.1)
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, code, nbits):                                   # Huffman codes go out high bit first
        for k in range(nbits - 1, -1, -1):
            self.put((code >> k) & 1, 1)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def _canonical(lengths):
    """symbol -> (code, length) of the canonical code with these lengths (RFC 1951, 3.2.2)"""
    codes, code = {}, 0
    for l in range(1, 16):
        for s, ls in enumerate(lengths):
            if ls == l:
                codes[s] = (code, l)
                code += 1
        code <<= 1
    return codes


def _dynamic_block(lit, dst, clc=None, body=(256,)):
    """one final dynamic block whose literal/length and distance code lengths are `lit`, `dst`, written one by one
    with a code length code of lengths `clc` (default: a complete one: 13 symbols of 4 bits, 6 of 5); `body`:
    literal/length symbols to emit"""
    if clc is None:
        clc = [4] * 13 + [5] * 6
    w = _Bits()
    w.put(1, 1)
    w.put(2, 2)
    w.put(len(lit) - 257, 5)
    w.put(len(dst) - 1, 5)
    w.put(19 - 4, 4)
    for s in (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15):
        w.put(clc[s], 3)
    cc = _canonical(clc)
    for l in list(lit) + list(dst):
        if l in cc:
            w.code(*cc[l])
    lc = _canonical(lit)
    for s in body:
        if s in lc:
            w.code(*lc[s])
    return w.done() + bytes(4)


@pytest.mark.parametrize("name,lit,dst,clc", [
    ("complete sets", {0: 1, 256: 1}, {0: 1, 1: 1}, None),
    ("one distance code of one bit", {0: 1, 256: 1}, {0: 1}, None),
    ("no distance code at all", {0: 1, 256: 1}, {}, None),
    ("one literal/length code of one bit", {256: 1}, {0: 1}, None),
    ("incomplete literal/length set", {0: 2, 1: 2, 256: 2}, {0: 1, 1: 1}, None),
    ("incomplete distance set of two codes", {0: 1, 256: 1}, {0: 2, 1: 2}, None),
    ("one distance code of two bits", {0: 1, 256: 1}, {0: 2}, None),
    ("over-subscribed literal/length set", {0: 1, 1: 1, 256: 1}, {0: 1, 1: 1}, None),
    ("over-subscribed distance set", {0: 1, 256: 1}, {0: 1, 1: 1, 2: 1}, None),
    ("incomplete code length code", {0: 1, 256: 1}, {0: 1, 1: 1}, [4] * 12 + [0] + [5] * 6),
    ("no end-of-block code", {0: 1, 1: 1}, {0: 1, 1: 1}, None),
])
def test_sets_of_code_lengths_are_accepted_where_zlib_accepts_them(infl, name, lit, dst, clc):
    """zlib's inflate_table (what the reference's inflate_block runs) refuses over-subscribed sets, and incomplete
    ones unless they are a single code of one bit outside the code length code; so does this decoder"""
    lit_l = [lit.get(s, 0) for s in range(257)]
    dst_l = [dst.get(s, 0) for s in range(max(1, max(dst) + 1 if dst else 1))]
    body = _dynamic_block(lit_l, dst_l, clc)
    try:
        want = zlib.decompressobj(-15).decompress(body)
        ok = True
    except zlib.error:
        want, ok = None, False
    rc, out = _run(infl, body, 0, 20)
    assert (rc == 0) == ok, (name, rc, ok)
    if ok:
        assert want == b""
