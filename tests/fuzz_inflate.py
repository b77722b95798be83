#!/usr/bin/env python
"""Randomised run of the device inflater (pg_bgzf_inflate) against zlib for a time budget: stretches of 1..400
BGZF blocks whose payloads, compression level and strategy are drawn at random (noise, few-symbol noise, repeats,
runs, BAM-like records, text; stored, fixed and dynamic blocks), block counts per wavefront 8 / 16 / 32, a prefix
of random length, a truncated last block.  Every output byte is compared.  Not collected by pytest.

    python tests/fuzz_inflate.py [seconds] [first_seed]
"""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from platypus_amd import native  # noqa: E402
from test_gpu_inflate import _bgzf  # noqa: E402


def payload(rng, size):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    if kind == 1:
        return rng.integers(0, int(rng.integers(2, 20)), size, dtype=np.uint8).tobytes()
    if kind == 2:
        unit = rng.integers(65, 91, int(rng.integers(1, 300)), dtype=np.uint8).tobytes()
        return (unit * (size // len(unit) + 1))[:size]
    if kind == 3:
        return bytes([int(rng.integers(0, 256))]) * size
    if kind == 4:                                                   # BAM-like: fixed fields, a name, packed bases, qualities
        recs = []
        while sum(map(len, recs)) < size:
            recs.append(b"\x2c\x01\x00\x00\x00\x00\x00\x00" + int(rng.integers(0, 1 << 28)).to_bytes(4, "little") +
                        b"r%d\x00" % int(rng.integers(0, 10 ** 7)) + rng.integers(0, 256, 75, dtype=np.uint8).tobytes() +
                        rng.integers(20, 42, 150, dtype=np.uint8).tobytes())
        return b"".join(recs)[:size]
    if kind == 5:
        words = [b"alpha", b"beta", b"gamma", b"\t", b"\n", b"0123456789", b"chr20"]
        out = b"".join(words[int(k)] for k in rng.integers(0, len(words), size // 3 + 1))
        return out[:size]
    return b""


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    inf = native.Inflater(0)
    t0 = last = time.perf_counter()
    n = blocks = bad = 0
    total = 0
    while time.perf_counter() - t0 < budget:
        rng = np.random.default_rng(seed)
        nb = int(rng.choice([1, 2, 7, 8, 9, 33, 100, 400]))
        level = int(rng.choice([0, 1, 1, 6, 6, 9]))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
        chunks = [payload(rng, int(rng.integers(1, 65281))) for _ in range(nb)]
        raw = _bgzf(chunks, level, strategy)
        prefix = rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8).tobytes()
        cut = len(raw) - int(rng.integers(0, 2)) * int(rng.integers(1, 12))
        inf.set_lanes(int(rng.choice([1, 1, 1, 8, 32])))
        got, eaten = inf.inflate(raw[:cut], prefix)
        whole = cut == len(raw)
        want = prefix + b"".join(chunks if whole else chunks[:-1])
        if got.tobytes() != want or (whole and eaten != len(raw)):
            bad += 1
            print("seed %d: DIFFERENT (blocks %d, level %d, strategy %d)" % (seed, nb, level, strategy), flush=True)
        n += 1
        blocks += nb
        total += len(want)
        seed += 1
        if time.perf_counter() - last > 20:
            last = time.perf_counter()
            print("%d stretches, %d blocks, %.0f MB, %d different" % (n, blocks, total / 1e6, bad), flush=True)
    print("%d stretches, %d blocks, %.0f MB inflated, %d different from zlib (seeds %d..%d)"
          % (n, blocks, total / 1e6, bad, int(sys.argv[2]) if len(sys.argv) > 2 else 1, seed - 1))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
