"""BGZF inflation on the device (pg_bgzf_inflate) against zlib: every byte equal.  The same decoder,
instantiated for the host, is checked without a GPU in tests/test_inflate_host.py."""
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bgzf(chunks, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, extra=b""):
    out = []
    for chunk in chunks:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        body = c.compress(chunk) + c.flush()
        xlen = 6 + len(extra)
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", xlen) + extra + b"BC\x02\x00" +
                   struct.pack("<H", len(body) + 12 + xlen + 8 - 1) + body + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


def _payloads(rng, n):
    out = []
    for k in range(n):
        size = int(rng.integers(1, 65281))
        kind = k % 6
        if kind == 0:
            d = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
        elif kind == 1:
            d = rng.integers(0, 4, size, dtype=np.uint8).tobytes()
        elif kind == 2:
            d = (b"ACGTTGCA" * 9000)[:size]
        elif kind == 3:
            d = bytes([int(rng.integers(0, 256))]) * size
        elif kind == 4:
            d = b"".join(rng.integers(30, 42, 150, dtype=np.uint8).tobytes() + b"name%d\x00" % i for i in range(size // 160 + 1))[:size]
        else:
            d = (b"%d\t" % k) * (size // 4 + 1)
            d = d[:size]
        out.append(d)
    return out


@pytest.mark.parametrize("lanes", [1, 8, 16, 32])
def test_inflate_equals_zlib(lanes):
    from platypus_amd import native
    rng = np.random.default_rng(100 + lanes)
    inf = native.Inflater(0)
    inf.set_lanes(lanes)
    for level, strategy in ((1, 0), (6, 0), (9, 0), (0, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
        chunks = _payloads(rng, 150)
        chunks.insert(7, b"")                                     # an empty block in the middle (the EOF marker is one)
        raw = _bgzf(chunks, level, strategy, extra=b"XY\x03\x00abc" if level == 9 else b"")
        want = b"".join(chunks)
        cut = len(raw) - 11                                       # the last block is incomplete
        prefix = b"left over"
        got, eaten = inf.inflate(raw[:cut], prefix)
        n_last = len(chunks[-1])
        assert eaten == native.bgzf_scan(raw[:cut])[0] and eaten < cut
        assert got.tobytes() == prefix + want[:len(want) - n_last]
        # the tail of one call's output as the next call's prefix (it lies in the buffer the call reuses)
        tail = got[len(got) - 1000:]
        keep = tail.tobytes()
        got2, eaten2 = inf.inflate(raw[eaten:], tail)
        assert eaten2 == len(raw) - eaten
        assert got2.tobytes() == keep + chunks[-1]
    inf.close()


def test_corrupt_blocks_are_refused():
    from platypus_amd import native
    rng = np.random.default_rng(7)
    inf = native.Inflater(0)
    chunks = _payloads(rng, 40)
    raw = bytearray(_bgzf(chunks, 6))
    with pytest.raises(native.PlatGpuError):
        inf.inflate(b"\x1f\x8b\x08\x00" + bytes(raw[4:]))        # no extra field: not BGZF
    # random damage inside deflate streams: either refused, or (when the damage happens to decode) the sizes still match
    for trial in range(30):
        bad = bytearray(raw)
        for _ in range(20):
            p = int(rng.integers(18, len(bad) - 8))
            bad[p] ^= 1 << int(rng.integers(0, 8))
        try:
            got, eaten = inf.inflate(bytes(bad))
        except native.PlatGpuError as e:
            assert "corrupt" in str(e) or "BGZF" in str(e)
        else:
            assert len(got) == sum(len(c) for c in chunks)
    got, _ = inf.inflate(bytes(raw))                              # the inflater is still usable
    assert got.tobytes() == b"".join(chunks)
    inf.close()


def test_bam_stretch_through_the_device_inflater(tmp_path):
    """a simulated BAM read with the device inflater = the same file read with the host library"""
    from platypus_amd import bamio, cbam, simulate
    rng = np.random.default_rng(3)
    G = 300000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 150, spacing=800)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=20, read_len=150, paired=True, insert=(500, 50))
    path = str(tmp_path / "s.bam")
    bamio.write_bam(path, [(b"sim", G)], reads, sample="S")
    host = bamio.BamReader(path)
    dev = bamio.BamReader(path, inflater=cbam.DeviceInflater(0))
    pool = bamio.InflaterPool(0, 2)
    pooled = bamio.BamReader(path, inflater=pool)
    for s, e in [(0, G), (100000, 200000), (16384, 16400), (G - 10, G)]:
        a, b, c = host.fetch(b"sim", s, e), dev.fetch(b"sim", s, e), pooled.fetch(b"sim", s, e)
        assert set(a) == set(b) == set(c) and len(a["pos"]) > 0
        for k in a:
            assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), (s, e, k)
    split = bamio.fetch_split(path, b"sim", 0, G, 5, pool)       # five pieces side by side borrow from two inflaters
    whole = host.fetch(b"sim", 0, G)
    for k in whole:
        assert np.array_equal(split[k], whole[k]), k
    for r in (host, dev, pooled):
        r.close()
    # without an index the reader scans from the first record: same records through the device inflater
    import os
    import shutil
    bare = str(tmp_path / "bare.bam")
    shutil.copy(path, bare)
    assert not os.path.exists(bare + ".bai")
    r1, r2 = bamio.BamReader(bare), bamio.BamReader(bare, inflater=pool)
    assert r2.linear is None
    a, b = r1.fetch(b"sim", 120000, 121000), r2.fetch(b"sim", 120000, 121000)
    assert len(a["pos"]) > 0
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    r1.close()
    r2.close()


def test_device_inflater_reads_a_file_written_by_the_reference_library(tmp_path):
    """a BAM written and indexed by the reference's vendored samtools (its own BGZF writer and compression level):
    the device inflater and the host library give the same records"""
    from oracle import samref
    if not samref.available():
        pytest.skip("oracle/_ref/libsamref.so not built")
    from platypus_amd import bamio
    rng = np.random.default_rng(8)
    recs = []
    for i, p in enumerate(np.sort(rng.integers(0, 300000, 6000))):
        L = int(rng.integers(30, 151))
        recs.append(dict(tid=0, pos=int(p), mapq=int(rng.integers(0, 61)), flag=int(rng.choice([0, 16, 99, 147])), name=b"q%d" % i,
                         cigar=np.array([(L << 4) | 0], dtype=np.uint32), seq=bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L)),
                         qual=bytes(int(q) for q in rng.integers(2, 41, L)), mtid=0, mpos=int(p) + 200, isize=300,
                         aux=b"NMC\x01" if i % 2 else b""))
    path = str(tmp_path / "theirs.bam")
    samref.write(path, [b"chr1"], [300200], b"@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:300200\n", recs)
    samref.index_build(path)
    host, dev = bamio.BamReader(path), bamio.BamReader(path, inflater=bamio.InflaterPool(0, 1))
    for s, e in [(0, 300200), (100000, 150000), (299000, 300200)]:
        a, b = host.fetch(b"chr1", s, e), dev.fetch(b"chr1", s, e)
        assert len(a["pos"]) > 10
        for k in a:
            assert np.array_equal(a[k], b[k]), (s, e, k)
    host.close()
    dev.close()
