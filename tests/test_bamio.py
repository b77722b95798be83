"""BAM writing and reading (BGZF blocks, header, records, .bai linear index) round trip, and region fetch
against a direct selection of the overlapping records."""
import gzip
import struct

import numpy as np
import pytest

from platypus_amd import bamio, simulate


def _ends(reads):
    ends = reads["pos"].copy()
    for i in range(len(ends)):
        for w in reads["cigar"][reads["cigarOff"][i]:reads["cigarOff"][i + 1]]:
            if (int(w) & 0xf) in (0, 2, 3):
                ends[i] += int(w) >> 4
    return ends


def _select(reads, keep):
    idx = np.flatnonzero(keep)
    if len(idx) == 0:
        return dict(pos=reads["pos"][:0])
    out = {k: reads[k][idx] for k in ("pos", "mapq", "flag")}
    out["seq"] = np.concatenate([reads["seq"][reads["off"][i]:reads["off"][i + 1]] for i in idx])
    out["qual"] = np.concatenate([reads["qual"][reads["off"][i]:reads["off"][i + 1]] for i in idx])
    out["off"] = np.concatenate([[0], np.cumsum(reads["off"][idx + 1] - reads["off"][idx])])
    out["cigar"] = np.concatenate([reads["cigar"][reads["cigarOff"][i]:reads["cigarOff"][i + 1]] for i in idx])
    out["cigarOff"] = np.concatenate([[0], np.cumsum(reads["cigarOff"][idx + 1] - reads["cigarOff"][idx])])
    return out


def test_bam_round_trip_and_region_fetch(tmp_path):
    rng = np.random.default_rng(4)
    G = 120000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 60, spacing=900)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=6, read_len=75)
    path = str(tmp_path / "s.bam")
    bamio.write_bam(path, [(b"other", 500), (b"sim", G)], dict(reads, refID=1), sample="NA1")
    assert gzip.open(path).read(4) == b"BAM\x01"                       # BGZF is valid multi-member gzip
    rd = bamio.BamReader(path)
    assert rd.refs == [b"other", b"sim"] and rd.lengths == [500, G] and rd.sample == "NA1"
    ends = _ends(reads)
    for (s, e) in [(0, G), (40000, 41000), (16384, 32768), (99990, 100010), (G - 50, G), (5, 6)]:
        got = rd.fetch(b"sim", s, e)
        want = _select(reads, (reads["pos"] < e) & (ends > s))
        assert len(got["pos"]) == len(want["pos"])
        for k in want:
            assert np.array_equal(got[k], want[k]), (s, e, k)
        assert np.all(got["insertSize"] == 0) and np.all(got["matePos"] == -1)
    assert len(rd.fetch(b"other", 0, 500)["pos"]) == 0
    # the same stretch read in pieces side by side: every record once, in the same order
    for (s, e, parts) in [(0, G, 7), (40000, 90000, 3), (16384, 16500, 4), (5, 6, 2)]:
        got, want = bamio.fetch_split(path, b"sim", s, e, parts), rd.fetch(b"sim", s, e)
        assert set(got) == set(want)
        for k in want:
            assert np.array_equal(got[k], want[k]), (s, e, parts, k)
    rd.close()
    # without the index the reader scans
    import os
    os.remove(path + ".bai")
    rd = bamio.BamReader(path)
    got = rd.fetch(b"sim", 40000, 41000)
    assert np.array_equal(got["pos"], _select(reads, (reads["pos"] < 41000) & (ends > 40000))["pos"])
    rd.close()


def test_packed_writer_equals_record_by_record_writer(tmp_path):
    """write_bam has two record encoders (whole arrays in Cython; one record at a time when names
    are given): same reads, same names -> the same BAM and .bai, byte for byte.  Also covers
    simulate.tile: repeats keep positions sorted and offsets consistent."""
    import filecmp
    from platypus_amd import simulate
    rng = np.random.default_rng(11)
    G = 40000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 30, spacing=600)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=6, read_len=101)      # odd length: half-filled last byte
    genome, truth, reads = simulate.tile(genome, truth, reads, 3)
    n = len(reads["pos"])
    assert len(genome) == 3 * G and (np.diff(reads["pos"]) >= 0).all()
    assert reads["off"][-1] == len(reads["seq"]) and reads["cigarOff"][-1] == len(reads["cigar"])
    a, b = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    bamio.write_bam(a, [(b"sim", 3 * G)], reads)
    bamio.write_bam(b, [(b"sim", 3 * G)], reads, names=["r%d" % i for i in range(n)])
    assert filecmp.cmp(a, b, shallow=False) and filecmp.cmp(a + ".bai", b + ".bai", shallow=False)
    back = bamio.BamReader(a).fetch(b"sim", G + 100, G + 5000)
    lo = np.searchsorted(reads["pos"], G + 100 - 101, side="left")
    assert len(back["pos"]) and set(back["pos"]) <= set(reads["pos"][lo:])


def test_block_inflation_in_c(tmp_path):
    """cbam.inflateBlocks: whole blocks are inflated behind the prefix, a cut-off last block stays
    unconsumed, damage is reported."""
    import zlib
    from platypus_amd import cbam

    def block(payload):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(payload) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25) + body +
                struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))

    a, b = bytes(range(256)) * 40, b"second block " * 1000
    raw = block(a) + block(b"") + block(b)
    u8 = lambda x: np.frombuffer(x, dtype=np.uint8)
    out, used = cbam.inflateBlocks(u8(raw), u8(b"left over"))
    assert used == len(raw) and out.tobytes() == b"left over" + a + b
    out, used = cbam.inflateBlocks(u8(raw[:-5]), u8(b""))
    assert used == len(block(a)) + len(block(b"")) and out.tobytes() == a
    out, used = cbam.inflateBlocks(u8(raw[:10]), u8(b"xy"))
    assert used == 0 and out.tobytes() == b"xy"
    damaged = bytearray(raw)
    damaged[40] ^= 0x55
    with pytest.raises(IOError):
        cbam.inflateBlocks(u8(bytes(damaged)), u8(b""))
    with pytest.raises(IOError):
        cbam.inflateBlocks(u8(b"not a gzip member at all....."), u8(b""))


def test_corrupt_records_are_reported():
    from platypus_amd import cbam
    rec = struct.pack("<iiiIIiiii", 32 + 3 + 4 + 2 + 4, 0, 100, (4681 << 16) | (60 << 8) | 3, (0 << 16) | 1, 4, -1, -1, 0) + \
        b"r0\x00" + struct.pack("<I", (4 << 4) | 0) + bytes([0x12, 0x48]) + bytes([30, 30, 30, 30])
    arrays, used, done = cbam.parseRecords(np.frombuffer(rec, dtype=np.uint8), 0, 0, 1000)
    assert used == len(rec) and arrays["seq"].tobytes() == b"ACGT" and list(arrays["pos"]) == [100]
    short = bytearray(rec)
    short[0:4] = struct.pack("<i", 8)                             # a block size no record can have
    with pytest.raises(IOError):
        cbam.parseRecords(np.frombuffer(bytes(short), dtype=np.uint8), 0, 0, 1000)
    long_seq = bytearray(rec)
    long_seq[20:24] = struct.pack("<i", 4000)                     # sequence longer than its record
    with pytest.raises(IOError):
        cbam.parseRecords(np.frombuffer(bytes(long_seq), dtype=np.uint8), 0, 0, 1000)


def test_region_bytes_follow_coverage(tmp_path):
    """BamReader.regionBytes: a stretch with three times the reads weighs about three times as much"""
    from platypus_amd import simulate
    rng = np.random.default_rng(5)
    G = 400000
    genome = simulate.random_genome(rng, G)
    lo = simulate.simulate_reads(rng, genome[:200000], [], coverage=5, read_len=100)
    hi = simulate.simulate_reads(rng, genome[200000:], [], coverage=15, read_len=100)
    hi["pos"] = hi["pos"] + 200000
    reads = {k: np.concatenate([lo[k], hi[k]]) for k in ("seq", "qual", "pos", "mapq", "flag", "cigar")}
    reads["off"] = np.concatenate([lo["off"], hi["off"][1:] + lo["off"][-1]])
    reads["cigarOff"] = np.concatenate([lo["cigarOff"], hi["cigarOff"][1:] + lo["cigarOff"][-1]])
    path = str(tmp_path / "w.bam")
    bamio.write_bam(path, [(b"sim", G)], reads)
    r = bamio.BamReader(path)
    a, b = r.regionBytes("sim", 0, 196608), r.regionBytes("sim", 212992, 399999)
    assert a > 0 and 2.2 < b / a < 3.8
    assert r.regionBytes("nope", 0, 10) is None


def test_pieces_keep_records_without_extent_on_their_edges(tmp_path):
    """records without an M, D or N operation (all '=' or all soft clip: their end is their position,
    samtoolsWrapper.pyx:773-774) and records without CIGAR that start exactly where a piece starts: the stretch read in
    pieces holds what the stretch read at once holds"""
    rng = np.random.default_rng(77)
    G = 8000
    genome = simulate.random_genome(rng, G)
    reads = simulate.simulate_reads(rng, genome, [], coverage=80, read_len=50)
    s, e, parts = 1000, 6000, 12
    edges = [s + (e - s) * k // parts for k in range(parts + 1)]
    coff, cig = reads["cigarOff"], reads["cigar"].copy()
    changed = 0
    for i in np.flatnonzero(np.isin(reads["pos"], edges)):
        if coff[i + 1] - coff[i] == 1:
            cig[coff[i]] = (cig[coff[i]] & ~np.uint32(0xf)) | np.uint32(7 if changed % 2 else 4)
            changed += 1
    assert changed > 6
    path = str(tmp_path / "edges.bam")
    bamio.write_bam(path, [(b"sim", G)], dict(reads, cigar=cig), sample="E")
    rd = bamio.BamReader(path)
    want = rd.fetch(b"sim", s, e)
    rd.close()
    got = bamio.fetch_split(path, b"sim", s, e, parts)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    on_edges = np.isin(want["pos"], edges[1:-1])
    ops = want["cigar"][want["cigarOff"][:-1][on_edges]] & 0xf
    assert ((ops == 7) | (ops == 4)).sum() >= 8                         # (those on the stretch's own start and end are outside it)
