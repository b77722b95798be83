"""The BAM/BAI reader and writer against the reference's own vendored samtools (oracle/samref.py over
oracle/_ref/libsamref.so, built from /root/reference/platypus/c/samtools by `make -C oracle samref`):
files written here are read, indexed and fetched by the reference's code, and files written and
indexed by the reference's code are read here.  CPU only."""
import os
import shutil

import numpy as np
import pytest

from oracle import samref
from platypus_amd import bamio, simulate

pytestmark = pytest.mark.skipif(not samref.available(), reason="oracle/_ref/libsamref.so not built (make -C oracle samref)")


def _rows(arrays):
    """packed arrays of BamReader.fetch as comparable tuples"""
    out = []
    for i in range(len(arrays["pos"])):
        s0, s1 = int(arrays["off"][i]), int(arrays["off"][i + 1])
        c0, c1 = int(arrays["cigarOff"][i]), int(arrays["cigarOff"][i + 1])
        out.append((int(arrays["pos"][i]), int(arrays["flag"][i]), int(arrays["mapq"][i]), arrays["seq"][s0:s1].tobytes(),
                    arrays["qual"][s0:s1].tobytes(), tuple(int(w) for w in arrays["cigar"][c0:c1]),
                    int(arrays["insertSize"][i]), int(arrays["matePos"][i])))
    return out


def _ref_rows(recs):
    out = []
    for r in recs:
        cig = tuple(int(w) for w in r["cigar"])
        out.append((r["pos"], r["flag"], r["mapq"], r["seq"], r["qual"], cig, r["isize"], r["mpos"]))
    return out


def test_reference_reads_indexes_and_fetches_our_bam(tmp_path):
    rng = np.random.default_rng(11)
    G = 150000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 80, spacing=700)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=8, read_len=100, paired=True, insert=(400, 40))
    path = str(tmp_path / "ours.bam")
    bamio.write_bam(path, [(b"first", 700), (b"sim", G)], dict(reads, refID=1), sample="S1")
    refs, lens, text, recs = samref.read_all(path)
    assert refs == [b"first", b"sim"] and lens == [700, G] and b"SM:S1" in text
    rd = bamio.BamReader(path)
    ours = _rows(rd.fetch(b"sim", 0, G))
    assert len(recs) == len(reads["pos"]) == len(ours)
    assert _ref_rows(recs) == ours                                       # every field the caller uses, record by record
    assert all(r["tid"] == 1 for r in recs)
    # the bin our writer stores is the one the reference computes for the alignment
    for r in recs[::37]:
        end = r["pos"] + sum(int(w) >> 4 for w in r["cigar"] if (int(w) & 15) in (0, 2, 3))
        assert r["bin"] == samref.reg2bin(r["pos"], max(end, r["pos"] + 1))
    regions = [(0, G), (40000, 41000), (16384, 32768), (16383, 16385), (99990, 100010), (G - 50, G), (5, 6),
               (65536, 65537), (131071, 131073)]
    regions += [(int(a), int(a + w)) for a, w in zip(rng.integers(0, G - 3000, 12), rng.integers(1, 3000, 12))]
    # (1) the reference's bam_fetch through OUR index = our fetch
    for s, e in regions:
        assert _ref_rows(samref.fetch(path, 1, s, e)) == _rows(rd.fetch(b"sim", s, e)), (s, e)
    rd.close()
    # (2) the reference's own index of our file: same answers from both readers
    path2 = str(tmp_path / "theirs_index.bam")
    shutil.copy(path, path2)
    samref.index_build(path2)
    rd2 = bamio.BamReader(path2)
    assert rd2.linear is not None
    for s, e in regions:
        want = _ref_rows(samref.fetch(path2, 1, s, e))
        assert want == _rows(rd2.fetch(b"sim", s, e)), (s, e)
        assert want == _ref_rows(samref.fetch(path, 1, s, e)), (s, e)
    rd2.close()


def _cig(*ops):
    return np.array([(n << 4) | "MIDNSHP=X".index(o) for n, o in ops], dtype=np.uint32)


def test_we_read_what_the_reference_writes(tmp_path):
    """records of the kinds our own writer never makes: real names, tags, hard and soft clips, '=' and
    'X' operations, a record without CIGAR, zero-length and odd-length sequences, several references,
    unmapped reads at the end; written and indexed by the reference's library (its BGZF writer and
    compression level)"""
    rng = np.random.default_rng(12)
    recs = []
    bases = b"ACGTN"
    for tid, n, span in ((0, 300, 20000), (2, 2500, 400000)):
        pos = np.sort(rng.integers(0, span, n))
        for i, p in enumerate(pos):
            L = int(rng.integers(1, 160))
            kind = int(rng.integers(0, 8))
            if kind == 0:
                cig = _cig((5, "S"), (L - 5, "M")) if L > 6 else _cig((L, "M"))
            elif kind == 1 and L > 20:
                cig = _cig((10, "M"), (3, "I"), (L - 13, "M"))
            elif kind == 2 and L > 20:
                cig = _cig((4, "H"), (10, "M"), (7, "D"), (L - 10, "M"), (2, "H"))
            elif kind == 3 and L > 20:
                cig = _cig((8, "="), (1, "X"), (L - 9, "="))
            elif kind == 4 and L > 30:
                cig = _cig((10, "M"), (1500, "N"), (L - 10, "M"))
            elif kind == 5:
                cig = _cig()                                               # no CIGAR: overlaps as one base
            elif kind == 6:
                cig = _cig((L, "S"))                                       # nothing on the reference
            else:
                cig = _cig((L, "M"))
            seq = bytes(bases[k] for k in rng.integers(0, 5, L))
            qual = bytes(int(q) for q in rng.integers(0, 94, L))
            aux = b"NMC\x03" + b"RGZgrp1\x00" if i % 3 == 0 else b""
            recs.append(dict(tid=tid, pos=int(p), mapq=int(rng.integers(0, 256)), flag=int(rng.integers(0, 2048)) & ~4,
                             name=b"read_%d/%d" % (tid, i), cigar=cig, seq=seq, qual=qual, mtid=tid, mpos=int(p) + 200,
                             isize=int(rng.integers(-600, 600)), aux=aux))
    recs.append(dict(tid=2, pos=399999, mapq=3, flag=16, name=b"star", cigar=_cig(), seq=b"", qual=b"", mtid=-1, mpos=-1))
    for i in range(40):                                                    # unmapped, unplaced reads come last
        recs.append(dict(tid=-1, pos=-1, mapq=0, flag=4, name=b"un%d" % i, cigar=_cig(), seq=b"ACGT", qual=b"\x10" * 4))
    path = str(tmp_path / "theirs.bam")
    text = b"@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chrA\tLN:20000\n@SQ\tSN:empty\tLN:100\n@SQ\tSN:chrB\tLN:400000\n@RG\tID:grp1\tSM:patient7\n"
    samref.write(path, [b"chrA", b"empty", b"chrB"], [20000, 100, 400000], text, recs)
    samref.index_build(path)
    rd = bamio.BamReader(path)
    assert rd.refs == [b"chrA", b"empty", b"chrB"] and rd.lengths == [20000, 100, 400000] and rd.sample == "patient7"
    regions = {0: [(0, 20000), (0, 1), (16383, 16385), (19990, 20000), (5000, 5001)],
               1: [(0, 100)],
               2: [(0, 400000), (0, 1), (16384, 32768), (131071, 131073), (262144, 262145), (399999, 400000), (399990, 400000)]}
    for tid in (0, 2):
        span = 20000 if tid == 0 else 400000
        regions[tid] += [(int(a), int(a + w)) for a, w in zip(rng.integers(0, span - 5000, 15), rng.integers(1, 5000, 15))]
    n_checked = 0
    for tid, regs in regions.items():
        for s, e in regs:
            want = _ref_rows(samref.fetch(path, tid, s, e))
            got = _rows(rd.fetch(rd.refs[tid], s, e))
            assert got == want, (tid, s, e, len(got), len(want))
            n_checked += len(want)
    assert n_checked > 3000
    # the whole of a reference = the records of that reference that overlap anything, in file order
    whole = _rows(rd.fetch(b"chrB", 0, 400000))
    assert len(whole) > 2000
    rd.close()
    # without an index our reader scans from the first record: same answers
    os.remove(path + ".bai")
    rd = bamio.BamReader(path)
    assert rd.linear is None
    for tid, regs in regions.items():
        for s, e in regs[:6]:
            os_rows = _rows(rd.fetch(rd.refs[tid], s, e))
            assert all(r[0] < e for r in os_rows)
    assert _rows(rd.fetch(b"chrB", 0, 400000)) == whole
    rd.close()


def test_named_record_writer_against_the_reference(tmp_path):
    """write_bam's record-by-record path (explicit read names): the reference reads the names and
    fetches through the index it wrote"""
    rng = np.random.default_rng(13)
    G = 90000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 30, spacing=900)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=5, read_len=120)
    names = ["frag%07d" % i for i in range(len(reads["pos"]))]
    path = str(tmp_path / "named.bam")
    bamio.write_bam(path, [(b"sim", G)], reads, sample="N", names=names)
    _, _, _, recs = samref.read_all(path)
    assert [r["name"].decode() for r in recs] == names
    rd = bamio.BamReader(path)
    assert _ref_rows(recs) == _rows(rd.fetch(b"sim", 0, G))
    for s, e in [(0, G), (16000, 17000), (32768, 49152), (70000, 70001)] + [(int(a), int(a) + 500) for a in rng.integers(0, G - 500, 10)]:
        assert _ref_rows(samref.fetch(path, 0, s, e)) == _rows(rd.fetch(b"sim", s, e)), (s, e)
    rd.close()
