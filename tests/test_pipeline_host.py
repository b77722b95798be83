"""Host steps of a region call (no GPU): record filtering and trimming (cwindow.pyx:74-229), candidate
generation (variant.pyx:253-615), standard form and filtering of candidates (chaplotype.pyx:532-665,
variantFilter.pyx:12-83), windows (window.py), haplotype enumeration (cfilter.pyx:22-78).  Expectations
are worked out from the reference's rules by hand or by construction of the simulated data."""
import math

import numpy as np
import pytest

from platypus_amd import pipeline, simulate
from platypus_amd.fastafile import FastaFile
from platypus_amd.variant import Variant, VariantCandidateGenerator


def _records(recs):
    """recs: (pos, seq, qual list, flag, cigar [(op, n)], insertSize, matePos, mapq)"""
    off = np.concatenate([[0], np.cumsum([len(r[1]) for r in recs])]).astype(np.int64)
    return dict(chrom=b"c", seq=np.frombuffer(b"".join(r[1] for r in recs), dtype=np.uint8),
                qual=np.concatenate([np.asarray(r[2], dtype=np.uint8) for r in recs]), off=off,
                pos=np.array([r[0] for r in recs], dtype=np.int32), mapq=np.array([r[7] for r in recs], dtype=np.int32),
                flag=np.array([r[3] for r in recs], dtype=np.int32),
                cigar=np.array([(n << 4) | op for r in recs for op, n in r[4]], dtype=np.uint32),
                cigarOff=np.concatenate([[0], np.cumsum([len(r[4]) for r in recs])]).astype(np.int64),
                insertSize=np.array([r[5] for r in recs], dtype=np.int32),
                matePos=np.array([r[6] for r in recs], dtype=np.int32))


def test_record_filtering_and_trimming():
    from platypus_amd import cwindow
    L = 30
    good = [30] * L
    recs = [
        (100, b"A" * L, good, 0, [(0, L)], 0, -1, 60),                                   # kept
        (100, b"A" * L, good, 0, [(0, L)], 0, -1, 60),                                   # same pos + length, unpaired: dropped
        (101, b"A" * L, good, 0, [(0, L)], 0, -1, 10),                                   # mapq < 20
        (102, b"A" * L, [10] * 15 + [30] * 15, 0, [(0, L)], 0, -1, 60),                  # only 15 good bases (< 20)
        (103, b"A" * L, good, 0x400, [(0, L)], 0, -1, 60),                               # duplicate flag
        (104, b"A" * L, good, 0x4, [(0, L)], 0, -1, 60),                                 # unmapped
        (105, b"A" * L, good, 0x1 | 0x20, [(0, L)], 20, 90, 60),                         # fragment shorter than the read
        (106, b"A" * L, [2, 3, 30, 4] + [30] * 23 + [4, 1, 2], 0x10, [(0, 10), (2, 5), (0, 20)], 0, -1, 60),   # trimmed ends
        (107, b"A" * L, good, 0x1 | 0x20, [(0, L)], 45, 122, 60),                        # fragment 45 < 2 * 30: 15 bases overlap
        (107, b"A" * L, good, 0x1 | 0x20, [(0, L)], 45, 300, 60),                        # same pos/len, other mate: kept
        (107, b"A" * L, good, 0x1 | 0x20, [(0, L)], 45, 300, 60),                        # same mate position: dropped
    ]
    r = _records(recs)
    buf = cwindow.bamReadBuffer.fromAlignments(b"c", 0, 1000, r["seq"], r["qual"], r["off"], r["pos"], r["mapq"], r["flag"],
                                               r["cigar"], r["cigarOff"], r["insertSize"], r["matePos"])
    info = [buf.readInfo(i) for i in range(buf.readCount)]
    assert [x[0] for x in info] == [100, 106, 107, 107]
    assert info[1][1] == 106 + 35 and info[1][4] == 1                                   # end counts M and D; reverse strand
    assert list(info[1][5]) == [0, 0, 30, 4] + [30] * 23 + [0, 0, 0]                     # heads / tails below 5 zeroed, inner 4 kept
    assert list(info[2][5]) == [30] * 15 + [0] * 15                                       # overlap with the mate zeroed
    assert list(info[3][5]) == [30] * 15 + [0] * 15


def test_candidates_from_cigars():
    ref = b"ACGTTGCAAGGCTTAACCGGATATCGCGATTACAGGCTAGCTAGGATCCAGTCATGCAAGTCC" * 8      # 504 bases
    fa = FastaFile(sequences={b"c": ref})
    L = 40
    def read(pos, edits, cigar, flag=0, lowq=()):
        s = bytearray(ref[pos:pos + 200])
        out, rp = bytearray(), 0
        for op, n in cigar:
            if op == 0:
                out += s[rp:rp + n]; rp += n
            elif op == 1:
                out += b"T" * n if s[rp:rp + 1] != b"T" else b"G" * n
            elif op == 2:
                rp += n
            elif op == 4:
                out += b"A" * n
        for i, b in edits:
            out[i] = b
        return (pos, bytes(out), [10 if i in lowq else 30 for i in range(len(out))], flag, cigar, 0, -1, 60)
    snp_alt = ord("A") if ref[210:211] != b"A" else ord("C")
    recs = [
        read(200, [(10, snp_alt)], [(0, L)]),                          # SNP at 210, forward
        read(201, [(9, snp_alt)], [(0, L)], flag=0x10),                # same SNP, reverse
        read(202, [(8, snp_alt)], [(0, L)], lowq=(8,)),                # low base quality: no SNP candidate
        read(220, [], [(0, 15), (1, 3), (0, 22)]),                     # insertion after base 234
        read(221, [], [(0, 14), (1, 3), (0, 23)], flag=0x10),          # the same insertion
        read(240, [], [(0, 20), (2, 4), (0, 20)]),                     # deletion of 260..263
        read(270, [], [(0, 2), (1, 2), (0, 2), (4, 34)]),              # insertion between matches shorter than minFlank
        read(300, [(1, ord("A") if ref[301:302] != b"A" else ord("C"))], [(0, 2), (2, 1), (0, 38)]),   # SNP in a 2-base match: ignored
    ]
    recs.sort(key=lambda r: r[0])
    buf = simulate.buffer_of(_records(recs), 0, 504)
    assert buf.readCount == len(recs)
    gen = VariantCandidateGenerator((b"c", 100, 400), fa)
    gen.addCandidatesFromBuffer(buf)
    cands = gen.getCandidates()
    by = {(v.refPos, v.removed, v.added): v for v in cands}
    snp = by[(210, ref[210:211], bytes([snp_alt]))]
    assert (snp.nSupportingReads, snp.nFwdReads, snp.nRevReads) == (2, 1, 1)
    # the sums of variant.pyx:57-76,107-119 over the two reads: weights from the base quality (30), read positions 11 and 10
    w = 1.0 - math.exp(-0.23025850929940459 * 30)
    assert snp.nFwdReadsWeighted == w and snp.nRevReadsWeighted == w
    # the mean of the squared read positions is an integer quotient (variant.pxd: int sumSqReadPos, int nSupportingReads)
    assert snp.rmsReadPos == math.sqrt((11 * 11 + 10 * 10) // 2) and (snp.minPos, snp.maxPos) == (210, 210)
    assert gen.getCandidates() == cands and gen.getCandidates()[1].nSupportingReads == cands[1].nSupportingReads   # no double counting
    ins_seq = b"TTT" if ref[235:236] != b"T" else b"GGG"
    ins = by[(234, b"", ins_seq)]
    assert (ins.nSupportingReads, ins.nFwdReads, ins.nRevReads) == (2, 1, 1)
    dele = by[(259, ref[260:264], b"")]
    assert dele.nSupportingReads == 1
    assert len(cands) == 4 and (301, ref[302:303], b"") in by            # the flanked deletion of base 302 in the last read
    assert cands == sorted(cands)
    # filter: minReads = 2 keeps the SNP and the insertion
    kept = pipeline.filterVariants(cands, fa, 150, 2, 0, 250)
    assert [(v.refPos, v.added) for v in kept if v.nAdded == 1 and v.nRemoved == 1] == [(210, bytes([snp_alt]))]
    assert len(kept) == 2


def test_standard_form_moves_indels_left():
    ref = (b"ACGTTGCAAGGCTTAACCGGATATCGCGATTACAGGCTAGC" * 5 + b"GATTTTTTTTC" + b"CAGTCATGCAAGTCCTAGGCATCGA" * 8 +
           b"ACACACACAG" + b"TTGACCATGCATGCAAGTTC" * 10)
    fa = FastaFile(sequences={b"c": ref})
    t0 = ref.index(b"GATTTTTTTTC") + 2                                   # first T of the run of 8
    # a T inserted after the 5th T is the same haplotype as a T inserted at the start of the run
    v = pipeline.convertVariantToStandardFormat(Variant(b"c", t0 + 4, b"", b"T", 3, 2, 60, 30, 5), fa, 150)
    assert (v.refPos, v.added, v.removed, v.nSupportingReads, v.nFwdReads) == (t0 - 1, b"T", b"", 3, 2)
    assert v.minPos == t0 - 1 and v.maxPos == t0 - 1 + 1 + 5
    # deleting the last two Ts == deleting the first two
    d = pipeline.convertVariantToStandardFormat(Variant(b"c", t0 + 5, b"TT", b""), fa, 150)
    assert (d.refPos, d.removed, d.added) == (t0 - 1, b"TT", b"")
    # a dinucleotide repeat: deleting the last AC == deleting the first
    a0 = ref.index(b"ACACACACAG")
    d = pipeline.convertVariantToStandardFormat(Variant(b"c", a0 + 5, b"AC", b""), fa, 150)
    assert (d.refPos, d.removed) == (a0 - 1, b"AC")
    # SNPs and variants in the first 100 bases are left alone (chaplotype.pyx:546-552)
    s = Variant(b"c", 300, b"A", b"C")
    assert pipeline.convertVariantToStandardFormat(s, fa, 150) is s
    e = Variant(b"c", 50, b"", b"G")
    assert pipeline.convertVariantToStandardFormat(e, fa, 150) is e
    # two spellings of one insertion are merged by the filter and their support added
    kept = pipeline.filterVariants(sorted([Variant(b"c", t0 + 1, b"", b"T", 1, 1, 60, 30, 5),
                                           Variant(b"c", t0 + 6, b"", b"T", 1, 0, 60, 30, 7)]), fa, 150, 2, 0, 250)
    assert len(kept) == 1 and (kept[0].refPos, kept[0].nSupportingReads, kept[0].nFwdReads, kept[0].nRevReads) == (t0 - 1, 2, 1, 1)
    # ... but only NEIGHBOURS of the sorted list merge (variantFilter.pyx:44): with another candidate between
    # the two spellings each is judged on its own support, whether that candidate is in the list (first call)
    # or was left out early for its low support and is known by its position only (second call)
    mk = lambda sup: [Variant(b"c", t0 + 1, b"", b"T", sup, 1, 60, 30, 5), Variant(b"c", t0 + 6, b"", b"T", sup, 0, 60, 30, 7)]
    weak = Variant(b"c", t0 + 3, ref[t0 + 3:t0 + 4], b"G", 1, 1, 60, 30, 5)
    assert pipeline.filterVariants(sorted(mk(1) + [weak]), fa, 150, 2, 0, 250) == []
    assert pipeline.filterVariants(sorted(mk(1)), fa, 150, 2, 0, 250, separators=[t0 + 3]) == []
    assert len(pipeline.filterVariants(sorted(mk(1)), fa, 150, 2, 0, 250, separators=[t0 - 40, t0 + 60])) == 1   # none between
    strong = Variant(b"c", t0 + 3, ref[t0 + 3:t0 + 4], b"G", 5, 3, 60, 30, 5)
    kept = pipeline.filterVariants(sorted(mk(1) + [strong]), fa, 150, 2, 0, 250)
    assert [(v.refPos, v.added) for v in kept] == [(t0 + 3, b"G")]
    # both spellings pass on their own: the reference's list holds the standard form twice and its set keeps one
    for kw, extra in ((dict(separators=[t0 + 3]), []), ({}, [weak])):
        kept = pipeline.filterVariants(sorted(mk(2) + extra), fa, 150, 2, 0, 250, **kw)
        assert [(v.refPos, v.nSupportingReads) for v in kept] == [(t0 - 1, 2), (t0 - 1, 2)]
        assert [(v.refPos, v.nSupportingReads) for v in sorted(dict.fromkeys(kept))] == [(t0 - 1, 2)]


def test_windows_and_haplotype_enumeration():
    ref = bytes(b"ACGT"[i % 4] for i in range(3000))
    fa = FastaFile(sequences={b"c": ref})
    opts = pipeline.Options(rlen=50)
    snp = lambda p: Variant(b"c", p, ref[p:p + 1], b"A" if ref[p:p + 1] != b"A" else b"C", 5, 3)
    v1, v2, v3 = snp(500), snp(500 + 1), snp(900)
    ins = Variant(b"c", 1200, b"", b"GG", 5, 3)          # maxPos = 1202
    v4 = snp(1202)                                        # interacts with the insertion
    v5 = snp(2000)
    wins = list(pipeline.WindowGenerator().WindowsAndVariants(b"c", 100, 2000, sorted([v1, v2, v3, ins, v4, v5]), opts))
    assert [(w["startPos"], w["endPos"], len(w["variants"])) for w in wins] == \
        [(500, 501, 1), (501, 502, 1), (900, 901, 1), (1200, 1202, 2), (1999, 2000, 1)]
    # two SNPs at one position share a window and cannot be combined
    alt2 = Variant(b"c", 500, ref[500:501], b"T" if ref[500:501] != b"T" else b"G", 5, 3)
    wins = list(pipeline.WindowGenerator().WindowsAndVariants(b"c", 100, 2000, sorted([v1, alt2]), opts))
    assert len(wins) == 1 and len(wins[0]["variants"]) == 2
    haps = pipeline.haplotypesInWindow(wins[0], 40, fa, opts)
    assert len(haps) == 2
    # an insertion and a SNP next to it: three haplotypes (each alone + both)
    w = dict(chromosome=b"c", startPos=1200, endPos=1203, variants=[ins, snp(1203)])
    haps = pipeline.haplotypesInWindow(w, 40, fa, opts)
    assert [len(h.variantsPy()) for h in haps] == [1, 1, 2]
    # three variants, two of them overlapping (a deletion over the SNP): 2^3 - 1 = 7 combinations, 5 valid
    dele = Variant(b"c", 1299, ref[1300:1303], b"", 5, 3)    # maxPos 1302
    w = dict(chromosome=b"c", startPos=1290, endPos=1320, variants=sorted([dele, snp(1301), snp(1310)]))
    haps = pipeline.haplotypesInWindow(w, 40, fa, opts)
    assert len(haps) == 5
    assert pipeline.haplotypesInWindow(w, 0, fa, opts) == []
    assert pipeline.haplotypesInWindow(dict(w, variants=[snp(1290 + 2 * i) for i in range(9)]), 40, fa, opts) == []   # 511 > 256


def test_simulator_alignments_are_consistent():
    rng = np.random.default_rng(2)
    genome = simulate.random_genome(rng, 6000)
    variants = simulate.plant_variants(rng, genome, 12)
    assert len(variants) >= 8
    reads = simulate.simulate_reads(rng, genome, variants, coverage=8, read_len=80, error_rate=0.0)
    fa = FastaFile(sequences={b"sim": genome})
    buf = simulate.buffer_of(reads, 0, 6000)
    gen = VariantCandidateGenerator((b"sim", 0, 6000), fa)
    gen.addCandidatesFromBuffer(buf)
    found = {(v.refPos, v.removed, v.added) for v in gen.getCandidates()}
    # error-free reads: every candidate is a planted variant, and every planted variant with coverage is found
    assert found <= {(p, r, a) for p, r, a, g in variants}
    assert len(found) >= len(variants) - 1


def test_early_snp_support_cut_changes_nothing():
    """getCandidates(minSnpSupport) followed by the filter == the filter over all candidates"""
    rng = np.random.default_rng(8)
    genome = simulate.random_genome(rng, 30000)
    variants = simulate.plant_variants(rng, genome, 30)
    reads = simulate.simulate_reads(rng, genome, variants, coverage=40, read_len=100, error_rate=0.02, error_qual=(15, 41))
    fa = FastaFile(sequences={b"sim": genome})
    buf = simulate.buffer_of(reads, 0, 30000)
    lists = []
    for cut in (0, 2):
        gen = VariantCandidateGenerator((b"sim", 0, 30000), fa)
        gen.addCandidatesFromBuffer(buf)
        raw = gen.getCandidates(cut)
        assert (len(gen.droppedSnpPositions) > 0) == (cut > 0)
        kept = pipeline.filterVariants(raw, fa, 100, 2, 0, 250, separators=gen.droppedSnpPositions)
        lists.append((len(raw), [(v.refPos, v.removed, v.added, v.nSupportingReads, v.nFwdReads, v.nFwdReadsWeighted,
                                  v.nRevReadsWeighted, v.rmsReadPos) for v in kept]))
    assert lists[0][1] == lists[1][1] and len(lists[0][1]) > 100
    assert lists[1][0] < lists[0][0] // 3                     # most single-read mismatches never became objects


def test_paired_simulation_is_consistent():
    """paired mode: mates point at each other, template lengths are +/- the same span, error-free
    reads give back exactly the planted variants, and (unlike unpaired reads that share a start,
    cwindow.pyx:104-113) almost no pair is dropped as a duplicate"""
    rng = np.random.default_rng(4)
    genome = simulate.random_genome(rng, 20000)
    variants = simulate.plant_variants(rng, genome, 25)
    reads = simulate.simulate_reads(rng, genome, variants, coverage=20, read_len=100, error_rate=0.0, paired=True,
                                    insert=(300, 30))
    n = len(reads["pos"])
    assert n % 2 == 0 and (np.diff(reads["pos"]) >= 0).all()
    fwd = (reads["flag"] & 0x10) == 0
    assert fwd.sum() == n // 2 and ((reads["flag"] & 0x1) != 0).all()
    assert (reads["insertSize"][fwd] > 0).all() and (reads["insertSize"][~fwd] < 0).all()
    assert sorted(reads["insertSize"][fwd]) == sorted(-reads["insertSize"][~fwd])
    assert sorted(reads["matePos"][fwd]) == sorted(reads["pos"][~fwd]) and sorted(reads["matePos"][~fwd]) == sorted(reads["pos"][fwd])
    assert abs(np.mean(reads["insertSize"][fwd]) - 300) < 10
    fa = FastaFile(sequences={b"sim": genome})
    buf = simulate.buffer_of(reads, 0, 20000)
    assert buf.readCount >= 0.99 * n
    gen = VariantCandidateGenerator((b"sim", 0, 20000), fa)
    gen.addCandidatesFromBuffer(buf)
    found = {(v.refPos, v.removed, v.added) for v in gen.getCandidates()}
    assert found <= {(p, r, a) for p, r, a, g in variants} and len(found) >= len(variants) - 1
    # the tiled form keeps mates consistent
    g2, t2, r2 = simulate.tile(genome, variants, reads, 2)
    assert sorted(r2["matePos"][n:]) == sorted(reads["matePos"] + 20000)


def test_binomial_statistics():
    """cpopulation.pyx:106-142 against exact values"""
    import math
    for n, k, p in [(10, 3, 0.5), (30, 0, 0.1), (14, 14, 0.9), (40, 7, 0.25)]:
        exact = math.comb(n, k) * p ** k * (1 - p) ** (n - k)
        assert abs(pipeline.binomial(k, n, p) - exact) < 2e-4 * exact + 1e-300      # Stirling series above 14
    assert pipeline.binomial(5, 5, 1) == 1.0 and pipeline.binomial(4, 5, 1) == 0.0 and pipeline.binomial(0, 7, 0) == 1.0
    assert abs(pipeline.logFactorial(10) - math.lgamma(11)) < 1e-12 and abs(pipeline.logFactorial(50) - math.lgamma(51)) < 1e-9
    ref = b"ACGTTTTTTGCAAAAAAGGCTTAACCGGATATCGCGATTACAGGCTAGCTAGGATCC" * 3
    fa = FastaFile(sequences={b"c": ref})
    t0 = ref.index(b"GTTTTTTG") + 1
    assert pipeline.homopolymerLength(Variant(b"c", t0 - 1, b"", b"T"), fa) == 6            # G | TTTTTT
    assert pipeline.homopolymerLength(Variant(b"c", t0 + 2, b"T", b"A"), fa) == 5           # TT [T] TTT: 2 + 3
    assert pipeline.homopolymerLength(Variant(b"c", 100, b"A", b"C"), fa) >= 1


def test_region_without_reads_gives_no_calls():
    """nothing to call, and nothing is asked of the device (this runs without one)"""
    from platypus_amd import bamio
    rng = np.random.default_rng(1)
    genome = simulate.random_genome(rng, 5000)
    fa = FastaFile(sequences={b"sim": genome})
    reads = bamio._concat([])
    reads["chrom"] = b"sim"
    buf = simulate.buffer_of(reads, 0, 5000)
    assert buf.readCount == 0
    assert pipeline.callRegion(b"sim", 0, 5000, [buf], fa, pipeline.Options(rlen=100)) == []


def test_other_letters_still_become_candidates():
    """a read letter that is neither A, C, G, T nor N goes through the object path (variant.pyx:309-365 makes no
    difference between letters); the table path beside it is unaffected"""
    ref = b"ACGTTGCAAGGCTTAACCGGATATCGCGATTACAGGCTAGCTAGGATCCAGTCATGCAAGTCC" * 8
    fa = FastaFile(sequences={b"c": ref})
    recs = []
    for pos, flag in ((200, 0), (203, 0x10)):
        s = bytearray(ref[pos:pos + 40])
        s[215 - pos] = ord("R")
        s[220 - pos] = ord("A") if ref[220:221] != b"A" else ord("C")
        recs.append((pos, bytes(s), [30] * 40, flag, [(0, 40)], 0, -1, 60))
    buf = simulate.buffer_of(_records(recs), 0, 504)
    gen = VariantCandidateGenerator((b"c", 100, 400), fa)
    gen.addCandidatesFromBuffer(buf)
    got = {(v.refPos, v.added): (v.nSupportingReads, v.nFwdReads) for v in gen.getCandidates()}
    assert got[(215, b"R")] == (2, 1) and got[(220, b"A" if ref[220:221] != b"A" else b"C")] == (2, 1) and len(got) == 2


def test_compiled_modules_are_not_older_than_their_sources():
    """an extension module shadows the source file of the same name: a build that is older than its source
    would run yesterday's code"""
    import os
    from platypus_amd import build_cython as bc
    for mod in bc.MODULES + bc.PY_MODULES:
        ext = bc._ext_path(mod)
        if os.path.exists(ext):
            assert os.path.getmtime(ext) >= os.path.getmtime(bc._source(bc.HERE, mod)), mod


def test_standard_form_equals_the_literal_restatement_on_random_indels():
    """the product compares base by base from the variant leftwards; the restatement builds the mutated sequence over
    the whole window as the reference does (chaplotype.pyx:532-665): the same standard forms for insertions and
    deletions in and around repeats, at the sequence's ends, of every length up to 40"""
    import numpy as np
    from oracle import candidates_oracle as co
    from oracle.vcf_oracle import Ref
    rng = np.random.default_rng(12)
    units = [b"A", b"AC", b"ACG", b"TTG", b"CAGT", b"GATTACA"]
    bits = []
    for k in range(400):
        bits.append(bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(rng.integers(5, 60)))]))
        u = units[int(rng.integers(0, len(units)))]
        bits.append(u * int(rng.integers(1, 25)))
    ref = b"".join(bits)
    fa, oref = FastaFile(sequences={b"c": ref}), Ref({b"c": ref})
    n = 0
    for k in range(3000):
        pos = int(rng.integers(0, len(ref) - 1))
        if rng.random() < 0.1:
            pos = int(rng.choice([0, 1, 50, 99, 100, 101, len(ref) - 2, len(ref) - 41, len(ref) - 3]))
        size = int(rng.integers(1, 41))
        if rng.random() < 0.5:
            # an insertion that copies what follows or precedes it (so that it can move), or random bases
            src = ref[pos + 1:pos + 1 + size] if rng.random() < 0.6 else bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size)])
            if not src:
                continue
            v = Variant(b"c", pos, b"", src, 2, 1, 60, 30, 7)
        else:
            rem = ref[pos + 1:pos + 1 + size]
            if not rem:
                continue
            v = Variant(b"c", pos, rem, b"", 2, 1, 60, 30, 7)
        try:
            want = co.standard_format(co.new_variant(b"c", v.refPos, v.removed, v.added, 2, 1, 60, 30, 7), oref)
        except Exception as e:                          # what the restatement refuses the product refuses
            with pytest.raises(Exception):
                pipeline.convertVariantToStandardFormat(v, fa, 150)
            continue
        got = pipeline.convertVariantToStandardFormat(v, fa, 150)
        assert (got.refPos, got.removed, got.added, got.minPos, got.maxPos) == (want.refPos, want.removed, want.added, want.minPos, want.maxPos), (k, v)
        n += 1
    assert n > 2500
