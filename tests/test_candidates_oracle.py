"""Read filtering and trimming, candidate generation and candidate filtering of the region caller against a
plain-Python restatement of the reference's text (oracle/candidates_oracle.py: cwindow.pyx:74-223,
variant.pyx:28-119,290-474,600-615, chaplotype.pyx:532-665, variantFilter.pyx:12-83, variantcaller.pyx:92-108),
on simulated reads made harder: duplicates, unmapped and low-quality records, short fragments, bad heads and
tails, 'N's, clips and skips.  Every field of every kept read and of every candidate must be equal.  CPU only."""
import numpy as np
import pytest

from oracle import candidates_oracle as co
from oracle.vcf_oracle import Ref
from platypus_amd import pipeline, simulate
from platypus_amd.fastafile import FastaFile
from platypus_amd.variant import VariantCandidateGenerator


def _harden(rng, reads, genome_len):
    """makes the simulated records exercise every rule: returns new packed arrays"""
    n = len(reads["pos"])
    flag = reads["flag"].copy()
    mapq = reads["mapq"].copy()
    qual = reads["qual"].copy()
    seq = reads["seq"].copy()
    isz = reads.get("insertSize", np.zeros(n, np.int32)).copy()
    mpos = reads.get("matePos", np.full(n, -1, np.int32)).copy()
    off = reads["off"]
    pick = lambda frac: np.flatnonzero(rng.random(n) < frac)
    flag[pick(0.02)] |= 0x400                                         # duplicates
    flag[pick(0.01)] |= 0x4                                           # unmapped
    mapq[pick(0.05)] = rng.integers(0, 20)                            # below minMapQual
    for i in pick(0.10):                                              # bad tails and heads (below 5, below 20)
        a, b = int(off[i]), int(off[i + 1])
        k = int(rng.integers(1, 30))
        qual[b - k:b] = rng.integers(0, 8, k)
        k = int(rng.integers(1, 12))
        qual[a:a + k] = rng.integers(0, 8, k)
    for i in pick(0.03):                                              # many bad bases: the read goes
        a, b = int(off[i]), int(off[i + 1])
        qual[a:b] = rng.integers(0, 25, b - a)
    for i in pick(0.05):                                              # 'N's
        a, b = int(off[i]), int(off[i + 1])
        seq[a + rng.integers(0, b - a, 3)] = ord("N")
    for i in pick(0.06):                                              # short fragments: overlap with the mate, or adapter
        L = int(off[i + 1] - off[i])
        size = int(rng.integers(L // 2, 2 * L + 10))
        if flag[i] & 0x10:
            flag[i] = (flag[i] | 0x1) & ~0x20
            isz[i] = -size
        else:
            flag[i] |= 0x1 | 0x20
            isz[i] = size
        mpos[i] = int(reads["pos"][i]) + 50
    # exact copies of records right behind the original (same position and length: the second one goes unless paired
    # with another mate position)
    dup = pick(0.03)
    order = np.sort(np.concatenate([np.arange(n), dup]))
    out = dict(chrom=reads["chrom"], pos=reads["pos"][order], mapq=mapq[order], flag=flag[order], insertSize=isz[order],
               matePos=mpos[order])
    lens = (off[1:] - off[:-1])[order]
    out["off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out["seq"] = np.concatenate([seq[off[i]:off[i + 1]] for i in order])
    out["qual"] = np.concatenate([qual[off[i]:off[i + 1]] for i in order])
    coff = reads["cigarOff"]
    clens = (coff[1:] - coff[:-1])[order]
    out["cigarOff"] = np.concatenate([[0], np.cumsum(clens)]).astype(np.int64)
    out["cigar"] = np.concatenate([reads["cigar"][coff[i]:coff[i + 1]] for i in order]).astype(np.uint32)
    twice = np.flatnonzero(order[1:] == order[:-1]) + 1
    for j in twice[::2]:
        out["matePos"][j] += 7                                       # half of the copies have another mate position
    return out


def _clip_some(rng, reads):
    """turns the first bases of some reads' leading match into a soft clip (the position moves along), puts a hard
    clip in front of others, and writes the matches of others with '=' and 'X' operations"""
    coff, cig = reads["cigarOff"], list(reads["cigar"])
    pos = reads["pos"].copy()
    new, new_off = [], [0]
    for i in range(len(pos)):
        ops = [int(w) for w in reads["cigar"][coff[i]:coff[i + 1]]]
        r = rng.random()
        if r < 0.05 and (ops[0] & 0xf) == 0 and (ops[0] >> 4) > 12:
            k = int(rng.integers(1, 8))
            ops = [(k << 4) | 4, (((ops[0] >> 4) - k) << 4) | 0] + ops[1:]
            pos[i] += k
        elif r < 0.08:
            ops = [(5 << 4) | 5] + ops
        elif r < 0.10:                                                # every match written as '=': nothing moves for those
            ops = [(w & ~0xf) | 7 if (w & 0xf) == 0 else w for w in ops]      # (variant.pyx:472-474), and the read has no extent
        elif r < 0.12 and (ops[0] & 0xf) == 0 and (ops[0] >> 4) > 12:  # '=', 'X', then the rest of the match
            ops = [(5 << 4) | 7, (1 << 4) | 8, (((ops[0] >> 4) - 6) << 4) | 0] + ops[1:]
        new.extend(ops)
        new_off.append(len(new))
    out = dict(reads)
    order = np.argsort(pos, kind="stable")                            # soft clips moved a few positions: sort again
    lens = (reads["off"][1:] - reads["off"][:-1])[order]
    out["pos"] = pos[order]
    for k in ("mapq", "flag", "insertSize", "matePos"):
        out[k] = reads[k][order]
    out["seq"] = np.concatenate([reads["seq"][reads["off"][i]:reads["off"][i + 1]] for i in order])
    out["qual"] = np.concatenate([reads["qual"][reads["off"][i]:reads["off"][i + 1]] for i in order])
    out["off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    new = np.array(new, dtype=np.uint32)
    out["cigar"] = np.concatenate([new[new_off[i]:new_off[i + 1]] for i in order])
    out["cigarOff"] = np.concatenate([[0], np.cumsum([new_off[i + 1] - new_off[i] for i in order])]).astype(np.int64)
    return out


def _fields(v):
    return (v.refName, v.refPos, v.removed, v.added, v.nSupportingReads, v.nFwdReads, v.nRevReads, v.minPos, v.maxPos,
            v.nFwdReadsWeighted, v.nRevReadsWeighted, v.rmsReadPos)


def _same_lists(ours, theirs, what):
    """equal as lists up to the order of candidates that tie in (name, position, length change)"""
    assert len(ours) == len(theirs), (what, len(ours), len(theirs))
    key = lambda f: (f[0], f[1], abs(len(f[3]) - len(f[2])))
    a, b = [_fields(v) for v in ours], [_fields(v) for v in theirs]
    assert [key(f) for f in a] == [key(f) for f in b], what
    assert sorted(a) == sorted(b), what


CASES = [dict(), dict(minFlank=10, minBaseQual=10), dict(strandFilter=1, minReads=3), dict(genIndels=0), dict(genSNPs=0, minReads=1),
         dict(maxSize=3, minMapQual=30, maxBadQualBases=8, minGoodQualBases=60)]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_buffer_candidates_and_filter_equal_the_restatement(case):
    rng = np.random.default_rng(900 + case)
    G = 60000
    genome = simulate.random_genome(rng, G, homopolymers=40)
    truth = simulate.plant_variants(rng, genome, 70, spacing=300, indel_frac=0.5)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=14, read_len=100, paired=(case % 2 == 0), insert=(260, 60),
                                    error_rate=0.01)
    reads = _clip_some(rng, _harden(rng, reads, G))
    opts = pipeline.Options(rlen=100, **CASES[case])
    start, end = 1000, G - 1000
    # --- the buffer: which records are kept, and their qualities after trimming
    buf = simulate.buffer_of(reads, start, end, minMapQual=opts.minMapQual, maxReads=int(opts.maxReads),
                             minGoodQualBases=opts.minGoodQualBases, maxBadQualBases=opts.maxBadQualBases)
    kept = co.buffer_reads(co.create_reads(reads), opts.minMapQual, int(opts.maxReads), opts.minGoodQualBases, opts.maxBadQualBases)
    pos, rend, mapq, rev, seqs, quals = buf.readArrays()
    assert len(kept) == len(pos) and 0.5 * len(reads["pos"]) < len(kept) < len(reads["pos"])
    assert pos == [r["pos"] for r in kept] and rend == [r["end"] for r in kept] and mapq == [r["mapq"] for r in kept]
    assert rev == [r["isReverse"] for r in kept] and seqs == [r["seq"] for r in kept]
    assert [bytes(q) for q in quals] == [bytes(r["qual"]) for r in kept]
    # --- raw candidates of all kept reads
    ref = Ref({b"sim": genome})
    fa = FastaFile(sequences={b"sim": genome})
    gen = VariantCandidateGenerator((b"sim", start, end), fa, opts.minMapQual, opts.minFlank, opts.minBaseQual, opts.genSNPs,
                                    opts.genIndels)
    gen.addCandidatesFromBuffer(buf)
    theirs = co.Generator((b"sim", start, end), ref, opts.minMapQual, opts.minFlank, opts.minBaseQual, opts.genSNPs, opts.genIndels)
    for r in kept:
        theirs.read(r)
    raw = theirs.candidates()
    _same_lists(gen.getCandidates(0), raw, "raw candidates")
    assert len(raw) > 300 or not (opts.genSNPs and opts.genIndels)
    # --- standard form, merging, cuts, set, order: the region's candidate list
    ours = pipeline.candidatesInRegion(b"sim", start, end, [buf], fa, opts)
    want = co.candidates_in_region(b"sim", start, end, [kept], ref, opts)
    _same_lists(ours, want, "filtered candidates")
    assert len(want) > 20
    # planted variants with support are among them
    have = {(v.refPos, v.removed, v.added) for v in want}
    wanted = [t for t in truth if (len(t[1]) == len(t[2]) and opts.genSNPs) or (len(t[1]) != len(t[2]) and opts.genIndels)]
    found = sum(1 for (p, rem, add, gt) in wanted if any(abs(h[0] - p) < 40 for h in have))
    assert found > 0.8 * len(wanted)


def test_two_buffers_and_hand_made_cigars():
    """two samples' buffers feed one generator; reads with N, S, H, P operations and indels without flanks"""
    rng = np.random.default_rng(77)
    ref_seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 3000))
    fa, ref = FastaFile(sequences={b"c": ref_seq}), Ref({b"c": ref_seq})
    L = 60
    recs = []
    for k in range(120):
        p = 300 + 15 * k
        kind = k % 6
        if kind == 0:
            cig, span = [(0, 20), (3, 100), (0, 40)], 160              # skipped reference: the reference moves the read too
        elif kind == 1:
            cig, span = [(4, 6), (0, 54)], 54
        elif kind == 2:
            cig, span = [(5, 9), (0, 30), (1, 2), (0, 28)], 58
        elif kind == 3:
            cig, span = [(0, 2), (2, 3), (0, 2), (1, 4), (0, 52)], 59    # indels without a flank of three on one side
        elif kind == 4:
            cig, span = [(0, 25), (6, 4), (0, 35)], 60
        else:
            cig, span = [(0, 30), (2, 5), (0, 30)], 65
        seq = bytearray()
        at = p
        for op, n in cig:
            if op == 0:
                seq += ref_seq[at:at + n]
                at += n
            elif op in (2, 3):
                at += n
            elif op in (1, 4):
                seq += bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
        seq = bytearray(seq[:L].ljust(L, b"A"))
        for j in rng.integers(0, L, 2):
            seq[j] = b"ACGT"[int(rng.integers(0, 4))]
        recs.append((p, bytes(seq), list(rng.integers(18, 40, L)), 0x10 if k % 3 == 0 else 0, cig, 0, -1, 50))
    from tests.test_pipeline_host import _records
    halves = [_records(recs[0::2]), _records(recs[1::2])]
    opts = pipeline.Options(rlen=L, minReads=1)
    bufs = [simulate.buffer_of(h, 0, 3000, minMapQual=20, maxReads=10 ** 6, minGoodQualBases=20, maxBadQualBases=20) for h in halves]
    kept = [co.buffer_reads(co.create_reads(h), 20, 10 ** 6, 20, 20) for h in halves]
    assert [b.readCount for b in bufs] == [len(k) for k in kept]
    ours = pipeline.candidatesInRegion(b"c", 0, 3000, bufs, fa, opts)
    want = co.candidates_in_region(b"c", 0, 3000, kept, ref, opts)
    _same_lists(ours, want, "two buffers")
    assert len(want) > 100
