"""A small diploid read simulator for tests and end-to-end timing: a random reference,
planted SNPs / insertions / deletions (heterozygous or homozygous), single-end reads drawn from
the two mutated copies with their true alignment (position + CIGAR), substitution errors and
per-base qualities.  Output is the packed record layout `bamReadBuffer.fromAlignments` takes."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_genome(rng, length, homopolymers=0):
    g = ACGT[rng.integers(0, 4, length)].copy()
    for _ in range(homopolymers):
        p = int(rng.integers(200, length - 200))
        g[p:p + int(rng.integers(5, 15))] = ACGT[rng.integers(0, 4)]
    return g.tobytes()


def plant_variants(rng, genome, n, margin=300, spacing=120, indel_frac=0.3):
    """n variants at least `spacing` apart: (pos, removed, added, genotype) with the package's
    conventions (SNP at its base; insertion after the base at pos; deletion of the bases after pos),
    genotype 1 = heterozygous, 2 = homozygous."""
    out, pos = [], margin
    while len(out) < n and pos < len(genome) - margin:
        pos += int(rng.integers(spacing, 2 * spacing))
        if pos >= len(genome) - margin:
            break
        gt = 1 if rng.random() < 0.6 else 2
        if rng.random() < indel_frac:
            k = int(rng.integers(1, 5))
            if rng.random() < 0.5:
                ins = ACGT[rng.integers(0, 4, k)].tobytes()
                if ins[0:1] == genome[pos + 1:pos + 2] or ins[-1:] == genome[pos:pos + 1]:
                    continue                              # would not be left-aligned: keep the test's truth simple
                out.append((pos, b"", ins, gt))
            else:
                rem = genome[pos + 1:pos + 1 + k]
                if genome[pos:pos + 1] == rem[-1:] or genome[pos + 1 + k:pos + 2 + k] == rem[0:1]:
                    continue
                out.append((pos, rem, b"", gt))
        else:
            ref = genome[pos:pos + 1]
            alt = [b for b in b"ACGT" if b != ref[0]][int(rng.integers(0, 3))]
            out.append((pos, ref, bytes([alt]), gt))
    return out


def _copy(genome, variants, which):
    """mutated copy + segments (ref_start, hap_start, length, op) with op in 'M', 'I', 'D'"""
    seq, segs, rp = [], [], 0
    hp = 0
    for pos, rem, add, gt in variants:
        if gt == 1 and which == 1:
            continue
        if len(rem) == len(add):                          # SNP
            n = pos - rp
            if n > 0:
                segs.append((rp, hp, n, "M")); seq.append(genome[rp:pos]); hp += n
            segs.append((pos, hp, len(add), "M")); seq.append(add); hp += len(add)
            rp = pos + len(rem)
        else:
            n = pos + 1 - rp
            segs.append((rp, hp, n, "M")); seq.append(genome[rp:pos + 1]); hp += n
            rp = pos + 1
            if add:
                segs.append((rp, hp, len(add), "I")); seq.append(add); hp += len(add)
            else:
                segs.append((rp, hp, len(rem), "D")); rp += len(rem)
    n = len(genome) - rp
    segs.append((rp, hp, n, "M")); seq.append(genome[rp:]); hp += n
    return b"".join(seq), segs


def _place(seq, segs, seg_starts, hs, he):
    """alignment of haplotype interval [hs, he): (reference position, merged CIGAR as (op, n) list) or None"""
    cigar, pos = [], None
    first = max(0, int(np.searchsorted(seg_starts, hs, side="right")) - 2)
    for rs, hps, n, op in segs[first:]:
        if hps >= he:
            break
        if op == "D":
            if pos is not None and hps > hs and hps < he:
                cigar.append((2, n))
            continue
        lo, hi = max(hs, hps), min(he, hps + n)
        if lo >= hi:
            continue
        if op == "M":
            if pos is None:
                pos = rs + (lo - hps)
            cigar.append((0, hi - lo))
        else:
            if pos is None:                            # read starts inside an insertion: soft clip it
                cigar.append((4, hi - lo))
            else:
                cigar.append((1, hi - lo))
    if pos is None:
        return None
    while cigar and cigar[-1][0] in (1, 2):             # an alignment does not end with an indel
        op, n = cigar.pop()
        if op == 1:
            cigar.append((4, n))
            break
    merged = []
    for op, n in cigar:
        if merged and merged[-1][0] == op:
            merged[-1] = (op, merged[-1][1] + n)
        else:
            merged.append((op, n))
    return pos, merged


def simulate_reads(rng, genome, variants, coverage=30, read_len=100, error_rate=0.005, chrom=b"sim",
                   error_qual=(5, 25), paired=False, insert=(500, 50)):
    """dict of packed record arrays, sorted by position.  paired=True draws fragments of
    insert[0] +- insert[1] bases and gives the forward read of their left end and the reverse-strand
    read of their right end (flags, mate position and template length as an aligner writes them)."""
    copies = [_copy(genome, variants, w) for w in (0, 1)]
    starts = [np.array([sg[1] for sg in segs]) for _, segs in copies]     # haplotype start of every segment
    n_reads = int(coverage * len(genome) / read_len)
    recs = []

    def bases(seq, hs, he):
        s = np.frombuffer(seq[hs:he], dtype=np.uint8).copy()
        q = rng.integers(25, 41, he - hs).astype(np.uint8)
        err = rng.random(he - hs) < error_rate
        s[err] = ACGT[rng.integers(0, 4, int(err.sum()))]
        q[err] = rng.integers(error_qual[0], error_qual[1], int(err.sum()))
        return s, q

    if not paired:
        which = rng.integers(0, 2, n_reads)
        for k in range(n_reads):
            seq, segs = copies[which[k]]
            hs = int(rng.integers(0, len(seq) - read_len))
            placed = _place(seq, segs, starts[which[k]], hs, hs + read_len)
            if placed is None:
                continue
            s, q = bases(seq, hs, hs + read_len)
            flag = 0x10 if rng.random() < 0.5 else 0
            recs.append((placed[0], s, q, flag, placed[1], -1, 0))
    else:
        which = rng.integers(0, 2, n_reads // 2)
        for k in range(n_reads // 2):
            seq, segs = copies[which[k]]
            flen = max(read_len, int(round(rng.normal(insert[0], insert[1]))))
            fs = int(rng.integers(0, len(seq) - flen))
            left = _place(seq, segs, starts[which[k]], fs, fs + read_len)
            right = _place(seq, segs, starts[which[k]], fs + flen - read_len, fs + flen)
            if left is None or right is None:
                continue
            right_end = right[0] + sum(n for op, n in right[1] if op in (0, 2))
            tlen = right_end - left[0]
            first_is_left = rng.random() < 0.5            # which of the two is read one
            s, q = bases(seq, fs, fs + read_len)
            recs.append((left[0], s, q, 0x1 | 0x2 | 0x20 | (0x40 if first_is_left else 0x80), left[1], right[0], tlen))
            s, q = bases(seq, fs + flen - read_len, fs + flen)
            recs.append((right[0], s, q, 0x1 | 0x2 | 0x10 | (0x80 if first_is_left else 0x40), right[1], left[0], -tlen))
    recs.sort(key=lambda r: r[0])
    off = np.concatenate([[0], np.cumsum([len(r[1]) for r in recs])]).astype(np.int64)
    cig = [((n << 4) | op) for r in recs for op, n in r[4]]
    coff = np.concatenate([[0], np.cumsum([len(r[4]) for r in recs])]).astype(np.int64)
    return dict(chrom=chrom, seq=np.concatenate([r[1] for r in recs]), qual=np.concatenate([r[2] for r in recs]),
                off=off, pos=np.array([r[0] for r in recs], dtype=np.int32),
                mapq=np.full(len(recs), 60, dtype=np.int32), flag=np.array([r[3] for r in recs], dtype=np.int32),
                cigar=np.array(cig, dtype=np.uint32), cigarOff=coff,
                matePos=np.array([r[5] for r in recs], dtype=np.int32),
                insertSize=np.array([r[6] for r in recs], dtype=np.int32))


def buffer_of(reads, start, end, **kw):
    from platypus_amd import cwindow
    return cwindow.bamReadBuffer.fromAlignments(reads["chrom"], start, end, reads["seq"], reads["qual"], reads["off"],
                                                reads["pos"], reads["mapq"], reads["flag"], reads["cigar"],
                                                reads["cigarOff"], insertSize=reads.get("insertSize"),
                                                matePos=reads.get("matePos"), **kw)


def tile(genome, variants, reads, copies):
    """`copies` repeats of one simulated stretch laid end to end as ONE longer reference: the
    sequence repeated, every planted variant and every read shifted by multiples of its length.
    Simulating is far slower than calling, so long references for timing runs are built this way;
    windows are local, so the repeats do not interact."""
    G, n = len(genome), len(reads["pos"])
    out = dict(chrom=reads["chrom"])
    for k in ("seq", "qual", "mapq", "flag", "cigar", "insertSize"):
        if k in reads:
            out[k] = np.tile(reads[k], copies)
    if "matePos" in reads:
        mp = reads["matePos"].astype(np.int64)
        out["matePos"] = np.concatenate([np.where(mp >= 0, mp + c * G, mp) for c in range(copies)]).astype(np.int32)
    out["pos"] = np.concatenate([reads["pos"].astype(np.int64) + c * G for c in range(copies)]).astype(np.int32)
    for k, base in (("off", "seq"), ("cigarOff", "cigar")):
        step = len(reads[base])
        out[k] = np.concatenate([reads[k][:-1] + c * step for c in range(copies)] + [np.array([copies * step])]).astype(np.int64)
    truth = [(p + c * G, rem, add, gt) for c in range(copies) for p, rem, add, gt in variants]
    return genome * copies, truth, out


# ---------------------------------------------------------------------------------------------
# Whole-chromosome simulation (end-to-end timing runs): the loops over reads are compiled
# (platypus_amd/csimulate.pyx), everything else is array arithmetic.
def chr_like_genome(rng, length, gc=0.44, homopolymer_every=4000, repeat_every=15000):
    """A chromosome-like sequence: GC content `gc`, a homopolymer run of 8-20 bases every
    `homopolymer_every` bases on average and a di- or trinucleotide repeat of 6-15 units every
    `repeat_every` (the low-complexity sequence real chromosomes hold and a uniform draw lacks)."""
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    g = ACGT[rng.choice(4, size=length, p=p)]
    n_h = max(0, length // homopolymer_every)
    at = rng.integers(300, max(301, length - 300), n_h)
    ln = rng.integers(8, 21, n_h)
    base = ACGT[rng.integers(0, 4, n_h)]
    for a, l, b in zip(at.tolist(), ln.tolist(), base.tolist()):
        g[a:a + l] = b
    n_r = max(0, length // repeat_every)
    at = rng.integers(300, max(301, length - 300), n_r)
    for a in at.tolist():
        unit = ACGT[rng.integers(0, 4, int(rng.integers(2, 4)))]
        if len(set(unit.tolist())) == 1:
            continue
        rep = np.tile(unit, int(rng.integers(6, 16)))
        g[a:a + len(rep)] = rep[:max(0, min(len(rep), length - 300 - a))] if a + len(rep) > length - 300 else rep
    return g.tobytes()


def left_align(genome, pos, rem, add):
    """the left-most equivalent of an insertion after / deletion behind `pos` (SNPs are returned as they are)"""
    if len(rem) == len(add):
        return pos, rem, add
    if add:
        while pos > 0 and add[-1:] == genome[pos:pos + 1]:
            add = genome[pos:pos + 1] + add[:-1]
            pos -= 1
        return pos, rem, add
    while pos > 0 and rem[-1:] == genome[pos:pos + 1]:
        rem = genome[pos:pos + 1] + rem[:-1]
        pos -= 1
    return pos, rem, add


def plant_variants_rate(rng, genome, rate=0.001, indel_frac=0.15, indel_extend=0.3, margin=300, het_frac=0.6667):
    """Variants at exponential gaps of mean 1 / rate (wgsim's parameters: mutation rate 0.001, 15 % indels whose
    length grows with probability 0.3 per base), each (pos, removed, added, genotype) in the package's
    conventions and in left-aligned form; an indel is kept at least its own length + 12 bases from the previous variant."""
    out, pos, last_end = [], margin, 0
    n = len(genome)
    while True:
        pos += 1 + int(rng.geometric(rate))
        if pos >= n - margin:
            break
        gt = 1 if rng.random() < het_frac else 2
        if rng.random() < indel_frac:
            k = 1
            while k < 30 and rng.random() < indel_extend:
                k += 1
            if rng.random() < 0.5:
                p, rem, add = left_align(genome, pos, b"", ACGT[rng.integers(0, 4, k)].tobytes())
            else:
                p, rem, add = left_align(genome, pos, genome[pos + 1:pos + 1 + k], b"")
            if p <= last_end + 12:
                continue
            out.append((p, rem, add, gt))
            last_end = max(pos, p) + k + 1
            pos = last_end
        else:
            if pos <= last_end + 1:
                continue
            ref = genome[pos:pos + 1]
            alt = [b for b in b"ACGT" if b != ref[0]][int(rng.integers(0, 3))]
            out.append((pos, ref, bytes([alt]), gt))
            last_end = pos
    return out


def _runs(seq, min_run=6):
    """(start, length) of the runs of at least min_run equal bases of a sequence"""
    a = np.frombuffer(seq, dtype=np.uint8)
    change = np.flatnonzero(a[1:] != a[:-1]) + 1
    starts = np.concatenate([[0], change])
    lens = np.diff(np.concatenate([starts, [len(a)]]))
    keep = lens >= min_run
    return starts[keep].astype(np.int64), lens[keep].astype(np.int32)


def simulate_pairs(seed, genome, variants, coverage=30, read_len=150, insert=(500, 50), chrom=b"sim",
                   fragments=None, run_err_scale=2e-5, threads=None):
    """Read pairs of a diploid sample (packed record arrays, sorted by position) with qualities that mean what they
    say, errors where the qualities are low and indel errors in homopolymer runs (csimulate.fill / place).
    `fragments`: (reference start, length) arrays to draw the fragments from instead of uniform positions (panels)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from platypus_amd import csimulate
    rng = np.random.default_rng(seed)
    L = read_len
    copies = [_copy(genome, variants, w) for w in (0, 1)]
    if fragments is None:
        n_frag = int(coverage * len(genome) / (2 * L))
        flen = np.maximum(L, np.rint(rng.normal(insert[0], insert[1], n_frag))).astype(np.int32)
        fstart = None
    else:
        fstart, flen = np.asarray(fragments[0], dtype=np.int64), np.asarray(fragments[1], dtype=np.int32)
        n_frag = len(fstart)
    which = rng.integers(0, 2, n_frag).astype(np.uint8)
    first_is_left = rng.random(n_frag) < 0.5
    frag_id = np.arange(n_frag, dtype=np.int64)
    parts = []
    haps = []
    for c in (0, 1):
        seq, segs = copies[c]
        haps.append(np.frombuffer(seq, dtype=np.uint8))
        seg_ref = np.array([s[0] for s in segs], dtype=np.int64)
        seg_hap = np.array([s[1] for s in segs], dtype=np.int64)
        seg_len = np.array([s[2] for s in segs], dtype=np.int64)
        seg_op = np.array([{"M": 0, "I": 1, "D": 2}[s[3]] for s in segs], dtype=np.uint8)
        mine = np.flatnonzero(which == c)
        if fstart is None:
            fs = (rng.random(len(mine)) * (len(seq) - flen[mine] - 4)).astype(np.int64)
        else:                                              # reference position -> position in this copy
            m = seg_op == 0
            mref, mhap = seg_ref[m], seg_hap[m]
            at = np.maximum(0, np.searchsorted(mref, fstart[mine], side="right") - 1)
            fs = np.clip(mhap[at] + (fstart[mine] - mref[at]), 0, len(seq) - flen[mine] - 4)
        rs, rl = _runs(seq)
        got = csimulate.place(seed, c, seg_ref, seg_hap, seg_len, seg_op, rs, rl, fs,
                              np.ascontiguousarray(flen[mine]), np.ascontiguousarray(frag_id[mine]), L, run_err_scale)
        got["frag"] = np.repeat(mine, 2)
        got["which"] = np.full(2 * len(mine), c, dtype=np.uint8)
        parts.append(got)
    cat = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    n = len(cat["pos"])
    ok = cat["pos"] >= 0
    ok = ok & np.repeat(ok[0::2] & ok[1::2], 2)             # both reads of a pair or neither
    side = np.tile(np.array([0, 1], dtype=np.uint8), n // 2)
    mate = np.arange(n) ^ 1
    mate_pos = cat["pos"][mate]
    tlen = np.repeat(cat["end"][1::2] - cat["pos"][0::2], 2)
    tlen = np.where(side == 0, tlen, -tlen)
    fil = first_is_left[cat["frag"]]
    flag = np.where(side == 0, 0x1 | 0x2 | 0x20 | np.where(fil, 0x40, 0x80), 0x1 | 0x2 | 0x10 | np.where(fil, 0x80, 0x40))
    keep = np.flatnonzero(ok)
    order = keep[np.argsort(cat["pos"][keep], kind="stable")]
    m = len(order)
    seq = np.empty(m * L, dtype=np.uint8)
    qual = np.empty(m * L, dtype=np.uint8)
    hs = np.ascontiguousarray(cat["hs"][order]); hlen = np.ascontiguousarray(cat["hlen"][order])
    ek = np.ascontiguousarray(cat["ekind"][order]); eo = np.ascontiguousarray(cat["eoff"][order])
    wh = np.ascontiguousarray(cat["which"][order]); rev = np.ascontiguousarray(side[order])
    rid = np.ascontiguousarray((2 * cat["frag"] + side)[order].astype(np.int64))
    threads = threads or min(16, os.cpu_count() or 1)
    cuts = [m * t // threads for t in range(threads + 1)]
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda t: csimulate.fill(seed, haps[0], haps[1], wh, hs, hlen, ek, eo, rev, rid, L, seq, qual,
                                               cuts[t], cuts[t + 1]), range(threads)))
    nc = cat["ncig"][order]
    cig = cat["cigar"][order]
    cigar = cig[np.arange(cig.shape[1])[None, :] < nc[:, None]]
    return dict(chrom=chrom, seq=seq, qual=qual, off=np.arange(m + 1, dtype=np.int64) * L,
                pos=cat["pos"][order].astype(np.int32), mapq=np.full(m, 60, dtype=np.int32),
                flag=flag[order].astype(np.int32), cigar=cigar.astype(np.uint32),
                cigarOff=np.concatenate([[0], np.cumsum(nc, dtype=np.int64)]),
                matePos=mate_pos[order].astype(np.int32), insertSize=tlen[order].astype(np.int32))
