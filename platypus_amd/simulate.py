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
