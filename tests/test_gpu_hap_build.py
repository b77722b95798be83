"""Haplotypes put together on the device from (reference stretch, variant list) -- pg_haps_build -- against the sequences
the reference's constructor makes (platypus/chaplotype.pyx:50-86, getMutatedSequence :283-353), restated in
oracle/vcf_oracle.py (Hap), and against the product's own host constructor; then the alignment results on such a batch
against those of the uploaded sequences."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _variants(rng, lo, hi, n, touching):
    """n sorted variants inside [lo, hi): substitutions, insertions, deletions; with `touching`, neighbours may touch or
    overlap (what Haplotype.valid would refuse is still a sequence the constructor makes)"""
    out, pos = [], lo
    for _ in range(n):
        pos += int(rng.integers(0 if touching else 8, 12))
        if pos >= hi - 12:
            break
        kind = int(rng.integers(0, 4))
        if kind == 0:
            nrem, added = 1, bytes(rng.choice(list(b"ACGT"), 1).tolist())
        elif kind == 1:
            k = int(rng.integers(2, 5))
            nrem, added = k, bytes(rng.choice(list(b"ACGT"), k).tolist())
        elif kind == 2:
            nrem, added = 0, bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 9))).tolist())
        else:
            nrem, added = int(rng.integers(1, 9)), b""
        out.append((pos, nrem, added))
        if not touching:
            pos += max(nrem, 1) + 1
    return out


def _pack(stretches, haps):
    """stretches: [(start, bytes)], haps: [(stretch index, [(pos, nRemoved, added)])] -> arguments of build_haps"""
    ref_off = np.zeros(len(stretches) + 1, dtype=np.int64)
    np.cumsum([len(s) for _, s in stretches], out=ref_off[1:])
    var_off = np.zeros(len(haps) + 1, dtype=np.int32)
    np.cumsum([len(vs) for _, vs in haps], out=var_off[1:])
    flat = [v for _, vs in haps for v in vs]
    added_off = np.zeros(len(flat) + 1, dtype=np.int32)
    np.cumsum([len(v[2]) for v in flat], out=added_off[1:])
    return dict(ref_seq=np.frombuffer(b"".join(s for _, s in stretches), dtype=np.uint8), ref_off=ref_off,
                ref_start=np.array([st for st, _ in stretches], dtype=np.int32),
                hap_stretch=np.array([s for s, _ in haps], dtype=np.int32), hap_var_off=var_off,
                var_pos=np.array([v[0] for v in flat], dtype=np.int32), var_n_removed=np.array([v[1] for v in flat], dtype=np.int32),
                var_added_off=added_off, var_added=np.frombuffer(b"".join(v[2] for v in flat), dtype=np.uint8))


@pytest.mark.parametrize("touching", [False, True])
def test_sequences_equal_the_constructor(touching):
    from oracle import vcf_oracle as vo
    from platypus_amd import chaplotype, native
    from platypus_amd.fastafile import FastaFile
    from platypus_amd.variant import Variant
    rng = np.random.default_rng(77 + touching)
    chrom = bytes(rng.choice(list(b"ACGT"), 6000).tolist())
    ref_o, ref_p = vo.Ref({b"c": chrom}), FastaFile(sequences={b"c": chrom})
    rlen = 100
    stretches, haps, want, want_p, want_start, want_varlen = [], [], [], [], [], []
    for w in range(40):
        ws = int(rng.integers(0, 5600))                        # windows near both ends of the sequence: the clamps of chaplotype.pyx:70-71
        we = ws + int(rng.integers(20, 300))
        if w == 0:
            ws, we = 10, 150
        if w == 1:
            ws, we = 5800, 5990
        proto = vo.Hap(ref_o, b"c", ws, we, (), rlen)
        stretches.append((proto.start, proto.seq))
        for h in range(int(rng.integers(1, 7))):
            vs = [] if h == 0 else _variants(rng, ws, min(we, len(chrom) - 20), int(rng.integers(1, 6)), touching)
            ovs = [vo.V(refName=b"c", refPos=p, removed=chrom[p + (0 if len(a) == nr else 1):p + (0 if len(a) == nr else 1) + nr], added=a, nAdded=len(a), nRemoved=nr)
                   for p, nr, a in vs]
            oh = vo.Hap(ref_o, b"c", ws, we, ovs, rlen)
            if oh.varSeqLen != 0 and len(oh.seq) < 15:
                continue
            haps.append((w, vs))
            want.append(oh.seq)
            want_start.append(oh.start)
            want_varlen.append(oh.varSeqLen)
            pvs = tuple(Variant(b"c", v.refPos, v.removed, v.added) for v in ovs)
            want_p.append(chaplotype.Haplotype(b"c", ws, we, pvs, ref_p, rlen).sequence)
    assert want == want_p                                      # the product's host constructor agrees with the restatement
    ctx = native.AlignContext(0)
    try:
        ctx.build_haps(**_pack(stretches, haps))
        seq, off = ctx.fetch_haps()
        got = [seq[off[i]:off[i + 1]].tobytes() for i in range(len(haps))]
        assert got == want
        # start positions and varSeqLen arrive where the alignment kernels read them: the same batch uploaded as
        # sequences gives the same results for a few reads
        reads = []
        for i, s in enumerate(want[:24]):
            o = int(rng.integers(0, max(1, len(s) - 130)))
            reads.append((i, o, s[o:o + 100]))
        reads = [r for r in reads if len(r[2]) == 100 and len(want[r[0]]) >= 140]
        rs = np.frombuffer(b"".join(r[2] for r in reads), dtype=np.uint8)
        rq = np.full(len(rs), 30, dtype=np.uint8)
        ro = np.arange(len(reads) + 1, dtype=np.int64) * 100
        rp = np.array([want_start[i] + o for i, o, _ in reads], dtype=np.int32)
        mq = np.full(len(reads), 60, dtype=np.int32)
        ph = np.array([i for i, _, _ in reads], dtype=np.int32)
        pr = np.arange(len(reads), dtype=np.int32)
        ctx.upload_reads(rs, rq, ro, rp, mq)
        built = ctx.align_pairs(ph, pr)
        hoff = np.zeros(len(want) + 1, dtype=np.int64)
        np.cumsum([len(s) for s in want], out=hoff[1:])
        ctx.upload_haps(np.frombuffer(b"".join(want), dtype=np.uint8), hoff, np.array(want_start, np.int32), np.array(want_varlen, np.int32))
        ctx.upload_reads(rs, rq, ro, rp, mq)
        uploaded = ctx.align_pairs(ph, pr)
        for a, b in zip(built, uploaded):
            assert np.array_equal(a, b)
    finally:
        ctx.close()


@pytest.mark.parametrize("homopolymers", [False, True])
def test_synthetic_batch_built_equals_uploaded(homopolymers):
    from platypus_amd import native, synth
    b = synth.make_batch(300, haps_per_window=5, reads_per_window=12, homopolymers=homopolymers, indel_hap_frac=0.5, seed=5)
    ctx = native.AlignContext(0)
    try:
        ctx.build_haps(**b["built"])
        seq, off = ctx.fetch_haps()
        assert np.array_equal(off, b["hap_off"]) and np.array_equal(seq, b["hap_seq"])
        ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
        built = ctx.align_pairs(b["pair_hap"], b["pair_read"])
        ctx.upload_batch(b)
        ctx.launch()
        uploaded = ctx.fetch()
        for x, y in zip(built, uploaded):
            assert np.array_equal(x, y)
    finally:
        ctx.close()


def test_refusals():
    from platypus_amd import native
    ctx = native.AlignContext(0)
    stretch = [(100, b"ACGT" * 50)]
    try:
        for vs, code in (([(99, 1, b"A")], native.PG_ERR_INVALID),            # before the stretch
                         ([(300, 1, b"A")], native.PG_ERR_INVALID),           # beyond it
                         ([(300, 0, b"AA")], native.PG_ERR_INVALID),          # the base an insertion follows is the first one outside
                         ([(295, 5, b"")], native.PG_ERR_INVALID),            # a deletion whose last base is that one
                         ([(150, 2, b"A")], native.PG_ERR_UNSUPPORTED)):      # chaplotype.pyx:340-342
            with pytest.raises(native.PlatGpuError) as e:
                ctx.build_haps(**_pack(stretch, [(0, vs)]))
            assert e.value.code == code
        with pytest.raises(native.PlatGpuError):
            ctx.fetch_haps()                                   # nothing on the device after a refusal
        ctx.build_haps(**_pack(stretch, [(0, []), (0, [(298, 1, b"T")]), (0, [(100, 0, b"GG")])]))
        seq, off = ctx.fetch_haps()
        s = stretch[0][1]
        assert seq[off[0]:off[1]].tobytes() == s
        assert seq[off[1]:off[2]].tobytes() == s[:198] + b"T" + s[199:]
        assert seq[off[2]:off[3]].tobytes() == s[:1] + b"GG" + s[1:]
    finally:
        ctx.close()
