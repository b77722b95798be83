"""pg_reads_upload_nibbles: reads whose bases come as a BAM record holds them (bam1_seq: two per byte, the 4-bit code of
bam_nt16_rev_table, platypus/c/samtools/bam.h) give the results of the same reads uploaded as characters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _results(ctx, b, nibbles, lead=0):
    """lead: bases of other reads in front of the batch's first one in the caller's arrays (offsets start there)"""
    from platypus_amd import native
    rng = np.random.default_rng(3)
    junk = np.frombuffer(native.NIBBLE_CODES, dtype=np.uint8)[rng.integers(0, 16, size=lead)]
    seq = np.concatenate([junk, b["read_seq"]])
    qual = np.concatenate([np.zeros(lead, dtype=np.uint8), b["read_qual"]])
    off = b["read_off"] + lead
    ctx.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
    if nibbles:
        ctx.upload_reads_nibbles(native.pack_nibbles(seq), qual, off, b["read_pos"], b["read_mapq"])
    else:
        ctx.upload_reads(seq, qual, off, b["read_pos"], b["read_mapq"])
    return ctx.align_pairs(b["pair_hap"], b["pair_read"])


@pytest.mark.parametrize("lead", [0, 1, 6])
def test_synthetic_batch(lead):
    from platypus_amd import native, synth
    b = synth.make_batch(200, haps_per_window=4, reads_per_window=16, n_frac=0.01, ragged=True, indel_hap_frac=0.5, seed=11 + lead)
    ctx = native.AlignContext(0)
    try:
        want = _results(ctx, b, False, lead)
        got = _results(ctx, b, True, lead)
        for x, y in zip(got, want):
            assert np.array_equal(x, y)
    finally:
        ctx.close()


def test_every_code_and_odd_lengths():
    """reads over all sixteen codes ('=' and the ambiguity letters included), odd lengths, an odd number of bases in all"""
    from platypus_amd import native
    rng = np.random.default_rng(5)
    letters = np.frombuffer(native.NIBBLE_CODES, dtype=np.uint8)
    hap = letters[rng.integers(0, 16, size=600)]
    lens = rng.integers(21, 120, size=41) | 1
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    assert off[-1] & 1
    starts = rng.integers(0, 400, size=len(lens))
    seq = np.concatenate([hap[s:s + n] for s, n in zip(starts, lens)]).copy()
    flip = rng.random(len(seq)) < 0.03
    seq[flip] = letters[rng.integers(0, 16, size=int(flip.sum()))]
    qual = rng.integers(0, 60, size=len(seq)).astype(np.uint8)
    pos = (1000 + starts).astype(np.int32)
    mapq = rng.integers(0, 61, size=len(lens)).astype(np.int32)
    ctx = native.AlignContext(0)
    try:
        out = []
        for nibbles in (False, True):
            ctx.upload_haps(hap, np.array([0, len(hap)], dtype=np.int64), [1000], [0])
            if nibbles:
                ctx.upload_reads_nibbles(native.pack_nibbles(seq), qual, off, pos, mapq)
            else:
                ctx.upload_reads(seq, qual, off, pos, mapq)
            out.append(ctx.align_pairs(np.zeros(len(lens), dtype=np.int32), np.arange(len(lens), dtype=np.int32)))
        for x, y in zip(*out):
            assert np.array_equal(x, y)
    finally:
        ctx.close()


def test_same_refusals():
    from platypus_amd import native
    ctx = native.AlignContext(0, 64)
    try:
        seq = np.frombuffer(b"ACGT" * 40, dtype=np.uint8)
        with pytest.raises(native.PlatGpuError) as e:              # a read longer than the context takes
            ctx.upload_reads_nibbles(native.pack_nibbles(seq), np.zeros(160, np.uint8), [0, 100, 160], [5, 6], [60, 60])
        assert e.value.code == native.PG_ERR_UNSUPPORTED
        with pytest.raises(native.PlatGpuError) as e:              # a quality above 127
            ctx.upload_reads_nibbles(native.pack_nibbles(seq[:60]), np.full(60, 200, np.uint8), [0, 60], [5], [60])
        assert e.value.code == native.PG_ERR_UNSUPPORTED
        with pytest.raises(ValueError):
            native.pack_nibbles(np.frombuffer(b"ACGU", dtype=np.uint8))
    finally:
        ctx.close()
