"""Parity of the HIP path (through the C ABI) with the CPU checker and with the
golden vectors made from the reference's own compiled routine.  Integer
results must be bit-exact; likelihoods must be the same doubles."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from platypus_amd import native
    c = native.AlignContext(0, 1024)
    yield c
    c.close()


def _oracle(b, **kw):
    from oracle import pyoracle as po
    from platypus_amd import synth
    return po.run_pairs(synth.oracle_view(b), threads=8, reuse_hash=True, **kw)


def _check_batch(ctx, b, exp_score=None, exp_anchor=None):
    from oracle import pyoracle as po
    ctx.upload_batch(b)
    ctx.launch()
    score, anchor, like = ctx.fetch()
    if exp_score is None:
        exp_score, exp_anchor = _oracle(b)
    bad = np.flatnonzero(score != exp_score)
    assert len(bad) == 0, ("score", len(bad), bad[:5], score[bad[:5]], exp_score[bad[:5]])
    bad = np.flatnonzero(anchor != exp_anchor)
    assert len(bad) == 0, ("anchor", len(bad), bad[:5], anchor[bad[:5]], exp_anchor[bad[:5]])
    mq = b["read_mapq"][b["pair_read"]]
    idx = np.linspace(0, len(score) - 1, min(len(score), 4000)).astype(int)
    exp_like = np.array([0.0 if score[i] < 0 else po.likelihood(score[i], mq[i]) for i in idx])
    assert np.array_equal(like[idx], exp_like)      # same doubles, not merely close
    return score, anchor, like


def test_golden_single_calls(ctx, golden_dir):
    """Every golden fastAlignmentRoutine call: seq1 becomes a haplotype of
    length len2+15 and the read is placed so that the slice starts at 0."""
    g = np.load(os.path.join(golden_dir, "align_golden.npz"))
    len2 = g["len2"].astype(np.int64)
    n = len(len2)
    b = dict(hap_seq=g["seq1"], hap_off=np.concatenate([[0], np.cumsum(len2 + 15)]).astype(np.int64),
             hap_start=np.full(n, 500, np.int32), hap_varlen=np.zeros(n, np.int32),
             read_seq=g["seq2"], read_qual=g["qual"],
             read_off=np.concatenate([[0], np.cumsum(len2)]).astype(np.int64),
             read_pos=np.full(n, 500, np.int32), read_mapq=np.full(n, 60, np.int32),
             pair_hap=np.arange(n, dtype=np.int32), pair_read=np.arange(n, dtype=np.int32))
    _check_batch(ctx, b, g["score"], np.zeros(n, np.int32))


def test_golden_windows(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "window_golden.npz"))
    for name in ("wgs", "panel"):
        b = {k.split("__", 1)[1]: g[k] for k in g.files if k.startswith(name + "__")}
        _check_batch(ctx, b, b["score"], b["anchor"])


@pytest.mark.parametrize("kw", [
    dict(n_windows=40, read_len=150, seed=1),
    dict(n_windows=30, read_len=100, indel_hap_frac=0.6, seed=2),
    dict(n_windows=30, read_len=250, hap_len=1100, indel_hap_frac=0.5, seed=3),
    dict(n_windows=30, read_len=150, homopolymers=True, n_frac=0.005, indel_hap_frac=0.9, seed=4),
    dict(n_windows=25, read_len=120, ragged=True, indel_hap_frac=0.5, seed=5),
    dict(n_windows=20, read_len=36, hap_len=300, haps_per_window=7, reads_per_window=50, seed=6),
])
def test_synthetic_windows_vs_checker(ctx, kw):
    from platypus_amd import synth
    _check_batch(ctx, synth.make_batch(**kw))


@pytest.mark.parametrize("kernel", ["bits", "table"])
@pytest.mark.parametrize("kw", [
    dict(n_windows=40, read_len=150, indel_hap_frac=1.0, seed=31),
    # reads the candidate positions cannot decide (a tenth of the bases wrong, every other read with an indel of its
    # own): the exact count over every position
    dict(n_windows=24, read_len=150, indel_hap_frac=1.0, sub_rate=0.1, read_indel_frac=0.5, seed=32),
    dict(n_windows=24, read_len=100, indel_hap_frac=1.0, homopolymers=True, n_frac=0.05, sub_rate=0.1, seed=33),
    dict(n_windows=12, read_len=263, hap_len=1500, indel_hap_frac=1.0, reads_per_window=70, sub_rate=0.05, seed=34),
    dict(n_windows=30, read_len=40, hap_len=200, indel_hap_frac=1.0, ragged=True, sub_rate=0.1, seed=35),
    dict(n_windows=6, read_len=150, haps_per_window=12, reads_per_window=300, indel_hap_frac=1.0, seed=36),
])
def test_both_anchoring_kernels(kernel, kw):
    """the 7-mer anchoring by exact bit-parallel counts (the default for reads of up to 263 bases) and by the
    position table in LDS: the reference's anchors from either"""
    from platypus_amd import native, synth
    c = native.AlignContext(0, 512)
    try:
        c.set_option("vote_kernel", kernel)
        b = synth.make_batch(**kw)
        rng = np.random.default_rng(kw["seed"])
        # some reads moved off their place, a few far off either end of the haplotypes
        nR = len(b["read_pos"])
        shift = np.where(rng.random(nR) < 0.15, rng.integers(-40, 41, nR), 0)
        shift = np.where(rng.random(nR) < 0.03, rng.integers(-kw.get("hap_len", 750), kw.get("hap_len", 750), nR), shift)
        b["read_pos"] = (b["read_pos"].astype(np.int64) + shift).astype(np.int32)
        _check_batch(c, b)
    finally:
        c.close()


@pytest.mark.parametrize("kernel", ["lane", "band"])
@pytest.mark.parametrize("kw", [
    dict(n_windows=40, read_len=150, seed=71),
    dict(n_windows=30, read_len=100, indel_hap_frac=0.6, homopolymers=True, n_frac=0.01, seed=72),
    dict(n_windows=25, read_len=120, ragged=True, indel_hap_frac=0.5, seed=73),
    dict(n_windows=20, read_len=36, hap_len=300, haps_per_window=7, reads_per_window=50, ragged=True, seed=74),
    dict(n_windows=6, read_len=700, hap_len=2500, reads_per_window=20, indel_hap_frac=0.5, homopolymers=True, seed=75),
    dict(n_windows=150, read_len=151, haps_per_window=5, reads_per_window=33, sub_rate=0.05, read_indel_frac=0.4, seed=76),
])
def test_both_alignment_kernels(kernel, kw):
    """the banded DP with one lane per pair (large launches) and with eight lanes per pair (small launches): the
    reference's scores from either, whatever the launch size"""
    from platypus_amd import native, synth
    c = native.AlignContext(0, 1024)
    try:
        c.set_option("align_kernel", kernel)
        b = synth.make_batch(**kw)
        rng = np.random.default_rng(kw["seed"])
        nR = len(b["read_pos"])
        shift = np.where(rng.random(nR) < 0.15, rng.integers(-40, 41, nR), 0)
        shift = np.where(rng.random(nR) < 0.03, rng.integers(-kw.get("hap_len", 750), kw.get("hap_len", 750), nR), shift)
        b["read_pos"] = (b["read_pos"].astype(np.int64) + shift).astype(np.int32)
        _check_batch(c, b)
    finally:
        c.close()


def test_edge_cases(ctx):
    """Short reads, reads running off the haplotype end on both paths, a
    haplotype shorter than a 7-mer, repeated 7-mers only, no votes."""
    rng = np.random.default_rng(9)
    A = np.frombuffer(b"ACGT", np.uint8)
    haps, reads, quals = [], [], []
    hstart, hvar, rpos = [], [], []
    ph, pr = [], []

    def add_hap(s, start, var):
        haps.append(np.frombuffer(s, np.uint8) if isinstance(s, bytes) else s)
        hstart.append(start); hvar.append(var)
        return len(haps) - 1

    def add_read(s, pos, q=30):
        s = np.frombuffer(s, np.uint8) if isinstance(s, bytes) else s
        reads.append(s); quals.append(np.full(len(s), q, np.uint8)); rpos.append(pos)
        return len(reads) - 1

    long_hap = A[rng.integers(0, 4, 400)]
    h0 = add_hap(long_hap, 1000, 0)
    h1 = add_hap(long_hap, 1000, 3)                       # same bases, forced through the vote
    h2 = add_hap(b"ACGTACGTTGCAAGT", 1000, 2)              # 15 bases: shortest indel haplotype taken
    h3 = add_hap(np.tile(np.frombuffer(b"AC", np.uint8), 100), 1000, 1)   # every 7-mer repeated
    h4 = add_hap(b"ACGTACGTACGTACG", 1000, 0)
    for L in (1, 2, 6, 7, 8, 9, 15, 16, 17, 40):
        r = add_read(long_hap[100:100 + L].copy(), 1100)
        for h in (h0, h1, h2, h3, h4):
            ph.append(h); pr.append(r)
    # runs off the end: assertion path (no indel) and N-extension path (indel)
    for st in (330, 360, 385, 392, 399):
        r = add_read(np.resize(long_hap[st:], 60).copy(), 1000 + st)
        for h in (h0, h1):
            ph.append(h); pr.append(r)
    # read far left / far right of the haplotype start
    for pos in (-50, 0, 1003, 1008, 5000, 2_000_000_000):
        r = add_read(long_hap[0:50].copy(), pos)
        for h in (h0, h1, h3):
            ph.append(h); pr.append(r)
    # unrelated read, all-N read, quality extremes
    r = add_read(A[rng.integers(0, 4, 80)], 1100, 93); ph += [h0, h1]; pr += [r, r]
    r = add_read(b"N" * 50, 1100, 0); ph += [h0, h1]; pr += [r, r]
    r = add_read(long_hap[50:130].copy(), 1050, 127); ph += [h0, h1]; pr += [r, r]
    b = dict(hap_seq=np.concatenate(haps),
             hap_off=np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.int64),
             hap_start=np.array(hstart, np.int32), hap_varlen=np.array(hvar, np.int32),
             read_seq=np.concatenate(reads), read_qual=np.concatenate(quals),
             read_off=np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64),
             read_pos=np.array(rpos, np.int32), read_mapq=np.full(len(reads), 37, np.int32),
             pair_hap=np.array(ph, np.int32), pair_read=np.array(pr, np.int32))
    score, anchor, like = _check_batch(ctx, b)
    assert (score == -1).any() and (like[score == -1] == 0.0).all()


def test_custom_parameters(ctx):
    from platypus_amd import synth
    b = synth.make_batch(n_windows=10, read_len=100, homopolymers=True, indel_hap_frac=0.5, seed=21)
    tab = bytes([33 + v for v in (50, 40, 35, 30, 22, 15, 9, 4)])
    try:
        ctx.set_params(5, 1, tab)
        exp = _oracle(b, gapextend=5, nucprior=1, table=tab)
        _check_batch(ctx, b, *exp)
    finally:
        ctx.set_params()


def test_dense_block_and_errors(ctx):
    from platypus_amd import native, synth
    b = synth.make_batch(n_windows=3, reads_per_window=10, read_len=80, hap_len=400, seed=31)
    ctx.upload_batch(b)
    ctx.pairs_block(4, 8, 10, 20)               # window 1: haplotypes 4..7 x reads 10..19
    ctx.launch()
    s, a, _ = ctx.fetch()
    ctx.upload_pairs(b["pair_hap"], b["pair_read"])
    ctx.launch()
    s_all, a_all, _ = ctx.fetch()
    sel = (b["pair_hap"] >= 4) & (b["pair_hap"] < 8)
    assert np.array_equal(s, s_all[sel]) and np.array_equal(a, a_all[sel])
    with pytest.raises(native.PlatGpuError) as e:
        ctx.upload_haps(np.frombuffer(b"ACGTA", np.uint8), [0, 5], [0], [2])
    assert e.value.code == native.PG_ERR_UNSUPPORTED
    ctx.upload_batch(b)
    with pytest.raises(native.PlatGpuError) as e:
        ctx.upload_pairs([0, 99999], [0, 0])
    assert e.value.code == native.PG_ERR_INVALID
    bad_q = b["read_qual"].copy(); bad_q[5] = 200
    with pytest.raises(native.PlatGpuError) as e:
        ctx.upload_reads(b["read_seq"], bad_q, b["read_off"], b["read_pos"], b["read_mapq"])
    assert e.value.code == native.PG_ERR_UNSUPPORTED
    ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
    s0, a0, l0 = ctx.align_pairs(np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert len(s0) == 0 and len(a0) == 0 and len(l0) == 0


def test_every_supported_byte_and_the_refused_ones(ctx):
    """Sequence bytes are compared for equality only (align.c:158-162), whatever they are: IUPAC codes,
    lower case, '=', digits (the read padding is '0', align.c:119-120), control bytes.  Every byte in
    1..31 and 33..126 is taken, in reads and haplotypes, on both anchoring paths; the rest is refused."""
    from platypus_amd import native
    rng = np.random.default_rng(17)
    alphabet = np.array(list(range(1, 32)) + list(range(33, 127)), np.uint8)
    n_h, hl, rl, per = 12, 400, 90, 30
    haps = [alphabet[rng.integers(0, len(alphabet), hl)] for _ in range(n_h)]
    for h in haps[:6]:                                           # half of them mostly ACGTN0 with runs
        h[:] = np.frombuffer(b"ACGTN0acgt=", np.uint8)[rng.integers(0, 11, hl)]
        h[100:130] = ord("0")
        h[200:215] = ord("N")
    reads, quals, pos, ph, pr = [], [], [], [], []
    for h, hp in enumerate(haps):
        for _ in range(per):
            o = int(rng.integers(0, hl - rl - 20))
            s = hp[o:o + rl].copy()
            sub = rng.random(rl) < 0.05
            s[sub] = alphabet[rng.integers(0, len(alphabet), int(sub.sum()))]
            if rng.random() < 0.3:
                s = np.concatenate([s[:40], s[43:]])
            reads.append(s); quals.append(rng.integers(0, 94, len(s)).astype(np.uint8)); pos.append(1000 + o)
            ph.append(h); pr.append(len(reads) - 1)
    b = dict(hap_seq=np.concatenate(haps), hap_off=(np.arange(n_h + 1) * hl).astype(np.int64),
             hap_start=np.full(n_h, 1000, np.int32), hap_varlen=np.array([0, 2] * (n_h // 2), np.int32),
             read_seq=np.concatenate(reads), read_qual=np.concatenate(quals),
             read_off=np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64),
             read_pos=np.array(pos, np.int32), read_mapq=np.full(len(reads), 50, np.int32),
             pair_hap=np.array(ph, np.int32), pair_read=np.array(pr, np.int32))
    _check_batch(ctx, b)
    for bad in (0, 32, 127, 200, 255):
        hs = b["hap_seq"].copy(); hs[7] = bad
        with pytest.raises(native.PlatGpuError) as e:
            ctx.upload_haps(hs, b["hap_off"], b["hap_start"], b["hap_varlen"])
        assert e.value.code == native.PG_ERR_UNSUPPORTED
        rs = b["read_seq"].copy(); rs[11] = bad
        with pytest.raises(native.PlatGpuError) as e:
            ctx.upload_reads(rs, b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
        assert e.value.code == native.PG_ERR_UNSUPPORTED
    with pytest.raises(native.PlatGpuError) as e:                 # run lengths are kept in 7 bits
        ctx.set_params(3, 2, bytes([40] * 129))
    assert e.value.code == native.PG_ERR_INVALID
    ctx.set_params(3, 2, bytes([97 - i // 2 for i in range(128)]))  # the longest table taken
    try:
        b2 = dict(b); b2["hap_seq"] = b["hap_seq"].copy(); b2["hap_seq"][500:700] = ord("A")   # runs up to 127
        from oracle import pyoracle as po
        from platypus_amd import synth
        exp = po.run_pairs(synth.oracle_view(b2), threads=8, reuse_hash=True, gapextend=3, nucprior=2,
                           table=bytes([97 - i // 2 for i in range(128)]))
        _check_batch(ctx, b2, *exp)
    finally:
        ctx.set_params()


def test_genotype_loglik(ctx):
    from oracle import pyoracle as po
    from platypus_amd import synth
    b = synth.make_batch(n_windows=6, haps_per_window=4, reads_per_window=37, read_len=100, seed=41)
    ctx.upload_batch(b)
    ctx.launch()
    _, _, like = ctx.fetch()
    H, R = 4, 37
    b1, b2 = [], []
    for w in range(6):
        for i in range(H):
            for j in range(i, H):
                b1.append((w * H + i) * R); b2.append((w * H + j) * R)
    got = ctx.genotype_loglik(b1, b2, np.full(len(b1), R, np.int32))
    exp = np.array([po.genotype_loglik(like[x:x + R], like[y:y + R]) for x, y in zip(b1, b2)])
    assert np.all(np.abs(got - exp) <= 1e-9 * np.abs(exp) + 1e-300)   # 1e-9 relative


def test_standalone_hash_sequence_and_map_read():
    """pg_hash_sequence / pg_map_read against the restatement of calign.pyx:99-158, bit for bit"""
    from platypus_amd import native
    from oracle import pyoracle as po
    rng = np.random.default_rng(123)
    ctx = native.AlignContext(0)
    try:
        for case in range(12):
            n = [0, 5, 7, 8, 60, 300, 900, 2500, 16383, 700, 700, 700][case]
            seq = bytes(b"ACGTN"[i] for i in rng.choice(5, size=n, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
            if case == 9:
                seq = (b"ACGTTGCA" * 100)[:700]                     # every 7-mer repeated
            tab = ctx.hash_sequence(seq)
            want = np.asarray(po.hash_sequence(seq), dtype=np.int32)
            assert np.array_equal(tab, want), case
            for _ in range(6):
                rl = int(rng.integers(1, 200))
                if n > rl + 20 and rng.random() < 0.8:
                    o = int(rng.integers(0, n - rl))
                    read = bytearray(seq[o:o + rl])
                    for j in np.flatnonzero(rng.random(rl) < 0.03):
                        read[j] = b"ACGT"[int(rng.integers(0, 4))]
                    if rl > 40 and rng.random() < 0.5:
                        del read[20:23]
                    read = bytes(read)
                else:
                    read = bytes(b"ACGT"[i] for i in rng.integers(0, 4, rl))
                assert ctx.map_read(read, n, tab) == po.map_read(read, n, want), (case, rl)
        with pytest.raises(native.PlatGpuError):
            ctx.hash_sequence(b"A" * 16384)
        bad = np.zeros(16384, dtype=np.int32); bad[:] = 5000
        with pytest.raises(native.PlatGpuError):
            ctx.map_read(b"ACGTACGTACGT", 100, bad)
    finally:
        ctx.close()


def test_longest_haplotypes_and_mixed_lengths(ctx):
    """Indel-bearing haplotypes at the upper length limit (16383, calign.pyx:112) beside short ones in one
    batch: the vote kernel's counters need the most LDS there; reads of 20..250 bases; 300 reads per haplotype."""
    from platypus_amd import synth
    rng = np.random.default_rng(77)
    A = np.frombuffer(b"ACGT", np.uint8)
    lens = [16383, 15000, 600, 16383, 90]
    haps = [A[rng.integers(0, 4, n)] for n in lens]
    haps[3][5000:5400] = A[0]                                   # a long homopolymer: many repeated 7-mers
    reads, quals, pos, ph, pr = [], [], [], [], []
    for h, hp in enumerate(haps):
        for _ in range(300):
            rl = int(rng.integers(20, 251))
            rl = min(rl, len(hp) - 20)
            o = int(rng.integers(0, len(hp) - rl))
            s = hp[o:o + rl].copy()
            if rl > 60 and rng.random() < 0.5:                   # an indel relative to the haplotype
                cut = int(rng.integers(20, rl - 20))
                s = np.concatenate([s[:cut], s[cut + int(rng.integers(1, 6)):]]) if rng.random() < 0.5 else \
                    np.concatenate([s[:cut], A[rng.integers(0, 4, int(rng.integers(1, 6)))], s[cut:]])
            sub = rng.random(len(s)) < 0.02
            s[sub] = A[rng.integers(0, 4, int(sub.sum()))]
            reads.append(s); quals.append(rng.integers(2, 42, len(s)).astype(np.uint8)); pos.append(100 + o)
            ph.append(h); pr.append(len(reads) - 1)
    b = dict(hap_seq=np.concatenate(haps), hap_off=np.concatenate([[0], np.cumsum(lens)]).astype(np.int64),
             hap_start=np.full(len(haps), 100, np.int32), hap_varlen=np.array([2, 1, 3, 4, 1], np.int32),
             read_seq=np.concatenate(reads), read_qual=np.concatenate(quals),
             read_off=np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64),
             read_pos=np.array(pos, np.int32), read_mapq=np.full(len(reads), 40, np.int32),
             pair_hap=np.array(ph, np.int32), pair_read=np.array(pr, np.int32))
    _check_batch(ctx, b)


def test_long_reads():
    """reads of 600..4000 bases (a context created for that length): the DP loop, the score range (the
    reference's int16 costs saturate nowhere but add up over thousands of columns) and the vote counters
    at lengths far beyond the usual 100-250"""
    from platypus_amd import native
    rng = np.random.default_rng(91)
    A = np.frombuffer(b"ACGT", np.uint8)
    lens = [9000, 12000, 6000]
    haps = [A[rng.integers(0, 4, n)] for n in lens]
    reads, quals, pos, ph, pr = [], [], [], [], []
    for h, hp in enumerate(haps):
        for _ in range(40):
            rl = int(rng.integers(600, 4001))
            o = int(rng.integers(0, len(hp) - rl - 20))
            s = hp[o:o + rl].copy()
            if rng.random() < 0.5:
                cut = int(rng.integers(100, rl - 100))
                s = np.concatenate([s[:cut], s[cut + int(rng.integers(1, 6)):]]) if rng.random() < 0.5 else \
                    np.concatenate([s[:cut], A[rng.integers(0, 4, int(rng.integers(1, 6)))], s[cut:]])
            sub = rng.random(len(s)) < float(rng.choice([0.0, 0.01, 0.05]))
            s[sub] = A[rng.integers(0, 4, int(sub.sum()))]
            reads.append(s); quals.append(rng.integers(2, 42, len(s)).astype(np.uint8)); pos.append(100 + o)
            ph.append(h); pr.append(len(reads) - 1)
    b = dict(hap_seq=np.concatenate(haps), hap_off=np.concatenate([[0], np.cumsum(lens)]).astype(np.int64),
             hap_start=np.full(len(haps), 100, np.int32), hap_varlen=np.array([0, 2, 0], np.int32),
             read_seq=np.concatenate(reads), read_qual=np.concatenate(quals),
             read_off=np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64),
             read_pos=np.array(pos, np.int32), read_mapq=np.full(len(reads), 40, np.int32),
             pair_hap=np.array(ph, np.int32), pair_read=np.array(pr, np.int32))
    long_ctx = native.AlignContext(0, max_read_len=4096)
    try:
        score, anchor, like = _check_batch(long_ctx, b)
        assert score.max() > 2000                      # far beyond anything a 150-base read can reach
    finally:
        long_ctx.close()
