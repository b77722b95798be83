"""Parity at full batch size through properties that do not need the CPU checker
to score every pair: results must not depend on how pairs are batched, ordered,
duplicated or translated, and a bounded sample is compared with the checker."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    from platypus_amd import native, synth
    b = synth.make_batch(n_windows=8192, haps_per_window=4, reads_per_window=32, read_len=150,
                         hap_len=750, indel_hap_frac=0.3, n_frac=0.0005, seed=2024)      # 2^20 pairs
    ctx = native.AlignContext(0)
    ctx.upload_batch(b)
    ctx.launch()
    s, a, l = ctx.fetch()
    yield b, ctx, s, a, l
    ctx.close()


def test_sample_against_checker(big):
    from oracle import pyoracle as po
    from platypus_amd import synth
    b, ctx, s, a, l = big
    n = len(s)
    lo = n // 2 - 40000
    es, ea = po.run_pairs(synth.oracle_view(b), threads=8, reuse_hash=True, lo=lo, hi=lo + 80000)
    assert np.array_equal(s[lo:lo + 80000], es) and np.array_equal(a[lo:lo + 80000], ea)
    assert (s >= 0).all() and (s <= 2048).all() and (l > 0).all() and (l <= 1.0).all()


def test_order_and_duplication_do_not_matter(big):
    b, ctx, s, a, l = big
    rng = np.random.default_rng(1)
    n = len(s)
    perm = rng.permutation(n)
    s2, a2, l2 = ctx.align_pairs(b["pair_hap"][perm], b["pair_read"][perm])
    assert np.array_equal(s2, s[perm]) and np.array_equal(a2, a[perm]) and np.array_equal(l2, l[perm])
    pick = rng.integers(0, n, 200000)
    pick = np.concatenate([pick, pick[:50000]])                  # with repeats
    s3, a3, l3 = ctx.align_pairs(b["pair_hap"][pick], b["pair_read"][pick])
    assert np.array_equal(s3, s[pick]) and np.array_equal(a3, a[pick]) and np.array_equal(l3, l[pick])


def test_dense_blocks_equal_pair_lists(big):
    b, ctx, s, a, l = big
    H, R = 4, 32
    for w in (0, 17, 8191):
        ctx.pairs_block(w * H, (w + 1) * H, w * R, (w + 1) * R)
        ctx.launch()
        sb, ab, lb = ctx.fetch()
        sl = slice(w * H * R, (w + 1) * H * R)
        assert np.array_equal(sb, s[sl]) and np.array_equal(ab, a[sl]) and np.array_equal(lb, l[sl])


def test_translation_invariance(big):
    """Shifting a window's coordinates (haplotype start and read positions alike) changes nothing."""
    from platypus_amd import native
    b, _, s, a, l = big
    c2 = native.AlignContext(0)
    try:
        shift = 123457
        c2.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"] + shift, b["hap_varlen"])
        c2.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"] + shift, b["read_mapq"])
        s2, a2, l2 = c2.align_pairs(b["pair_hap"][:300000], b["pair_read"][:300000])
        assert np.array_equal(s2, s[:300000]) and np.array_equal(a2, a[:300000]) and np.array_equal(l2, l[:300000])
    finally:
        c2.close()


def test_perfect_reads_score_zero_and_mapq_bounds_likelihood(big):
    """A read copied from the haplotype at its mapped place scores 0 whatever its qualities;
    the likelihood then is exactly 1, and never below the mapping-error floor otherwise."""
    import math
    from platypus_amd import native
    b, _, _, _, _ = big
    rng = np.random.default_rng(3)
    nh = 2000
    hap_ids = rng.integers(0, len(b["hap_off"]) - 1, nh)
    seqs, quals, pos = [], [], []
    for h in hap_ids:
        o0, o1 = int(b["hap_off"][h]), int(b["hap_off"][h + 1])
        st = int(rng.integers(16, o1 - o0 - 200))
        seqs.append(b["hap_seq"][o0 + st:o0 + st + 150])
        quals.append(rng.integers(0, 94, 150).astype(np.uint8))
        pos.append(int(b["hap_start"][h]) + st)
    c2 = native.AlignContext(0)
    try:
        c2.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], np.zeros_like(b["hap_varlen"]))
        mq = rng.integers(0, 61, nh).astype(np.int32)
        c2.upload_reads(np.concatenate(seqs), np.concatenate(quals), np.arange(nh + 1, dtype=np.int64) * 150,
                        np.array(pos, np.int32), mq)
        s2, a2, l2 = c2.align_pairs(hap_ids.astype(np.int32), np.arange(nh, dtype=np.int32))
        assert (s2 == 0).all() and (l2 == 1.0).all()
        # same reads against a wrong haplotype: likelihood >= P(mapped wrongly)
        s3, _, l3 = c2.align_pairs(((hap_ids + 7) % (len(b["hap_off"]) - 1)).astype(np.int32),
                                   np.arange(nh, dtype=np.int32))
        floor = np.array([math.exp(-0.23025850929940459 * q) for q in mq])
        ok = s3 >= 0
        assert (l3[ok] >= floor[ok]).all()
    finally:
        c2.close()
