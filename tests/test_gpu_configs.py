"""The two benchmark configurations other than single-sample WGS, at their own shapes, through the C ABI:

* the indel-heavy amplicon panel (every variant haplotype carries an indel, long homopolymer runs, 'N's):
  2^20 pairs, so that the 'N'-aware DP kernel and all-vote batches run at full size -- a bounded sample against
  the CPU checker plus size-independent properties;
* joint calling of 50 individuals with large haplotype sets: the 257 haplotype x 550 read alignment block,
  and windows of 50 individuals x >= 100 haplotypes through genotype likelihoods, EM and posteriors against
  the restatement of platypus/cpopulation.pyx:658-890,515-632.

Bars: scores and anchors bit-exact, likelihoods the same doubles; caller doubles 1e-9 relative, equal iteration
counts and calls."""
import numpy as np
import pytest

from platypus_amd import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope="module")
def panel():
    from platypus_amd import native
    # 1366 windows x 6 haplotypes x 128 reads = 1 049 088 pairs
    b = synth.make_batch(n_windows=1366, haps_per_window=6, reads_per_window=128, read_len=150, hap_len=750,
                         indel_hap_frac=1.0, homopolymers=True, n_frac=0.005, seed=404)
    ctx = native.AlignContext(0)
    ctx.upload_batch(b)
    ctx.launch()
    s, a, l = ctx.fetch()
    yield b, ctx, s, a, l
    ctx.close()


def test_panel_sample_against_checker(panel):
    from oracle import pyoracle as po
    b, ctx, s, a, l = panel
    n = len(s)
    assert n >= 1 << 20
    for lo in (0, n // 2 - 20000, n - 40000):
        es, ea = po.run_pairs(synth.oracle_view(b), threads=8, reuse_hash=True, lo=lo, hi=lo + 40000)
        assert np.array_equal(s[lo:lo + 40000], es) and np.array_equal(a[lo:lo + 40000], ea)
    mq = b["read_mapq"][b["pair_read"]]
    idx = np.linspace(0, n - 1, 3000).astype(int)
    assert np.array_equal(l[idx], np.array([po.likelihood(s[i], mq[i]) for i in idx]))
    assert (b["hap_varlen"][b["pair_hap"]] != 0).mean() > 0.8          # nearly every pair goes through the vote


def test_panel_order_duplication_and_blocks(panel):
    b, ctx, s, a, l = panel
    rng = np.random.default_rng(2)
    n = len(s)
    perm = rng.permutation(n)
    s2, a2, l2 = ctx.align_pairs(b["pair_hap"][perm], b["pair_read"][perm])
    assert np.array_equal(s2, s[perm]) and np.array_equal(a2, a[perm]) and np.array_equal(l2, l[perm])
    H, R = 6, 128
    ws = np.array([0, 3, 700, 1365])
    base = ctx.pairs_blocks(ws * H, [H] * 4, ws * R, [R] * 4)
    ctx.launch()
    sb, ab, lb = ctx.fetch()
    for k, w in enumerate(ws):
        sl, bl = slice(w * H * R, (w + 1) * H * R), slice(int(base[k]), int(base[k]) + H * R)
        assert np.array_equal(sb[bl], s[sl]) and np.array_equal(ab[bl], a[sl]) and np.array_equal(lb[bl], l[sl])


def test_joint_alignment_block_257_by_550():
    """one window of 257 haplotypes x 550 reads (50 individuals' 11 reads each) as a dense block"""
    from oracle import pyoracle as po
    from platypus_amd import native
    b = synth.make_batch(n_windows=2, haps_per_window=257, reads_per_window=550, read_len=150, hap_len=750,
                         indel_hap_frac=0.25, seed=505)
    ctx = native.AlignContext(0)
    try:
        ctx.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
        ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
        ctx.pairs_block(257, 514, 550, 1100)                                  # the second window
        ctx.launch()
        s, a, l = ctx.fetch()
        n = 257 * 550
        assert len(s) == n
        es, ea = po.run_pairs(synth.oracle_view(b), threads=8, reuse_hash=True, lo=n, hi=2 * n)
        assert np.array_equal(s, es) and np.array_equal(a, ea)
        mq = b["read_mapq"][b["pair_read"][n:]]
        idx = np.linspace(0, n - 1, 3000).astype(int)
        assert np.array_equal(l[idx], np.array([po.likelihood(s[i], mq[i]) for i in idx]))
    finally:
        ctx.close()


@pytest.mark.parametrize("H,n_windows", [(100, 3), (257, 1)])
def test_joint_caller_50_individuals(H, n_windows):
    """50 individuals x 11 reads each, 100 and 257 haplotypes (5050 / 33 153 genotypes per individual):
    genotype likelihoods, EM and calls against the CPU restatement"""
    from oracle import pyoracle as po
    from platypus_amd import native
    n_ind, R = 50, 550
    b = synth.make_batch(n_windows=n_windows, haps_per_window=H, reads_per_window=R, read_len=100, hap_len=500,
                         indel_hap_frac=0.25, seed=600 + H)
    off = (np.arange(n_ind + 1) * (R // n_ind)).astype(np.int32)
    ctx = native.AlignContext(0)
    try:
        ctx.upload_batch(b)
        ctx.launch()
        ctx.upload_windows([H] * n_windows, [n_ind] * n_windows, np.arange(n_windows, dtype=np.int64) * H * R,
                           np.tile(off, n_windows))
        ctx.caller_launch(100)
        res = ctx.caller_fetch()
        _, _, like = ctx.fetch()
        masks = np.zeros((n_windows, H), np.uint8); masks[:, 1] = 1; masks[:, H // 2] = 1
        vp = ctx.variant_posteriors(np.arange(n_windows), masks.reshape(-1), [1e-3] * n_windows)
    finally:
        ctx.close()
    n_gen = H * (H + 1) // 2
    for w in range(n_windows):
        post = po.caller_setup(like[w * H * R:(w + 1) * H * R].reshape(H, R), off)
        g_post = res["post"][res["cell_off"][w]:res["cell_off"][w + 1]].reshape(n_ind, n_gen)
        np.testing.assert_allclose(g_post, post, rtol=RTOL, atol=0)
        ref = po.caller_call(post, off, H, want_cisr=True)
        assert res["iters"][w] == ref["iters"]
        np.testing.assert_allclose(res["freq"][w * H:(w + 1) * H], ref["freq"], rtol=RTOL, atol=1e-150)
        g_best = res["best_gen"][w * n_ind:(w + 1) * n_ind]
        np.testing.assert_allclose(res["best_val"][w * n_ind:(w + 1) * n_ind], ref["best_val"], rtol=RTOL, atol=1e-300)
        for i in range(n_ind):
            if g_best[i] != ref["best_gen"][i]:
                c = ref["cisr"][i]
                assert abs(c[g_best[i]] - c[ref["best_gen"][i]]) <= 1e-12 * abs(c[ref["best_gen"][i]])
        want = po.variant_posterior(post, off, ref["freq"], masks[w], 1e-3)
        np.testing.assert_allclose(vp[w], want, rtol=1e-8, atol=1e-300)
