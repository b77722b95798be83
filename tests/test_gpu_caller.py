"""GPU parity of the genotype-likelihood / EM / variant-posterior stage (through the C ABI)
against the CPU restatement of platypus/cpopulation.pyx:515-632,658-890.

Tolerance: 1e-9 relative on every double (the device's log/exp differ from libm in the last
bits and sums over genotypes are regrouped); iteration counts and genotype calls must be equal
(a call may differ only where the checker's own csr values tie to 1e-12)."""
import numpy as np
import pytest

from platypus_amd import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def split_reads(rng, n_reads, n_ind, empty=None):
    """ascending offsets 0..n_reads for n_ind individuals; individual `empty` gets no reads"""
    cuts = np.sort(rng.integers(1, n_reads, size=n_ind - 1)) if n_ind > 1 else np.empty(0, dtype=np.int64)
    off = np.concatenate([[0], cuts, [n_reads]]).astype(np.int32)
    if empty is not None and n_ind > 1:
        off[empty + 1] = off[empty]
        off = np.maximum.accumulate(off)
        off[-1] = n_reads
    return off


def run_case(n_windows, H, R, n_ind, seed, indel_frac=0.3, empty=None, read_len=60, hap_len=300, options=()):
    from platypus_amd import native
    from oracle import pyoracle as po
    rng = np.random.default_rng(seed)
    b = synth.make_batch(n_windows=n_windows, haps_per_window=H, reads_per_window=R, read_len=read_len,
                         hap_len=hap_len, indel_hap_frac=indel_frac, seed=seed)
    ctx = native.AlignContext(0)
    try:
        for name, value in options:
            ctx.set_option(name, value)
        ctx.upload_batch(b)
        ctx.launch()
        offs = [split_reads(rng, R, n_ind, empty if (w % 3 == 0) else None) for w in range(n_windows)]
        ctx.upload_windows([H] * n_windows, [n_ind] * n_windows,
                           np.arange(n_windows, dtype=np.int64) * H * R, np.concatenate(offs))
        ctx.caller_launch(100)
        res = ctx.caller_fetch()
        _, _, like = ctx.fetch()
        n_gen = H * (H + 1) // 2
        # variants: haplotype h > 0 carries "its" variant; plus one variant shared by two haplotypes
        var_win, masks, priors = [], [], []
        for w in range(n_windows):
            for h in range(1, H):
                m = np.zeros(H, dtype=np.uint8); m[h] = 1
                if h == 1 and H > 2:
                    m[2] = 1
                var_win.append(w); masks.append(m); priors.append(10.0 ** -rng.integers(1, 5))
        gpu_vp = ctx.variant_posteriors(var_win, np.concatenate(masks), priors)
    finally:
        ctx.close()

    vi = 0
    for w in range(n_windows):
        L = like[w * H * R:(w + 1) * H * R].reshape(H, R)
        post = po.caller_setup(L, offs[w])
        g_post = res["post"][res["cell_off"][w]:res["cell_off"][w + 1]].reshape(n_ind, n_gen)
        np.testing.assert_allclose(g_post, post, rtol=RTOL, atol=0)
        ref = po.caller_call(post, offs[w], H, want_cisr=True)
        assert res["iters"][w] == ref["iters"], (w, res["iters"][w], ref["iters"])
        g_freq = res["freq"][res["hap_off"][w]:res["hap_off"][w + 1]]
        # frequencies that have decayed below 1e-150 are products of terms at the edge of the normal range
        # of a double (denormal factors carry few digits): their leading digits depend on the order of
        # additions for the reference as for any restatement; no output can show them (FR is printed "%1.4f")
        np.testing.assert_allclose(g_freq, ref["freq"], rtol=RTOL, atol=1e-150)
        # the largest change is a difference of two frequencies of order one: its absolute error is a few 1e-16
        np.testing.assert_allclose(res["max_change"][w], ref["max_change"], rtol=1e-6, atol=1e-14)
        g_best = res["best_gen"][res["ind_off"][w]:res["ind_off"][w + 1]]
        g_val = res["best_val"][res["ind_off"][w]:res["ind_off"][w + 1]]
        np.testing.assert_allclose(g_val, ref["best_val"], rtol=RTOL, atol=1e-300)
        for i in range(n_ind):
            if g_best[i] != ref["best_gen"][i]:
                c = ref["cisr"][i]
                assert g_best[i] >= 0 and abs(c[g_best[i]] - c[ref["best_gen"][i]]) <= 1e-12 * abs(c[ref["best_gen"][i]])
        for h in range(1, H):
            want = po.variant_posterior(post, offs[w], ref["freq"], masks[vi], priors[vi])
            np.testing.assert_allclose(gpu_vp[vi], want, rtol=1e-8, atol=1e-300)
            vi += 1


def test_single_sample_windows():
    run_case(n_windows=96, H=4, R=32, n_ind=1, seed=11)


def test_few_individuals_with_an_empty_one():
    run_case(n_windows=24, H=6, R=60, n_ind=4, seed=12, empty=1)


def test_many_haplotypes_many_individuals():
    run_case(n_windows=3, H=70, R=240, n_ind=20, seed=13, indel_frac=0.2)


@pytest.mark.parametrize("H", [100, 130, 257])
def test_wide_windows_all_column_chunk_shapes(H):
    """haplotype counts that exercise 2, 3 and 4 + 1 column chunks of the EM's matrix-vector product"""
    run_case(n_windows=2, H=H, R=36, n_ind=3, seed=30 + H, indel_frac=0.1, read_len=40, hap_len=200)


def test_very_wide_window():
    """600 haplotypes (180 300 genotypes): several 256-column passes, large LDS rows, both EM paths"""
    run_case(n_windows=1, H=600, R=20, n_ind=2, seed=61, indel_frac=0.05, read_len=30, hap_len=120)
    run_case(n_windows=1, H=600, R=24, n_ind=5, seed=62, indel_frac=0.05, read_len=30, hap_len=120)


def test_hundreds_of_reads_per_individual():
    """the product of an individual's per-read terms leaves the range of a double long before 300 reads:
    the device keeps mantissa and exponent apart and renormalises every 128 factors (both the per-cell
    kernel and the tiled one, whose read chunks here are longer than 128)"""
    run_case(2, 4, 640, 2, seed=31)
    run_case(1, 26, 520, 2, seed=32)


def test_two_haplotypes():
    run_case(n_windows=40, H=2, R=16, n_ind=2, seed=14)


def test_window_errors():
    from platypus_amd import native
    b = synth.make_batch(n_windows=2, haps_per_window=3, reads_per_window=8, read_len=40, hap_len=200, seed=5)
    ctx = native.AlignContext(0)
    try:
        ctx.upload_batch(b)
        with pytest.raises(native.PlatGpuError):          # no alignment launched yet
            ctx.upload_windows([3, 3], [1, 1], [0, 24], [0, 8, 0, 8]); ctx.caller_launch()
        ctx.launch()
        with pytest.raises(native.PlatGpuError):          # reaches outside the pair list
            ctx.upload_windows([3, 3], [1, 1], [0, 30], [0, 8, 0, 8])
        with pytest.raises(native.PlatGpuError):          # no reads at all
            ctx.upload_windows([3], [2], [0], [0, 0, 0])
        with pytest.raises(native.PlatGpuError):          # offsets must ascend
            ctx.upload_windows([3], [2], [0], [0, 9, 8])
    finally:
        ctx.close()


@pytest.mark.parametrize("H,n_ind", [(7, 3), (2, 16), (3, 9), (4, 11), (5, 5), (6, 4), (8, 2), (2, 1), (4, 16)])
def test_small_window_kernel_equals_general_kernel(H, n_ind):
    """Windows of at most 8 haplotypes take a one-wavefront-per-window kernel that handles 64 / H^2 individuals
    side by side; its sums have the shape of the general kernel's, so both return the same doubles -- for one
    individual per step (H >= 6) as for sixteen (H = 2), whole steps or a ragged last one, with individuals that
    have no reads.  (Up to 16 individuals: beyond that the general kernel gives a wavefront several individuals
    and adds their shares per wavefront first.)"""
    from platypus_amd import native
    R, nw = 4 * n_ind + 24, 50
    b = synth.make_batch(n_windows=nw, haps_per_window=H, reads_per_window=R, read_len=50, hap_len=260,
                         indel_hap_frac=0.3, seed=21 + H)
    rng = np.random.default_rng(3)
    offs = np.concatenate([split_reads(rng, R, n_ind, (w % max(1, n_ind)) if w % 4 == 0 else None) for w in range(nw)])
    out = {}
    for which in ("small", "big"):
        ctx = native.AlignContext(0)
        try:
            ctx.set_option("em_kernel", which)
            ctx.set_option("em_small", "fast")       # the regrouped small-window kernel (the default adds in the reference's order)
            ctx.upload_batch(b)
            ctx.launch()
            ctx.upload_windows([H] * nw, [n_ind] * nw, np.arange(nw, dtype=np.int64) * H * R, offs)
            ctx.caller_launch(100)
            out[which] = ctx.caller_fetch()
        finally:
            ctx.close()
    for k in ("freq", "iters", "max_change", "best_gen", "best_val", "post"):
        assert np.array_equal(out["small"][k], out["big"][k]), k


def test_small_windows_with_many_individuals_against_checker():
    """the side-by-side slots of the small-window kernel over many steps: 2 to 5 haplotypes, 23 to 70 individuals,
    some of them without reads"""
    run_case(n_windows=12, H=2, R=160, n_ind=70, seed=61, empty=33)
    run_case(n_windows=12, H=3, R=120, n_ind=41, seed=62, empty=0)
    run_case(n_windows=12, H=4, R=120, n_ind=23, seed=63, empty=22)
    run_case(n_windows=12, H=5, R=100, n_ind=30, seed=64)


@pytest.mark.parametrize("which", ["big", "multi"])
def test_em_kernel_paths_against_checker(which):
    """the one-workgroup-per-window kernel and the launch-pair-per-iteration path, each forced"""
    opt = [("em_kernel", which)]
    run_case(n_windows=5, H=40, R=120, n_ind=9, seed=41, indel_frac=0.2, empty=2, options=opt)
    run_case(n_windows=7, H=5, R=40, n_ind=3, seed=42, options=opt)


@pytest.mark.parametrize("step", ["full", "sym"])
def test_em_step_kernels_against_checker(step):
    """the launch-pair path's step from whole matrices and from their upper triangle (whose last workgroup per
    window does the update), forced: matrices narrower than a wavefront, of one and a bit and of several
    wavefronts' width, an individual without reads, one to sixteen wavefronts per workgroup"""
    opt = [("em_kernel", "multi"), ("em_step", step)]
    run_case(n_windows=5, H=40, R=120, n_ind=9, seed=44, indel_frac=0.2, empty=2, options=opt)
    run_case(n_windows=3, H=65, R=90, n_ind=5, seed=45, empty=0, read_len=40, hap_len=200, options=opt)
    run_case(n_windows=2, H=130, R=80, n_ind=3, seed=46, read_len=40, hap_len=200, options=opt + [("em_step_waves", "16")])
    run_case(n_windows=4, H=9, R=40, n_ind=4, seed=47, empty=3, options=opt + [("em_step_waves", "1")])
    # widths just past whole wavefronts (the last columns come from the other half's rows), and the widest of those
    run_case(n_windows=2, H=129, R=60, n_ind=3, seed=48, read_len=40, hap_len=200, options=opt)
    run_case(n_windows=2, H=80, R=60, n_ind=2, seed=49, empty=1, read_len=40, hap_len=200, options=opt + [("em_step_waves", "3")])
    run_case(n_windows=1, H=64, R=60, n_ind=2, seed=50, read_len=40, hap_len=200, options=opt)


def test_em_iteration_cap():
    """max_iters is honoured by every path (the reference's default is 100, cpopulation.pyx:741)"""
    from platypus_amd import native
    from oracle import pyoracle as po
    H, R, n_ind, nw = 12, 60, 5, 4
    b = synth.make_batch(n_windows=nw, haps_per_window=H, reads_per_window=R, read_len=50, hap_len=260, seed=43)
    off = np.linspace(0, R, n_ind + 1).astype(np.int32)
    for which in ("big", "multi"):
        ctx = native.AlignContext(0)
        try:
            ctx.set_option("em_kernel", which)
            ctx.upload_batch(b)
            ctx.launch()
            ctx.upload_windows([H] * nw, [n_ind] * nw, np.arange(nw, dtype=np.int64) * H * R, np.tile(off, nw))
            _, _, like = ctx.fetch()
            for cap in (0, 1, 3, 9):
                ctx.caller_launch(cap)
                res = ctx.caller_fetch()
                for w in range(nw):
                    post = po.caller_setup(like[w * H * R:(w + 1) * H * R].reshape(H, R), off)
                    ref = po.caller_call(post, off, H, max_iters=cap)
                    assert res["iters"][w] == ref["iters"] <= cap
                    np.testing.assert_allclose(res["freq"][w * H:(w + 1) * H], ref["freq"], rtol=RTOL, atol=1e-300)
                    np.testing.assert_allclose(res["best_val"][w * n_ind:(w + 1) * n_ind], ref["best_val"], rtol=RTOL, atol=1e-300)
        finally:
            ctx.close()


@pytest.mark.parametrize("which", ["cell", "tile"])
def test_genotype_likelihood_kernels(which):
    """one thread per cell from global memory, and the LDS-tiled kernel (incl. several read chunks per individual)"""
    opt = [("gl_kernel", which)]
    run_case(n_windows=2, H=257, R=60, n_ind=2, seed=51, indel_frac=0.1, read_len=40, hap_len=200, options=opt)
    run_case(n_windows=9, H=4, R=33, n_ind=3, seed=52, empty=0, options=opt)
    run_case(n_windows=3, H=30, R=90, n_ind=4, seed=53, indel_frac=0.2, options=opt)


@pytest.mark.parametrize("H,R,n_ind,W", [(2, 20, 1, 300), (3, 24, 1, 200), (4, 32, 1, 400), (4, 40, 7, 60), (5, 30, 3, 80),
                                         (8, 40, 1, 50), (8, 60, 12, 30), (2, 100, 50, 20), (8, 200, 50, 6), (6, 36, 9, 40),
                                         (10, 40, 1, 50), (12, 48, 1, 40), (16, 60, 1, 30), (16, 48, 3, 20), (9, 45, 20, 20)])
def test_small_windows_in_the_reference_order_give_its_doubles(H, R, n_ind, W):
    """k_em_exact adds every sum of EMiteration in the reference's order: run on the genotype posteriors the device
    itself computed, the CPU restatement of Caller.call returns the SAME doubles -- frequencies, largest change, genotype
    values -- and the same iteration counts and calls; some individuals have no reads"""
    from platypus_amd import native
    from oracle import pyoracle as po
    rng = np.random.default_rng(H * 1000 + R)
    b = synth.make_batch(n_windows=W, haps_per_window=H, reads_per_window=R, read_len=60, hap_len=300, indel_hap_frac=0.3,
                         seed=H * 77 + n_ind)
    ctx = native.AlignContext(0)
    try:
        ctx.upload_batch(b)
        ctx.launch()
        offs = [split_reads(rng, R, n_ind, 1 if (w % 4 == 0 and n_ind > 2) else None) for w in range(W)]
        ctx.upload_windows([H] * W, [n_ind] * W, np.arange(W, dtype=np.int64) * H * R, np.concatenate(offs))
        ctx.caller_launch(100)
        res = ctx.caller_fetch()
        ctx.set_option("em_small", "fast")
        ctx.caller_launch(100)
        fast = ctx.caller_fetch()
    finally:
        ctx.close()
    n_gen = H * (H + 1) // 2
    n_half = 0
    for w in range(W):
        post = res["post"][res["cell_off"][w]:res["cell_off"][w + 1]].reshape(n_ind, n_gen)
        ref = po.caller_call(post, offs[w], H, literal=(w % 2 == 0))
        assert res["iters"][w] == ref["iters"]
        f = res["freq"][res["hap_off"][w]:res["hap_off"][w + 1]]
        assert np.array_equal(f, ref["freq"]), (w, f, ref["freq"])
        assert res["max_change"][w] == ref["max_change"]
        assert np.array_equal(res["best_gen"][res["ind_off"][w]:res["ind_off"][w + 1]], ref["best_gen"])
        assert np.array_equal(res["best_val"][res["ind_off"][w]:res["ind_off"][w + 1]], ref["best_val"])
        n_half += int(np.sum(f == 0.5))
        # the regrouped kernel on the same input: same counts, values to 1e-9
        assert fast["iters"][w] == ref["iters"]
        np.testing.assert_allclose(fast["freq"][fast["hap_off"][w]:fast["hap_off"][w + 1]], ref["freq"], rtol=RTOL, atol=1e-150)
    print("frequencies that are exactly 0.5: %d" % n_half)


@pytest.mark.parametrize("options", [(), (("em_kernel", "big"),), (("em_kernel", "multi"),), (("em_kernel", "multi"), ("em_step", "full")),
                                     (("em_kernel", "multi"), ("em_stage_max", "0"))])
def test_mixed_batch_small_windows_keep_the_reference_order(options):
    """a batch with windows of up to eight haplotypes AND wider ones: the small ones still get the reference's order (the
    CPU restatement returns their doubles from the device's genotype posteriors), the wider ones agree to 1e-9, whichever
    kernels take the wider ones"""
    from platypus_amd import native
    from oracle import pyoracle as po
    H, R, n_ind, W = 20, 48, 5, 36
    rng = np.random.default_rng(404)
    b = synth.make_batch(n_windows=W, haps_per_window=H, reads_per_window=R, read_len=60, hap_len=300, indel_hap_frac=0.3, seed=404)
    n_haps = [int(x) for x in rng.choice([2, 3, 4, 8, 9, 12, 20], size=W)]
    n_haps[0], n_haps[1] = 20, 4                                      # both kinds are there (with 5 individuals 9 and 12 are light enough too)
    ctx = native.AlignContext(0)
    try:
        for name, value in options:
            ctx.set_option(name, value)
        ctx.upload_batch(b)
        ctx.launch()
        offs = [split_reads(rng, R, n_ind, 2 if w % 5 == 0 else None) for w in range(W)]
        ctx.upload_windows(n_haps, [n_ind] * W, np.arange(W, dtype=np.int64) * H * R, np.concatenate(offs))
        ctx.caller_launch(100)
        res = ctx.caller_fetch()
    finally:
        ctx.close()
    n_small = 0
    for w in range(W):
        n = n_haps[w]
        n_gen = n * (n + 1) // 2
        post = res["post"][res["cell_off"][w]:res["cell_off"][w + 1]].reshape(n_ind, n_gen)
        ref = po.caller_call(post, offs[w], n)
        f = res["freq"][res["hap_off"][w]:res["hap_off"][w + 1]]
        best = res["best_gen"][res["ind_off"][w]:res["ind_off"][w + 1]]
        val = res["best_val"][res["ind_off"][w]:res["ind_off"][w + 1]]
        assert res["iters"][w] == ref["iters"], (w, n)
        if n <= 8 or (n <= 16 and n_ind * (n_gen + 1) <= 2000):      # the windows k_em_exact takes
            n_small += 1
            assert np.array_equal(f, ref["freq"]) and res["max_change"][w] == ref["max_change"], (w, n)
            assert np.array_equal(best, ref["best_gen"]) and np.array_equal(val, ref["best_val"]), (w, n)
        else:
            np.testing.assert_allclose(f, ref["freq"], rtol=RTOL, atol=1e-150)
            np.testing.assert_allclose(val, ref["best_val"], rtol=RTOL, atol=1e-300)
    assert 0 < n_small < W
