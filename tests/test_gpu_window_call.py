"""pg_align_window (one window per call: one block up, one kernel to make the arrays, anchoring, alignment, one
block back) against the five-call path on the same windows and against the CPU checker: the same integers,
the same doubles, the same errors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from platypus_amd import native
    c = native.AlignContext(0, 1024)
    yield c
    c.close()


def _window(b, w):
    """window w of a packed batch as the arguments of align_window (offsets are NOT rebased: the entry point
    takes them relative to its first one, as the upload calls do)"""
    m = b["meta"]
    H, R = m["haps_per_window"], m["reads_per_window"]
    h0, r0 = w * H, w * R
    return (b["hap_seq"], b["hap_off"][h0:h0 + H + 1], b["hap_start"][h0:h0 + H], b["hap_varlen"][h0:h0 + H],
            b["read_seq"], b["read_qual"], b["read_off"][r0:r0 + R + 1], b["read_pos"][r0:r0 + R], b["read_mapq"][r0:r0 + R])


def _both(ctx, b):
    m = b["meta"]
    W, H, R = m["windows"], m["haps_per_window"], m["reads_per_window"]
    got = [ctx.align_window(*_window(b, w)) for w in range(W)]
    ctx.upload_batch(b)
    ctx.launch()
    s, a, l = ctx.fetch()
    for w in range(W):
        sel = slice(w * H * R, (w + 1) * H * R)
        assert np.array_equal(got[w][0], s[sel]), ("score", w)
        assert np.array_equal(got[w][1], a[sel]), ("anchor", w)
        assert np.array_equal(got[w][2], l[sel]), ("likelihood", w)
    return s, a


@pytest.mark.parametrize("kw", [
    dict(n_windows=12, read_len=150, seed=41),
    dict(n_windows=8, read_len=150, indel_hap_frac=1.0, sub_rate=0.1, read_indel_frac=0.5, seed=42),
    dict(n_windows=8, read_len=100, homopolymers=True, n_frac=0.02, indel_hap_frac=0.7, seed=43),
    dict(n_windows=8, read_len=120, ragged=True, indel_hap_frac=0.5, seed=44),
    dict(n_windows=6, read_len=36, hap_len=300, haps_per_window=7, reads_per_window=50, ragged=True, seed=45),
    dict(n_windows=3, read_len=250, hap_len=1500, haps_per_window=12, reads_per_window=300, indel_hap_frac=0.6, seed=46),
    dict(n_windows=4, read_len=150, haps_per_window=1, reads_per_window=1, indel_hap_frac=1.0, seed=47),
])
def test_window_calls_equal_the_batch_path_and_the_checker(ctx, kw):
    from oracle import pyoracle as po
    from platypus_amd import synth
    b = synth.make_batch(**kw)
    rng = np.random.default_rng(kw["seed"])
    nR = len(b["read_pos"])
    shift = np.where(rng.random(nR) < 0.15, rng.integers(-40, 41, nR), 0)      # some reads off their place
    b["read_pos"] = (b["read_pos"].astype(np.int64) + shift).astype(np.int32)
    s, a = _both(ctx, b)
    es, ea = po.run_pairs(synth.oracle_view(b), threads=8, reuse_hash=True)
    assert np.array_equal(s, es) and np.array_equal(a, ea)


def test_reads_of_every_short_length_and_off_the_ends(ctx):
    rng = np.random.default_rng(5)
    A = np.frombuffer(b"ACGT", np.uint8)
    hap = A[rng.integers(0, 4, 400)]
    haps = [hap, hap.copy(), np.tile(np.frombuffer(b"AC", np.uint8), 100), np.frombuffer(b"ACGTACGTTGCAAGT", np.uint8)]
    hvar = [0, 3, 1, 2]
    reads, pos = [], []
    for L in (1, 2, 6, 7, 8, 9, 15, 16, 17, 40):
        reads.append(hap[100:100 + L].copy()); pos.append(1100)
    for st in (330, 360, 385, 392, 399):
        reads.append(np.resize(hap[st:], 60).copy()); pos.append(1000 + st)
    for p in (-50, 0, 1003, 5000, 2_000_000_000):
        reads.append(hap[0:50].copy()); pos.append(p)
    reads.append(np.frombuffer(b"N" * 50, np.uint8)); pos.append(1100)
    quals = [np.full(len(r), 30, np.uint8) for r in reads]
    hap_off = np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.int64)
    read_off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    args = (np.concatenate(haps), hap_off, np.full(4, 1000, np.int32), np.array(hvar, np.int32), np.concatenate(reads),
            np.concatenate(quals), read_off, np.array(pos, np.int64).astype(np.int32), np.full(len(reads), 37, np.int32))
    s, a, l = ctx.align_window(*args)
    nh, nr = 4, len(reads)
    ctx.upload_haps(*args[:4]); ctx.upload_reads(*args[4:])
    ctx.pairs_block(0, nh, 0, nr)
    ctx.launch()
    s2, a2, l2 = ctx.fetch()
    assert np.array_equal(s, s2) and np.array_equal(a, a2) and np.array_equal(l, l2)
    assert (s == -1).any() and (l[s == -1] == 0.0).all()


def test_custom_parameters_and_a_smaller_window_after_a_larger_one(ctx):
    from platypus_amd import synth
    big = synth.make_batch(n_windows=1, read_len=150, haps_per_window=9, reads_per_window=200, indel_hap_frac=0.8,
                           homopolymers=True, seed=51)
    small = synth.make_batch(n_windows=2, read_len=80, hap_len=400, haps_per_window=3, reads_per_window=5, indel_hap_frac=0.8,
                             ragged=True, seed=52)
    tab = bytes([33 + v for v in (50, 40, 35, 30, 22, 15, 9, 4)])
    try:
        ctx.set_params(5, 1, tab)
        _both(ctx, big)
        _both(ctx, small)
    finally:
        ctx.set_params()
    _both(ctx, small)


def test_long_reads_go_through_the_upload_calls(ctx):
    from platypus_amd import synth
    b = synth.make_batch(n_windows=2, read_len=400, hap_len=1600, haps_per_window=3, reads_per_window=12, indel_hap_frac=1.0, seed=53)
    _both(ctx, b)


def test_errors_are_those_of_the_upload_calls(ctx):
    from platypus_amd import native, synth
    b = synth.make_batch(n_windows=1, reads_per_window=10, read_len=80, hap_len=400, seed=54)
    args = list(_window(b, 0))

    def code_and_text(call):
        with pytest.raises(native.PlatGpuError) as e:
            call()
        return e.value.code, str(e.value)

    def both(mutate, upload):
        a = list(args)
        mutate(a)
        c1, t1 = code_and_text(lambda: ctx.align_window(*a))
        c2, t2 = code_and_text(lambda: upload(a))
        assert c1 == c2, (t1, t2)
        return t1, t2

    up_h = lambda a: ctx.upload_haps(*a[:4])
    up_r = lambda a: ctx.upload_reads(*a[4:])

    def bad_qual(a): a[5] = a[5].copy(); a[5][5] = 200
    def bad_read_byte(a): a[4] = a[4].copy(); a[4][11] = 32
    def bad_hap_byte(a): a[0] = a[0].copy(); a[0][7] = 0
    def bad_mapq(a): a[8] = a[8].copy(); a[8][3] = 300
    def long_read(a): a[6] = a[6].copy(); a[6][-1] += 2000; a[4] = np.resize(a[4], int(a[6][-1]) + 1); a[5] = np.resize(a[5], int(a[6][-1]) + 1)
    def short_indel_hap(a): a[1] = np.array([0, 5], np.int64); a[2] = a[2][:1]; a[3] = np.array([2], np.int32)
    t1, t2 = both(bad_qual, up_r); assert t1 == t2
    both(bad_read_byte, up_r)
    both(bad_hap_byte, up_h)
    t1, t2 = both(bad_mapq, up_r); assert t1 == t2
    t1, t2 = both(long_read, up_r); assert t1 == t2
    t1, t2 = both(short_indel_hap, up_h); assert t1 == t2
    # an empty window: nothing to do, nothing written
    s, a, l = ctx.align_window(args[0], args[1][:1], args[2][:0], args[3][:0], *args[4:])
    assert len(s) == 0 and len(a) == 0 and len(l) == 0
    # and the context still works
    _both(ctx, b)


def test_results_stay_on_the_device_for_what_follows(ctx):
    """the genotype sums, the caller stage and a later fetch read the window call's likelihoods where it left them"""
    from platypus_amd import synth
    b = synth.make_batch(n_windows=1, read_len=100, haps_per_window=5, reads_per_window=40, indel_hap_frac=0.5, seed=61)
    H, R = 5, 40
    g1 = np.repeat(np.arange(H), H) * R
    g2 = np.tile(np.arange(H), H) * R
    nr = np.full(H * H, R, np.int32)

    def after(call):
        call()
        s, a, l = ctx.fetch()
        gl = ctx.genotype_loglik(g1, g2, nr)
        ctx.upload_windows([H], [1], [0], [0, R])
        ctx.caller_launch(50)
        out = ctx.caller_fetch()
        stats = ctx.anchor_stats()
        return s, a, l, gl, out, stats

    def batch():
        ctx.upload_batch(b)
        ctx.launch()

    x = after(lambda: ctx.align_window(*_window(b, 0)))
    y = after(batch)
    for u, v in zip(x[:4], y[:4]):
        assert np.array_equal(u, v)
    for k in x[4]:
        assert np.array_equal(np.asarray(x[4][k]), np.asarray(y[4][k])), k
    assert x[5][0] == y[5][0]
