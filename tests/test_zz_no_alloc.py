"""pg_align_pairs_noalloc (device-resident pair indices and outputs, every intermediate array
in the library's fixed device array) against pg_align_pairs on the same inputs, through the
C ABI.  Bar: scores and anchors bit-exact, likelihoods the same doubles.  Two calls in a
row on one context, the second with a smaller input, so that whatever the first call left in the
fixed array would show.  The caller's device memory comes from tests/hipmem.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


POISON = np.frombuffer(b"\xa5" * 4, np.int32)[0]        # what an output element holds before the call


def _noalloc(ctx, hap_idx, read_idx):
    from tests.hipmem import DeviceArray
    n = len(hap_idx)
    d_h, d_r = DeviceArray(n, np.int32, hap_idx), DeviceArray(n, np.int32, read_idx)
    # poisoned outputs: every element must be written by the call
    d_s, d_a, d_l = DeviceArray(n, np.int32), DeviceArray(n, np.int32), DeviceArray(n, np.float64)
    try:
        ctx.align_pairs_noalloc(n, d_h.ptr, d_r.ptr, d_s.ptr, d_a.ptr, d_l.ptr)
        return d_s.numpy(), d_a.numpy(), d_l.numpy()
    finally:
        for d in (d_h, d_r, d_s, d_a, d_l):
            d.free()


def _same(got, want):
    for g, w, name in zip(got, want, ("score", "anchor", "likelihood")):
        bad = np.flatnonzero(g != w)
        assert len(bad) == 0, (name, len(bad), bad[:5], g[bad[:5]], w[bad[:5]])


@pytest.mark.parametrize("ragged", [False, True])
def test_noalloc_equals_align_pairs_twice_in_a_row(ragged):
    from platypus_amd import native, synth
    b = synth.make_batch(n_windows=60, haps_per_window=5, reads_per_window=40, read_len=120, hap_len=500,
                         indel_hap_frac=0.6, homopolymers=True, n_frac=0.003, ragged=ragged, seed=71 + ragged)
    ctx = native.AlignContext(0)
    try:
        ctx.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
        ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
        rng = np.random.default_rng(5)
        ph, pr = b["pair_hap"], b["pair_read"]
        # first call: the whole list, shuffled; second call: a smaller, different selection with repeats
        first = rng.permutation(len(ph))
        second = rng.integers(0, len(ph), size=len(ph) // 7 + 3)
        for sel in (first, second):
            want = ctx.align_pairs(ph[sel], pr[sel])
            got = _noalloc(ctx, ph[sel], pr[sel])
            _same(got, want)
        assert len(second) < len(first)
        # and once more in the other order of entry points: the fixed array is used first
        got = _noalloc(ctx, ph[first[:1000]], pr[first[:1000]])
        want = ctx.align_pairs(ph[first[:1000]], pr[first[:1000]])
        _same(got, want)
    finally:
        ctx.close()


def test_noalloc_leaves_the_launched_results_alone_and_checks_its_limits():
    from platypus_amd import native, synth
    from tests.hipmem import DeviceArray
    b = synth.make_batch(n_windows=10, haps_per_window=4, reads_per_window=16, read_len=80, hap_len=400,
                         indel_hap_frac=0.5, seed=73)
    ctx = native.AlignContext(0)
    try:
        ctx.upload_batch(b)
        ctx.launch()
        before = ctx.fetch()
        got = _noalloc(ctx, b["pair_hap"][:100], b["pair_read"][:100])
        after = ctx.fetch()                              # results of pg_align_launch are still there
        _same(after, before)
        _same(got, [x[:100] for x in before])
        # an index outside the uploaded batches: refused, nothing written
        bad_h = b["pair_hap"][:50].copy(); bad_h[17] = 10 ** 6
        d_h, d_r = DeviceArray(50, np.int32, bad_h), DeviceArray(50, np.int32, b["pair_read"][:50])
        d_s, d_a = DeviceArray(50, np.int32), DeviceArray(50, np.int32)
        with pytest.raises(native.PlatGpuError) as e:
            ctx.align_pairs_noalloc(50, d_h.ptr, d_r.ptr, d_s.ptr, d_a.ptr, None)
        assert e.value.code == native.PG_ERR_INVALID and "pair 17" in str(e.value)
        assert (d_s.numpy() == POISON).all() and (d_a.numpy() == POISON).all()
        # more pairs than the fixed array is sized for
        with pytest.raises(native.PlatGpuError) as e:
            ctx.align_pairs_noalloc(native.NOALLOC_MAX_PAIRS + 1, d_h.ptr, d_r.ptr, d_s.ptr, d_a.ptr, None)
        assert e.value.code == native.PG_ERR_UNSUPPORTED
        # likelihood output is optional; the call after a refused one works
        got = _noalloc(ctx, b["pair_hap"][100:130], b["pair_read"][100:130])
        _same(got, [x[100:130] for x in before])
    finally:
        ctx.close()


def test_noalloc_at_its_largest_size():
    """PG_NOALLOC_MAX_PAIRS pairs of a ragged batch: the arrays of the fixed scratch are used to their ends"""
    from platypus_amd import native, synth
    b = synth.make_batch(n_windows=64, haps_per_window=8, reads_per_window=64, read_len=100, hap_len=420,
                         indel_hap_frac=0.5, ragged=True, seed=75)
    ctx = native.AlignContext(0)
    try:
        ctx.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
        ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
        reps = native.NOALLOC_MAX_PAIRS // len(b["pair_hap"])
        ph, pr = np.tile(b["pair_hap"], reps), np.tile(b["pair_read"], reps)
        assert len(ph) == native.NOALLOC_MAX_PAIRS
        want = ctx.align_pairs(ph, pr)
        _same(_noalloc(ctx, ph, pr), want)
        _same(_noalloc(ctx, ph[:777], pr[:777]), [x[:777] for x in want])
    finally:
        ctx.close()


def test_noalloc_takes_no_device_memory():
    """free device memory as the runtime reports it is the same before and after a call (the first call of a process
    may load code objects, so one call is made before the measurement)"""
    from platypus_amd import native, synth
    from tests.hipmem import DeviceArray, free_bytes
    b = synth.make_batch(n_windows=300, haps_per_window=5, reads_per_window=40, read_len=120, hap_len=500,
                         indel_hap_frac=0.6, ragged=True, seed=77)
    ctx = native.AlignContext(0)
    try:
        ctx.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
        ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
        n = len(b["pair_hap"])
        d = [DeviceArray(n, np.int32, b["pair_hap"]), DeviceArray(n, np.int32, b["pair_read"]),
             DeviceArray(n, np.int32), DeviceArray(n, np.int32), DeviceArray(n, np.float64)]
        try:
            ctx.align_pairs_noalloc(1000, *[x.ptr for x in d])          # a small launch (eight lanes per pair) ...
            ctx.align_pairs_noalloc(n, *[x.ptr for x in d])             # ... and a large one (one lane per pair)
            before = free_bytes()
            ctx.align_pairs_noalloc(n, *[x.ptr for x in d])
            ctx.align_pairs_noalloc(1000, *[x.ptr for x in d])
            assert free_bytes() == before
        finally:
            for x in d:
                x.free()
    finally:
        ctx.close()


def test_overlapping_calls_on_one_gpu_are_refused_not_raced():
    """two host threads with a context each on the same GPU: every call either gives the right answer or is refused
    with PG_ERR_STATE (the GPU's fixed array was in use); nothing else, and never a wrong number"""
    import threading
    from platypus_amd import native, synth
    from tests.hipmem import DeviceArray
    b = synth.make_batch(n_windows=400, haps_per_window=5, reads_per_window=40, read_len=120, hap_len=500,
                         indel_hap_frac=0.6, ragged=True, seed=78)
    n = len(b["pair_hap"])
    ref_ctx = native.AlignContext(0)
    try:
        ref_ctx.upload_batch(b)
        ref_ctx.launch()
        want = ref_ctx.fetch()
    finally:
        ref_ctx.close()
    outcomes = [[], []]
    errors = []

    def worker(slot):
        try:
            ctx = native.AlignContext(0)
            try:
                ctx.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
                ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
                d = [DeviceArray(n, np.int32, b["pair_hap"]), DeviceArray(n, np.int32, b["pair_read"]),
                     DeviceArray(n, np.int32), DeviceArray(n, np.int32), DeviceArray(n, np.float64)]
                try:
                    for _ in range(40):
                        try:
                            ctx.align_pairs_noalloc(n, *[x.ptr for x in d])
                        except native.PlatGpuError as e:
                            assert e.code == native.PG_ERR_STATE, str(e)
                            outcomes[slot].append("refused")
                            continue
                        got = (d[2].numpy(), d[3].numpy(), d[4].numpy())
                        _same(got, want)
                        outcomes[slot].append("right")
                finally:
                    for x in d:
                        x.free()
            finally:
                ctx.close()
        except BaseException as e:      # noqa: B036  (reported by the main thread)
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert all(len(o) == 40 for o in outcomes)
    assert any(x == "right" for o in outcomes for x in o)
