"""The Cython 3 mirror of the reference's calign / cwindow / chaplotype /
cgenotype interfaces.  Window pointers are host logic (CPU test); everything
that aligns runs on the GPU and is compared with the CPU checker."""
import numpy as np
import pytest


def _buffers(b, H, R):
    """One bamReadBuffer per window of a synthetic batch (reads sorted by pos)."""
    from platypus_amd import cwindow
    bufs = []
    W = len(b["hap_off"]) // H if False else b["meta"]["windows"]
    for w in range(W):
        lo, hi = w * R, (w + 1) * R
        order = np.argsort(b["read_pos"][lo:hi], kind="stable") + lo
        seq = np.concatenate([b["read_seq"][b["read_off"][r]:b["read_off"][r + 1]] for r in order])
        qual = np.concatenate([b["read_qual"][b["read_off"][r]:b["read_off"][r + 1]] for r in order])
        lens = (b["read_off"][order + 1] - b["read_off"][order])
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        pos = b["read_pos"][order]
        bufs.append((order, cwindow.bamReadBuffer(b"chr20", int(pos.min()), int(pos.max()) + 1000, seq, qual, off,
                                                  pos, (pos + lens).astype(np.int32), b["read_mapq"][order])))
    return bufs


def test_window_pointers_match_reference_rule():
    """setWindowPointers against a direct restatement of cwindow.pyx:242-267."""
    from platypus_amd import cwindow
    rng = np.random.default_rng(5)
    n = 400
    pos = np.sort(rng.integers(1000, 6000, n)).astype(np.int32)
    lens = rng.integers(30, 151, n)
    end = (pos + lens + rng.integers(-5, 30, n)).astype(np.int32)      # deletions / soft clips move the end
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seq = np.full(off[-1], 65, np.uint8)
    buf = cwindow.bamReadBuffer(b"1", 1000, 7000, seq, seq, off, pos, end, np.full(n, 60, np.int32))
    longest = int((end - pos).max())
    for _ in range(300):
        s = int(rng.integers(900, 6100)); e = s + int(rng.integers(1, 400))
        lo = int(np.searchsorted(pos, max(1, s - longest), side="left"))
        hi = int(np.searchsorted(pos, e, side="left"))
        while lo < n and end[lo] <= s:
            lo += 1
        assert buf.setWindow(s, e) == (lo, min(hi, n))
    empty = cwindow.bamReadBuffer(b"1", 0, 10, np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.int64),
                                  np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert empty.setWindow(5, 9) == (0, 0)


@pytest.mark.gpu
def test_haplotype_align_reads_prefetch_and_genotypes():
    from oracle import pyoracle as po
    from platypus_amd import calign, cgenotype, chaplotype, synth
    H, R = 4, 30
    b = synth.make_batch(n_windows=5, haps_per_window=H, reads_per_window=R, read_len=100, hap_len=520,
                         indel_hap_frac=0.5, seed=77)
    exp_s, exp_a = po.run_pairs(synth.oracle_view(b), reuse_hash=True)
    mq = b["read_mapq"]
    for w, (order, buf) in enumerate(_buffers(b, H, R)):
        haps = []
        for h in range(w * H, (w + 1) * H):
            seq = b["hap_seq"][b["hap_off"][h]:b["hap_off"][h + 1]].tobytes()
            haps.append(chaplotype.Haplotype.fromSequence(b"chr20", int(b["hap_start"][h]), int(b["hap_start"][h]) + len(seq),
                                             seq, int(b["hap_varlen"][h])))
        buf.setWindow(int(b["hap_start"][w * H]) + 200, int(b["hap_start"][w * H]) + 320)
        lo, hi = buf.windowRange()
        sel = order[lo:hi]                                  # original read ids inside the window
        def expected(h):
            return np.array([po.likelihood(exp_s[h * R + (r - w * R)], mq[r]) for r in sel])
        # (1) the reference's call pattern: one haplotype, one individual, cache on second call
        first = haps[0].alignReadsPy(0, buf, 1)
        assert np.array_equal(first, expected(w * H)) and haps[0].isCached(0)
        assert np.array_equal(haps[0].alignReadsPy(0, buf, 1), first)
        # (2) the batch hook fills every other cache in one launch
        chaplotype.prefetchWindow(haps, [buf])
        for k, hp in enumerate(haps):
            assert hp.isCached(0)
            assert np.array_equal(hp.alignReadsPy(0, buf, 1), expected(w * H + k))
        # (3) genotype log-likelihoods: same arithmetic and order as cgenotype.pyx:58-61
        gts = cgenotype.allGenotypes(haps)
        assert len(gts) == H * (H + 1) // 2
        ll = cgenotype.genotypeLogLikelihoods(haps, gts, [buf])
        g = 0
        for i in range(H):
            for j in range(i, H):
                want = po.genotype_loglik(expected(w * H + i), expected(w * H + j))
                np.testing.assert_allclose(ll[0, g], want, rtol=1e-9)       # device log, fixed-tree sum
                np.testing.assert_allclose(gts[g].calculateDataLikelihoodPy(buf, 0, 1), want, rtol=1e-9)
                g += 1
    # single-pair entry point with the reference's signature
    h, r = 1, 3
    got = calign.alignNoTracebackPy(b["read_seq"][b["read_off"][r]:b["read_off"][r + 1]].tobytes(),
                                    b["read_qual"][b["read_off"][r]:b["read_off"][r + 1]].tobytes(),
                                    int(b["read_pos"][r]), int(mq[r]),
                                    b["hap_seq"][b["hap_off"][h]:b["hap_off"][h + 1]].tobytes(),
                                    int(b["hap_start"][h]), int(b["hap_varlen"][h]), po.HOMOPOLQ)
    assert got == po.likelihood(exp_s[h * R + r], mq[r])
    calign.closeDevice()


class _Var(object):
    """stand-in for platypus/variant.pyx Variant: hashable, comparable, with a prior"""
    def __init__(self, name, prior):
        self.name, self.prior = name, prior

    def calculatePrior(self):
        return self.prior

    def __eq__(self, other):
        return self.name == other.name

    def __hash__(self):
        return hash(self.name)


def _buffer_of(b, ids):
    from platypus_amd import cwindow
    ids = np.asarray(ids, dtype=np.int64)
    if len(ids) == 0:
        z = np.zeros(0, np.int32)
        return cwindow.bamReadBuffer(b"chr20", 0, 10, np.zeros(0, np.uint8), np.zeros(0, np.uint8),
                                     np.zeros(1, np.int64), z, z, z)
    ids = ids[np.argsort(b["read_pos"][ids], kind="stable")]
    seq = np.concatenate([b["read_seq"][b["read_off"][r]:b["read_off"][r + 1]] for r in ids])
    qual = np.concatenate([b["read_qual"][b["read_off"][r]:b["read_off"][r + 1]] for r in ids])
    lens = b["read_off"][ids + 1] - b["read_off"][ids]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pos = b["read_pos"][ids]
    buf = cwindow.bamReadBuffer(b"chr20", int(pos.min()), int(pos.max()) + 1000, seq, qual, off, pos,
                                (pos + lens).astype(np.int32), b["read_mapq"][ids])
    buf.setWindow(int(pos.min()), int(pos.max()) + 1)       # every read of the buffer is in the window
    return ids, buf


@pytest.mark.gpu
def test_caller_setup_call_and_variant_posterior():
    """Caller.setup / Caller.call / Population.calculatePosterior (cpopulation.pyx) against the CPU checker."""
    from oracle import pyoracle as po
    from platypus_amd import calign, cgenotype, chaplotype, cpopulation, synth
    H, R = 5, 48
    b = synth.make_batch(n_windows=4, haps_per_window=H, reads_per_window=R, read_len=80, hap_len=400,
                         indel_hap_frac=0.4, seed=78)
    exp_s, _ = po.run_pairs(synth.oracle_view(b), reuse_hash=True)
    mq = b["read_mapq"]
    for w in range(4):
        variants = [_Var("v%d" % k, 10.0 ** -(1 + k % 3)) for k in range(H)]
        haps = []
        for k, h in enumerate(range(w * H, (w + 1) * H)):
            seq = b["hap_seq"][b["hap_off"][h]:b["hap_off"][h + 1]].tobytes()
            vs = () if k == 0 else ((variants[k],) if k != 2 else (variants[1], variants[2]))
            haps.append(chaplotype.Haplotype.fromSequence(b"chr20", int(b["hap_start"][h]), int(b["hap_start"][h]) + len(seq),
                                             seq, int(b["hap_varlen"][h]), vs))
        reads = np.arange(w * R, (w + 1) * R)
        groups = [reads[0::3], reads[:0], reads[1::3], reads[2::3]]      # the second individual has no reads
        ids, bufs = [], []
        for gsel in groups:
            if len(gsel):
                i, bf = _buffer_of(b, gsel)
            else:
                i, bf = np.zeros(0, np.int64), _buffer_of(b, gsel)
            ids.append(i); bufs.append(bf)
        assert [bf.readCount for bf in bufs] == [len(i) for i in ids]
        order = np.concatenate(ids)
        off = np.concatenate([[0], np.cumsum([len(i) for i in ids])]).astype(np.int32)
        like = np.array([[po.likelihood(exp_s[h * R + (r - w * R)], mq[r]) for r in order]
                         for h in range(w * H, (w + 1) * H)])
        gts = cgenotype.allGenotypes(haps)
        caller = cpopulation.Caller()
        caller.setupPy(haps, gts, bufs)
        pop = caller.callPy()
        post = po.caller_setup(like, off)
        ref = po.caller_call(post, off, H, want_cisr=True)
        np.testing.assert_allclose(np.array(pop.genotypePosteriorsPy()), post, rtol=1e-9)
        np.testing.assert_allclose(pop.frequenciesPy(), ref["freq"], rtol=1e-9, atol=1e-300)
        assert pop.nIterations == ref["iters"]
        assert pop.haplotypeIndexesPy() == [(a, c) for a in range(H) for c in range(a, H)]
        for i in range(4):
            if ref["best_gen"][i] < 0:
                assert pop.genotypeCalls[i] is None and pop.genotypeLikelihoods[i] == 0.0
            else:
                if pop.genotypeCalls[i] is not gts[ref["best_gen"][i]]:
                    # two genotypes whose values agree to the last few bits (the genotype posteriors of device and
                    # checker differ in theirs): either is the largest, as in tests/test_gpu_caller.py
                    got = [k for k, g in enumerate(gts) if g is pop.genotypeCalls[i]][0]
                    c = ref["cisr"][i]
                    assert abs(c[got] - c[ref["best_gen"][i]]) <= 1e-12 * abs(c[ref["best_gen"][i]])
                np.testing.assert_allclose(pop.genotypeLikelihoods[i], ref["best_val"][i], rtol=1e-9)
        for k in (1, 2, 4):
            mask = [1 if variants[k] in hp.variantsPy() else 0 for hp in haps]
            want = po.variant_posterior(post, off, ref["freq"], mask, variants[k].prior)
            np.testing.assert_allclose(pop.calculatePosteriorPy(variants[k]), want, rtol=1e-8)
        # the haplotypes' caches were filled by setup, as the reference's loop would have
        assert all(hp.isCached(i) for hp in haps for i in range(4))
    with pytest.raises(ValueError):
        cpopulation.Caller().setupPy(haps, gts[::-1], bufs)
    calign.closeDevice()


@pytest.mark.gpu
def test_window_batch_equals_window_by_window():
    """WindowBatch (collect -> one alignment launch + one genotype/EM launch -> consume) returns, window for
    window, what Caller.setup / Caller.call return one window at a time."""
    from platypus_amd import calign, cgenotype, chaplotype, cpopulation, synth
    H, R, W = 4, 36, 12
    b = synth.make_batch(n_windows=W, haps_per_window=H, reads_per_window=R, read_len=70, hap_len=360,
                         indel_hap_frac=0.4, seed=79)
    wins = []
    for w in range(W):
        variants = [_Var("w%dv%d" % (w, k), 1e-3) for k in range(H)]
        haps = []
        for k, h in enumerate(range(w * H, (w + 1) * H)):
            seq = b["hap_seq"][b["hap_off"][h]:b["hap_off"][h + 1]].tobytes()
            haps.append(chaplotype.Haplotype.fromSequence(b"chr20", int(b["hap_start"][h]), int(b["hap_start"][h]) + len(seq),
                                             seq, int(b["hap_varlen"][h]), () if k == 0 else (variants[k],)))
        reads = np.arange(w * R, (w + 1) * R)
        bufs = [_buffer_of(b, reads[0::2])[1], _buffer_of(b, reads[1::2])[1]]
        wins.append((haps, bufs, variants))
    # one window at a time
    single = []
    for haps, bufs, variants in wins:
        caller = cpopulation.Caller()
        caller.setupPy(haps, cgenotype.allGenotypes(haps), bufs)
        pop = caller.callPy()
        single.append((pop.frequenciesPy(), pop.genotypePosteriorsPy(), pop.nIterations, list(pop.genotypeLikelihoods),
                       [None if g is None else (g.haplotypes[0], g.haplotypes[1]) for g in pop.genotypeCalls],
                       [pop.calculatePosteriorPy(v) for v in variants[1:]]))
    # all windows in one batch; the buffers' window pointers may move on after add()
    batch = cpopulation.WindowBatch()
    for haps, bufs, _ in wins:
        assert batch.add(haps, bufs) == len(batch) - 1
        for bf in bufs:
            bf.setWindow(0, 1)
    pops = batch.run()
    assert len(pops) == W and batch.timings["windows"] == W
    post = batch.posteriors([(w, v) for w in range(W) for v in wins[w][2][1:]])
    k = 0
    for w, pop in enumerate(pops):
        f, gp, it, gl, calls, vp = single[w]
        assert pop.frequenciesPy() == f and pop.genotypePosteriorsPy() == gp and pop.nIterations == it
        assert list(pop.genotypeLikelihoods) == gl
        assert [None if g is None else (g.haplotypes[0], g.haplotypes[1]) for g in pop.genotypeCalls] == calls
        assert post[k:k + len(vp)] == vp
        k += len(vp)
    # a Population of an earlier call can no longer reach its results on the device
    caller = cpopulation.Caller()
    caller.setupPy(wins[0][0], cgenotype.allGenotypes(wins[0][0]), wins[0][1])
    with pytest.raises(ValueError):
        cpopulation.WindowBatch().add(wins[0][0], wins[0][1])      # window pointers were moved: no reads
    for bf in wins[0][1]:
        bf.setWindow(0, 2 ** 30)
    caller.setupPy(wins[0][0], cgenotype.allGenotypes(wins[0][0]), wins[0][1])
    caller.callPy()
    with pytest.raises(calign.PlatGpuError):
        pops[0].calculatePosteriorPy(wins[0][2][1])
    calign.closeDevice()


@pytest.mark.gpu
def test_site_likelihoods_second_alignment_site():
    """WindowBatch.siteLikelihoods: AA/AB/BB log-likelihoods of fresh reference/variant haplotypes over a
    window's reads (variantcaller.pyx:371-378,422-427), all sites in one launch, against the CPU checker."""
    from oracle import pyoracle as po
    from platypus_amd import calign, chaplotype, cpopulation
    from platypus_amd.fastafile import FastaFile
    from platypus_amd.variant import Variant
    rng = np.random.default_rng(91)
    L, G = 60, 4000
    genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, G))
    ref = FastaFile(sequences={b"c": genome})
    batch = cpopulation.WindowBatch()
    items, expected = [], []
    for w in range(6):
        ws = 400 + 500 * w
        var = [Variant(b"c", ws + 20, genome[ws + 20:ws + 21], b"A" if genome[ws + 20:ws + 21] != b"A" else b"C"),
               Variant(b"c", ws + 20, b"", b"GTG"),
               Variant(b"c", ws + 20, genome[ws + 21:ws + 23], b"")][w % 3]
        refHap = chaplotype.Haplotype(b"c", ws, ws + 40, (), ref, L)
        varHap = chaplotype.Haplotype(b"c", ws, ws + 40, (var,), ref, L)
        # reads sampled from either haplotype, two individuals (the second one empty in window 4)
        seqs, quals, poss = [], [], []
        for r in range(24):
            src = (refHap, varHap)[r % 2].sequence
            o = int(rng.integers(2 * L - 50, 2 * L - 5))
            s = bytearray(src[o:o + L])
            for j in np.flatnonzero(rng.random(L) < 0.02):
                s[j] = b"ACGT"[int(rng.integers(0, 4))]
            seqs.append(bytes(s)); quals.append(bytes(rng.integers(10, 41, L).astype(np.uint8)))
            poss.append(refHap.startPosPy + o)
        order = np.argsort(poss, kind="stable")
        def make_buffer(ids):
            from platypus_amd import cwindow
            if len(ids) == 0:
                z = np.zeros(0, np.int32)
                return cwindow.bamReadBuffer(b"c", 0, 10, np.zeros(0, np.uint8), np.zeros(0, np.uint8),
                                             np.zeros(1, np.int64), z, z, z)
            seq = np.frombuffer(b"".join(seqs[i] for i in ids), dtype=np.uint8)
            qual = np.frombuffer(b"".join(quals[i] for i in ids), dtype=np.uint8)
            pos = np.array([poss[i] for i in ids], dtype=np.int32)
            buf = cwindow.bamReadBuffer(b"c", int(pos.min()), int(pos.max()) + 100, seq, qual,
                                        np.arange(len(ids) + 1, dtype=np.int64) * L, pos, pos + L,
                                        np.full(len(ids), 50, np.int32))
            buf.setWindow(int(pos.min()), int(pos.max()) + 1)
            return buf
        groups = [list(order[:14]), [] if w == 4 else list(order[14:])]
        bufs = [make_buffer(g) for g in groups]
        batch.add([refHap, varHap], bufs)
        items.append((w, refHap, varHap))
        exp = []
        for g in groups:
            if not g:
                exp.append(None)
                continue
            rows = []
            for hp in (refHap, varHap):
                row = []
                for i in g:
                    sc, _ = po.align_no_traceback(seqs[i], quals[i], poss[i], hp.sequence, hp.startPosPy,
                                                  hp.varSeqLenPy)
                    row.append(po.likelihood(sc, 50))
                rows.append(np.array(row))
            exp.append((po.genotype_loglik(rows[0], rows[0]), po.genotype_loglik(rows[0], rows[1]),
                        po.genotype_loglik(rows[1], rows[1])))
        expected.append(exp)
    pops = batch.run()
    got = batch.siteLikelihoods(items)
    for g_item, e_item in zip(got, expected):
        assert len(g_item) == 2
        for g, e in zip(g_item, e_item):
            if e is None:
                assert g is None
            else:
                np.testing.assert_allclose(g, e, rtol=1e-9)
    # the genotype/EM results of run() are still reachable
    assert 0.0 <= pops[0].calculatePosteriorPy(items[0][2].variantsPy()[0]) <= 1.0
    calign.closeDevice()


@pytest.mark.gpu
def test_hash_sequence_and_map_read_cdefs():
    """the reference's cdef hash_sequence / map_read (calign.pxd:5, calign.pyx:99-158) through the Cython layer"""
    from oracle import pyoracle as po
    from platypus_amd import calign
    rng = np.random.default_rng(8)
    seq = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 800))
    want = po.hash_sequence(seq)
    assert calign.hashSequencePy(seq) == list(want)
    for o in (0, 13, 400, 700):
        read = seq[o:o + 90]
        assert calign.mapReadPy(read, seq) == po.map_read(read, len(seq), want) == o
    assert calign.mapReadPy(b"ACG", seq) == 0
    calign.closeDevice()


def test_reference_call_sites_compile_against_the_layer():
    """tests/cy_dropin/refcalls.pyx cimports platypus_amd's .pxd files unchanged and makes the calls of
    cgenotype.pyx:55-56, variantcaller.pyx:130 and :373-374 with the reference's argument lists; it is built
    with the layer (platypus_amd.build_cython), so importing it proves the interface still fits."""
    from tests.cy_dropin import refcalls
    for name in ("windowHaplotypes", "calculateDataLikelihood", "perReadLikelihoods", "anchors"):
        assert callable(getattr(refcalls, name))


@pytest.mark.gpu
def test_reference_call_sites_against_checker():
    """The same calls on the GPU: the reference's constructor form, alignReads through the genotype sum of
    cgenotype.pyx:58-61, alignNoTraceback read by read, hash_sequence + map_read -- against the CPU checker."""
    from oracle import pyoracle as po
    from platypus_amd import calign, cwindow
    from platypus_amd.fastafile import FastaFile
    from platypus_amd.variant import Variant
    from tests.cy_dropin import refcalls
    rng = np.random.default_rng(191)
    L, G = 80, 3000
    genome = bytes(b"ACGT"[i] for i in rng.integers(0, 4, G))
    ref = FastaFile(sequences={b"c": genome})
    for case, var in enumerate([Variant(b"c", 1020, genome[1020:1021], b"A" if genome[1020:1021] != b"A" else b"C"),
                                Variant(b"c", 1020, b"", b"GTGA"), Variant(b"c", 1020, genome[1021:1024], b"")]):
        refHaplotype, testHap, refHap, varHap = refcalls.windowHaplotypes(b"c", 1000, 1040, var, ref, L)
        # variantcaller.pyx:373-374 builds over the padded bounds with read length 0: same stretch, same sequence
        assert (refHap.startPosPy, refHap.endPosPy) == (1000 - 2 * L, 1040 + 2 * L) == (testHap.startPosPy, testHap.endPosPy)
        assert refHap.sequence == refHaplotype.sequence == genome[1000 - 2 * L:1040 + 2 * L]
        assert varHap.sequence == testHap.sequence and varHap.varSeqLenPy == abs(var.nAdded - var.nRemoved)
        seqs, quals, poss = [], [], []
        for r in range(40):
            src = (refHap, varHap)[r % 2].sequence
            o = int(rng.integers(2 * L - 70, 2 * L - 5))
            s = bytearray(src[o:o + L])
            for j in np.flatnonzero(rng.random(L) < 0.02):
                s[j] = b"ACGT"[int(rng.integers(0, 4))]
            seqs.append(bytes(s)); quals.append(bytes(rng.integers(5, 41, L).astype(np.uint8))); poss.append(refHap.startPosPy + o)
        order = np.argsort(poss, kind="stable")
        seqs = [seqs[i] for i in order]; quals = [quals[i] for i in order]; poss = [poss[i] for i in order]
        pos = np.array(poss, np.int32)
        buf = cwindow.bamReadBuffer(b"c", int(pos.min()), int(pos.max()) + 100, np.frombuffer(b"".join(seqs), np.uint8),
                                    np.frombuffer(b"".join(quals), np.uint8), np.arange(41, dtype=np.int64) * L, pos, pos + L,
                                    np.full(40, 45, np.int32))
        buf.setWindow(int(pos.min()), int(pos.max()) + 1)
        rows = {}
        for name, hp in (("ref", refHap), ("var", varHap)):
            exp = [po.align_no_traceback(seqs[i], quals[i], poss[i], hp.sequence, hp.startPosPy, hp.varSeqLenPy)
                   for i in range(40)]
            rows[name] = np.array([po.likelihood(sc, 45) for sc, _ in exp])
            assert refcalls.perReadLikelihoods(hp, buf) == list(rows[name])            # same doubles
            tab = po.hash_sequence(hp.sequence)
            assert refcalls.anchors(hp, buf) == [po.map_read(seqs[i], len(hp.sequence), tab) for i in range(40)]
        for a, b in (("ref", "ref"), ("ref", "var"), ("var", "var")):
            got = refcalls.calculateDataLikelihood({"ref": refHap, "var": varHap}[a], {"ref": refHap, "var": varHap}[b], buf, 0, 1)
            want = po.genotype_loglik(rows[a], rows[b])
            assert abs(got - want) <= 1e-9 * abs(want)
    calign.closeDevice()
