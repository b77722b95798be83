# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""Caller / Population of platypus/cpopulation.pyx with the numeric work on the GPU.

``Caller.setup`` (cpopulation.pyx:658-739) scores every haplotype of the window
against every individual's window reads in one alignment launch; ``Caller.call``
(cpopulation.pyx:741-805) then runs genotype log-likelihoods, their
normalisation and the EM (cpopulation.pyx:807-890) on the device over the
likelihoods that are still resident there, and ``Population.calculatePosterior``
(cpopulation.pyx:515-632) evaluates a variant against those results, also on
the device.  Nothing is computed on the host; there is no CPU fallback.

The VCF-formatting methods of the reference's Population (vcfINFO, vcfFILTER,
vcfVARIANTS, cpopulation.pyx:189-513) are host string work downstream of the
path and are not mirrored.
"""
from libc.stdint cimport int32_t, int64_t, uint8_t
from libc.stdlib cimport malloc, calloc, free
from libc.string cimport memcpy, memset
from libc.math cimport exp, floor
import time

from platypus_amd.calign cimport deviceContext, ensureReadLength
from platypus_amd.calign import PlatGpuError
from platypus_amd.chaplotype cimport Haplotype, prefetchWindowC
from platypus_amd.cgenotype cimport DiploidGenotype, generateAllGenotypesFromHaplotypeList
from platypus_amd.calign cimport setParams
from platypus_amd.samtoolsWrapper cimport cAlignedRead
from platypus_amd.cwindow cimport bamReadBuffer
from platypus_amd.platgpu cimport *


cdef int check(pg_ctx* ctx, int rc) except -1:
    if rc != PG_OK:
        msg = pg_last_error_string(ctx)
        raise PlatGpuError("libplatgpu error %d: %s" % (rc, msg.decode() if msg != NULL else ""))
    return 0


# Results of the genotype / EM kernels stay on the device until the next windows are
# uploaded; a Population remembers which upload it came from.
cdef long g_generation = 0
cdef double mLTOT = -0.23025850929940459


cdef class Population:
    """Result of Caller.call: frequencies, normalised genotype likelihoods, calls."""

    def __dealloc__(self):
        cdef int index
        if self.genotypePosteriors != NULL:
            for index in range(self.nIndividuals):
                free(self.genotypePosteriors[index])
            free(self.genotypePosteriors)
        if self.haplotypeIndexes != NULL:
            if self.nGenotypes > 0:
                free(self.haplotypeIndexes[0])                 # the rows are one block
            free(self.haplotypeIndexes)
        free(self.frequencies)
        free(self.nReads)

    cdef double calculatePosterior(self, object var) except? -1.0:
        """Posterior that `var` segregates (cpopulation.pyx:515-632).  `var` needs
        calculatePrior() and must be comparable with the entries of Haplotype.variants."""
        cdef pg_ctx* ctx = deviceContext()
        cdef int32_t win = self.windowIndex
        cdef double prior = var.calculatePrior()
        cdef double out = 0.0
        cdef int i
        cdef Haplotype hap
        if self.nHaplotypes > 1000:
            raise ValueError("Too many haplotypes (%s). This should never happen" % self.nHaplotypes)
        if self.generation != g_generation:
            raise PlatGpuError("this Population's results have left the device (another window or batch was "
                               "called since); evaluate variants before the next call")
        cdef uint8_t* mask = <uint8_t*>malloc(self.nHaplotypes)
        try:
            for i in range(self.nHaplotypes):
                hap = self.haplotypes[i]
                mask[i] = 1 if var in hap.variants else 0        # cpopulation.pyx:562
            check(ctx, pg_variant_posteriors(ctx, 1, &win, mask, &prior, &out))
        finally:
            free(mask)
        return out

    def calculatePosteriorPy(self, var):
        return self.calculatePosterior(var)

    def frequenciesPy(self):
        return [self.frequencies[i] for i in range(self.nHaplotypes)]

    def genotypePosteriorsPy(self):
        if self.nIndividuals > 0 and self.genotypePosteriors[0] == NULL:
            raise ValueError("the genotype posteriors of this batch were left on the device (keepPosteriors=False)")
        return [[self.genotypePosteriors[i][g] for g in range(self.nGenotypes)] for i in range(self.nIndividuals)]

    def haplotypeIndexesPy(self):
        return [(self.haplotypeIndexes[g][0], self.haplotypeIndexes[g][1]) for g in range(self.nGenotypes)]


cdef class Caller:
    def __init__(self, options=None):
        self.options = options
        self.nReads = NULL

    def __dealloc__(self):
        free(self.nReads)

    cdef setup(self, list haplotypes, list genotypes, int nInd, int ploidy, int verbosity, list readBuffers):
        """cpopulation.pyx:658-739.  The genotype list must be the one
        generateAllGenotypesFromHaplotypeList makes of `haplotypes` (the only list the
        reference ever passes, variantcaller.pyx:153-155,178-185)."""
        cdef int i, g, a, b
        cdef DiploidGenotype gt
        cdef bamReadBuffer buf
        assert ploidy == 2
        self.ploidy = ploidy
        self.nGenotypes = len(genotypes)
        self.haplotypes = haplotypes
        self.genotypes = genotypes
        self.nHaplotypes = len(haplotypes)
        self.nIndividuals = nInd
        self.readBuffers = readBuffers
        self.verbosity = verbosity
        if self.nGenotypes != self.nHaplotypes * (self.nHaplotypes + 1) // 2:
            raise ValueError("genotypes must be all pairs of the haplotypes (cgenotype.pyx:90-96)")
        g = 0
        for a in range(self.nHaplotypes):
            for b in range(a, self.nHaplotypes):
                gt = genotypes[g]
                if gt.hap1 is not haplotypes[a] or gt.hap2 is not haplotypes[b]:
                    raise ValueError("genotypes must be in the order of generateAllGenotypesFromHaplotypeList")
                g += 1
        free(self.nReads)
        self.nReads = <int*>malloc((nInd if nInd > 0 else 1) * sizeof(int))
        for i in range(nInd):
            buf = readBuffers[i]
            self.nReads[i] = <int>(buf.windowEnd - buf.windowStart)   # cpopulation.pyx:703-704
        # one launch: all haplotypes x all individuals' window reads; the likelihood matrix
        # stays on the device (and fills the haplotypes' caches, as alignReads would)
        prefetchWindowC(haplotypes, readBuffers, True)

    cdef Population call(self, int maxIters=100):
        """cpopulation.pyx:741-805."""
        cdef pg_ctx* ctx = deviceContext()
        cdef int nH = self.nHaplotypes, nI = self.nIndividuals, nG = self.nGenotypes
        cdef int i, g, a, b
        cdef int32_t n_haps = nH, n_inds = nI, iters = 0
        cdef int64_t like_base = 0
        cdef double change = 0.0
        cdef int32_t* off = <int32_t*>malloc((nI + 1) * sizeof(int32_t))
        cdef int32_t* best = <int32_t*>malloc((nI + 1) * sizeof(int32_t))
        cdef double* val = <double*>malloc((nI + 1) * sizeof(double))
        cdef double* post = <double*>malloc((<size_t>nI * nG + 1) * sizeof(double))
        cdef Population pop = Population.__new__(Population)
        pop.nHaplotypes = nH; pop.nIndividuals = nI; pop.nGenotypes = nG
        pop.haplotypes = self.haplotypes; pop.readBuffers = self.readBuffers; pop.verbosity = self.verbosity
        pop.frequencies = <double*>calloc(nH, sizeof(double))
        pop.nReads = <int*>malloc((nI if nI > 0 else 1) * sizeof(int))
        pop.genotypePosteriors = <double**>calloc(nI, sizeof(double*))
        pop.haplotypeIndexes = <int**>calloc(nG, sizeof(int*))
        try:
            off[0] = 0
            for i in range(nI):
                off[i + 1] = off[i] + self.nReads[i]
                pop.nReads[i] = self.nReads[i]
            global g_generation
            g_generation += 1
            pop.generation = g_generation
            pop.windowIndex = 0
            check(ctx, pg_windows_upload(ctx, 1, &n_haps, &n_inds, &like_base, off))
            check(ctx, pg_caller_launch(ctx, maxIters))
            check(ctx, pg_caller_fetch(ctx, pop.frequencies, &iters, &change, best, val, post))
            pop.nIterations = iters
            pop.maxChange = change
            pop.genotypeCalls = []
            pop.genotypeLikelihoods = []
            for i in range(nI):
                pop.genotypePosteriors[i] = <double*>malloc(nG * sizeof(double))
                for g in range(nG):
                    pop.genotypePosteriors[i][g] = post[<size_t>i * nG + g]
                if best[i] < 0:                                   # cpopulation.pyx:777-780
                    pop.genotypeCalls.append(None)
                    pop.genotypeLikelihoods.append(0.0)
                else:
                    pop.genotypeCalls.append(self.genotypes[best[i]])
                    pop.genotypeLikelihoods.append(val[i])
            fillHaplotypeIndexes(pop.haplotypeIndexes, nH, nG)
        finally:
            free(off); free(best); free(val); free(post)
        return pop

    def setupPy(self, list haplotypes, list genotypes, list readBuffers, int verbosity=0):
        self.setup(haplotypes, genotypes, len(readBuffers), 2, verbosity, readBuffers)

    def callPy(self, int maxIters=100):
        return self.call(maxIters)


cdef inline long roundHalfUp(double x) noexcept:
    """round() of the Python 2 the reference runs on (halves away from zero), for x >= 0; the fraction x - floor(x) is
    exact, x + 0.5 would not always be"""
    cdef double r = floor(x)
    return <long>r + 1 if x - r >= 0.5 else <long>r


cdef int fillHaplotypeIndexes(int** rows, int nH, int nG) except -1:
    """[genotype] -> its two haplotype indexes, the rows cut from one block (freed through rows[0])"""
    cdef int a, b, g = 0
    cdef int* block
    if nG <= 0:
        return 0
    block = <int*>malloc(2 * <size_t>nG * sizeof(int))
    if block == NULL:
        raise MemoryError()
    for a in range(nH):
        for b in range(a, nH):
            rows[g] = block + 2 * g
            rows[g][0] = a
            rows[g][1] = b
            g += 1
    return 0


cdef Population makePopulation(list haplotypes, list genotypes, list readBuffers, int nI, const int32_t* nReads,
                               const double* freq, const double* post, const int32_t* best, const double* val,
                               int iters, double change, int windowIndex):
    """A Population from the flat results of one window of a batch."""
    cdef int nH = len(haplotypes), nG = nH * (nH + 1) // 2, i, g, a, b
    cdef Population pop = Population.__new__(Population)
    pop.nHaplotypes = nH; pop.nIndividuals = nI; pop.nGenotypes = nG
    pop.haplotypes = haplotypes; pop.readBuffers = readBuffers; pop.verbosity = 0
    pop.frequencies = <double*>malloc((nH if nH > 0 else 1) * sizeof(double))
    memcpy(pop.frequencies, freq, nH * sizeof(double))
    pop.nReads = <int*>malloc((nI if nI > 0 else 1) * sizeof(int))
    pop.genotypePosteriors = <double**>calloc(nI, sizeof(double*))
    pop.haplotypeIndexes = <int**>calloc(nG, sizeof(int*))
    pop.nIterations = iters
    pop.maxChange = change
    pop.windowIndex = windowIndex
    pop.generation = g_generation
    pop.genotypeCalls = []
    pop.genotypeLikelihoods = []
    cdef dict made = {}
    for i in range(nI):
        pop.nReads[i] = nReads[i]
        if post != NULL:
            pop.genotypePosteriors[i] = <double*>malloc(nG * sizeof(double))
            memcpy(pop.genotypePosteriors[i], post + <size_t>i * nG, nG * sizeof(double))
        if best[i] < 0:                                   # cpopulation.pyx:777-780
            pop.genotypeCalls.append(None)
            pop.genotypeLikelihoods.append(0.0)
        else:
            if genotypes is not None:
                pop.genotypeCalls.append(genotypes[best[i]])
            else:                                         # only the called genotypes are made, once each (cgenotype.pyx:90-96: first index major)
                gt = made.get(best[i])
                if gt is None:
                    a = 0
                    g = best[i]
                    while g >= nH - a:
                        g -= nH - a
                        a += 1
                    gt = made[best[i]] = DiploidGenotype(2, [haplotypes[a], haplotypes[a + g]])
                pop.genotypeCalls.append(gt)
            pop.genotypeLikelihoods.append(val[i])
    fillHaplotypeIndexes(pop.haplotypeIndexes, nH, nG)
    return pop


cdef class WindowBatch:
    """Collects windows and calls them all in one device batch.

    The reference handles one window at a time: set the window pointers, build the
    haplotypes, Caller.setup, Caller.call, write the calls (variantcaller.pyx:116-187).
    Here the loop is split: ``add`` takes a window's haplotypes and the individuals'
    read buffers (whose window pointers are snapshotted, so the caller may move on to the
    next window), ``run`` uploads everything collected, runs ONE alignment launch and ONE
    genotype/EM launch for all windows, and returns a Population per window, in the order
    of the ``add`` calls.  ``posteriors`` evaluates many (window, variant) pairs in one call.
    """

    def __init__(self):
        self.windows = []
        self.populations = []
        self.pairBase = []
        self.timings = {}

    def __len__(self):
        return len(self.windows)

    def add(self, list haplotypes, list readBuffers, start=None, end=None):
        """Returns the window's index in this batch.  The window's reads are the buffers' current window
        ranges, or those of [start, end) when given (platypus/cwindow.pyx:242-267); they are kept as
        (first read, count) per buffer.  A window without any read is refused (the reference divides by
        zero there, cpopulation.pyx:883)."""
        cdef bamReadBuffer buf
        cdef int i, n, tot = 0
        cdef list counts = []
        cdef list firsts = []
        cdef Haplotype hap
        if len(haplotypes) < 1 or len(haplotypes) > 1000:
            raise ValueError("a window needs 1..1000 haplotypes (cpopulation.pyx:546)")
        for hp in haplotypes:
            # what the device refuses, found here so that only this window is lost (the reference skips a window
            # that fails, variantcaller.pyx:189-191): the 7-mer table of calign.pyx:112,132 ends at 16383 bases,
            # and below 15 the reference reads past its own 'N' extension (calign.pyx:73-77)
            hap = hp
            if hap.varSeqLen != 0 and (hap.hapLen < 15 or hap.hapLen > 16383):
                raise ValueError("a haplotype with indels of %d bases (15..16383 are taken)" % hap.hapLen)
        if len(readBuffers) > 4096:
            raise ValueError("more than 4096 read buffers in one window")
        for i in range(len(readBuffers)):
            buf = readBuffers[i]
            if buf.maxReadLen > 16384:
                raise ValueError("a read of %d bases (at most 16384 are taken)" % buf.maxReadLen)
            if start is not None:
                buf.setWindowPointers(start, end)
            n = <int>(buf.windowEnd - buf.windowStart)
            counts.append(n)
            firsts.append(<int>(buf.windowStart - buf.cReads))
            tot += n
        if tot == 0:
            raise ValueError("the window has no reads")
        self.windows.append((haplotypes, readBuffers, counts, firsts))
        return len(self.windows) - 1

    def coverage(self, int w, int varPos):
        """(TC, TCR, TCF) of cpopulation.pyx:322-346: the window's reads, over all individuals, that
        span position varPos (pos <= varPos <= end), and how many of them are on the reverse /
        forward strand."""
        cdef cAlignedRead* rd
        cdef bamReadBuffer buf
        cdef int tc = 0, tcr = 0, bi, r, lo, n
        haps, bufs, counts, firsts = self.windows[w]
        for bi in range(len(bufs)):
            buf = bufs[bi]; lo = firsts[bi]; n = counts[bi]
            for r in range(n):
                rd = buf.cReads[lo + r]
                if rd.pos <= varPos and rd.end >= varPos:
                    tc += 1
                    if rd.isReverse:
                        tcr += 1
        return tc, tcr, tc - tcr

    def siteStatistics(self, int w, int varPos, bytes added, int badReadsWindow):
        """The read-level counts of Population.vcfINFO for one variant (cpopulation.pyx:274-346) over the
        window's reads of all individuals: (TC, TCR, TCF, weighted TCR, weighted TCF, minimum base
        qualities).  The last item lists, for every read that spans varPos and shows the variant's
        single added base there, the smallest quality in a window of badReadsWindow bases around it
        (empty unless `added` is one base, cpopulation.pyx:290-310)."""
        cdef cAlignedRead* rd
        cdef bamReadBuffer buf
        cdef int tc = 0, tcr = 0, ws, we, k, mn, bi, r, lo, n
        cdef double wr = 0.0, wf = 0.0
        cdef list mins = []
        cdef bint single = len(added) == 1
        cdef char addedChar = added[0] if single else 0
        haps, bufs, counts, firsts = self.windows[w]
        for bi in range(len(bufs)):
            buf = bufs[bi]; lo = firsts[bi]; n = counts[bi]
            for r in range(n):
                rd = buf.cReads[lo + r]
                if rd.pos <= varPos and rd.end >= varPos:
                    tc += 1
                    if rd.isReverse:
                        tcr += 1
                        wr += 1.0 - exp(mLTOT * rd.mapq)
                    else:
                        wf += 1.0 - exp(mLTOT * rd.mapq)
                    if single and varPos - rd.pos < rd.rlen and rd.seq[varPos - rd.pos] == addedChar:
                        ws = max(0, varPos - rd.pos - (badReadsWindow - 1) // 2)
                        we = min(rd.rlen, varPos - rd.pos + (badReadsWindow - 1) // 2)
                        mn = 100
                        for k in range(ws, we):
                            if rd.qual[k] < mn:
                                mn = rd.qual[k]
                        mins.append(mn)
        return tc, tcr, tc - tcr, wr, wf, mins

    def siteInfo(self, int w, variant, double freq, refFile, int badReadsWindow, list minQuals, list fwdWeight, binomialSum):
        """The INFO statistics of one variant (cpopulation.pyx:239-389, chaplotype.pyx:366-432) in one typed pass:
        (TC, TCR, TCF, HP, SC, RMP, MMLQ, FPV, RPV, ABPV), the three p-values as their '%1.2e' texts.  `minQuals` and
        `fwdWeight` are the window's running list of minimum base qualities and its running weighted forward coverage:
        the reference never clears the first between the variants of a window (cpopulation.pyx:231,310) and resets the
        second under a misspelt name (cpopulation.pyx:269), so both carry over here as they do there.  `binomialSum(upto,
        size, prob)` is the caller's sum of binomial terms.  The reference sequence around the variant is read from the
        file's cache where the request lies inside it (fastafile.pyx:148-150), through getSequence otherwise."""
        cdef int refPos = variant.refPos, tc, tcr, tcf, nl, nr, n, hp, lo, hi
        cdef double wr, wf, nw, fwdW = variant.nFwdReadsWeighted, revW = variant.nRevReadsWeighted, p
        cdef bytes added = variant.added, seq, left, right
        cdef const unsigned char* c
        cdef Py_ssize_t cacheBegin, cacheEnd, a
        tc, tcr, tcf, wr, wf, mins = self.siteStatistics(w, refPos, added, badReadsWindow)
        fwdWeight[0] += wf
        wf = fwdWeight[0]
        if mins:
            minQuals.extend(mins)
            minQuals.sort()
        med = minQuals[len(minQuals) // 2] if minQuals else 100
        # homopolymer run touching the position (chaplotype.pyx:366-402) and the 21 bases around it
        cache = refFile._cache
        if (cache is not None and cache[0] == variant.refName and refPos - 20 >= cache[1] and refPos + 21 < cache[2]):
            seq = cache[3]
            cacheBegin = cache[1]
            c = <const unsigned char*>seq
            a = refPos - cacheBegin                                   # index of the variant's position
            nl = 1                                                    # bases equal to the one just left of it, leftwards (20 at most)
            while nl < 20 and c[a - 1 - nl] == c[a - 1]:
                nl += 1
            nr = 1
            while nr < 20 and c[a + 1 + nr] == c[a + 1]:
                nr += 1
            hp = max(nl, nr) if c[a - 1] != c[a + 1] else nl + nr
            sc = seq[a - 10:a + 11]
        else:
            left = refFile.getSequence(variant.refName, refPos - 20, refPos)
            right = refFile.getSequence(variant.refName, refPos + 1, refPos + 21)
            if len(left) == 0 or len(right) == 0:
                hp = 0
            else:
                nl = <int>(len(left) - len(left.rstrip(left[-1:])))
                nr = <int>(len(right) - len(right.lstrip(right[:1])))
                hp = max(nl, nr) if left[-1:] != right[:1] else nl + nr
            sc = refFile.getSequence(variant.refName, refPos - 10, refPos + 11)
        nw = fwdW + revW
        if tc > 0 and nw / <double>tc >= freq / 2.0:                  # cpopulation.pyx:367-373
            ab = 1.0
        else:
            ab = binomialSum(roundHalfUp(nw), tc, freq / 2.0)
        p = freq if freq < 0.99 else 0.99
        fwd = binomialSum(roundHalfUp(fwdW), roundHalfUp(wf), p)
        rev = binomialSum(roundHalfUp(revW), roundHalfUp(wr), p)
        return (tc, tcr, tcf, hp, sc, round(variant.rmsReadPos, 2),
                med if (variant.nAdded == 1 and variant.nRemoved == 1) else None,
                "%1.2e" % fwd, "%1.2e" % rev, "%1.2e" % ab)

    def readCounts(self, int w):
        """reads per individual of window w"""
        return list(self.windows[w][2])

    def run(self, int maxIters=100, bint keepPosteriors=True):
        """One alignment launch and one genotype/EM launch for all windows added so far; returns their Populations.
        keepPosteriors=False leaves the genotype posteriors on the device (variant posteriors and genotype calls
        are computed there anyway): a caller that only needs calls saves the largest transfer of the batch."""
        cdef int nW = len(self.windows), w, h, i, r, n, lo, cnt, longest = 1
        for w in range(nW):
            for b in self.windows[w][1]:
                longest = max(longest, (<bamReadBuffer>b).maxReadLen)
        ensureReadLength(longest)                         # may replace the context: before anything is uploaded
        cdef pg_ctx* ctx = deviceContext()
        cdef int64_t nHapTot = 0, nReadTot = 0, nIndTot = 0, hapBytes = 0, readBytes = 0, nCells = 0
        cdef Haplotype hap
        cdef cAlignedRead* rd
        cdef bamReadBuffer buf
        cdef int64_t nInc = 0, nPairs = 0, pi = 0
        cdef dict slots = {}
        cdef list slotBufs = []
        cdef int slot
        cdef int32_t* m
        cdef int32_t* maps[4096]
        if nW == 0:
            return []
        t0 = time.perf_counter()
        for w in range(nW):
            haps, bufs, counts, firsts = self.windows[w]
            nHapTot += len(haps)
            nIndTot += len(bufs)
            nCells += <int64_t>len(bufs) * (len(haps) * (len(haps) + 1) // 2)
            for hp in haps:
                hap = hp
                hapBytes += hap.hapLen
            for i in range(len(bufs)):
                buf = bufs[i]; lo = firsts[i]; n = counts[i]
                nInc += n
                nPairs += <int64_t>len(haps) * n
                slot = slots.get(id(buf), -1)
                if slot < 0:                                      # a read that lies in several windows is uploaded once:
                    slot = len(slotBufs)                          # per buffer, a map from its reads to their place in the batch
                    slots[id(buf)] = slot
                    slotBufs.append(buf)
                    if slot >= 4096:
                        raise ValueError("more than 4096 read buffers in one batch")
                    maps[slot] = <int32_t*>malloc((buf.nReads + 1) * sizeof(int32_t))
                    memset(maps[slot], 0xff, (buf.nReads + 1) * sizeof(int32_t))
                m = maps[slot]
                for r in range(n):
                    if m[lo + r] < 0:
                        m[lo + r] = 0
                        nReadTot += 1
                        readBytes += buf.cReads[lo + r].rlen
        cdef int64_t* hoff = <int64_t*>malloc((nHapTot + 1) * sizeof(int64_t))
        cdef int32_t* hstart = <int32_t*>malloc((nHapTot + 1) * sizeof(int32_t))
        cdef int32_t* hvl = <int32_t*>malloc((nHapTot + 1) * sizeof(int32_t))
        cdef char* hbuf = <char*>malloc(hapBytes + 1)
        cdef int64_t* roff = <int64_t*>malloc((nReadTot + 1) * sizeof(int64_t))
        cdef int32_t* rpos = <int32_t*>malloc((nReadTot + 1) * sizeof(int32_t))
        cdef int32_t* rmq = <int32_t*>malloc((nReadTot + 1) * sizeof(int32_t))
        cdef char* sbuf = <char*>malloc(readBytes + 1)
        cdef char* qbuf = <char*>malloc(readBytes + 1)
        cdef int32_t* b_hlo = <int32_t*>malloc((nW + 1) * sizeof(int32_t))
        cdef int32_t* b_nh = <int32_t*>malloc((nW + 1) * sizeof(int32_t))
        cdef int32_t* b_rlo = <int32_t*>malloc((nW + 1) * sizeof(int32_t))
        cdef int32_t* b_nr = <int32_t*>malloc((nW + 1) * sizeof(int32_t))
        cdef int32_t* w_ni = <int32_t*>malloc((nW + 1) * sizeof(int32_t))
        cdef int64_t* base = <int64_t*>malloc((nW + 1) * sizeof(int64_t))
        cdef int32_t* ioff = <int32_t*>malloc((nIndTot + nW + 1) * sizeof(int32_t))
        cdef int32_t* nreads = <int32_t*>malloc((nIndTot + 1) * sizeof(int32_t))
        cdef double* freq = <double*>malloc((nHapTot + 1) * sizeof(double))
        cdef int32_t* iters = <int32_t*>malloc((nW + 1) * sizeof(int32_t))
        cdef double* change = <double*>malloc((nW + 1) * sizeof(double))
        cdef int32_t* best = <int32_t*>malloc((nIndTot + 1) * sizeof(int32_t))
        cdef double* val = <double*>malloc((nIndTot + 1) * sizeof(double))
        cdef double* post = <double*>malloc(((nCells if keepPosteriors else 0) + 1) * sizeof(double))
        cdef int64_t hi = 0, ri = 0, ii = 0, oi = 0, ci = 0, nCellsDP = 0
        cdef int32_t* p_hap = <int32_t*>malloc((nPairs + 1) * sizeof(int32_t))
        cdef int32_t* p_read = <int32_t*>malloc((nPairs + 1) * sizeof(int32_t))
        cdef int hl
        cdef float kms[5]
        cdef int64_t votes[3]
        cdef list out = []
        global g_generation
        try:
            hoff[0] = 0
            roff[0] = 0
            for slot in range(len(slotBufs)):                     # the batch's reads: every buffer's used reads, in buffer order
                buf = slotBufs[slot]
                m = maps[slot]
                for r in range(buf.nReads):
                    if m[r] >= 0:
                        rd = buf.cReads[r]
                        m[r] = <int32_t>ri
                        memcpy(sbuf + roff[ri], rd.seq, rd.rlen)
                        memcpy(qbuf + roff[ri], rd.qual, rd.rlen)
                        roff[ri + 1] = roff[ri] + rd.rlen
                        rpos[ri] = rd.pos
                        rmq[ri] = rd.mapq
                        ri += 1
            for w in range(nW):
                haps, bufs, counts, firsts = self.windows[w]
                b_hlo[w] = <int32_t>hi
                b_nh[w] = len(haps)
                w_ni[w] = len(bufs)
                base[w] = pi
                for hp in haps:
                    hap = hp
                    memcpy(hbuf + hoff[hi], hap.cHaplotypeSequence, hap.hapLen)
                    hoff[hi + 1] = hoff[hi] + hap.hapLen
                    hstart[hi] = hap.startPos
                    hvl[hi] = hap.varSeqLen
                    hi += 1
                ioff[oi] = 0
                n = 0
                for i in range(len(bufs)):
                    n += <int>counts[i]
                    ioff[oi + i + 1] = n
                    nreads[ii + i] = <int>counts[i]
                for i in range(len(bufs)):                        # the first haplotype's pairs: the window's reads by individual
                    buf = bufs[i]; lo = firsts[i]; cnt = counts[i]
                    m = maps[<int>slots[id(buf)]]
                    for r in range(cnt):
                        p_hap[pi] = b_hlo[w]
                        p_read[pi] = m[lo + r]
                        nCellsDP += <int64_t>len(haps) * 16 * (roff[m[lo + r] + 1] - roff[m[lo + r]] + 8)   # align.c:106
                        pi += 1
                for hl in range(1, len(haps)):                    # the other haplotypes: the same reads again
                    for r in range(n):
                        p_hap[pi] = b_hlo[w] + hl
                        p_read[pi] = p_read[base[w] + r]
                        pi += 1
                oi += len(bufs) + 1
                ii += len(bufs)
                b_nr[w] = n
            hap = self.windows[0][0][0]
            setParams(hap.cHomopolQ, 3, 2)                       # chaplotype.pyx:264
            t1 = time.perf_counter()
            check(ctx, pg_haps_upload(ctx, <int>nHapTot, hbuf, hoff, hstart, hvl))
            check(ctx, pg_reads_upload(ctx, <int>nReadTot, sbuf, qbuf, roff, rpos, rmq))
            check(ctx, pg_pairs_upload(ctx, nPairs, p_hap, p_read))
            t2 = time.perf_counter()
            check(ctx, pg_align_launch(ctx))
            g_generation += 1
            check(ctx, pg_windows_upload(ctx, nW, b_nh, w_ni, base, ioff))
            check(ctx, pg_caller_launch(ctx, maxIters))
            check(ctx, pg_caller_fetch(ctx, freq, iters, change, best, val, post if keepPosteriors else NULL))
            t3 = time.perf_counter()
            for w in range(5):                                    # anchor, alignment, genotype, setup, EM (platgpu.h)
                kms[w] = 0.0
                pg_kernel_time_ms(ctx, w, &kms[w])
            votes[0] = votes[1] = votes[2] = 0
            pg_anchor_stats(ctx, votes)                           # pairs anchored by votes; of those, counted by a wavefront
            self.pairBase = [base[w] for w in range(nW)]          # pair index of (haplotype 0, read 0) of every window
            hi = 0; ii = 0; ci = 0
            for w in range(nW):
                haps, bufs, counts, firsts = self.windows[w]
                out.append(makePopulation(haps, None, bufs, len(bufs), nreads + ii, freq + hi, post + ci if keepPosteriors else NULL,
                                          best + ii, val + ii, iters[w], change[w], w))
                hi += len(haps)
                ii += len(bufs)
                ci += <int64_t>len(bufs) * (len(haps) * (len(haps) + 1) // 2)
            t4 = time.perf_counter()
            self.timings = dict(pack=t1 - t0, upload=t2 - t1, device=t3 - t2, unpack=t4 - t3,
                                windows=nW, haplotypes=int(nHapTot), reads=int(nReadTot), read_incidences=int(nInc), pairs=int(nPairs),
                                cells=int(nCellsDP), vote_pairs=int(votes[0]), vote_named=int(votes[1]), vote_full=int(votes[2]), anchor_ms=kms[0], align_ms=kms[1], setup_ms=kms[3], em_ms=kms[4])
        finally:
            free(hoff); free(hstart); free(hvl); free(hbuf); free(roff); free(rpos); free(rmq); free(sbuf); free(qbuf)
            free(b_hlo); free(b_nh); free(b_rlo); free(b_nr); free(w_ni); free(base); free(ioff); free(nreads)
            free(p_hap); free(p_read)
            for slot in range(len(slotBufs)):
                free(maps[slot])
            free(freq); free(iters); free(change); free(best); free(val); free(post)
        self.populations = out
        return out

    def siteLikelihoods(self, list items):
        """The second alignment site of the reference, batched: for every emitted single-variant
        site the reference builds a fresh reference haplotype and a fresh variant haplotype and
        computes, per individual, the log-likelihoods of the genotypes AA, AB, BB over the
        window's reads (variantcaller.pyx:371-378,422-427).  `items` is a list of
        (window index, refHap, varHap); all of them are aligned in ONE launch and summed in ONE
        kernel.  Returns, per item, a list over individuals of (AA, AB, BB) -- unrounded; the
        reference rounds to two decimals -- or None for an individual without reads.
        The genotype/EM results of `run` stay valid."""
        cdef pg_ctx* ctx = deviceContext()
        cdef int nItems = len(items), it, i, w, n, k, bi, r, lo, cnt
        cdef int64_t hapBytes = 0, readBytes = 0, nReadTot = 0, nRows = 0, ri = 0, row = 0
        cdef Haplotype hap
        cdef cAlignedRead* rd
        cdef bamReadBuffer buf
        cdef dict readBase = {}
        cdef list order = []
        if nItems == 0:
            return []
        for w, refHap, varHap in items:
            if w not in readBase:
                readBase[w] = -1
                order.append(w)
            hap = refHap
            hapBytes += hap.hapLen
            hap = varHap
            hapBytes += hap.hapLen
            nRows += 3 * len(self.windows[w][1])
        for w in order:
            haps_w, bufs_w, counts_w, firsts_w = self.windows[w]
            for bi in range(len(bufs_w)):
                buf = bufs_w[bi]; lo = firsts_w[bi]; cnt = counts_w[bi]
                nReadTot += cnt
                for r in range(cnt):
                    readBytes += buf.cReads[lo + r].rlen
        cdef int64_t* hoff = <int64_t*>malloc((2 * nItems + 1) * sizeof(int64_t))
        cdef int32_t* hstart = <int32_t*>malloc((2 * nItems + 1) * sizeof(int32_t))
        cdef int32_t* hvl = <int32_t*>malloc((2 * nItems + 1) * sizeof(int32_t))
        cdef char* hbuf = <char*>malloc(hapBytes + 1)
        cdef int64_t* roff = <int64_t*>malloc((nReadTot + 1) * sizeof(int64_t))
        cdef int32_t* rpos = <int32_t*>malloc((nReadTot + 1) * sizeof(int32_t))
        cdef int32_t* rmq = <int32_t*>malloc((nReadTot + 1) * sizeof(int32_t))
        cdef char* sbuf = <char*>malloc(readBytes + 1)
        cdef char* qbuf = <char*>malloc(readBytes + 1)
        cdef int32_t* b_hlo = <int32_t*>malloc((nItems + 1) * sizeof(int32_t))
        cdef int32_t* b_nh = <int32_t*>malloc((nItems + 1) * sizeof(int32_t))
        cdef int32_t* b_rlo = <int32_t*>malloc((nItems + 1) * sizeof(int32_t))
        cdef int32_t* b_nr = <int32_t*>malloc((nItems + 1) * sizeof(int32_t))
        cdef int64_t* base = <int64_t*>malloc((nItems + 1) * sizeof(int64_t))
        cdef int64_t* g1 = <int64_t*>malloc((nRows + 1) * sizeof(int64_t))
        cdef int64_t* g2 = <int64_t*>malloc((nRows + 1) * sizeof(int64_t))
        cdef int32_t* gn = <int32_t*>malloc((nRows + 1) * sizeof(int32_t))
        cdef double* gout = <double*>malloc((nRows + 1) * sizeof(double))
        cdef int64_t bRef, bVar
        cdef list out = []
        try:
            roff[0] = 0
            for w in order:                                       # reads of every window involved, once
                readBase[w] = ri
                haps_w, bufs_w, counts_w, firsts_w = self.windows[w]
                for bi in range(len(bufs_w)):
                    buf = bufs_w[bi]; lo = firsts_w[bi]; cnt = counts_w[bi]
                    for r in range(cnt):
                        rd = buf.cReads[lo + r]
                        memcpy(sbuf + roff[ri], rd.seq, rd.rlen)
                        memcpy(qbuf + roff[ri], rd.qual, rd.rlen)
                        roff[ri + 1] = roff[ri] + rd.rlen
                        rpos[ri] = rd.pos
                        rmq[ri] = rd.mapq
                        ri += 1
            hoff[0] = 0
            for it in range(nItems):
                w, refHap, varHap = items[it]
                for k, hp in enumerate((refHap, varHap)):
                    hap = hp
                    memcpy(hbuf + hoff[2 * it + k], hap.cHaplotypeSequence, hap.hapLen)
                    hoff[2 * it + k + 1] = hoff[2 * it + k] + hap.hapLen
                    hstart[2 * it + k] = hap.startPos
                    hvl[2 * it + k] = hap.varSeqLen
                b_hlo[it] = 2 * it
                b_nh[it] = 2
                b_rlo[it] = <int32_t>readBase[w]
                b_nr[it] = sum(self.windows[w][2])
            hap = items[0][1]
            setParams(hap.cHomopolQ, 3, 2)
            check(ctx, pg_haps_upload(ctx, 2 * nItems, hbuf, hoff, hstart, hvl))
            check(ctx, pg_reads_upload(ctx, <int>nReadTot, sbuf, qbuf, roff, rpos, rmq))
            check(ctx, pg_pairs_blocks(ctx, nItems, b_hlo, b_nh, b_rlo, b_nr, base))
            check(ctx, pg_align_launch(ctx))
            for it in range(nItems):
                w = items[it][0]
                n = 0
                for cnt in self.windows[w][2]:
                    bRef = base[it] + n
                    bVar = base[it] + b_nr[it] + n
                    g1[row] = bRef; g2[row] = bRef; gn[row] = cnt          # AA
                    g1[row + 1] = bRef; g2[row + 1] = bVar; gn[row + 1] = cnt  # AB
                    g1[row + 2] = bVar; g2[row + 2] = bVar; gn[row + 2] = cnt  # BB
                    row += 3
                    n += <int>cnt
            check(ctx, pg_genotype_loglik(ctx, nRows, g1, g2, gn, gout))
            row = 0
            for it in range(nItems):
                w = items[it][0]
                res = []
                for cnt in self.windows[w][2]:
                    res.append((gout[row], gout[row + 1], gout[row + 2]) if cnt > 0 else None)
                    row += 3
                out.append(res)
        finally:
            free(hoff); free(hstart); free(hvl); free(hbuf); free(roff); free(rpos); free(rmq); free(sbuf); free(qbuf)
            free(b_hlo); free(b_nh); free(b_rlo); free(b_nr); free(base); free(g1); free(g2); free(gn); free(gout)
        return out

    def siteLikelihoodsResident(self, list items):
        """siteLikelihoods for sites whose two haplotypes are among their window's own: `items` is a list of
        (window index, index of the reference haplotype, index of the variant haplotype) into the window's
        haplotype list.  The fresh haplotypes the reference builds there (variantcaller.pyx:371-378) span the
        window's haplotypes' coordinates and carry the same variants, i.e. they are the same sequences against
        the same reads: their likelihood rows are the ones `run` left on the device, and only the AA / AB / BB
        sums are computed.  Valid until the next alignment launch of this process."""
        cdef pg_ctx* ctx = deviceContext()
        cdef int nItems = len(items), it, w, hr, hv, cnt
        cdef int64_t nRows = 0, row = 0, n, nTot, b0, bRef, bVar
        cdef Population pop
        cdef list out = []
        if nItems == 0:
            return []
        for w, hr, hv in items:
            pop = self.populations[w]
            if pop.generation != g_generation:
                raise PlatGpuError("the batch's results have left the device")
            nRows += 3 * len(self.windows[w][1])
        cdef int64_t* g1 = <int64_t*>malloc((nRows + 1) * sizeof(int64_t))
        cdef int64_t* g2 = <int64_t*>malloc((nRows + 1) * sizeof(int64_t))
        cdef int32_t* gn = <int32_t*>malloc((nRows + 1) * sizeof(int32_t))
        cdef double* gout = <double*>malloc((nRows + 1) * sizeof(double))
        try:
            for it in range(nItems):
                w, hr, hv = items[it]
                counts = self.windows[w][2]
                nTot = sum(counts)
                b0 = self.pairBase[w]
                n = 0
                for cnt in counts:
                    bRef = b0 + hr * nTot + n
                    bVar = b0 + hv * nTot + n
                    g1[row] = bRef; g2[row] = bRef; gn[row] = cnt          # AA
                    g1[row + 1] = bRef; g2[row + 1] = bVar; gn[row + 1] = cnt  # AB
                    g1[row + 2] = bVar; g2[row + 2] = bVar; gn[row + 2] = cnt  # BB
                    row += 3
                    n += cnt
            check(ctx, pg_genotype_loglik(ctx, nRows, g1, g2, gn, gout))
            row = 0
            for it in range(nItems):
                res = []
                for cnt in self.windows[items[it][0]][2]:
                    res.append((gout[row], gout[row + 1], gout[row + 2]) if cnt > 0 else None)
                    row += 3
                out.append(res)
        finally:
            free(g1); free(g2); free(gn); free(gout)
        return out

    def posteriors(self, list items):
        """[(window index, variant), ...] -> posteriors (cpopulation.pyx:515-632), one device call."""
        cdef pg_ctx* ctx = deviceContext()
        cdef int nV = len(items), v, k, nH
        cdef int64_t tot = 0
        cdef Population pop
        cdef Haplotype hap
        if nV == 0:
            return []
        for w, var in items:
            pop = self.populations[w]
            if pop.generation != g_generation:
                raise PlatGpuError("the batch's results have left the device")
            tot += pop.nHaplotypes
        cdef int32_t* win = <int32_t*>malloc(nV * sizeof(int32_t))
        cdef double* prior = <double*>malloc(nV * sizeof(double))
        cdef double* res = <double*>malloc(nV * sizeof(double))
        cdef uint8_t* mask = <uint8_t*>malloc(tot + 1)
        tot = 0
        try:
            for v in range(nV):
                w, var = items[v]
                pop = self.populations[w]
                win[v] = w
                prior[v] = var.calculatePrior()
                for k in range(pop.nHaplotypes):
                    hap = pop.haplotypes[k]
                    mask[tot + k] = 1 if var in hap.variants else 0
                tot += pop.nHaplotypes
            check(ctx, pg_variant_posteriors(ctx, nV, win, mask, prior, res))
            return [res[v] for v in range(nV)]
        finally:
            free(win); free(prior); free(res); free(mask)
