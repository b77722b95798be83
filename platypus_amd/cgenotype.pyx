# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""Diploid genotypes over GPU-scored haplotypes (platypus/cgenotype.pyx).

``calculateDataLikelihood`` keeps the reference's signature (cgenotype.pyx:48-63):
the two likelihood rows come from Haplotype.alignReads (device-scored, cached) and
their log-sum is taken on the device as well -- one small launch per call, so it is
the compatible form, not the fast one.  ``genotypeLogLikelihoods`` is the batched
form of Caller.setup's double loop (cpopulation.pyx:698-725): one alignment launch
for the window, one kernel for all (individual, genotype) sums over the likelihoods
that stayed on the device.  Nothing is summed on the host.
"""
import numpy as np

from libc.stdint cimport int32_t, int64_t
from libc.stdlib cimport malloc, free

from platypus_amd.chaplotype cimport Haplotype, prefetchWindowC
from platypus_amd.cwindow cimport bamReadBuffer
from platypus_amd.samtoolsWrapper cimport cAlignedRead
from platypus_amd.calign cimport deviceContext
from platypus_amd.calign import PlatGpuError
from platypus_amd.platgpu cimport *


cdef class DiploidGenotype:
    def __init__(self, int ploidy, list haplotypes):
        if ploidy != 2 or len(haplotypes) != 2:
            raise ValueError("Diploid genotype must have ploidy == 2. Supplied ploidy = %s. nHaplotypes = %s"
                             % (ploidy, len(haplotypes)))
        self.haplotypes = haplotypes
        self.hap1 = haplotypes[0]
        self.hap2 = haplotypes[1]

    cdef double calculateDataLikelihood(DiploidGenotype self, cAlignedRead** start, cAlignedRead** end,
                                        int individualIndex, int nReads, int nIndividuals) except? -1.0:
        cdef double likelihood = 0.0
        cdef double* arr1 = self.hap1.alignReads(individualIndex, start, end, nReads, nIndividuals)
        cdef double* arr2 = self.hap2.alignReads(individualIndex, start, end, nReads, nIndividuals)
        cdef pg_ctx* ctx = deviceContext()
        cdef int rc = pg_genotype_loglik_rows(ctx, nReads, arr1, arr2, &likelihood)
        if rc != PG_OK:
            msg = pg_last_error_string(ctx)
            raise PlatGpuError("libplatgpu error %d: %s" % (rc, msg.decode() if msg != NULL else ""))
        return likelihood

    def calculateDataLikelihoodPy(self, bamReadBuffer buf, int individualIndex, int nIndividuals):
        return self.calculateDataLikelihood(buf.windowStart, buf.windowEnd, individualIndex,
                                            <int>(buf.windowEnd - buf.windowStart), nIndividuals)


cdef list generateAllGenotypesFromHaplotypeList(int ploidy, list haplotypes):
    """platypus/cgenotype.pyx:66-101: all unordered pairs, first index major."""
    cdef list genotypes = []
    cdef int nHaplotypes = len(haplotypes), indexOne, indexTwo
    if ploidy != 2:
        raise ValueError("Platypus only works for diploid genotypes")
    for indexOne in range(nHaplotypes):
        for indexTwo in range(indexOne, nHaplotypes):
            genotypes.append(DiploidGenotype(ploidy, [haplotypes[indexOne], haplotypes[indexTwo]]))
    return genotypes


def allGenotypes(list haplotypes):
    return generateAllGenotypesFromHaplotypeList(2, haplotypes)


def genotypeLogLikelihoods(list haplotypes, list genotypes, list readBuffers):
    """log-likelihood[individual, genotype] as Caller.setup computes it
    (cpopulation.pyx:698-725) before normalisation: one alignment launch for the
    whole window, then one kernel for all sums; 0.0 where an individual has no reads
    (cpopulation.pyx:717-718)."""
    cdef int nI = len(readBuffers), nG = len(genotypes), nH = len(haplotypes), i, g, n, nTot = 0, base = 0
    cdef bamReadBuffer buf
    cdef DiploidGenotype gt
    cdef pg_ctx* ctx = deviceContext()
    cdef dict index = {}
    for i in range(nH):
        index[id(haplotypes[i])] = i
    prefetchWindowC(haplotypes, readBuffers, True)       # likelihoods resident: pair(h, r) = h * nTot + r
    for i in range(nI):
        buf = readBuffers[i]
        nTot += <int>(buf.windowEnd - buf.windowStart)
    out = np.zeros((nI, nG), dtype=np.float64)
    cdef double[:, ::1] o = out
    cdef int64_t* b1 = <int64_t*>malloc((<size_t>nI * nG + 1) * sizeof(int64_t))
    cdef int64_t* b2 = <int64_t*>malloc((<size_t>nI * nG + 1) * sizeof(int64_t))
    cdef int32_t* nr = <int32_t*>malloc((<size_t>nI * nG + 1) * sizeof(int32_t))
    cdef int rc
    try:
        for i in range(nI):
            buf = readBuffers[i]
            n = <int>(buf.windowEnd - buf.windowStart)
            for g in range(nG):
                gt = genotypes[g]
                b1[i * nG + g] = <int64_t>index[id(gt.hap1)] * nTot + base
                b2[i * nG + g] = <int64_t>index[id(gt.hap2)] * nTot + base
                nr[i * nG + g] = n
            base += n
        if nTot > 0 and nG > 0:
            rc = pg_genotype_loglik(ctx, <int64_t>nI * nG, b1, b2, nr, &o[0, 0])
            if rc != PG_OK:
                msg = pg_last_error_string(ctx)
                raise PlatGpuError("libplatgpu error %d: %s" % (rc, msg.decode() if msg != NULL else ""))
    finally:
        free(b1); free(b2); free(nr)
    return out
