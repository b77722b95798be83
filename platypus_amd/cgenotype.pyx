# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""Diploid genotypes over GPU-scored haplotypes (platypus/cgenotype.pyx).

``calculateDataLikelihood`` keeps the reference's arithmetic (cgenotype.pyx:58-61:
log(0.5*(l1+l2)) summed in read order with the C library's log) over the two
cached likelihood rows, which come from the device -- the reads are never
aligned here.  ``genotypeLogLikelihoods`` is the batched form of
Caller.setup's double loop (cpopulation.pyx:698-725).
"""
import numpy as np

from libc.math cimport log

from platypus_amd.chaplotype cimport Haplotype, prefetchWindowC
from platypus_amd.cwindow cimport bamReadBuffer
from platypus_amd.samtoolsWrapper cimport cAlignedRead


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
        cdef Py_ssize_t readIndex
        cdef double* arr1 = self.hap1.alignReads(individualIndex, start, end, nReads, nIndividuals)
        cdef double* arr2 = self.hap2.alignReads(individualIndex, start, end, nReads, nIndividuals)
        for readIndex in range(nReads):
            likelihood += log(0.5 * (arr1[readIndex] + arr2[readIndex]))
        return likelihood


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
    (cpopulation.pyx:698-725) before normalisation: one prefetch launch for the
    whole window, then the per-genotype sums; 0.0 where an individual has no reads
    (cpopulation.pyx:717-718)."""
    cdef int nI = len(readBuffers), nG = len(genotypes), i, g, n
    cdef bamReadBuffer buf
    cdef DiploidGenotype gt
    prefetchWindowC(haplotypes, readBuffers)
    out = np.zeros((nI, nG), dtype=np.float64)
    cdef double[:, ::1] o = out
    for i in range(nI):
        buf = readBuffers[i]
        n = <int>(buf.windowEnd - buf.windowStart)
        for g in range(nG):
            gt = genotypes[g]
            if n > 0:
                o[i, g] = gt.calculateDataLikelihood(buf.windowStart, buf.windowEnd, i, n, nI)
    return out
