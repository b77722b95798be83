# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""TEST MODULE.  A stand-in for the reference's callers of the alignment path: it cimports
platypus_amd's calign / chaplotype / cwindow declarations unchanged and makes exactly the calls the
reference makes, with the reference's argument lists:

    platypus/variantcaller.pyx:130       Haplotype(chrom, start, end, (), refFile, options.rlen)
    platypus/variantcaller.pyx:373-374   Haplotype(testHap.refName, testHap.startPos, testHap.endPos, (theVar,), refFile, 0)
    platypus/cgenotype.pyx:55-61         hap.alignReads(individualIndex, start, end, nReads, nIndividuals)
    platypus/chaplotype.pyx:264          alignNoTraceback(read, seq, startPos, varSeqLen, cHomopolQ, 3, 2, hapLen, True)
    platypus/calign.pyx:66-70            hash_sequence(hap), map_read(read, hapLen, table), free(table)
    platypus/cfilter.pyx:58-73           hap.valid()

If the mirror's interface drifts from the reference's, this module stops compiling."""
from libc.math cimport log
from libc.stdlib cimport free

from platypus_amd.samtoolsWrapper cimport cAlignedRead
from platypus_amd.calign cimport alignNoTraceback, hash_sequence, map_read
from platypus_amd.chaplotype cimport Haplotype
from platypus_amd.cwindow cimport bamReadBuffer


def windowHaplotypes(bytes chrom, int windowStart, int windowEnd, variant, refFile, int rlen):
    """variantcaller.pyx:130 and :373-374: the window's reference haplotype, then fresh reference and
    variant haplotypes over its padded bounds with maxReadLength = 0"""
    cdef Haplotype refHaplotype = Haplotype(chrom, windowStart, windowEnd, (), refFile, rlen)
    cdef Haplotype testHap = Haplotype(chrom, windowStart, windowEnd, (variant,), refFile, rlen)
    cdef Haplotype refHap = Haplotype(testHap.refName, testHap.startPos, testHap.endPos, (), refFile, 0)
    cdef Haplotype varHap = Haplotype(testHap.refName, testHap.startPos, testHap.endPos, (variant,), refFile, 0)
    assert refHaplotype.valid() and testHap.valid() and varHap.valid()
    return refHaplotype, testHap, refHap, varHap


def calculateDataLikelihood(Haplotype hap1, Haplotype hap2, bamReadBuffer theBuffer, int individualIndex, int nIndividuals):
    """cgenotype.pyx:48-63 as it stands, over theBuffer.windowStart .. windowEnd (variantcaller.pyx:422-425)"""
    cdef cAlignedRead** start = theBuffer.windowStart
    cdef cAlignedRead** end = theBuffer.windowEnd
    cdef int nReads = <int>(end - start)
    cdef double likelihood = 0.0
    cdef Py_ssize_t readIndex = 0
    cdef double* arr1 = hap1.alignReads(individualIndex, start, end, nReads, nIndividuals)
    cdef double* arr2 = hap2.alignReads(individualIndex, start, end, nReads, nIndividuals)
    for readIndex in range(nReads):
        likelihood += log(0.5 * (arr1[readIndex] + arr2[readIndex]))
    return likelihood


def perReadLikelihoods(Haplotype hap, bamReadBuffer theBuffer):
    """the loop body of chaplotype.pyx:259-268, one alignNoTraceback call per read"""
    cdef cAlignedRead** start = theBuffer.windowStart
    cdef cAlignedRead** end = theBuffer.windowEnd
    out = []
    while start != end:
        out.append(alignNoTraceback(start[0], hap.cHaplotypeSequence, hap.startPos, hap.varSeqLen, hap.cHomopolQ,
                                    3, 2, hap.hapLen, True))
        start += 1
    return out


def anchors(Haplotype hap, bamReadBuffer theBuffer):
    """calign.pyx:66-70: indexOfReadIntoHap of every window read"""
    cdef cAlignedRead** start = theBuffer.windowStart
    cdef cAlignedRead** end = theBuffer.windowEnd
    cdef int* table = hash_sequence(hap.cHaplotypeSequence)
    out = []
    while start != end:
        out.append(map_read(start[0].seq, hap.hapLen, table))
        start += 1
    free(table)
    return out
