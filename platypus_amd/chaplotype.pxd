# Declarations of the haplotype layer.  Attribute and method names are those of the part of
# platypus/chaplotype.pxd:7-34 that the alignment path touches.
from platypus_amd.samtoolsWrapper cimport cAlignedRead

cdef int prefetchWindowC(list haplotypes, list readBuffers, bint always=*) except -1
cpdef bint variantsCompatible(tuple variants) except -1


cdef class Haplotype:
    # where and what
    cdef bytes refName, haplotypeSequence
    cdef int startPos, endPos, hapLen, varSeqLen, hash
    cdef tuple variants
    cdef object refFile
    # what the aligner is fed from
    cdef char* cHaplotypeSequence
    cdef char* cHomopolQ
    # likelihoods per individual, filled by the device
    cdef int lenCache
    cdef double** likesByIndCache
    cdef void setUp(self, bytes refName, int startPos, int endPos, bytes haplotypeSequence, int varSeqLen,
                    tuple variants)
    cpdef int valid(self) except -1
    cdef int ensureCache(self, int nIndividuals) except -1
    cdef double* alignReads(self, int individualIndex, cAlignedRead** start, cAlignedRead** end, int nReads,
                            int nIndividuals) except NULL
