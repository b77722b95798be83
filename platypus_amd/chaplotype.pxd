from platypus_amd.samtoolsWrapper cimport cAlignedRead

# the attributes and the method of the reference's Haplotype that the
# alignment path touches, platypus/chaplotype.pxd:7-34
cdef class Haplotype:
    cdef bytes refName
    cdef int startPos
    cdef int endPos
    cdef tuple variants
    cdef bytes haplotypeSequence
    cdef char* cHaplotypeSequence
    cdef double** likesByIndCache
    cdef int lenCache
    cdef int hash
    cdef int hapLen
    cdef int varSeqLen
    cdef char* cHomopolQ
    cdef int ensureCache(self, int nIndividuals) except -1
    cdef double* alignReads(
            self,
            int individualIndex,
            cAlignedRead** start,
            cAlignedRead** end,
            int nReads,
            int nIndividuals
    ) except NULL

cdef int prefetchWindowC(list haplotypes, list readBuffers, bint always=*) except -1
