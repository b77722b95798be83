from platypus_amd.samtoolsWrapper cimport cAlignedRead

# same attribute block as the reference, platypus/cwindow.pxd:5-16
cdef class bamReadBuffer:
    cdef bytes chromBytes
    cdef char* chrom
    cdef int startBase
    cdef int endBase
    cdef int nReads
    cdef int longestRead
    cdef cAlignedRead** cReads
    cdef cAlignedRead** start
    cdef cAlignedRead** end
    cdef cAlignedRead** windowStart
    cdef cAlignedRead** windowEnd
    cdef object keepAlive
    cdef setWindowPointers(self, int start, int end)

cdef int bisectReadsLeft(cAlignedRead** reads, int testPos, int nReads)
