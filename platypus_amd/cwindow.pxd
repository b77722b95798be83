# Declarations of the read buffer.  Attribute names are those of platypus/cwindow.pxd:5-16.
from platypus_amd.samtoolsWrapper cimport cAlignedRead

cdef int bisectReadsLeft(cAlignedRead** reads, int testPos, int nReads)


cdef class bamReadBuffer:
    cdef bytes chromBytes
    cdef char* chrom
    cdef int startBase, endBase, nReads, longestRead
    cdef int maxReadLen                             # longest sequence among the buffered reads
    cdef cAlignedRead** cReads                      # all buffered reads, sorted by position
    cdef cAlignedRead** start                       # == cReads
    cdef cAlignedRead** end                         # one past the last read
    cdef cAlignedRead** windowStart                 # the current window's reads: [windowStart, windowEnd)
    cdef cAlignedRead** windowEnd
    cdef object keepAlive
    cdef char* slabBytes                            # fromAlignments: bases and qualities of all reads in one block,
    cdef int* slabInts                              # CIGAR operations and lengths in another,
    cdef cAlignedRead* slabReads                    # the records in a third
    cdef setWindowPointers(self, int start, int end)
