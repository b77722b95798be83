# attribute block of platypus/variant.pxd (the Variant class only)
cdef class Variant:
    cdef public bytes refName
    cdef public bytes added
    cdef public bytes removed
    cdef public int refPos
    cdef public int minPos
    cdef public int maxPos
    cdef public int nAdded
    cdef public int nRemoved
    cdef public int nSupportingReads
    cdef public int nFwdReads
    cdef public int nRevReads
    cdef public double nFwdReadsWeighted
    cdef public double nRevReadsWeighted
    cdef public double freqPrior
    cdef public double rmsReadPos
    cdef public long long sumSqReadPos
    cdef long hashValue
    cpdef double calculatePrior(self) except? -1.0
    cpdef addVariant(self, Variant other)
