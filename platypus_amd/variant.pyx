# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""The Variant record of platypus/variant.pyx:19-250: position, removed and added
sequence, support counters, prior.  Candidate generation from CIGAR strings
(variant.pyx:253-615) is upstream of the alignment path and not part of this package."""
from libc.math cimport exp, sqrt

cdef double mLTOT = -0.23025850929940459    # minus log ten over ten


cdef class Variant:
    def __init__(self, bytes refName, int refPos, bytes removed, bytes added, int nSupportingReads=0,
                 int nFwdReads=0, int mapQ=0, int baseQ=0, int readPos=0):
        refPos = max(0, refPos)                                # variant.pyx:38
        self.refName = refName
        self.minPos = refPos
        self.nAdded = <int>len(added)
        self.nRemoved = <int>len(removed)
        self.freqPrior = 0.0
        if self.nRemoved == 1 and self.nAdded == 1:            # SNP, variant.pyx:48-49
            self.maxPos = refPos
        elif self.nRemoved == self.nAdded:                     # multi-nucleotide substitution
            self.maxPos = refPos + (self.nAdded - 1)
        else:                                                  # indel / complex
            self.maxPos = refPos + self.nRemoved + self.nAdded
        self.refPos = refPos
        self.removed = removed
        self.added = added
        self.nSupportingReads = nSupportingReads
        self.nFwdReads = nFwdReads
        self.nRevReads = nSupportingReads - nFwdReads
        self.hashValue = -1
        self.sumSqReadPos = <double>readPos * readPos
        self.rmsReadPos = 0.0
        if nSupportingReads > 0:
            self.rmsReadPos = sqrt(self.sumSqReadPos / nSupportingReads)
        if nFwdReads == 1:                                     # variant.pyx:71-76
            self.nFwdReadsWeighted = 1.0 - exp(mLTOT * baseQ)
            self.nRevReadsWeighted = 0.0
        else:
            self.nFwdReadsWeighted = 0.0
            self.nRevReadsWeighted = 1.0 - exp(mLTOT * baseQ)

    cpdef double calculatePrior(self) except? -1.0:
        """variant.pyx:78-105"""
        if self.freqPrior != 0.0:
            return self.freqPrior
        if self.nAdded == 1 and self.nRemoved == 1:
            return 1e-3
        if self.nAdded == self.nRemoved:
            return 5e-5 * (0.1 ** self.nAdded)
        if self.nAdded > 0 and self.nRemoved == 0:
            return 1e-4 * (0.33 ** self.nAdded)
        if self.nAdded == 0 and self.nRemoved > 0:
            return 1e-4 * (0.8 ** self.nRemoved)
        raise ValueError("Do not know how to calculate prior for variant %s." % self)

    cpdef addVariant(self, Variant other):
        """variant.pyx:107-119"""
        self.nSupportingReads += other.nSupportingReads
        self.nFwdReads += other.nFwdReads
        self.nRevReads += other.nRevReads
        self.nFwdReadsWeighted += other.nFwdReadsWeighted
        self.nRevReadsWeighted += other.nRevReadsWeighted
        self.sumSqReadPos += other.sumSqReadPos
        if self.nSupportingReads > 0:
            self.rmsReadPos = sqrt(self.sumSqReadPos / self.nSupportingReads)

    def __hash__(self):
        if self.hashValue == -1:
            self.hashValue = hash((self.refName, self.refPos, self.removed, self.added))
        return self.hashValue

    def _key(self):
        return (self.refName, self.refPos, abs(self.nAdded - self.nRemoved))

    def __richcmp__(self, other, int op):
        """variant.pyx:143-228: ordering by (name, position, |length change|); equality by
        name, position, removed and added sequence."""
        if not isinstance(other, Variant):
            return NotImplemented
        cdef Variant o = other
        if op == 2 or op == 3:
            same = (self.refName == o.refName and self.refPos == o.refPos and self.removed == o.removed
                    and self.added == o.added)
            return same if op == 2 else not same
        a, b = self._key(), o._key()
        if op == 0:
            return a < b
        if op == 1:
            return a <= b
        if op == 4:
            return a > b
        return a >= b

    def __str__(self):
        s = "Var(%s:%s" % (self.refName.decode(), self.refPos)
        if self.nRemoved > 0:
            s += " -" + self.removed.decode()
        if self.nAdded > 0:
            s += " +" + self.added.decode()
        return s + " nReads = %s, nFwd = %s, nRev = %s)" % (self.nSupportingReads, self.nFwdReads, self.nRevReads)

    def __repr__(self):
        return self.__str__()
