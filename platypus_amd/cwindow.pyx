# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""The read buffer of one sample and its window pointers.

Only the part of platypus/cwindow.pyx that defines which reads belong to a
window is mirrored: the sorted ``cAlignedRead**`` array, ``bisectReadsLeft``
(cwindow.pyx:30-48) and ``setWindowPointers`` (cwindow.pyx:242-267).  The
reference fills the buffer from a BAM file (cwindow.pyx:156-229); here it is
filled from packed arrays (bases, raw qualities, offsets, pos, end, mapq), which
is what a BAM reader or a simulator hands over.
"""
import numpy as np

from libc.stdlib cimport malloc, free, calloc
from libc.string cimport memcpy

from platypus_amd.samtoolsWrapper cimport cAlignedRead


cdef int bisectReadsLeft(cAlignedRead** reads, int testPos, int nReads):
    cdef int low = 0
    cdef int high = nReads
    cdef int mid = 0
    while low < high:
        mid = (low + high) // 2
        if reads[mid].pos < testPos:
            low = mid + 1
        else:
            high = mid
    return low


cdef class bamReadBuffer:
    def __init__(self, bytes chrom, int start, int end, seq, qual, off, pos, readEnd, mapq):
        """Reads must be sorted by pos (the BAM order the reference relies on)."""
        cdef const unsigned char[::1] s = np.ascontiguousarray(seq, dtype=np.uint8)
        cdef const unsigned char[::1] q = np.ascontiguousarray(qual, dtype=np.uint8)
        cdef const long long[::1] o = np.ascontiguousarray(off, dtype=np.int64)
        cdef const int[::1] p = np.ascontiguousarray(pos, dtype=np.int32)
        cdef const int[::1] e = np.ascontiguousarray(readEnd, dtype=np.int32)
        cdef const int[::1] m = np.ascontiguousarray(mapq, dtype=np.int32)
        cdef int i, n = <int>(o.shape[0] - 1), rl
        cdef cAlignedRead* r
        self.chromBytes = chrom
        self.chrom = self.chromBytes
        self.startBase = start
        self.endBase = end
        self.nReads = n
        self.longestRead = 0
        self.cReads = <cAlignedRead**>malloc((n + 1) * sizeof(cAlignedRead*))
        for i in range(n):
            rl = <int>(o[i + 1] - o[i])
            r = <cAlignedRead*>calloc(1, sizeof(cAlignedRead))
            r.seq = <char*>calloc(rl + 1, 1)
            r.qual = <char*>calloc(rl + 1, 1)
            memcpy(r.seq, &s[o[i]], rl)
            memcpy(r.qual, &q[o[i]], rl)
            r.pos = p[i]
            r.rlen = rl
            r.end = e[i]
            r.mapq = m[i]
            if i > 0 and p[i] < p[i - 1]:
                raise ValueError("reads are not sorted by position")
            if r.end - r.pos > self.longestRead:      # cwindow.pyx:220-221
                self.longestRead = r.end - r.pos
            self.cReads[i] = r
        self.start = self.cReads
        self.end = self.cReads + n
        self.windowStart = self.cReads
        self.windowEnd = self.cReads + n

    def __dealloc__(self):
        cdef int i
        if self.cReads != NULL:
            for i in range(self.nReads):
                free(self.cReads[i].seq)
                free(self.cReads[i].qual)
                free(self.cReads[i])
            free(self.cReads)

    cdef setWindowPointers(self, int start, int end):
        """platypus/cwindow.pyx:242-267: reads with pos in [max(1, start - longestRead), end),
        minus the LEADING ones that end at or before the window start."""
        if self.nReads == 0:
            self.windowStart = self.cReads
            self.windowEnd = self.cReads
            return
        cdef int firstOverlapStart = max(1, start - self.longestRead)
        cdef int startPosOfReads = bisectReadsLeft(self.cReads, firstOverlapStart, self.nReads)
        cdef int endPosOfReads = bisectReadsLeft(self.cReads, end, self.nReads)
        while startPosOfReads < self.nReads and self.cReads[startPosOfReads].end <= start:
            startPosOfReads += 1
        self.windowStart = self.cReads + startPosOfReads
        self.windowEnd = self.cReads + endPosOfReads if self.cReads + endPosOfReads < self.end else self.end
        if startPosOfReads > endPosOfReads:
            raise RuntimeError("This should never happen. Read start pointer > read end pointer!!")

    # ---- Python-visible helpers (the reference's attributes are cdef-only) ----
    def setWindow(self, int start, int end):
        self.setWindowPointers(start, end)
        return (<int>(self.windowStart - self.cReads), <int>(self.windowEnd - self.cReads))

    def windowRange(self):
        return (<int>(self.windowStart - self.cReads), <int>(self.windowEnd - self.cReads))

    @property
    def readCount(self):
        return self.nReads
