# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""BAM record decoding into the packed arrays `bamReadBuffer.fromAlignments` takes (the
fields createRead extracts, platypus/samtoolsWrapper.pyx:699-789).  Input is a stretch of
uncompressed BAM alignment records; the BGZF and index handling is in platypus_amd/bamio.py."""
import numpy as np

from libc.stdint cimport int32_t, uint32_t, uint16_t, uint8_t, int64_t
from libc.string cimport memcpy

cdef const char* BAM_BASES = b"=ACMGRSVTWYHKDBN"


def parseRecords(const unsigned char[::1] data, int refID, int start, int end):
    """Records of reference `refID` that overlap [start, end) (what Samfile.fetch yields,
    platypus/cwindow.pyx:171-172), in file order.  Returns (arrays, bytes consumed, done):
    `done` is True once a record beyond the region (or of a later reference) was met; a
    truncated last record is left unconsumed."""
    cdef Py_ssize_t n = data.shape[0], p = 0
    cdef int32_t block, rid, pos, lseq, mrid, mpos, tlen, rend
    cdef uint32_t binmq, flagnc, cw
    cdef int lname, mapq, ncig, flag, k, op, ln
    cdef bint done = False
    # first pass: count
    cdef Py_ssize_t nrec = 0, nbases = 0, ncigar = 0, q
    cdef list starts = []
    while p + 4 <= n:
        memcpy(&block, &data[p], 4)
        if p + 4 + block > n:
            break
        memcpy(&rid, &data[p + 4], 4)
        memcpy(&pos, &data[p + 8], 4)
        memcpy(&binmq, &data[p + 12], 4)
        memcpy(&flagnc, &data[p + 16], 4)
        memcpy(&lseq, &data[p + 20], 4)
        lname = binmq & 0xff
        ncig = flagnc & 0xffff
        if rid > refID or rid < 0 or (rid == refID and pos >= end):
            done = True
            break
        if rid == refID:
            rend = pos
            q = p + 36 + lname
            for k in range(ncig):
                memcpy(&cw, &data[q + 4 * k], 4)
                op = cw & 0xf
                if op == 0 or op == 2 or op == 3 or op == 7 or op == 8:
                    rend += cw >> 4
            if rend > start or (rend == pos and pos >= start):
                starts.append(p)
                nrec += 1
                nbases += lseq
                ncigar += ncig
        p += 4 + block
    consumed = p
    seq = np.empty(nbases, dtype=np.uint8)
    qual = np.empty(nbases, dtype=np.uint8)
    off = np.zeros(nrec + 1, dtype=np.int64)
    apos = np.empty(nrec, dtype=np.int32)
    amapq = np.empty(nrec, dtype=np.int32)
    aflag = np.empty(nrec, dtype=np.int32)
    cigar = np.empty(ncigar, dtype=np.uint32)
    coff = np.zeros(nrec + 1, dtype=np.int64)
    isize = np.empty(nrec, dtype=np.int32)
    matepos = np.empty(nrec, dtype=np.int32)
    cdef unsigned char[::1] vs = seq, vq = qual
    cdef long long[::1] vo = off, vco = coff
    cdef int[::1] vp = apos, vm = amapq, vf = aflag, vi = isize, vmp = matepos
    cdef unsigned int[::1] vc = cigar
    cdef Py_ssize_t i, b = 0, c = 0, j
    for i in range(nrec):
        p = starts[i]
        memcpy(&pos, &data[p + 8], 4)
        memcpy(&binmq, &data[p + 12], 4)
        memcpy(&flagnc, &data[p + 16], 4)
        memcpy(&lseq, &data[p + 20], 4)
        memcpy(&mpos, &data[p + 28], 4)
        memcpy(&tlen, &data[p + 32], 4)
        lname = binmq & 0xff
        ncig = flagnc & 0xffff
        vp[i] = pos
        vm[i] = (binmq >> 8) & 0xff
        vf[i] = flagnc >> 16
        vi[i] = tlen
        vmp[i] = mpos
        q = p + 36 + lname
        for k in range(ncig):
            memcpy(&cw, &data[q + 4 * k], 4)
            op = cw & 0xf
            if op == 7 or op == 8:                      # '=' and 'X' are matches to this caller
                cw = (cw & 0xfffffff0u)
            vc[c + k] = cw
        c += ncig
        vco[i + 1] = c
        q += 4 * ncig
        for j in range(lseq):
            vs[b + j] = BAM_BASES[(data[q + (j >> 1)] >> (4 * (1 - (j & 1)))) & 0xf]
        q += (lseq + 1) >> 1
        for j in range(lseq):
            vq[b + j] = data[q + j]
        b += lseq
        vo[i + 1] = b
    return (dict(seq=seq, qual=qual, off=off, pos=apos, mapq=amapq, flag=aflag, cigar=cigar, cigarOff=coff,
                 insertSize=isize, matePos=matepos), consumed, done)
