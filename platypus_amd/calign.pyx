# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""Read-to-haplotype alignment on an MI355X behind the reference's calign interface.

``alignNoTraceback`` keeps the signature of platypus/calign.pyx:36 but sends the
pair to the HIP kernels of libplatgpu.so; ``alignBatch`` is the form the rest of
this package uses: every haplotype of a window against every read of an
individual in one launch.  The alignment path keeps its 7-mer table in LDS inside
the device's anchoring kernel; ``hash_sequence`` / ``map_read`` (platypus/calign.pyx:99-158)
exist for callers of those cdef functions and run on the device as well.  There is
no CPU fallback.
"""
import os

from libc.stdint cimport int32_t, int64_t
from libc.stdlib cimport malloc, free
from libc.string cimport memcpy, strlen

from platypus_amd.samtoolsWrapper cimport cAlignedRead
from platypus_amd.platgpu cimport *

cdef pg_ctx* g_ctx = NULL
cdef bytes g_table = b""
cdef int g_ge = -1
cdef int g_np = -1


class PlatGpuError(RuntimeError):
    pass


cdef int check(int rc) except -1:
    if rc != PG_OK:
        msg = pg_last_error_string(g_ctx)
        raise PlatGpuError("libplatgpu error %d: %s" % (rc, msg.decode() if msg != NULL else ""))
    return 0


cdef int g_max_read_len = 0          # read length the context was (or will be) created for; 0: not decided yet


cdef int ensureReadLength(int n) except -1:
    """Make sure the device context takes reads of n bases: a context made for shorter reads is
    replaced (device haplotypes carry that many trailing 'N's, so the limit is fixed at creation).
    Results of earlier launches are gone after a replacement."""
    global g_max_read_len
    if n > 16384:
        raise ValueError("a read of %d bases: the device path takes at most 16384" % n)
    if g_max_read_len == 0:
        g_max_read_len = max(512, int(os.environ.get("PLATGPU_MAX_READ_LEN", "0")))
    if n > g_max_read_len:
        closeDevice()
        while g_max_read_len < n:
            g_max_read_len *= 2
        g_max_read_len = min(g_max_read_len, 16384)
    return 0


cdef pg_ctx* deviceContext() except NULL:
    """One device context per process (the unit runner.py:132 spreads over workers).
    PLATGPU_DEVICE selects the GPU; the longest read accepted starts at 512 (or PLATGPU_MAX_READ_LEN)
    and grows with ensureReadLength."""
    global g_ctx
    cdef int rc
    if g_ctx == NULL:
        ensureReadLength(1)
        rc = pg_ctx_create(int(os.environ.get("PLATGPU_DEVICE", "0")), g_max_read_len, &g_ctx)
        if rc != PG_OK:
            msg = pg_last_error_string(NULL)
            g_ctx = NULL
            raise PlatGpuError("libplatgpu error %d: %s" % (rc, msg.decode() if msg != NULL else ""))
    return g_ctx


cdef int setParams(char* table, int gapextend, int nucprior) except -1:
    """Alignment parameters travel with every call in the reference
    (chaplotype.pyx:264); the device keeps them, so only changes are sent."""
    global g_table, g_ge, g_np
    cdef bytes t = table
    deviceContext()
    if t != g_table or gapextend != g_ge or nucprior != g_np:
        check(pg_set_params(g_ctx, gapextend, nucprior, table, <int>len(t)))
        g_table = t
        g_ge = gapextend
        g_np = nucprior
    return 0


cdef int alignBatch(int nHaps, char** hapSeqs, int* hapStarts, int* varSeqLens, int* hapLens,
                    cAlignedRead** start, cAlignedRead** end, char* homopolgapq,
                    int gapextend, int nucprior, double* likelihoods, int* scores,
                    int* anchors) except -1:
    cdef int nReads = <int>(end - start)
    cdef int h, r
    cdef int64_t tot = 0
    if nHaps <= 0 or nReads <= 0:
        return 0
    cdef int longest = 1
    for r in range(nReads):
        if start[r].rlen > longest:
            longest = start[r].rlen
    ensureReadLength(longest)
    setParams(homopolgapq, gapextend, nucprior)
    # haplotypes, concatenated
    cdef int64_t* hoff = <int64_t*>malloc((nHaps + 1) * sizeof(int64_t))
    hoff[0] = 0
    for h in range(nHaps):
        hoff[h + 1] = hoff[h] + hapLens[h]
    cdef char* hbuf = <char*>malloc(hoff[nHaps] + 1)
    for h in range(nHaps):
        memcpy(hbuf + hoff[h], hapSeqs[h], hapLens[h])
    # reads: the five fields of cAlignedRead the path uses, packed
    cdef int64_t* roff = <int64_t*>malloc((nReads + 1) * sizeof(int64_t))
    cdef int32_t* rpos = <int32_t*>malloc(nReads * sizeof(int32_t))
    cdef int32_t* rmq = <int32_t*>malloc(nReads * sizeof(int32_t))
    roff[0] = 0
    for r in range(nReads):
        roff[r + 1] = roff[r] + start[r].rlen
        rpos[r] = start[r].pos
        rmq[r] = start[r].mapq
    cdef char* sbuf = <char*>malloc(roff[nReads] + 1)
    cdef char* qbuf = <char*>malloc(roff[nReads] + 1)
    for r in range(nReads):
        memcpy(sbuf + roff[r], start[r].seq, start[r].rlen)
        memcpy(qbuf + roff[r], start[r].qual, start[r].rlen)
    try:
        # one block up, one kernel for the device arrays, anchoring, alignment, one block back
        check(pg_align_window(g_ctx, nHaps, hbuf, hoff, <int32_t*>hapStarts, <int32_t*>varSeqLens, nReads, sbuf, qbuf, roff,
                              rpos, rmq, <int32_t*>scores, <int32_t*>anchors, likelihoods))
    finally:
        free(hoff); free(hbuf); free(roff); free(rpos); free(rmq); free(sbuf); free(qbuf)
    return 0


cdef double alignNoTraceback(cAlignedRead* read, char* hapSeq, int hapStart, int varSeqLen,
                             char* homopolgapq_or_localgapopen, int gapextend, int nucprior,
                             int hapLen, int useAsHomopolGapQ) except? -1.0:
    """One pair, the reference's call (platypus/calign.pyx:36-78).  Correct but a
    whole launch per pair: use alignBatch / Haplotype.alignReads for throughput."""
    cdef double like = 0.0
    cdef cAlignedRead* one = read
    if not useAsHomopolGapQ:
        raise PlatGpuError("local gap-open strings (useAsHomopolGapQ = 0) are not used by this fork "
                           "(chaplotype.pyx:61) and not implemented")
    alignBatch(1, &hapSeq, &hapStart, &varSeqLen, &hapLen, &one, &one + 1,
               homopolgapq_or_localgapopen, gapextend, nucprior, &like, NULL, NULL)
    return like


cdef int* hash_sequence(char* sequence) except NULL:
    """platypus/calign.pyx:99-123; the caller frees the table."""
    cdef pg_ctx* ctx = deviceContext()
    cdef int* h = <int*>malloc(16384 * sizeof(int))
    cdef int rc = pg_hash_sequence(ctx, sequence, <int>strlen(sequence), <int32_t*>h)
    if rc != PG_OK:
        free(h)
        check(rc)
    return h


cdef int map_read(char* read, int sequence_length, int* sequence_hash) except? -2147483647:
    """platypus/calign.pyx:126-158"""
    cdef pg_ctx* ctx = deviceContext()
    cdef int32_t pos = 0
    check(pg_map_read(ctx, read, <int>strlen(read), sequence_length, <int32_t*>sequence_hash, &pos))
    return pos


def hashSequencePy(bytes sequence):
    cdef int* h = hash_sequence(sequence)
    try:
        return [h[i] for i in range(16384)]
    finally:
        free(h)


def mapReadPy(bytes read, bytes sequence):
    cdef int* h = hash_sequence(sequence)
    try:
        return map_read(read, <int>len(sequence), h)
    finally:
        free(h)


def openDevice(int maxReadLength=0):
    """Create the device context now rather than at the first call that needs it (a caller that has to wait for its
    input anyway creates it meanwhile)."""
    if maxReadLength > 0:
        ensureReadLength(maxReadLength)
    deviceContext()


def closeDevice():
    """Release the device context (tests; a worker process just exits)."""
    global g_ctx, g_table, g_ge, g_np
    if g_ctx != NULL:
        pg_ctx_destroy(g_ctx)
        g_ctx = NULL
        g_table = b""
        g_ge = g_np = -1


def alignNoTracebackPy(bytes readSeq, bytes readQual, int readPos, int mapq, bytes hapSeq, int hapStart,
                       int varSeqLen, bytes table, int gapextend=3, int nucprior=2):
    """Python-visible form of alignNoTraceback for tests (the reference's is cdef-only)."""
    cdef cAlignedRead r
    r.seq = readSeq
    r.qual = readQual
    r.pos = readPos
    r.rlen = <int>len(readSeq)
    r.end = readPos + r.rlen
    r.mapq = mapq
    return alignNoTraceback(&r, hapSeq, hapStart, varSeqLen, table, gapextend, nucprior, <int>len(hapSeq), 1)
