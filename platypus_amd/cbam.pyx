# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""BAM record decoding into the packed arrays `bamReadBuffer.fromAlignments` takes (the
fields createRead extracts, platypus/samtoolsWrapper.pyx:699-789).  Input is a stretch of
uncompressed BAM alignment records; the BGZF and index handling is in platypus_amd/bamio.py."""
import numpy as np

from libc.stdint cimport int32_t, uint32_t, uint16_t, uint8_t, int64_t
from libc.string cimport memcpy
from libc.stdlib cimport malloc, realloc, free

cdef const char* BAM_BASES = b"=ACMGRSVTWYHKDBN"
cdef uint16_t PAIRS[256]                                 # packed byte -> its two base characters, first base in the low byte
cdef int _pi
for _pi in range(256):
    PAIRS[_pi] = <uint16_t>(<unsigned char>BAM_BASES[_pi >> 4]) | (<uint16_t>(<unsigned char>BAM_BASES[_pi & 15]) << 8)


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
    # first pass: count (without the interpreter lock: the caller's other thread keeps running)
    cdef Py_ssize_t nrec = 0, nbases = 0, ncigar = 0, q, cap = 4096
    cdef Py_ssize_t* starts = <Py_ssize_t*>malloc(cap * sizeof(Py_ssize_t))
    cdef Py_ssize_t* grown
    cdef bint oom = starts == NULL, corrupt = False
    with nogil:
        while p + 4 <= n and not oom:
            memcpy(&block, &data[p], 4)
            if block < 32:
                corrupt = True
                break
            if p + 4 + block > n:
                break
            memcpy(&rid, &data[p + 4], 4)
            memcpy(&pos, &data[p + 8], 4)
            memcpy(&binmq, &data[p + 12], 4)
            memcpy(&flagnc, &data[p + 16], 4)
            memcpy(&lseq, &data[p + 20], 4)
            lname = binmq & 0xff
            ncig = flagnc & 0xffff
            if lseq < 0 or 32 + lname + 4 * <Py_ssize_t>ncig + ((<Py_ssize_t>lseq + 1) >> 1) + lseq > block:
                corrupt = True
                break
            if rid > refID or rid < 0 or (rid == refID and pos >= end):
                done = True
                break
            if rid == refID:
                rend = pos
                q = p + 36 + lname
                for k in range(ncig):
                    memcpy(&cw, &data[q + 4 * k], 4)
                    op = cw & 0xf
                    if op == 0 or op == 2 or op == 3:         # what bam_calend counts (c/samtools/bam.c:17-27)
                        rend += cw >> 4
                if ncig == 0:                               # is_overlap, c/samtools/bam_index.c:556-561
                    rend = pos + 1
                if rend > start:
                    if nrec == cap:
                        grown = <Py_ssize_t*>realloc(starts, 2 * cap * sizeof(Py_ssize_t))
                        if grown == NULL:
                            oom = True
                            break
                        starts = grown
                        cap *= 2
                    starts[nrec] = p
                    nrec += 1
                    nbases += lseq
                    ncigar += ncig
            p += 4 + block
    if oom:
        free(starts)
        raise MemoryError()
    if corrupt:
        free(starts)
        raise IOError("corrupt BAM record at byte %d of the stretch" % p)
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
    with nogil:
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
                memcpy(&cw, &data[q + 4 * k], 4)            # '=' and 'X' stay what they are: to the reference they are
                vc[c + k] = cw                              # "other kinds of flag" that move nothing (variant.pyx:472-474)
            c += ncig
            vco[i + 1] = c
            q += 4 * ncig
            for j in range(lseq >> 1):                      # two bases per packed byte, one 2-byte store
                memcpy(&vs[b + 2 * j], &PAIRS[data[q + j]], 2)
            if lseq & 1:
                vs[b + lseq - 1] = BAM_BASES[data[q + (lseq >> 1)] >> 4]
            q += (lseq + 1) >> 1
            if lseq:
                memcpy(&vq[b], &data[q], lseq)
            b += lseq
            vo[i + 1] = b
    free(starts)
    return (dict(seq=seq, qual=qual, off=off, pos=apos, mapq=amapq, flag=aflag, cigar=cigar, cigarOff=coff,
                 insertSize=isize, matePos=matepos), consumed, done)


cdef inline int reg2bin(int beg, int end) noexcept nogil:
    end -= 1
    if beg >> 14 == end >> 14:
        return 4681 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return 585 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return 73 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return 9 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return 1 + (beg >> 26)
    return 0


def encodeRecords(int refID, const unsigned char[::1] seq, const unsigned char[::1] qual, const long long[::1] off,
                  const int[::1] pos, const int[::1] mapq, const int[::1] flag, const unsigned int[::1] cigar,
                  const long long[::1] cigarOff, const int[::1] insertSize, const int[::1] matePos, long long firstName=0):
    """The inverse of parseRecords for records without tags: uncompressed BAM alignment records
    (block_size included) of reads named r<firstName + i>.  Returns (bytes of all records, int64
    offsets of the records in them [n + 1], int32 end position of every alignment, int32 bin)."""
    cdef Py_ssize_t n = pos.shape[0], i, j, k, p = 0, total = 0
    cdef int lseq, ncig, lname, rend, b, op
    cdef uint32_t cw, w
    cdef int32_t v
    cdef long long name_id
    cdef char name[24]
    cdef unsigned char code[256]
    for i in range(256):
        code[i] = 15
    for i in range(16):
        code[<unsigned char>BAM_BASES[i]] = <unsigned char>i
    # sizes first: the name length depends on the number of digits
    recOff = np.zeros(n + 1, dtype=np.int64)
    ends = np.empty(n, dtype=np.int32)
    bins = np.empty(n, dtype=np.int32)
    cdef long long[::1] vo = recOff
    cdef int[::1] ve = ends, vb = bins
    for i in range(n):
        name_id = firstName + i
        lname = 3                                         # 'r', one digit, the terminating zero
        while name_id >= 10:
            name_id //= 10
            lname += 1
        lseq = <int>(off[i + 1] - off[i])
        ncig = <int>(cigarOff[i + 1] - cigarOff[i])
        total += 36 + lname + 4 * ncig + ((lseq + 1) >> 1) + lseq
        vo[i + 1] = total
    blob = np.zeros(total, dtype=np.uint8)
    cdef unsigned char[::1] out = blob
    for i in range(n):
        p = vo[i]
        lseq = <int>(off[i + 1] - off[i])
        ncig = <int>(cigarOff[i + 1] - cigarOff[i])
        lname = 0
        name_id = firstName + i
        while True:                                       # digits, last first
            name[lname] = <char>(48 + name_id % 10)
            lname += 1
            name_id //= 10
            if name_id == 0:
                break
        rend = pos[i]
        for k in range(ncig):
            cw = cigar[cigarOff[i] + k]
            op = cw & 0xf
            if op == 0 or op == 2 or op == 3:
                rend += cw >> 4
        ve[i] = rend
        b = reg2bin(pos[i], rend if rend > pos[i] else pos[i] + 1)
        vb[i] = b
        v = <int32_t>(vo[i + 1] - p - 4); memcpy(&out[p], &v, 4)
        v = refID; memcpy(&out[p + 4], &v, 4)
        v = pos[i]; memcpy(&out[p + 8], &v, 4)
        w = (<uint32_t>b << 16) | (<uint32_t>(mapq[i] & 0xff) << 8) | <uint32_t>(lname + 2); memcpy(&out[p + 12], &w, 4)
        w = (<uint32_t>flag[i] << 16) | <uint32_t>ncig; memcpy(&out[p + 16], &w, 4)
        v = lseq; memcpy(&out[p + 20], &v, 4)
        v = refID if matePos[i] >= 0 else -1; memcpy(&out[p + 24], &v, 4)
        v = matePos[i]; memcpy(&out[p + 28], &v, 4)
        v = insertSize[i]; memcpy(&out[p + 32], &v, 4)
        p += 36
        out[p] = 114                                      # 'r'
        for k in range(lname):
            out[p + 1 + k] = <unsigned char>name[lname - 1 - k]
        out[p + 1 + lname] = 0
        p += lname + 2
        for k in range(ncig):
            cw = cigar[cigarOff[i] + k]
            memcpy(&out[p + 4 * k], &cw, 4)
        p += 4 * ncig
        j = off[i]
        for k in range(lseq):
            if k & 1:
                out[p + (k >> 1)] |= code[seq[j + k]]
            else:
                out[p + (k >> 1)] = code[seq[j + k]] << 4
        p += (lseq + 1) >> 1
        for k in range(lseq):
            out[p + k] = qual[j + k]
    return blob, recOff, ends, bins


cdef extern from "zlib.h":
    ctypedef struct z_stream:
        const unsigned char* next_in
        unsigned int avail_in
        unsigned char* next_out
        unsigned int avail_out
        void* zalloc
        void* zfree
        void* opaque
    int inflateInit2(z_stream* strm, int windowBits) nogil
    int inflate(z_stream* strm, int flush) nogil
    int inflateReset(z_stream* strm) nogil
    int inflateEnd(z_stream* strm) nogil
    enum:
        Z_OK
        Z_STREAM_END
        Z_FINISH


cdef extern from "dlfcn.h":
    void* dlopen(const char* name, int flags) nogil
    void* dlsym(void* handle, const char* symbol) nogil
    enum:
        RTLD_NOW

# libdeflate inflates BGZF blocks two to three times as fast as zlib.  It is looked up at run time
# (the image ships the library without its header); zlib is used when it is not there.
ctypedef void* (*ld_alloc_t)() noexcept nogil
ctypedef int (*ld_inflate_t)(void* d, const void* src, size_t n_src, void* dst, size_t n_dst, size_t* n_out) noexcept nogil
ctypedef void (*ld_free_t)(void* d) noexcept nogil
cdef ld_alloc_t ld_alloc = NULL
cdef ld_inflate_t ld_inflate = NULL
cdef ld_free_t ld_free = NULL


cdef void find_libdeflate():
    global ld_alloc, ld_inflate, ld_free
    cdef void* h = dlopen(b"libdeflate.so.0", RTLD_NOW)
    if h == NULL:
        return
    cdef void* a = dlsym(h, b"libdeflate_alloc_decompressor")
    cdef void* f = dlsym(h, b"libdeflate_deflate_decompress")
    cdef void* r = dlsym(h, b"libdeflate_free_decompressor")
    if a != NULL and f != NULL and r != NULL:
        ld_alloc = <ld_alloc_t>a
        ld_inflate = <ld_inflate_t>f
        ld_free = <ld_free_t>r


import os as _os
if not _os.environ.get("PLATYPUS_AMD_ZLIB_ONLY"):
    find_libdeflate()


def inflateLibrary():
    """which library inflateBlocks uses"""
    return "libdeflate" if ld_inflate != NULL else "zlib"


def inflateBlocks(const unsigned char[::1] raw, const unsigned char[::1] prefix):
    """Every complete BGZF block at the start of `raw`, inflated one after the other behind a copy of
    `prefix` (what the caller had left over), in one call that does not hold the interpreter lock.
    Returns (uint8 array, bytes of `raw` consumed); a trailing incomplete block is left unconsumed."""
    cdef Py_ssize_t n = raw.shape[0], p = 0, k, q, total = prefix.shape[0], xlen, bsize, at
    cdef Py_ssize_t nblocks = 0
    cdef uint32_t isize
    cdef int bad = 0, rc
    # sizes first
    while p + 18 <= n:
        if raw[p] != 0x1f or raw[p + 1] != 0x8b or raw[p + 2] != 8 or raw[p + 3] != 4:
            raise IOError("not a BGZF block")
        xlen = raw[p + 10] | (raw[p + 11] << 8)
        if p + 12 + xlen > n:
            break
        bsize = -1
        k = p + 12
        while k + 4 <= p + 12 + xlen:
            if raw[k] == 66 and raw[k + 1] == 67:
                bsize = raw[k + 4] | (raw[k + 5] << 8)
            k += 4 + (raw[k + 2] | (raw[k + 3] << 8))
        if bsize < 0:
            raise IOError("BGZF block without a BC field")
        if bsize + 1 < 12 + xlen + 8:
            raise IOError("corrupt BGZF block")
        if p + bsize + 1 > n:
            break
        memcpy(&isize, &raw[p + bsize + 1 - 4], 4)
        total += isize
        nblocks += 1
        p += bsize + 1
    consumed = p
    out = np.empty(total, dtype=np.uint8)
    cdef unsigned char[::1] vo = out
    cdef z_stream zs
    if prefix.shape[0]:
        memcpy(&vo[0], &prefix[0], prefix.shape[0])
    if nblocks == 0:
        return out, consumed
    cdef void* dec = NULL
    cdef size_t got = 0
    if ld_inflate != NULL:
        with nogil:
            dec = ld_alloc()
            if dec == NULL:
                bad = 1
            else:
                p = 0
                at = prefix.shape[0]
                for k in range(nblocks):
                    xlen = raw[p + 10] | (raw[p + 11] << 8)
                    bsize = -1
                    q = p + 12
                    while q + 4 <= p + 12 + xlen:
                        if raw[q] == 66 and raw[q + 1] == 67:
                            bsize = raw[q + 4] | (raw[q + 5] << 8)
                        q += 4 + (raw[q + 2] | (raw[q + 3] << 8))
                    memcpy(&isize, &raw[p + bsize + 1 - 4], 4)
                    if isize:
                        rc = ld_inflate(dec, &raw[p + 12 + xlen], <size_t>(bsize + 1 - 12 - xlen - 8), &vo[at], isize, &got)
                        if rc != 0 or got != isize:
                            bad = 2
                            break
                    at += isize
                    p += bsize + 1
                ld_free(dec)
    else:
      with nogil:
        zs.zalloc = NULL; zs.zfree = NULL; zs.opaque = NULL
        zs.next_in = NULL; zs.avail_in = 0
        if inflateInit2(&zs, -15) != Z_OK:
            bad = 1
        else:
            p = 0
            at = prefix.shape[0]
            for k in range(nblocks):
                xlen = raw[p + 10] | (raw[p + 11] << 8)
                bsize = -1
                q = p + 12
                while q + 4 <= p + 12 + xlen:
                    if raw[q] == 66 and raw[q + 1] == 67:
                        bsize = raw[q + 4] | (raw[q + 5] << 8)
                    q += 4 + (raw[q + 2] | (raw[q + 3] << 8))
                memcpy(&isize, &raw[p + bsize + 1 - 4], 4)
                if isize:
                    zs.next_in = &raw[p + 12 + xlen]
                    zs.avail_in = <unsigned int>(bsize + 1 - 12 - xlen - 8)
                    zs.next_out = &vo[at]
                    zs.avail_out = isize
                    rc = inflate(&zs, Z_FINISH)
                    if rc != Z_STREAM_END or zs.avail_out != 0:
                        bad = 2
                        break
                    inflateReset(&zs)
                at += isize
                p += bsize + 1
            inflateEnd(&zs)
    if bad:
        raise IOError("corrupt BGZF block" if bad == 2 else "zlib could not be initialised")
    return out, consumed


cdef extern from "platgpu.h":
    ctypedef struct pg_inflater:
        pass
    int pg_inflater_create(int device_ordinal, pg_inflater** out) nogil
    int pg_inflater_destroy(pg_inflater* h) nogil
    const char* pg_inflater_error(const pg_inflater* h) nogil
    int pg_inflater_input(pg_inflater* h, size_t nbytes, void** buf) nogil
    int pg_bgzf_inflate(pg_inflater* h, const void* raw, size_t n_raw, const void* prefix, size_t n_prefix,
                        void** out, size_t* n_out, size_t* consumed) nogil


cdef class DeviceInflater:
    """inflateBlocks on the GPU (pg_bgzf_inflate: one lane per BGZF block).  The array a call returns is a
    view of the inflater's page-locked buffer and is valid until the next call; one thread at a time."""
    cdef pg_inflater* h

    def __cinit__(self, int device=0):
        self.h = NULL
        if pg_inflater_create(device, &self.h) != 0:
            raise RuntimeError("pg_inflater_create: " + pg_inflater_error(NULL).decode())

    def __dealloc__(self):
        if self.h != NULL:
            pg_inflater_destroy(self.h)
            self.h = NULL

    def inputBuffer(self, size_t n):
        """uint8 view (n bytes) of the inflater's page-locked input buffer: file bytes read into it go to the
        device without a staging copy.  Growing keeps the content; earlier views are invalid after a growth."""
        cdef void* buf = NULL
        if pg_inflater_input(self.h, n, &buf) != 0:
            raise MemoryError(pg_inflater_error(self.h).decode())
        if n == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.asarray(<unsigned char[:n]> (<unsigned char*>buf))

    def inflateBlocks(self, const unsigned char[::1] raw, const unsigned char[::1] prefix):
        cdef void* out = NULL
        cdef size_t n_out = 0, eaten = 0
        cdef const unsigned char* rp = &raw[0] if raw.shape[0] else NULL
        cdef const unsigned char* pp = &prefix[0] if prefix.shape[0] else NULL
        cdef size_t nr = raw.shape[0], npre = prefix.shape[0]
        cdef int rc
        with nogil:
            rc = pg_bgzf_inflate(self.h, rp, nr, pp, npre, &out, &n_out, &eaten)
        if rc != 0:
            raise IOError(pg_inflater_error(self.h).decode())
        if n_out == 0:
            return np.zeros(0, dtype=np.uint8), eaten
        return np.asarray(<unsigned char[:n_out]> (<unsigned char*>out)), eaten
