"""TEST INFRASTRUCTURE ONLY -- ctypes access to the reference's own vendored samtools 0.1.x
(platypus/c/samtools/{bgzf,bam,bam_index,bam_aux,kstring,sam_header,bam_import,sam,bam_pileup,faidx,razf}.c),
compiled from the sources where they lie by `make -C oracle samref` into oracle/_ref/libsamref.so.
It is what the reference reads and indexes BAM files with (platypus/samtoolsWrapper.pyx:60-161),
so it pins the product's BAM/BAI reader and writer (platypus_amd/bamio.py, cbam.pyx): files written
by one side are read by the other.  Nothing in the product imports this module."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libsamref.so")


def available():
    return os.path.exists(LIB_PATH)


class Core(C.Structure):                                   # bam1_core_t, c/samtools/bam.h:164-173
    _fields_ = [("tid", C.c_int32), ("pos", C.c_int32), ("bin_mq_nl", C.c_uint32), ("flag_nc", C.c_uint32),
                ("l_qseq", C.c_int32), ("mtid", C.c_int32), ("mpos", C.c_int32), ("isize", C.c_int32)]


class Bam1(C.Structure):                                   # bam1_t, c/samtools/bam.h:189-193
    _fields_ = [("core", Core), ("l_aux", C.c_int), ("data_len", C.c_int), ("m_data", C.c_int),
                ("data", C.POINTER(C.c_uint8))]


class Header(C.Structure):                                 # bam_header_t, c/samtools/bam.h:87-94
    _fields_ = [("n_targets", C.c_int32), ("target_name", C.POINTER(C.c_char_p)), ("target_len", C.POINTER(C.c_uint32)),
                ("dict", C.c_void_p), ("hash", C.c_void_p), ("rg2lib", C.c_void_p), ("l_text", C.c_size_t),
                ("n_text", C.c_size_t), ("text", C.c_char_p)]


_FETCH_F = C.CFUNCTYPE(C.c_int, C.POINTER(Bam1), C.c_void_p)
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.bgzf_open.restype = C.c_void_p
        L.bgzf_open.argtypes = [C.c_char_p, C.c_char_p]
        L.bgzf_close.argtypes = [C.c_void_p]
        L.bam_header_read.restype = C.POINTER(Header)
        L.bam_header_read.argtypes = [C.c_void_p]
        L.bam_header_write.argtypes = [C.c_void_p, C.POINTER(Header)]
        L.bam_read1.argtypes = [C.c_void_p, C.POINTER(Bam1)]
        L.bam_write1.argtypes = [C.c_void_p, C.POINTER(Bam1)]
        L.bam_index_build.argtypes = [C.c_char_p]
        L.bam_index_load.restype = C.c_void_p
        L.bam_index_load.argtypes = [C.c_char_p]
        L.bam_index_destroy.argtypes = [C.c_void_p]
        L.bam_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, _FETCH_F]
        L.bam_header_destroy.argtypes = [C.POINTER(Header)]
        _lib = L
    return _lib


_BASES = b"=ACMGRSVTWYHKDBN"


def _record(b):
    """one bam1_t as a dict of plain Python values (the fields createRead reads,
    platypus/samtoolsWrapper.pyx:699-789, plus name and aux bytes)"""
    c = b.core
    l_qname = c.bin_mq_nl >> 24                            # bit fields fill from the low end on x86-64 gcc
    n_cigar = c.flag_nc >> 16
    data = C.string_at(b.data, b.data_len)
    p = l_qname
    cigar = np.frombuffer(data, dtype="<u4", count=n_cigar, offset=p).copy()
    p += 4 * n_cigar
    packed = data[p:p + (c.l_qseq + 1) // 2]
    seq = bytes(_BASES[(packed[j >> 1] >> (4 * (1 - (j & 1)))) & 15] for j in range(c.l_qseq))
    p += (c.l_qseq + 1) // 2
    qual = data[p:p + c.l_qseq]
    p += c.l_qseq
    return dict(tid=c.tid, pos=c.pos, bin=c.bin_mq_nl & 0xffff, mapq=(c.bin_mq_nl >> 16) & 0xff, flag=c.flag_nc & 0xffff,
                name=data[:l_qname - 1], cigar=cigar, seq=seq, qual=qual, mtid=c.mtid, mpos=c.mpos, isize=c.isize,
                aux=data[p:])


def read_all(path):
    """(reference names, lengths, header text, list of records) through bam_header_read / bam_read1"""
    L = lib()
    fp = L.bgzf_open(path.encode(), b"r")
    if not fp:
        raise IOError("bgzf_open failed: " + path)
    h = L.bam_header_read(fp)
    refs = [h.contents.target_name[i] for i in range(h.contents.n_targets)]
    lens = [int(h.contents.target_len[i]) for i in range(h.contents.n_targets)]
    text = C.string_at(h.contents.text, h.contents.l_text) if h.contents.l_text else b""
    b = Bam1()
    recs = []
    while L.bam_read1(fp, C.byref(b)) >= 0:
        recs.append(_record(b))
    C.CDLL(None).free(b.data)
    L.bam_header_destroy(h)
    L.bgzf_close(fp)
    return refs, lens, text, recs


def index_build(path):
    """bam_index_build: writes path + '.bai'"""
    if lib().bam_index_build(path.encode()) != 0:
        raise IOError("bam_index_build failed")


def fetch(path, tid, beg, end):
    """records bam_fetch hands to its callback for (tid, [beg, end)), index from path + '.bai'"""
    L = lib()
    fp = L.bgzf_open(path.encode(), b"r")
    idx = L.bam_index_load(path.encode())
    if not fp or not idx:
        raise IOError("cannot open %s or its index" % path)
    out = []

    def cb(bp, _):
        out.append(_record(bp.contents))
        return 0
    L.bam_fetch(fp, idx, tid, beg, end, None, _FETCH_F(cb))
    L.bam_index_destroy(idx)
    L.bgzf_close(fp)
    return out


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def write(path, refs, lengths, text, records):
    """a BAM written by bam_header_write / bam_write1 (the library's own BGZF writer and compression
    level).  records: dicts with tid, pos, mapq, flag, name, cigar (uint32 words), seq, qual, mtid, mpos,
    isize and optional aux bytes, in file order."""
    L = lib()
    fp = L.bgzf_open(path.encode(), b"w")
    if not fp:
        raise IOError("bgzf_open failed: " + path)
    h = Header()
    h.n_targets = len(refs)
    names = (C.c_char_p * len(refs))(*refs)
    lens = (C.c_uint32 * len(refs))(*lengths)
    h.target_name = C.cast(names, C.POINTER(C.c_char_p))
    h.target_len = C.cast(lens, C.POINTER(C.c_uint32))
    h.text = text
    h.l_text = len(text)
    L.bam_header_write(fp, C.byref(h))
    code = {ch: i for i, ch in enumerate(_BASES)}
    for r in records:
        seq = r["seq"]
        packed = bytearray((len(seq) + 1) // 2)
        for j, ch in enumerate(seq):
            packed[j >> 1] |= code.get(ch, 15) << (4 * (1 - (j & 1)))
        cigar = np.asarray(r["cigar"], dtype="<u4")
        aux = r.get("aux", b"")
        data = r["name"] + b"\x00" + cigar.tobytes() + bytes(packed) + bytes(r["qual"]) + aux
        end = r["pos"] + sum(int(w) >> 4 for w in cigar if (int(w) & 15) in (0, 2, 3))
        b = Bam1()
        b.core.tid, b.core.pos = r["tid"], r["pos"]
        b.core.bin_mq_nl = reg2bin(r["pos"], max(end, r["pos"] + 1)) | (r["mapq"] << 16) | ((len(r["name"]) + 1) << 24)
        b.core.flag_nc = r["flag"] | (len(cigar) << 16)
        b.core.l_qseq = len(seq)
        b.core.mtid, b.core.mpos, b.core.isize = r.get("mtid", -1), r.get("mpos", -1), r.get("isize", 0)
        b.l_aux, b.data_len, b.m_data = len(aux), len(data), len(data)
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        b.data = C.cast(buf, C.POINTER(C.c_uint8))
        L.bam_write1(fp, C.byref(b))
    L.bgzf_close(fp)
