"""ctypes front-end of the CPU checker -- TEST INFRASTRUCTURE ONLY.

Loads ``oracle/_build/libplaty_oracle.so`` (the repo's scalar C restatement of
the Platypus 0.1.5 alignment path) and, when it has been built,
``oracle/_ref/libalign_ref.so`` (the reference's own ``fastAlignmentRoutine``
compiled from its sources, platypus/c/align.c:18).  Only tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` import
this module; the product package never does.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libplaty_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libalign_ref.so")

# chaplotype.pyx:40 -- the literal equals the computed table of chaplotype.pyx:37-38
HOMOPOLQ = b"NKJHFA=854210/.-,,+**))((('''&&&%%%$$$$#####\"\"\"\"\""

ASSERT_FAILED = -1


class _Batch(C.Structure):
    _fields_ = [
        ("hap_seq", C.c_char_p), ("hap_off", C.c_void_p), ("hap_start", C.c_void_p),
        ("hap_varlen", C.c_void_p),
        ("read_seq", C.c_char_p), ("read_qual", C.c_char_p), ("read_off", C.c_void_p),
        ("read_pos", C.c_void_p),
        ("pair_hap", C.c_void_p), ("pair_read", C.c_void_p),
        ("table", C.c_char_p),
        ("gapextend", C.c_int), ("nucprior", C.c_int), ("reuse_hash", C.c_int),
    ]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle not built: run `make -C oracle`")
        L = C.CDLL(ORACLE_SO)
        L.po_fast_align.restype = C.c_int
        L.po_fast_align.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_char_p]
        L.po_gap_open.restype = None
        L.po_gap_open.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_void_p]
        L.po_hash7.restype = C.c_uint
        L.po_hash7.argtypes = [C.c_char_p]
        L.po_hash_sequence.restype = None
        L.po_hash_sequence.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.po_map_read.restype = C.c_int
        L.po_map_read.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.po_likelihood.restype = C.c_double
        L.po_likelihood.argtypes = [C.c_int, C.c_int]
        L.po_align_no_traceback.restype = C.c_int
        L.po_align_no_traceback.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                            C.c_char_p, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_char_p, C.c_void_p,
                                            C.POINTER(C.c_int)]
        L.po_genotype_loglik.restype = C.c_double
        L.po_genotype_loglik.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.po_run_pairs.restype = None
        L.po_run_pairs.argtypes = [C.POINTER(_Batch), C.c_long, C.c_long, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
        L.po_n_genotypes.restype = C.c_int
        L.po_n_genotypes.argtypes = [C.c_int]
        L.po_caller_setup.restype = None
        L.po_caller_setup.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_caller_call.restype = C.c_int
        L.po_caller_call.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
        L.po_variant_posterior.restype = C.c_double
        L.po_variant_posterior.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_double]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The reference's own compiled routine (platypus/c/align.h:6)."""
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.fastAlignmentRoutine.restype = C.c_int
        R.fastAlignmentRoutine.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_char_p, C.c_char]
        _ref = R
    return _ref


def fast_align(seq1, seq2, qual2, gapextend=3, nucprior=2, table=HOMOPOLQ):
    """Restatement of fastAlignmentRoutine; seq1 must be len(seq2)+15 bytes."""
    return lib().po_fast_align(bytes(seq1), bytes(seq2), bytes(qual2), len(seq1), len(seq2),
                               gapextend, nucprior, table)


def ref_fast_align(seq1, seq2, qual2, gapextend=3, nucprior=2, table=HOMOPOLQ):
    return ref().fastAlignmentRoutine(bytes(seq1), bytes(seq2), bytes(qual2), len(seq1),
                                      len(seq2), gapextend, nucprior, table, b"\x01")


def gap_open(seq1, table=HOMOPOLQ):
    out = np.zeros(len(seq1), dtype=np.int16)
    lib().po_gap_open(bytes(seq1), len(seq1), table, out.ctypes.data)
    return out


def hash_sequence(seq):
    tab = np.zeros(16384, dtype=np.int32)
    lib().po_hash_sequence(bytes(seq), len(seq), tab.ctypes.data)
    return tab


def map_read(read, seqlen, tab):
    return lib().po_map_read(bytes(read), len(read), seqlen, tab.ctypes.data)


def likelihood(score, mapq):
    return lib().po_likelihood(int(score), int(mapq))


def align_no_traceback(read, qual, rpos, hap, hap_start, var_seq_len,
                       gapextend=3, nucprior=2, table=HOMOPOLQ):
    a = C.c_int(0)
    sc = lib().po_align_no_traceback(bytes(read), bytes(qual), len(read), rpos, bytes(hap),
                                     len(hap), hap_start, var_seq_len, gapextend, nucprior,
                                     table, None, C.byref(a))
    return sc, a.value


def genotype_loglik(l1, l2):
    l1 = np.ascontiguousarray(l1, dtype=np.float64)
    l2 = np.ascontiguousarray(l2, dtype=np.float64)
    return lib().po_genotype_loglik(l1.ctypes.data, l2.ctypes.data, len(l1))


def run_pairs(batch, use_ref=False, threads=1, reuse_hash=False, lo=0, hi=None,
              gapextend=3, nucprior=2, table=HOMOPOLQ):
    """Score pairs [lo, hi) of a packed batch (dict of numpy arrays with keys
    hap_seq, hap_off, hap_start, hap_varlen, read_seq, read_qual, read_off,
    read_pos, pair_hap, pair_read).  ``use_ref`` routes the DP through the
    reference's own compiled routine.  Returns (score, anchor) int32 arrays
    of length hi-lo.  ``threads`` splits the range over OS threads (ctypes
    releases the GIL)."""
    L = lib()
    n = len(batch["pair_hap"])
    hi = n if hi is None else hi
    keep = {k: np.ascontiguousarray(batch[k]) for k in batch}
    assert keep["hap_off"].dtype == np.int64 and keep["read_off"].dtype == np.int64
    for k in ("hap_start", "hap_varlen", "read_pos", "pair_hap", "pair_read"):
        assert keep[k].dtype == np.int32, k
    b = _Batch()
    b.hap_seq = keep["hap_seq"].ctypes.data_as(C.c_char_p)
    b.hap_off = keep["hap_off"].ctypes.data
    b.hap_start = keep["hap_start"].ctypes.data
    b.hap_varlen = keep["hap_varlen"].ctypes.data
    b.read_seq = keep["read_seq"].ctypes.data_as(C.c_char_p)
    b.read_qual = keep["read_qual"].ctypes.data_as(C.c_char_p)
    b.read_off = keep["read_off"].ctypes.data
    b.read_pos = keep["read_pos"].ctypes.data
    b.pair_hap = keep["pair_hap"].ctypes.data
    b.pair_read = keep["pair_read"].ctypes.data
    b.table = table
    b.gapextend, b.nucprior, b.reuse_hash = gapextend, nucprior, int(reuse_hash)
    score = np.zeros(n, dtype=np.int32)
    anchor = np.zeros(n, dtype=np.int32)
    fn = None
    if use_ref:
        fn = C.cast(ref().fastAlignmentRoutine, C.c_void_p)
    bounds = np.linspace(lo, hi, threads + 1).astype(np.int64)

    def work(i):
        L.po_run_pairs(C.byref(b), int(bounds[i]), int(bounds[i + 1]), fn,
                       score.ctypes.data, anchor.ctypes.data)

    if threads == 1:
        work(0)
    else:
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    return score[lo:hi], anchor[lo:hi]


# ---- genotype likelihood / EM / variant posterior stage (platypus/cpopulation.pyx) ----

def caller_setup(like, ind_read_off):
    """Caller.setup (cpopulation.pyx:658-739): like is the window's [n_hap, n_reads] likelihood matrix;
    returns the normalised genotype likelihoods [n_ind, n_gen]."""
    like = np.ascontiguousarray(like, dtype=np.float64)
    off = np.ascontiguousarray(ind_read_off, dtype=np.int32)
    n_hap, n_reads = like.shape
    n_ind = len(off) - 1
    post = np.zeros((n_ind, lib().po_n_genotypes(n_hap)), dtype=np.float64)
    lib().po_caller_setup(n_hap, n_ind, n_reads, off.ctypes.data, like.ctypes.data, post.ctypes.data)
    return post


def caller_call(post, ind_read_off, n_hap, max_iters=100, literal=False, want_cisr=False):
    """Caller.call (cpopulation.pyx:741-805): returns dict(freq, iters, max_change, best_gen, best_val[, cisr])."""
    post = np.ascontiguousarray(post, dtype=np.float64)
    off = np.ascontiguousarray(ind_read_off, dtype=np.int32)
    n_ind = len(off) - 1
    freq = np.zeros(n_hap, dtype=np.float64)
    cisr = np.zeros_like(post) if want_cisr else None
    mc = C.c_double(0.0)
    best = np.zeros(n_ind, dtype=np.int32)
    val = np.zeros(n_ind, dtype=np.float64)
    iters = lib().po_caller_call(n_hap, n_ind, off.ctypes.data, post.ctypes.data, max_iters, int(literal),
                                 freq.ctypes.data, cisr.ctypes.data if want_cisr else None, C.byref(mc),
                                 best.ctypes.data, val.ctypes.data)
    out = dict(freq=freq, iters=iters, max_change=mc.value, best_gen=best, best_val=val)
    if want_cisr:
        out["cisr"] = cisr
    return out


def variant_posterior(post, ind_read_off, freq, hap_has_var, prior):
    """Population.calculatePosterior (cpopulation.pyx:515-632)."""
    post = np.ascontiguousarray(post, dtype=np.float64)
    off = np.ascontiguousarray(ind_read_off, dtype=np.int32)
    freq = np.ascontiguousarray(freq, dtype=np.float64)
    m = np.ascontiguousarray(hap_has_var, dtype=np.uint8)
    return lib().po_variant_posterior(len(freq), len(off) - 1, off.ctypes.data, post.ctypes.data,
                                      freq.ctypes.data, m.ctypes.data, float(prior))
