"""ctypes binding of libplatgpu.so (include/platgpu.h).

The library is the product: if it is missing or does not load, importing this
module raises -- there is no Python or CPU fallback for the alignment path.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLATGPU_LIB points at another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("PLATGPU_LIB") or os.path.join(_HERE, "lib", "libplatgpu.so")

PG_OK = 0
PG_ERR_NO_DEVICE, PG_ERR_INVALID, PG_ERR_HIP, PG_ERR_UNSUPPORTED, PG_ERR_STATE = -1, -2, -3, -4, -5
PG_SCORE_ASSERT_FAILED = -1
PG_K_ANCHOR, PG_K_ALIGN, PG_K_GENOTYPE, PG_K_CALLER_SETUP, PG_K_CALLER_EM = 0, 1, 2, 3, 4
ABI_VERSION = 8
NOALLOC_MAX_PAIRS, NOALLOC_MAX_INDEL_HAPS = 262144, 65536      # PG_NOALLOC_MAX_* of include/platgpu.h

# every symbol include/platgpu.h declares: (restype, argtypes)
_i32p, _i64p, _f64p = C.c_void_p, C.c_void_p, C.c_void_p
SYMBOLS = {
    "pg_abi_version": (C.c_int, []),
    "pg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pg_ctx_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pg_ctx_destroy": (C.c_int, [C.c_void_p]),
    "pg_last_error_string": (C.c_char_p, [C.c_void_p]),
    "pg_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "pg_host_free": (C.c_int, [C.c_void_p]),
    "pg_set_params": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "pg_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "pg_haps_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64p, _i32p, _i32p]),
    "pg_haps_build": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64p, _i32p, C.c_int, _i32p, _i32p, _i32p, _i32p, _i32p, C.c_void_p]),
    "pg_haps_fetch": (C.c_int, [C.c_void_p, C.c_void_p, _i64p]),
    "pg_reads_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _i64p, _i32p, _i32p]),
    "pg_reads_upload_nibbles": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _i64p, _i32p, _i32p]),
    "pg_pairs_upload": (C.c_int, [C.c_void_p, C.c_int64, _i32p, _i32p]),
    "pg_pairs_block": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pg_pairs_blocks": (C.c_int, [C.c_void_p, C.c_int, _i32p, _i32p, _i32p, _i32p, _i64p]),
    "pg_align_launch": (C.c_int, [C.c_void_p]),
    "pg_sync": (C.c_int, [C.c_void_p]),
    "pg_results_fetch": (C.c_int, [C.c_void_p, _i32p, _i32p, _f64p]),
    "pg_results_fetch_async": (C.c_int, [C.c_void_p, _i32p, _i32p, _f64p]),
    "pg_align_pairs": (C.c_int, [C.c_void_p, C.c_int64, _i32p, _i32p, _i32p, _i32p, _f64p]),
    "pg_align_window": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64p, _i32p, _i32p, C.c_int, C.c_void_p, C.c_void_p, _i64p,
                                  _i32p, _i32p, _i32p, _i32p, _f64p]),
    "pg_align_pairs_noalloc": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pg_kernel_time_ms": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "pg_anchor_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "pg_genotype_loglik": (C.c_int, [C.c_void_p, C.c_int64, _i64p, _i64p, _i32p, _f64p]),
    "pg_hash_sequence": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, _i32p]),
    "pg_map_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, _i32p, C.POINTER(C.c_int32)]),
    "pg_genotype_loglik_rows": (C.c_int, [C.c_void_p, C.c_int, _f64p, _f64p, C.POINTER(C.c_double)]),
    "pg_windows_upload": (C.c_int, [C.c_void_p, C.c_int, _i32p, _i32p, _i64p, _i32p]),
    "pg_caller_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_caller_fetch": (C.c_int, [C.c_void_p, _f64p, _i32p, _f64p, _i32p, _f64p, _f64p]),
    "pg_variant_posteriors": (C.c_int, [C.c_void_p, C.c_int, _i32p, C.c_void_p, _f64p, _f64p]),
    "pg_valu_int16_peak": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "pg_valu_vop2_rate": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "pg_inflater_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pg_inflater_destroy": (C.c_int, [C.c_void_p]),
    "pg_inflater_error": (C.c_char_p, [C.c_void_p]),
    "pg_inflater_set_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_inflater_last_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "pg_inflater_input": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pg_bgzf_scan": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "pg_bgzf_inflate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
}


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError("libplatgpu.so is not built (%s): run `python -m platypus_amd.build`" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype, fn.argtypes = res, args
    if lib.pg_abi_version() != ABI_VERSION:
        raise ImportError("libplatgpu.so ABI %d, binding expects %d" % (lib.pg_abi_version(), ABI_VERSION))
    return lib


LIB = load_library()


class PlatGpuError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libplatgpu error %d: %s" % (code, msg))
        self.code = code


def _ptr(a):
    return None if a is None else a.ctypes.data


def _arr(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def pinned_array(shape, dtype):
    """numpy array in page-locked host memory (pg_host_alloc); freed when the array and its views are gone"""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    p = C.c_void_p()
    rc = LIB.pg_host_alloc(max(1, n * dtype.itemsize), C.byref(p))
    if rc != PG_OK:
        raise PlatGpuError(rc, (LIB.pg_last_error_string(None) or b"").decode())
    buf = (C.c_char * max(1, n * dtype.itemsize)).from_address(p.value)
    a = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
    _PINNED[p.value] = (buf, weakref.finalize(a.base if a.base is not None else a, LIB.pg_host_free, p.value))
    return a


_PINNED = {}


NIBBLE_CODES = b"=ACMGRSVTWYHKDBN"                     # bam_nt16_rev_table


def pack_nibbles(seq):
    """bases (uint8 characters of NIBBLE_CODES) two per byte, the first in the high half: the form
    pg_reads_upload_nibbles takes"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    code = np.full(256, 255, dtype=np.uint8)
    code[np.frombuffer(NIBBLE_CODES, dtype=np.uint8)] = np.arange(16, dtype=np.uint8)
    c = code[seq]
    if (c == 255).any():
        raise ValueError("a base outside %r" % NIBBLE_CODES)
    if len(c) & 1:
        c = np.concatenate([c, np.zeros(1, dtype=np.uint8)])
    return (c[0::2] << 4) | c[1::2]


def device_count():
    """gfx950 devices this process can open (initialises the HIP runtime)"""
    n = C.c_int(0)
    LIB.pg_device_count(C.byref(n))
    return n.value


class AlignContext(object):
    """One device context (include/platgpu.h: pg_ctx)."""

    def __init__(self, device=0, max_read_len=0):
        h = C.c_void_p()
        rc = LIB.pg_ctx_create(device, max_read_len, C.byref(h))
        if rc != PG_OK:
            raise PlatGpuError(rc, (LIB.pg_last_error_string(None) or b"").decode())
        self._h = h
        self.n_pairs = 0
        self.n_haps = 0

    def close(self):
        if self._h is not None:
            LIB.pg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != PG_OK:
            raise PlatGpuError(rc, (LIB.pg_last_error_string(self._h) or b"").decode())

    def set_params(self, gapextend=3, nucprior=2, table=None):
        if table is None:
            table = b"NKJHFA=854210/.-,,+**))((('''&&&%%%$$$$#####\"\"\"\"\""
        self._check(LIB.pg_set_params(self._h, gapextend, nucprior, table, len(table)))

    def set_option(self, name, value):
        """kernel-choice overrides for tests and measurements (include/platgpu.h: pg_set_option)"""
        self._check(LIB.pg_set_option(self._h, name.encode(), str(value).encode()))

    def upload_haps(self, seq, off, start_pos, var_seq_len):
        seq = _arr(seq, np.uint8); off = _arr(off, np.int64)
        sp = _arr(start_pos, np.int32); vl = _arr(var_seq_len, np.int32)
        self._check(LIB.pg_haps_upload(self._h, len(off) - 1, _ptr(seq), _ptr(off), _ptr(sp), _ptr(vl)))
        self.n_haps = len(off) - 1

    def build_haps(self, ref_seq, ref_off, ref_start, hap_stretch, hap_var_off, var_pos, var_n_removed, var_added_off,
                   var_added):
        """the haplotype batch from reference stretches and variant lists, put together on the device
        (include/platgpu.h: pg_haps_build)"""
        rs = _arr(ref_seq, np.uint8); ro = _arr(ref_off, np.int64); st = _arr(ref_start, np.int32)
        hs = _arr(hap_stretch, np.int32); hv = _arr(hap_var_off, np.int32)
        vp = _arr(var_pos, np.int32); vr = _arr(var_n_removed, np.int32); vo = _arr(var_added_off, np.int32)
        va = _arr(var_added, np.uint8)
        assert len(hv) == len(hs) + 1 and len(st) == len(ro) - 1 and len(vo) == len(vp) + 1 == len(vr) + 1
        self._check(LIB.pg_haps_build(self._h, len(st), _ptr(rs), _ptr(ro), _ptr(st), len(hs), _ptr(hs), _ptr(hv), _ptr(vp),
                                      _ptr(vr), _ptr(vo), _ptr(va)))
        self.n_haps = len(hs)

    def fetch_haps(self):
        """(sequence bytes, offsets) of the haplotype batch on the device"""
        off = np.empty(self.n_haps + 1, dtype=np.int64)
        self._check(LIB.pg_haps_fetch(self._h, None, _ptr(off)))
        seq = np.empty(int(off[-1]), dtype=np.uint8)
        self._check(LIB.pg_haps_fetch(self._h, _ptr(seq), None))
        return seq, off

    def upload_reads(self, seq, qual, off, pos, mapq):
        seq = _arr(seq, np.uint8); qual = _arr(qual, np.uint8); off = _arr(off, np.int64)
        pos = _arr(pos, np.int32); mapq = _arr(mapq, np.int32)
        self._check(LIB.pg_reads_upload(self._h, len(off) - 1, _ptr(seq), _ptr(qual), _ptr(off), _ptr(pos), _ptr(mapq)))

    def upload_reads_nibbles(self, seq4, qual, off, pos, mapq):
        """reads whose bases come two per byte in BAM's 4-bit code (include/platgpu.h: pg_reads_upload_nibbles)"""
        seq4 = _arr(seq4, np.uint8); qual = _arr(qual, np.uint8); off = _arr(off, np.int64)
        pos = _arr(pos, np.int32); mapq = _arr(mapq, np.int32)
        assert len(seq4) >= (int(off[-1]) + 1) // 2
        self._check(LIB.pg_reads_upload_nibbles(self._h, len(off) - 1, _ptr(seq4), _ptr(qual), _ptr(off), _ptr(pos), _ptr(mapq)))

    def upload_batch(self, b):
        """Haplotypes, reads and pairs of a packed batch (see platypus_amd.synth)."""
        self.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
        self.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
        self.upload_pairs(b["pair_hap"], b["pair_read"])

    def upload_pairs(self, hap_idx, read_idx):
        hi = _arr(hap_idx, np.int32); ri = _arr(read_idx, np.int32)
        assert len(hi) == len(ri)
        self._check(LIB.pg_pairs_upload(self._h, len(hi), _ptr(hi), _ptr(ri)))
        self.n_pairs = len(hi)

    def pairs_block(self, hap_lo, hap_hi, read_lo, read_hi):
        self._check(LIB.pg_pairs_block(self._h, hap_lo, hap_hi, read_lo, read_hi))
        self.n_pairs = (hap_hi - hap_lo) * (read_hi - read_lo)

    def pairs_blocks(self, hap_lo, n_haps, read_lo, n_reads):
        """dense blocks back to back; returns the index of each block's first pair"""
        hl = _arr(hap_lo, np.int32); nh = _arr(n_haps, np.int32)
        rl = _arr(read_lo, np.int32); nr = _arr(n_reads, np.int32)
        base = np.zeros(len(hl), dtype=np.int64)
        self._check(LIB.pg_pairs_blocks(self._h, len(hl), _ptr(hl), _ptr(nh), _ptr(rl), _ptr(nr), _ptr(base)))
        self.n_pairs = int((nh.astype(np.int64) * nr).sum())
        return base

    def launch(self):
        self._check(LIB.pg_align_launch(self._h))

    def sync(self):
        self._check(LIB.pg_sync(self._h))

    def fetch(self, likelihood=True):
        n = self.n_pairs
        score = np.empty(n, dtype=np.int32); anchor = np.empty(n, dtype=np.int32)
        like = np.empty(n, dtype=np.float64) if likelihood else None
        self._check(LIB.pg_results_fetch(self._h, _ptr(score), _ptr(anchor), _ptr(like)))
        return score, anchor, like

    def align_pairs(self, hap_idx, read_idx):
        self.upload_pairs(hap_idx, read_idx)
        if self.n_pairs == 0:
            z = np.empty(0, dtype=np.int32)
            return z, z.copy(), np.empty(0, dtype=np.float64)
        self.launch()
        return self.fetch()

    def align_window(self, hap_seq, hap_off, hap_start, hap_varlen, read_seq, read_qual, read_off, read_pos, read_mapq,
                     likelihood=True):
        """every haplotype against every read in one call (pair h * n_reads + r); see include/platgpu.h"""
        hs = _arr(hap_seq, np.uint8); ho = _arr(hap_off, np.int64); sp = _arr(hap_start, np.int32); vl = _arr(hap_varlen, np.int32)
        rs = _arr(read_seq, np.uint8); rq = _arr(read_qual, np.uint8); ro = _arr(read_off, np.int64)
        rp = _arr(read_pos, np.int32); mq = _arr(read_mapq, np.int32)
        nh, nr = len(ho) - 1, len(ro) - 1
        n = nh * nr
        score = np.empty(n, dtype=np.int32); anchor = np.empty(n, dtype=np.int32)
        like = np.empty(n, dtype=np.float64) if likelihood else None
        self._check(LIB.pg_align_window(self._h, nh, _ptr(hs), _ptr(ho), _ptr(sp), _ptr(vl), nr, _ptr(rs), _ptr(rq), _ptr(ro),
                                        _ptr(rp), _ptr(mq), _ptr(score), _ptr(anchor), _ptr(like)))
        self.n_pairs = n                      # the results stay on the device (fetch, genotype_loglik, the caller stage)
        return score, anchor, like

    def align_pairs_noalloc(self, n_pairs, d_hap_idx, d_read_idx, d_score, d_anchor, d_like):
        """device addresses (integers) of the caller's arrays; see include/platgpu.h"""
        self._check(LIB.pg_align_pairs_noalloc(self._h, n_pairs, d_hap_idx, d_read_idx, d_score, d_anchor, d_like))

    def kernel_ms(self, which=PG_K_ALIGN):
        ms = C.c_float(0)
        self._check(LIB.pg_kernel_time_ms(self._h, which, C.byref(ms)))
        return ms.value

    def anchor_stats(self):
        """(pairs the last launch anchored by 7-mer votes; of those: pairs a wavefront counted over the positions their
        7-mers name; pairs counted over every position)"""
        a = (C.c_int64 * 3)()
        self._check(LIB.pg_anchor_stats(self._h, a))
        return a[0], a[1], a[2]

    def valu_int16_peak(self):
        v = C.c_double(0)
        self._check(LIB.pg_valu_int16_peak(self._h, C.byref(v)))
        return v.value

    def valu_vop2_rate(self):
        v = C.c_double(0)
        self._check(LIB.pg_valu_vop2_rate(self._h, C.byref(v)))
        return v.value

    def genotype_loglik(self, base1, base2, n_reads):
        b1 = _arr(base1, np.int64); b2 = _arr(base2, np.int64); nr = _arr(n_reads, np.int32)
        out = np.empty(len(b1), dtype=np.float64)
        self._check(LIB.pg_genotype_loglik(self._h, len(b1), _ptr(b1), _ptr(b2), _ptr(nr), _ptr(out)))
        return out

    def hash_sequence(self, seq):
        table = np.empty(16384, dtype=np.int32)
        self._check(LIB.pg_hash_sequence(self._h, bytes(seq), len(seq), _ptr(table)))
        return table

    def map_read(self, read, seq_len, table):
        t = _arr(table, np.int32)
        pos = C.c_int32(0)
        self._check(LIB.pg_map_read(self._h, bytes(read), len(read), seq_len, _ptr(t), C.byref(pos)))
        return pos.value

    def genotype_loglik_rows(self, row1, row2):
        r1 = _arr(row1, np.float64); r2 = _arr(row2, np.float64)
        assert len(r1) == len(r2)
        out = C.c_double(0)
        self._check(LIB.pg_genotype_loglik_rows(self._h, len(r1), _ptr(r1), _ptr(r2), C.byref(out)))
        return out.value

    # ---- genotype likelihoods, EM, variant posteriors over windows ----

    def upload_windows(self, n_haps, n_inds, like_base, ind_read_off):
        nh = _arr(n_haps, np.int32); ni = _arr(n_inds, np.int32)
        lb = _arr(like_base, np.int64); io = _arr(ind_read_off, np.int32)
        assert len(nh) == len(ni) == len(lb) and len(io) == int(ni.sum()) + len(ni)
        self._check(LIB.pg_windows_upload(self._h, len(nh), _ptr(nh), _ptr(ni), _ptr(lb), _ptr(io)))
        self._win = (nh.copy(), ni.copy())

    def caller_launch(self, max_iters=100):
        self._check(LIB.pg_caller_launch(self._h, max_iters))

    def caller_fetch(self, genotype_post=True):
        """dict of flat arrays (see include/platgpu.h: pg_caller_fetch) plus the offsets to slice them."""
        nh, ni = self._win
        ng = nh.astype(np.int64) * (nh + 1) // 2
        hap_off = np.concatenate([[0], np.cumsum(nh, dtype=np.int64)])
        ind_off = np.concatenate([[0], np.cumsum(ni, dtype=np.int64)])
        cell_off = np.concatenate([[0], np.cumsum(ni * ng, dtype=np.int64)])
        freq = np.empty(int(hap_off[-1]), dtype=np.float64)
        iters = np.empty(len(nh), dtype=np.int32)
        change = np.empty(len(nh), dtype=np.float64)
        best = np.empty(int(ind_off[-1]), dtype=np.int32)
        val = np.empty(int(ind_off[-1]), dtype=np.float64)
        post = np.empty(int(cell_off[-1]), dtype=np.float64) if genotype_post else None
        self._check(LIB.pg_caller_fetch(self._h, _ptr(freq), _ptr(iters), _ptr(change), _ptr(best), _ptr(val), _ptr(post)))
        return dict(freq=freq, iters=iters, max_change=change, best_gen=best, best_val=val, post=post,
                    hap_off=hap_off, ind_off=ind_off, cell_off=cell_off, n_gen=ng)

    def variant_posteriors(self, var_window, hap_has_var, prior):
        vw = _arr(var_window, np.int32); m = _arr(hap_has_var, np.uint8); pr = _arr(prior, np.float64)
        out = np.empty(len(vw), dtype=np.float64)
        self._check(LIB.pg_variant_posteriors(self._h, len(vw), _ptr(vw), _ptr(m), _ptr(pr), _ptr(out)))
        return out


def bgzf_scan(raw):
    """(bytes the complete BGZF blocks at the start of `raw` occupy, bytes they inflate to, their number); host only"""
    raw = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8))
    eaten, total, nb = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
    rc = LIB.pg_bgzf_scan(_ptr(raw), raw.size, C.byref(eaten), C.byref(total), C.byref(nb))
    if rc != PG_OK:
        raise PlatGpuError(rc, "not a BGZF block")
    return eaten.value, total.value, nb.value


class Inflater(object):
    """BGZF inflation on the device (pg_inflater_*, pg_bgzf_inflate)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = LIB.pg_inflater_create(device, C.byref(h))
        if rc != PG_OK:
            raise PlatGpuError(rc, (LIB.pg_inflater_error(None) or b"").decode())
        self._h = h

    def close(self):
        if self._h:
            LIB.pg_inflater_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lanes(self, lanes):
        if LIB.pg_inflater_set_lanes(self._h, lanes) != PG_OK:
            raise ValueError("lanes must be 8, 16 or 32")

    def last_ms(self):
        ms = C.c_float(0)
        LIB.pg_inflater_last_ms(self._h, C.byref(ms))
        return ms.value

    def input_buffer(self, n):
        """uint8 view of the inflater's page-locked input buffer, n bytes"""
        buf = C.c_void_p()
        rc = LIB.pg_inflater_input(self._h, n, C.byref(buf))
        if rc != PG_OK:
            raise PlatGpuError(rc, (LIB.pg_inflater_error(self._h) or b"").decode())
        return np.frombuffer((C.c_uint8 * n).from_address(buf.value), dtype=np.uint8) if n else np.zeros(0, np.uint8)

    def inflate(self, raw, prefix=b""):
        """(uint8 array: prefix + the inflated complete blocks at the start of raw, bytes of raw consumed).  The array
        is a view of the inflater's own page-locked buffer: valid until the next call."""
        raw = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8))
        prefix = np.ascontiguousarray(np.frombuffer(prefix, dtype=np.uint8))
        out, n_out, eaten = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
        rc = LIB.pg_bgzf_inflate(self._h, _ptr(raw), raw.size, _ptr(prefix), prefix.size, C.byref(out), C.byref(n_out), C.byref(eaten))
        if rc != PG_OK:
            raise PlatGpuError(rc, (LIB.pg_inflater_error(self._h) or b"").decode())
        if n_out.value == 0:
            return np.zeros(0, dtype=np.uint8), eaten.value
        buf = (C.c_uint8 * n_out.value).from_address(out.value)
        return np.frombuffer(buf, dtype=np.uint8), eaten.value
