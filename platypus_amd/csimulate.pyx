# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""The inner loops of the whole-chromosome read simulator (platypus_amd/simulate.py: simulate_pairs):
placement of a read of a mutated copy on the reference (position + CIGAR) and the bases and qualities of a
read.  Test and benchmark data only; nothing of the calling path uses it."""
import numpy as np

from libc.stdint cimport int32_t, int64_t, uint32_t, uint64_t, uint8_t, int8_t

cdef enum:
    MAXC = 12                   # CIGAR operations kept per read (more: the read is dropped)


cdef inline uint64_t mix(uint64_t z) noexcept nogil:        # splitmix64
    z += <uint64_t>0x9e3779b97f4a7c15ULL
    z = (z ^ (z >> 30)) * <uint64_t>0xbf58476d1ce4e5b9ULL
    z = (z ^ (z >> 27)) * <uint64_t>0x94d049bb133111ebULL
    return z ^ (z >> 31)


cdef inline uint64_t step(uint64_t* s) noexcept nogil:      # xorshift64*
    cdef uint64_t x = s[0]
    x ^= x >> 12
    x ^= x << 25
    x ^= x >> 27
    s[0] = x
    return x * <uint64_t>0x2545f4914f6cdd1dULL


cdef inline double unit(uint64_t* s) noexcept nogil:
    return <double>(step(s) >> 11) * (1.0 / 9007199254740992.0)


def place(uint64_t seed, int copy_id,
          const int64_t[::1] seg_ref, const int64_t[::1] seg_hap, const int64_t[::1] seg_len, const uint8_t[::1] seg_op,
          const int64_t[::1] run_start, const int32_t[::1] run_len,
          const int64_t[::1] frag_start, const int32_t[::1] frag_len, const int64_t[::1] frag_id, int read_len,
          double run_err_scale):
    """Both reads of every fragment of one mutated copy, placed on the reference.

    Segments describe the copy: (reference start, copy start, length, op) with op 0 = M, 1 = I, 2 = D, in order.
    A read may carry one sequencing indel error inside a homopolymer run of the copy (`run_start`, `run_len`:
    runs of at least six equal bases): with probability run_err_scale * len^3 (at most 0.12) per run it reads one
    base too few (two times in three) or one too many, written at the run's left end as an aligner would.
    Returns per read (two per fragment, left then right): pos (int32, -1 = unplaceable), the copy interval
    [hs, hs + hlen), CIGAR words [n, MAXC] and their count, error kind (0, 1 = base missing, 2 = base extra) and
    the error's offset in the copy interval, the alignment's end on the reference."""
    cdef Py_ssize_t nf = frag_start.shape[0], f, k, side, i, first, lo, hi, mid, nseg = seg_ref.shape[0], nrun = run_start.shape[0]
    cdef Py_ssize_t n = 2 * nf
    pos = np.full(n, -1, dtype=np.int32)
    end = np.zeros(n, dtype=np.int32)
    hs_out = np.zeros(n, dtype=np.int64)
    hlen_out = np.zeros(n, dtype=np.int32)
    cig = np.zeros((n, MAXC), dtype=np.uint32)
    ncig = np.zeros(n, dtype=np.int32)
    ekind = np.zeros(n, dtype=np.int8)
    eoff = np.zeros(n, dtype=np.int32)
    cdef int32_t[::1] vpos = pos, vend = end, vhlen = hlen_out, vncig = ncig, veoff = eoff
    cdef int64_t[::1] vhs = hs_out
    cdef uint32_t[:, ::1] vcig = cig
    cdef int8_t[::1] vek = ekind
    cdef int64_t hs, he, a, b, rs, hps, ln, rpos, rend, e_at
    cdef int op, nc, kind, L = read_len, cop, clen, started
    cdef uint64_t rng
    cdef double p
    cdef uint32_t ops[64]
    cdef int lens[64]
    cdef int64_t hoff[64]
    cdef int no, j, placed_err, cut
    with nogil:
        for f in range(nf):
            for side in range(2):
                k = 2 * f + side
                rng = mix(seed ^ mix(<uint64_t>(2 * frag_id[f] + side) * 4 + <uint64_t>copy_id + 17))
                if rng == 0:
                    rng = 1
                hs = frag_start[f] if side == 0 else frag_start[f] + frag_len[f] - L
                # ---- a sequencing indel error inside a homopolymer run of the copy? ----
                kind = 0
                e_at = -1
                if nrun > 0 and run_err_scale > 0:
                    lo = 0; hi = nrun
                    while lo < hi:
                        mid = (lo + hi) >> 1
                        if run_start[mid] < hs + 12:
                            lo = mid + 1
                        else:
                            hi = mid
                    i = lo
                    while i < nrun and run_start[i] + run_len[i] <= hs + L - 14:
                        p = run_err_scale * <double>run_len[i] * run_len[i] * run_len[i]
                        if p > 0.12:
                            p = 0.12
                        if unit(&rng) < p:
                            kind = 1 if unit(&rng) < 0.6667 else 2
                            e_at = run_start[i]                 # the run's first base
                            break
                        i += 1
                he = hs + L + (1 if kind == 1 else (-1 if kind == 2 else 0))
                # ---- the copy interval [hs, he) as CIGAR operations ----
                lo = 0; hi = nseg
                while lo < hi:                                   # first segment that ends after hs
                    mid = (lo + hi) >> 1
                    if seg_hap[mid] + (seg_len[mid] if seg_op[mid] != 2 else 0) <= hs:
                        lo = mid + 1
                    else:
                        hi = mid
                first = lo
                if first > 0:
                    first -= 1
                no = 0
                rpos = -1
                rend = -1
                i = first
                while i < nseg and no < 60:
                    hps = seg_hap[i]; ln = seg_len[i]; op = seg_op[i]; rs = seg_ref[i]
                    if hps >= he:
                        break
                    if op == 2:
                        if rpos >= 0 and hps > hs and hps < he:
                            ops[no] = 2; lens[no] = <int>ln; hoff[no] = hps; no += 1
                        i += 1
                        continue
                    a = hs if hs > hps else hps
                    b = he if he < hps + ln else hps + ln
                    if a >= b:
                        i += 1
                        continue
                    if op == 0:
                        if rpos < 0:
                            rpos = rs + (a - hps)
                        ops[no] = 0; lens[no] = <int>(b - a); hoff[no] = a; no += 1
                    else:
                        ops[no] = 4 if rpos < 0 else 1           # a read that starts inside an insertion: soft clip
                        lens[no] = <int>(b - a); hoff[no] = a; no += 1
                    i += 1
                if rpos < 0 or no >= 60:
                    continue
                while no > 0 and (ops[no - 1] == 1 or ops[no - 1] == 2):     # an alignment does not end with an indel
                    if ops[no - 1] == 1:
                        ops[no - 1] = 4
                        break
                    no -= 1
                # ---- the error splits the M operation it lies in (with room on both sides) ----
                placed_err = 0
                nc = 0
                rend = rpos
                for j in range(no):
                    cop = ops[j]; clen = lens[j]
                    if kind and not placed_err and cop == 0 and e_at >= hoff[j] + 10 and e_at + 12 <= hoff[j] + clen:
                        cut = <int>(e_at - hoff[j])
                        if nc + 3 > MAXC:
                            nc = MAXC + 1
                            break
                        if nc > 0 and (vcig[k, nc - 1] & 0xf) == 0:
                            vcig[k, nc - 1] += (<uint32_t>cut << 4)
                        else:
                            vcig[k, nc] = (<uint32_t>cut << 4); nc += 1
                        if kind == 1:                            # the read lacks the base at e_at: a deletion
                            vcig[k, nc] = (1 << 4) | 2; nc += 1
                            vcig[k, nc] = (<uint32_t>(clen - cut - 1) << 4); nc += 1
                        else:                                    # the read repeats it: an insertion
                            vcig[k, nc] = (1 << 4) | 1; nc += 1
                            vcig[k, nc] = (<uint32_t>(clen - cut) << 4); nc += 1
                        placed_err = 1
                        rend += clen
                        continue
                    if nc > 0 and (vcig[k, nc - 1] & 0xf) == <uint32_t>cop:
                        vcig[k, nc - 1] += (<uint32_t>clen << 4)
                    else:
                        if nc >= MAXC:
                            nc = MAXC + 1
                            break
                        vcig[k, nc] = (<uint32_t>clen << 4) | <uint32_t>cop; nc += 1
                    if cop == 0 or cop == 2:
                        rend += clen
                if nc > MAXC:
                    continue
                if kind and not placed_err:                      # no room for the error beside a variant: the read is given up
                    continue
                vpos[k] = <int32_t>rpos
                vend[k] = <int32_t>rend
                vhs[k] = hs
                vhlen[k] = <int32_t>(he - hs)
                vncig[k] = nc
                vek[k] = kind
                veoff[k] = <int32_t>(e_at - hs) if kind else 0
    return dict(pos=pos, end=end, hs=hs_out, hlen=hlen_out, cigar=cig, ncig=ncig, ekind=ekind, eoff=eoff)


def fill(uint64_t seed, const uint8_t[::1] hap0, const uint8_t[::1] hap1, const uint8_t[::1] which,
         const int64_t[::1] hs, const int32_t[::1] hlen, const int8_t[::1] ekind, const int32_t[::1] eoff,
         const uint8_t[::1] reverse, const int64_t[::1] read_id, int read_len,
         uint8_t[::1] seq, uint8_t[::1] qual, Py_ssize_t lo, Py_ssize_t hi):
    """Bases and qualities of reads lo..hi-1 (rows of read_len bytes in seq / qual), without the interpreter lock.

    Quality model: a base is good (Q30-41) with probability 0.85, middling (Q20-29) 0.06, poor (Q2-19) 0.09; over the
    last 15 cycles of a read poor bases are four times as frequent, and one read in fifty ends in 5-30 cycles of Q2.
    A base is misread with the probability its quality states (at most 0.5), so errors sit where the qualities are
    low: about 0.02 per base overall.  Reverse-strand reads have their cycles counted from the right."""
    cdef Py_ssize_t k, j, L = read_len, src, cyc, tail
    cdef uint64_t rng, r
    cdef const uint8_t* hap
    cdef int q, cls
    cdef uint32_t u, v, thr_poor
    cdef uint8_t b
    cdef double perr[42]
    cdef int qq
    for qq in range(42):
        perr[qq] = 10.0 ** (-qq / 10.0)
        if perr[qq] > 0.5:
            perr[qq] = 0.5
    cdef const uint8_t* acgt = b"ACGT"
    with nogil:
        for k in range(lo, hi):
            rng = mix(seed ^ mix(<uint64_t>read_id[k] * 2 + 1))
            if rng == 0:
                rng = 1
            hap = &hap0[0] if which[k] == 0 else &hap1[0]
            tail = 0
            if (step(&rng) & 63) == 0:                            # (about one read in fifty after the next test)
                if (step(&rng) % 100) < 78:
                    tail = 5 + <Py_ssize_t>(step(&rng) % 26)
            for j in range(L):
                # source base in the copy
                src = j
                if ekind[k] == 1 and j >= eoff[k]:
                    src = j + 1                                   # the base at eoff is skipped
                elif ekind[k] == 2 and j > eoff[k]:
                    src = j - 1                                   # the base at eoff is read twice
                b = hap[hs[k] + src]
                cyc = (L - 1 - j) if reverse[k] else j
                r = step(&rng)
                u = <uint32_t>(r & 0xfffff)                       # class
                v = <uint32_t>((r >> 20) & 0xffffff)              # error
                thr_poor = 94372 if cyc < L - 15 else 377487      # 0.09 / 0.36 of 2^20
                if cyc >= L - tail:
                    q = 2
                elif u < thr_poor:
                    q = 2 + <int>((r >> 44) % 18)
                elif u < thr_poor + 62915:                        # 0.06
                    q = 20 + <int>((r >> 44) % 10)
                else:
                    q = 30 + <int>((r >> 44) % 12)
                if <double>v < perr[q] * 16777216.0:
                    b = acgt[(r >> 50) & 3]                       # (one time in four the same base again)
                seq[k * L + j] = b
                qual[k * L + j] = <uint8_t>q
