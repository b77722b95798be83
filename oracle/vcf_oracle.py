"""TEST INFRASTRUCTURE ONLY -- never imported by the product.

CPU restatement of the reference's per-window calling procedure and VCF output, from the
candidate list of a region to the text of its VCF body lines:

    platypus/window.py:21-129              windows of interacting variants
    platypus/cwindow.pyx:242-302           read range of a window, haplotypes of a window
    platypus/cfilter.pyx:23-78             every valid combination of the window's variants
    platypus/chaplotype.pyx:50-86,189-237,283-461   haplotype sequence, valid(), HP / SC
    platypus/variantcaller.pyx:116-191     the window loop, the pre-filter for crowded windows
    platypus/cpopulation.pyx:189-513       vcfINFO, vcfFILTER, vcfVARIANTS (sums in the reference's order)
    platypus/variantcaller.pyx:251-502     outputCallToVCF, refAndAlt
    platypus/vcf.py:297-328,710-739        the text of a line

Alignment, likelihood, Caller.setup / Caller.call and calculatePosterior come from the C
restatements (oracle/pyoracle.py: platypus/calign.pyx:36-158, c/align.c, cpopulation.pyx:515-890).

Parity unpinned: the reference program cannot run here and ships no VCF, so this file follows its
source text.  Where the reference's result depends on the iteration order of a Python 2 dict or set
(the order of `set([refHaplotype] + haplotypes)`, variantcaller.pyx:170; `for variant in INFO`,
cpopulation.pyx:263; `for v in FILTER`, cpopulation.pyx:450) no restatement can be exact: the order
used here is stated at each place (haplotypes in the order they are made; variants sorted).
"""
import decimal
import math
from itertools import combinations

import numpy as np

from oracle import pyoracle as po

mLTOT = -0.23025850929940459


class V(object):
    """the fields of platypus/variant.pyx:19-75 the procedure reads"""
    __slots__ = ("refName", "refPos", "removed", "added", "nAdded", "nRemoved", "minPos", "maxPos", "nSupportingReads",
                 "nFwdReads", "nRevReads", "nFwdReadsWeighted", "nRevReadsWeighted", "rmsReadPos")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def same(self):                      # equality, variant.pyx:203-215
        return (self.refName, self.refPos, self.removed, self.added)

    def order(self):                     # ordering, variant.pyx:165-201
        return (self.refName, self.refPos, abs(self.nAdded - self.nRemoved))

    def prior(self):                     # variant.pyx:78-105
        if self.nAdded == 1 and self.nRemoved == 1:
            return 1e-3
        if self.nAdded == self.nRemoved:
            return 5e-5 * (0.1 ** self.nAdded)
        if self.nAdded > 0 and self.nRemoved == 0:
            return 1e-4 * (0.33 ** self.nAdded)
        if self.nAdded == 0 and self.nRemoved > 0:
            return 1e-4 * (0.8 ** self.nRemoved)
        raise ValueError("no prior for this variant")


class Ref(object):
    """fastafile.pyx:91-174 over sequences held in memory"""
    def __init__(self, sequences):
        self.s = sequences

    def length(self, name):
        return len(self.s[name])

    def seq(self, name, begin, end):
        begin = max(0, begin)
        end = min(len(self.s[name]) - 1, end)                        # fastafile.pyx:154-155
        if end < begin:
            raise IndexError("Cannot have beginPos = %s, endPos = %s" % (begin, end))
        return self.s[name][begin:end].upper()

    def char(self, name, pos):
        return self.s[name][pos:pos + 1].upper()


class Reads(object):
    """one sample's buffered reads, sorted by position (what a bamReadBuffer holds after filtering and trimming)"""
    def __init__(self, pos, end, mapq, reverse, seqs, quals):
        self.pos, self.end, self.mapq, self.reverse, self.seqs, self.quals = pos, end, mapq, reverse, seqs, quals
        self.longest = max([e - p for p, e in zip(pos, end)], default=0)    # cwindow.pyx:220-221
        self.lo = self.hi = 0

    def set_window(self, start, end):
        """cwindow.pyx:242-267"""
        import bisect
        first = max(1, start - self.longest)
        lo = bisect.bisect_left(self.pos, first)
        hi = min(bisect.bisect_left(self.pos, end), len(self.pos))
        while lo < len(self.pos) and self.end[lo] <= start:                  # leading reads that end before the window
            lo += 1
        if lo > hi:
            raise RuntimeError("read start pointer > read end pointer")
        self.lo, self.hi = lo, hi
        return hi - lo


class Hap(object):
    def __init__(self, ref, name, start, end, variants, rlen):
        """chaplotype.pyx:50-86"""
        self.ref, self.name, self.variants = ref, name, tuple(variants)
        self.varSeqLen = sum(abs(v.nAdded - v.nRemoved) for v in variants)
        self.start = max(0, start - 2 * rlen)
        self.end = min(end + 2 * rlen, ref.length(name))
        self.seq = self.mutated()

    def mutated(self):
        """chaplotype.pyx:283-353"""
        ref, name = self.ref, self.name
        if not self.variants:
            return ref.seq(name, self.start, self.end)
        cur = self.start
        bits = [ref.seq(name, cur, self.variants[0].refPos)]
        cur = self.variants[0].refPos
        for v in self.variants:
            if v.refPos > cur:
                bits.append(ref.seq(name, cur, v.refPos))
                cur = v.refPos
            if v.nAdded == v.nRemoved:
                bits.append(v.added)
                cur += v.nAdded
            elif v.nAdded == 0 and v.nRemoved > 0:
                if v.refPos == cur:
                    bits.append(ref.char(name, v.refPos))
                    cur += 1
                cur += v.nRemoved
            elif v.nAdded > 0 and v.nRemoved == 0:
                if v.refPos == cur:
                    bits.append(ref.char(name, v.refPos))
                    cur += 1
                bits.append(v.added)
            else:
                raise ValueError("variant type not supported in getMutatedSequence()")
        if cur < self.end:
            bits.append(ref.seq(name, cur, self.end))
        return b"".join(bits)

    def valid(self):
        """chaplotype.pyx:189-237"""
        vs = self.variants
        for this, nxt in zip(vs, vs[1:]):
            if this.minPos > nxt.minPos:
                raise ValueError("Variants out of order in haplotype!")
            if this.maxPos >= nxt.minPos:
                if (this.nAdded > 0 and this.nRemoved == 0) and (nxt.nAdded == 0 and nxt.nRemoved == 0):
                    continue
                return False
        return True

    def has(self, v):
        return any(v.same() == u.same() for u in self.variants)

    def homopolymer(self, v):
        """chaplotype.pyx:366-404"""
        left = self.ref.seq(v.refName, v.refPos - 20, v.refPos)
        right = self.ref.seq(v.refName, v.refPos + 1, v.refPos + 21)
        if len(left) == 0 or len(right) == 0:
            return 0
        nl = nr = 0
        for c in reversed(left):
            if c != left[-1]:
                break
            nl += 1
        for c in right:
            if c != right[0]:
                break
            nr += 1
        return max(nl, nr) if left[-1] != right[0] else nl + nr

    def context(self, v):
        return self.ref.seq(v.refName, v.refPos - 10, v.refPos + 11)      # chaplotype.pyx:405-410


def windows_and_variants(chrom, start, end, sorted_variants, opts):
    """window.py:21-129"""
    by_pos = {}
    for v in sorted_variants:
        if v.refName == chrom and start <= v.refPos <= end:
            by_pos.setdefault(v.refPos, []).append(v)
    bunches = []
    for pos in sorted(by_pos):
        var_list = by_pos[pos]
        if bunches and max(v.maxPos for v in bunches[-1]) >= min(v.minPos for v in var_list):
            bunches[-1].extend(var_list)
        else:
            bunches.append(list(var_list))
    for vs in bunches:
        if opts.callOnlyIndels and not any(v.nAdded != v.nRemoved for v in vs):
            continue
        ws, we = max(vs[0].minPos, start), min(vs[-1].maxPos, end)
        if ws == we:
            if we == end:
                ws -= 1
            else:
                we += 1
        yield dict(chromosome=chrom, startPos=ws, endPos=we, variants=vs)


def haplotypes_in_window(window, n_reads, ref, opts):
    """cwindow.pyx:270-302 + cfilter.pyx:23-78 (the reference haplotype is added by the caller)"""
    chrom, ws, we, vs = window["chromosome"], window["startPos"], window["endPos"], window["variants"]
    ref_hap = Hap(ref, chrom, ws, we, (), opts.rlen)
    if n_reads == 0:
        return [ref_hap]
    if (n_reads * opts.rlen) / (we - ws + opts.rlen) > opts.maxReads:
        return [ref_hap]
    make = lambda sub: Hap(ref, chrom, ws, we, sub, opts.rlen)
    n = len(vs)
    if n == 1:
        return [make((vs[0],))]
    if n == 2:
        out = [make((vs[0],)), make((vs[1],))]
        if vs[0].refPos != vs[1].refPos:
            both = make((vs[0], vs[1]))
            if both.valid():
                out.append(both)
        return out
    if 2 ** n - 1 > opts.maxHaplotypes:
        return []
    out = []
    for k in range(1, n + 1):
        for sub in combinations(vs, k):
            h = make(sub)
            if h.valid():
                out.append(h)
    return out


def binomial(x, size, prob):
    """cpopulation.pyx:143-160 (its own copy, with the x == size == 0 case)"""
    def log_factorial(n):
        if n < 15:
            return sum(math.log(i) for i in range(1, n + 1))
        y = float(n)
        return (y * math.log(y) + math.log(2.0 * math.pi * y) / 2 - y + (y ** -1) / 12 - (y ** -3) / 360 + (y ** -5) / 1260
                - (y ** -7) / 1680 + (y ** -9) / 1188)
    if x == size and prob == 1:
        return 1.0
    if x != size and prob == 1:
        return 0.0
    if x == 0 and prob == 0:
        return 1.0
    if x == 0 and size == 0:
        return 1.0
    coeff = log_factorial(size) - (log_factorial(x) + log_factorial(size - x))
    clog = lambda p: math.log(p) if p > 0.0 else float("-inf")       # the C library's log(0)
    return math.exp(coeff + (x * clog(prob) + (size - x) * clog(1.0 - prob)))    # cpopulation.pyx:157-160


def py2_round(x, nd=0):
    """round() of Python 2.7: the exact binary value rounded to nd decimals, halves away from zero
    (floatobject.c float_round: correctly rounded through dtoa, not by scaling)"""
    if math.isinf(x) or math.isnan(x):                 # returned as they are (a log-likelihood of -inf is printed "-inf")
        return x
    q = decimal.Decimal(x).quantize(decimal.Decimal(1).scaleb(-nd), rounding=decimal.ROUND_HALF_UP)
    return float(q)


def py2_str(x):
    """str() of Python 2 for the values a line holds: floats with 12 significant digits"""
    if isinstance(x, float):
        s = "%.12g" % x
        if "." not in s and "e" not in s and "n" not in s and "i" not in s:
            s += ".0"
        return s
    if isinstance(x, bytes):
        return x.decode()
    return str(x)


class Population(object):
    """the result of Caller.setup + Caller.call for one window (cpopulation.pyx:658-805)"""

    def __init__(self, haps, samples, opts):
        self.haps, self.samples, self.opts = haps, samples, opts
        rows = []
        off = [0]
        for s in samples:
            off.append(off[-1] + (s.hi - s.lo))
        self.off = np.array(off, np.int32)
        self.window = [(s.lo, s.hi) for s in samples]
        for h in haps:                                               # chaplotype.pyx:259-268 per individual
            row = []
            for s in samples:
                for r in range(s.lo, s.hi):
                    sc, _ = po.align_no_traceback(s.seqs[r], s.quals[r], s.pos[r], h.seq, h.start, h.varSeqLen)
                    row.append(0.0 if sc < 0 else po.likelihood(sc, s.mapq[r]))
            rows.append(row)
        self.like = np.array(rows, np.float64).reshape(len(haps), off[-1])
        self.post = po.caller_setup(self.like, self.off)
        res = po.caller_call(self.post, self.off, len(haps), max_iters=opts.maxIters)
        self.freq = res["freq"]
        n = len(haps)
        self.genotypes = [(a, b) for a in range(n) for b in range(a, n)]      # cgenotype.pyx:90-96
        self.calls = [None if g < 0 else self.genotypes[g] for g in res["best_gen"]]
        self.call_values = res["best_val"]
        self.min_quals = []            # listOfMinBaseQuals is never emptied (cpopulation.pyx:233,311)
        self.weighted_fwd = 0.0        # weightedTCFVal is reset under another name (cpopulation.pyx:269)

    def posterior(self, v):
        mask = np.array([h.has(v) for h in self.haps], np.uint8)
        return po.variant_posterior(self.post, self.off, self.freq, mask, v.prior())

    def vcf_variants(self):
        """cpopulation.pyx:479-513: position -> variants of the haplotypes that some individual was called with"""
        out = {}
        for i, h in enumerate(self.haps):
            if self.freq[i] < 0.0:
                continue
            if not any(g is not None and i in g for g in self.calls):
                continue
            for v in h.variants:
                out.setdefault(v.refPos, {})[v.same()] = v
        return {p: list(d.values()) for p, d in out.items()}

    def vcf_info(self):
        """cpopulation.pyx:189-397.  `for variant in INFO` runs in sorted variant order here."""
        o = self.opts
        info, order = {}, {}
        for hi, h in enumerate(self.haps):
            for v in h.variants:
                k = v.same()
                if k not in info:
                    pp = "%.0f" % (-10.0 * math.log10(max(1e-20, 1.0 - self.posterior(v))))
                    info[k] = dict(HP=[h.homopolymer(v)], PP=[pp], TR=[v.nSupportingReads], NR=[v.nRevReads], NF=[v.nFwdReads],
                                   FR=[self.freq[hi]], SC=[h.context(v)], RMP=[py2_round(v.rmsReadPos, 2)])
                    order[k] = v
                else:
                    info[k]["FR"][0] += self.freq[hi]
        for k in sorted(info, key=lambda k: order[k].order()):
            v = order[k]
            tc = tcr = tcf = 0
            w_rev = 0.0
            for s, (lo, hi) in zip(self.samples, self.window):
                for r in range(lo, hi):
                    if s.pos[r] <= v.refPos <= s.end[r]:
                        if len(v.added) != 1:
                            continue
                        if s.seqs[r][v.refPos - s.pos[r]:v.refPos - s.pos[r] + 1] != v.added:
                            continue
                        ws = max(0, v.refPos - s.pos[r] - (o.badReadsWindow - 1) // 2)
                        we = min(len(s.seqs[r]), v.refPos - s.pos[r] + (o.badReadsWindow - 1) // 2)
                        mn = 100
                        for j in range(ws, we):
                            mn = min(mn, s.quals[r][j])
                        self.min_quals.append(mn)
            self.min_quals.sort()
            med = self.min_quals[len(self.min_quals) // 2] if self.min_quals else 100
            for s, (lo, hi) in zip(self.samples, self.window):
                for r in range(lo, hi):
                    if s.pos[r] <= v.refPos <= s.end[r]:
                        tc += 1
                        if s.reverse[r]:
                            tcr += 1
                            w_rev += 1.0 - math.exp(mLTOT * s.mapq[r])
                        else:
                            tcf += 1
                            self.weighted_fwd += 1.0 - math.exp(mLTOT * s.mapq[r])
            d = info[k]
            if len(v.added) == 1 and len(v.removed) == 1:
                d["MMLQ"] = [med]
            d["TC"], d["TCR"], d["TCF"] = [tc], [tcr], [tcf]
            fr = d["FR"][0]
            d["FR"][0] = "%1.4f" % fr
            ab = fwd = rev = 0.0
            weighted = v.nFwdReadsWeighted + v.nRevReadsWeighted
            ratio = weighted / tc if tc else (float("inf") if weighted > 0 else float("nan"))   # C double division
            if ratio >= fr / 2.0:
                ab = 1.0
            else:
                for i in range(int(py2_round(weighted)) + 1):
                    ab += binomial(i, tc, fr / 2.0)
            for i in range(int(py2_round(v.nFwdReadsWeighted)) + 1):
                fwd += binomial(i, int(py2_round(self.weighted_fwd)), min(0.99, fr))
            for i in range(int(py2_round(v.nRevReadsWeighted)) + 1):
                rev += binomial(i, int(py2_round(w_rev)), min(0.99, fr))
            d["FPV"], d["RPV"], d["ABPV"] = ["%1.2e" % fwd], ["%1.2e" % rev], ["%1.2e" % ab]
        self._order = order
        return info

    def vcf_filter(self, info):
        """cpopulation.pyx:399-477.  The flags go to the variant `for v in FILTER` ends with: the last in sorted order here."""
        o = self.opts
        keys = sorted(info, key=lambda k: self._order[k].order())
        flt = {k: [] for k in keys}
        n = n_sb = n_ab = n_bad = 0
        for k in keys:
            d = info[k]
            nf, nr = int(d["NF"][0]), int(d["NR"][0])
            med = int(d.get("MMLQ", [100])[0])
            if med < o.badReadsThreshold and not (nf > 0 and nr > 0):
                n_bad += 1
            n += 1
            if nf == 0 and int(d["TCF"][0]) != 0 and float(d["FPV"][0]) < o.sbThreshold:
                n_sb += 1
            if nr == 0 and int(d["TCR"][0]) != 0 and float(d["RPV"][0]) < o.sbThreshold:
                n_sb += 1
            if float(d["ABPV"][0]) < o.abThreshold:
                n_ab += 1
        last = keys[-1]
        if n_bad == n:
            flt[last].append("badReads")
        if n_sb == n:
            flt[last].append("sb")
        if n_ab == n:
            flt[last].append("ab")
        return flt


def ref_and_alt(chrom, pos, variants, ref):
    """variantcaller.pyx:442-502"""
    non_snp = any(v.nRemoved != 1 or v.nAdded != 1 for v in variants)
    indel = any(v.nRemoved != v.nAdded for v in variants)
    if not non_snp:
        return ref.char(chrom, pos), [v.added for v in variants]
    rlen = max(v.nRemoved for v in variants)
    REF = ref.seq(chrom, pos, pos + rlen + 1) if indel else ref.seq(chrom, pos, pos + rlen)
    alts = []
    for v in variants:
        seq = list(REF.decode())
        if v.nRemoved == v.nAdded:
            seq[0:len(v.added)] = v.added.decode()
        else:
            seq[1:1 + v.nRemoved] = v.added.decode()
        alts.append("".join(seq).encode())
    return REF, alts


def format_data(data, key=True, value=True, separator=":", missing=None):
    """vcf.py:297-328 (missing: the field's missing value, -1 for INFO fields, '.' for FORMAT fields)"""
    data = {k: ["." if x == missing else x for x in vs] for k, vs in data.items()}
    sdata = sorted((k, v) for k, v in data.items() if k != "GT")
    if "GT" in data:
        sdata = [("GT", ["".join(py2_str(x) for x in g) for g in data["GT"]])] + sdata
    out = []
    for k, v in sdata:
        if key and value:
            out.append(k + "=" + ",".join(py2_str(x) for x in v) if v else k)
        elif key:
            out.append(k)
        else:
            out.append(",".join(py2_str(x) for x in v) if v else ".")
    while len(out) > 1 and len(out[-1].replace(",", "").replace(".", "")) == 0:
        out = out[:-1]
    return separator.join(out)


def output_call(pop, ref, opts, lines):
    """variantcaller.pyx:251-439; appends one text line per written position"""
    var_dict = pop.vcf_variants()
    info = pop.vcf_info()
    flt = pop.vcf_filter(info) if info else {}
    n_ind = len(pop.samples)
    n_reads_this_individual = None                                   # unbound until the first single-variant position
    for POS in sorted(var_dict):
        variants = sorted([v for v in var_dict[POS] if float(info[v.same()]["PP"][0]) >= opts.minPosterior],
                          key=lambda v: v.order())
        if not variants:
            continue
        chrom = variants[0].refName
        REF, alt = ref_and_alt(chrom, POS, variants, ref)
        lineinfo = info[variants[0].same()]
        linefilter = []
        FR, PP, NF, NR, TR = [], [], [], [], []
        for v in variants:
            d = info[v.same()]
            linefilter.extend(flt[v.same()])
            FR.extend(d["FR"]); PP.extend(d["PP"]); NR.extend(d["NR"]); NF.extend(d["NF"]); TR.extend(d["TR"])
        lineinfo["FR"], lineinfo["PP"], lineinfo["NF"], lineinfo["NR"], lineinfo["TR"] = FR, PP, NF, NR, TR
        qual = max(int(pp) for pp in PP)
        linefilter = sorted(set(linefilter))                          # list(set(...)): order not defined in the reference
        the_var = variants[0]
        test = pop.haps[0]
        ref_hap = Hap(ref, test.name, test.start, test.end, (), 0)
        var_hap = Hap(ref, test.name, test.start, test.end, (the_var,), 0)
        sample_cols = []
        for i in range(n_ind):
            g = pop.calls[i]
            if g is None:
                sample_cols.append(dict(GT=[[".", "/", "."]], GL=[-1, -1, -1], GQ=[-1], NR=[0]))
                continue
            gq = int(-10.0 * math.log10(max(1e-10, 1.0 - pop.call_values[i])))
            a = b = 0
            for vi, v in enumerate(variants):
                if pop.haps[g[0]].has(v):
                    a = vi + 1
                if pop.haps[g[1]].has(v):
                    b = vi + 1
            if len(variants) > 1:
                gl = [-1, -1, -1]
            else:
                s = pop.samples[i]
                lo, hi = pop.window[i]
                n_reads_this_individual = hi - lo
                rows = []
                for h in (ref_hap, var_hap):
                    row = []
                    for r in range(lo, hi):
                        sc, _ = po.align_no_traceback(s.seqs[r], s.quals[r], s.pos[r], h.seq, h.start, h.varSeqLen)
                        row.append(0.0 if sc < 0 else po.likelihood(sc, s.mapq[r]))
                    rows.append(np.array(row, np.float64))
                gl = [py2_round(po.genotype_loglik(rows[x], rows[y]), 2) for x, y in ((0, 0), (0, 1), (1, 1))]
            if n_reads_this_individual is None:
                # variantcaller.pyx:436 reads a local that only the single-variant branch (:422) assigns
                raise UnboundLocalError("local variable 'nReadsThisIndividual' referenced before assignment")
            sample_cols.append(dict(GT=[[str(a), "/", str(b)]], GL=gl, GQ=[gq], NR=[n_reads_this_individual]))
        cols = [chrom.decode(), str(POS + 1), ".", REF.decode(), ",".join(x.decode() for x in alt) if alt else ".",
                str(qual), ";".join(linefilter) if linefilter else "PASS",
                format_data(lineinfo, separator=";", missing=-1), "GT:GL:GQ:NR"]
        for d in sample_cols:
            cols.append(format_data(d, key=False, missing="."))
        lines.append("\t".join(cols))


def call_region(chrom, start, end, variants, samples, ref, opts, skipped=None):
    """variantcaller.pyx:116-191 over the sorted candidates of one buffer; returns the VCF body lines.
    `skipped` collects (window start, window end, exception) of the windows the reference would log and skip."""
    lines = []
    for w in windows_and_variants(chrom, start, end, variants, opts):
        ws, we = w["startPos"], w["endPos"]
        vs = w["variants"]
        ref_hap = Hap(ref, chrom, ws, we, (), opts.rlen)
        try:
            n_reads = sum(s.set_window(ws, we) for s in samples)
            if len(vs) > opts.maxVariants:
                if len(vs) > 50:
                    vs = sorted(sorted(vs, key=lambda v: v.nSupportingReads, reverse=True)[:50], key=lambda v: v.order())
                pre = Population([ref_hap] + [Hap(ref, chrom, ws, we, (v,), opts.rlen) for v in vs], samples, opts)
                # varsByPost.sort(reverse=True) on (posterior, variant): variants that compare equal keep their order
                by_post = sorted(((pre.posterior(v), v.order(), i) for i, v in enumerate(vs)), key=lambda t: t[:2], reverse=True)
                vs = sorted((vs[i] for _, _, i in by_post[:opts.maxVariants]), key=lambda v: v.order())
                w["variants"] = vs
            var_haps = haplotypes_in_window(w, n_reads, ref, opts)
            unique, seen = [], set()
            for h in [ref_hap] + var_haps:                            # list(set(...)), variantcaller.pyx:170: order of creation here
                if (h.start, h.seq) not in seen:
                    seen.add((h.start, h.seq))
                    unique.append(h)
            if len(unique) <= 1:
                continue
            pop = Population(unique, samples, opts)                   # Caller.setup + Caller.call
            output_call(pop, ref, opts, lines)
        except Exception as e:                                        # variantcaller.pyx:189-191
            if skipped is not None:
                skipped.append((ws, we, e))
            continue
    return lines
