"""TEST INFRASTRUCTURE ONLY -- never imported by the product.

CPU restatement, in plain Python, of what the reference does with the records of a BAM region before the
window loop: which reads a buffer keeps and how it trims them, which variant candidates their CIGAR
strings and mismatches give, and what is left of them after standard form, merging and the support cuts:

    platypus/samtoolsWrapper.pyx:699-789   createRead: the fields of a record the caller reads
    platypus/cwindow.pyx:74-148            checkAndTrimRead
    platypus/cwindow.pyx:182-223           the buffer's loop over the fetched records
    platypus/variant.pyx:28-76,107-119     Variant: fields, addVariant (integer mean of the squared read positions)
    platypus/variant.pyx:290-474           candidates of one read (CIGAR walk, mismatches of a match segment)
    platypus/variant.pyx:600-615           all reads; the sorted candidate list
    platypus/chaplotype.pyx:532-665        convertVariantToStandardFormat
    platypus/variantFilter.pyx:12-83       filterVariants
    platypus/variantcaller.pyx:92-108      the generator over all buffers, the set, the sorted list

Parity unpinned: the reference program cannot run here, so this file follows its source text.  One place
depends on the iteration order of a Python 2 dict (`sorted(self.variantHeap.values())`, variant.pyx:615:
candidates of one position and one length change come out in hash order); here they keep the order in
which the reads produced them.
"""
import math

from oracle.vcf_oracle import Hap, V, mLTOT


class CV(V):
    """V with the integer sum of squared read positions (variant.pxd: `int sumSqReadPos`)"""
    __slots__ = ("sumSq",)


# ---- records -> reads ---------------------------------------------------------------------------------
def create_reads(arrays):
    """samtoolsWrapper.pyx:699-789 for every record of packed arrays (what BamReader.fetch returns; CIGAR
    words as in the file)"""
    out = []
    off, coff = arrays["off"], arrays["cigarOff"]
    for i in range(len(arrays["pos"])):
        seq = bytes(arrays["seq"][off[i]:off[i + 1]])
        qual = [int(q) if q < 128 else int(q) - 256 for q in arrays["qual"][off[i]:off[i + 1]]]   # char: signed
        flag = int(arrays["flag"][i])
        ops = [(int(w) & 0xf, int(w) >> 4) for w in arrays["cigar"][coff[i]:coff[i + 1]]]
        end = int(arrays["pos"][i]) + sum(n for op, n in ops if op in (0, 2, 3))
        out.append(dict(seq=seq, qual=qual, pos=int(arrays["pos"][i]), rlen=len(seq), mapq=int(arrays["mapq"][i]), cigar=ops,
                        end=end, nQualsBelow20=sum(1 for q in qual if q < 20), isReverse=bool(flag & 0x10),
                        isDuplicate=bool(flag & 0x400), isUnMapped=bool(flag & 0x4), mateIsReverse=bool(flag & 0x20),
                        isPaired=bool(flag & 0x1), insertSize=int(arrays["insertSize"][i]), matePos=int(arrays["matePos"][i])))
    return out


def check_and_trim(r, last, maxBadQualBases, minGoodQualBases):
    """cwindow.pyx:74-148; qualities are changed in place"""
    if r["rlen"] - r["nQualsBelow20"] < minGoodQualBases:
        return False
    if r["nQualsBelow20"] > maxBadQualBases:
        return False
    if r["isUnMapped"] or r["isDuplicate"]:
        return False
    if r["insertSize"] > 0 and not r["isReverse"] and r["mateIsReverse"] and r["insertSize"] < r["rlen"]:
        return False
    if r["insertSize"] < 0 and r["isReverse"] and not r["mateIsReverse"] and -1 * r["insertSize"] < r["rlen"]:
        return False
    if last is not None and r["pos"] == last["pos"] and r["rlen"] == last["rlen"]:
        if r["isPaired"]:
            if last["matePos"] == r["matePos"]:
                return False
        else:
            return False
    q, n = r["qual"], r["rlen"]
    for index in range(1, n + 1):
        if q[n - index] < 5:
            q[n - index] = 0
        else:
            break
    for index in range(n):
        if q[index] < 5:
            q[index] = 0
        else:
            break
    if r["insertSize"] > 0 and not r["isReverse"] and r["mateIsReverse"] and r["insertSize"] < 2 * n:
        for index in range(1, (2 * n - r["insertSize"]) + 1):
            q[n - index] = 0
    return True


def buffer_reads(reads, minMapQual, maxReads, minGoodQualBases, maxBadQualBases):
    """cwindow.pyx:182-223: the reads a buffer keeps, in file order, trimmed"""
    kept = []
    for r in reads:
        if r["mapq"] < minMapQual:
            continue
        if not check_and_trim(r, kept[-1] if kept else None, maxBadQualBases, minGoodQualBases):
            continue
        kept.append(r)
        if len(kept) > maxReads:
            break
    return kept


# ---- reads -> candidates ------------------------------------------------------------------------------
def new_variant(refName, refPos, removed, added, nSupportingReads, nFwdReads, mapQ, baseQ, readPos):
    """variant.pyx:28-76"""
    refPos = max(0, refPos)
    nAdded, nRemoved = len(added), len(removed)
    if nRemoved == 1 and nAdded == 1:
        maxPos = refPos
    elif nRemoved == nAdded:
        maxPos = refPos + (nAdded - 1)
    else:
        maxPos = refPos + nRemoved + nAdded
    v = CV(refName=refName, refPos=refPos, removed=removed, added=added, nAdded=nAdded, nRemoved=nRemoved, minPos=refPos,
          maxPos=maxPos, nSupportingReads=nSupportingReads, nFwdReads=nFwdReads, nRevReads=nSupportingReads - nFwdReads,
          nFwdReadsWeighted=0.0, nRevReadsWeighted=0.0, rmsReadPos=0.0)
    v.sumSq = readPos * readPos
    if nSupportingReads > 0:
        v.rmsReadPos = math.sqrt(v.sumSq // nSupportingReads)        # both are C ints: an integer quotient
    w = 1.0 - math.exp(mLTOT * baseQ)
    if nFwdReads == 1:
        v.nFwdReadsWeighted = w
    else:
        v.nRevReadsWeighted = w
    return v


def add_variant(this, other):
    """variant.pyx:107-119"""
    this.nSupportingReads += other.nSupportingReads
    this.nFwdReads += other.nFwdReads
    this.nRevReads += other.nRevReads
    this.nFwdReadsWeighted += other.nFwdReadsWeighted
    this.nRevReadsWeighted += other.nRevReadsWeighted
    this.sumSq += other.sumSq
    if this.nSupportingReads > 0:
        this.rmsReadPos = math.sqrt(this.sumSq // this.nSupportingReads)


class Generator(object):
    """variant.pyx:253-474,600-615"""

    def __init__(self, region, ref, minMapQual, minFlank, minBaseQual, genSNPs=1, genIndels=1):
        self.rname, self.ref = region[0], ref
        self.minFlank, self.minBaseQual, self.genSNPs, self.genIndels = minFlank, minBaseQual, genSNPs, genIndels
        self.refSeqStart = max(1, region[1] - 2000)
        self.refSeqEnd = min(region[2] + 2000, ref.length(region[0]))
        self.refSeq = ref.seq(self.rname, self.refSeqStart, self.refSeqEnd)
        self.heap = {}                                               # insertion-ordered (see the header)

    def add(self, var):
        known = self.heap.get(var.same())
        if known is None:
            self.heap[var.same()] = var
        else:
            add_variant(known, var)

    def segment(self, r, readOffset, refOffset, n):
        """variant.pyx:309-365"""
        for index in range(n):
            readIndex = index + readOffset
            refIndex = (index + refOffset + r["pos"]) - self.refSeqStart
            if refIndex < 0 or refIndex >= len(self.refSeq) or readIndex >= r["rlen"]:
                continue                 # the reference reads outside its arrays here (undefined): nothing is compared
            readChar, refChar = r["seq"][readIndex:readIndex + 1], self.refSeq[refIndex:refIndex + 1]
            baseQual = r["qual"][readIndex]
            if readChar != refChar and readChar != b"N" and refChar != b"N" and baseQual >= self.minBaseQual:
                self.add(new_variant(self.rname, refIndex + self.refSeqStart, refChar, readChar, 1, 0 if r["isReverse"] else 1,
                                     r["mapq"], baseQual, readIndex + 1))

    def read(self, r):
        """variant.pyx:367-474"""
        readStart, ops = r["pos"], r["cigar"]
        nFwd = 0 if r["isReverse"] else 1
        refOffset = readOffset = 0
        n = len(ops)

        def flanked(k):
            if k > 0 and ops[k - 1][0] == 0 and ops[k - 1][1] >= self.minFlank:
                return True
            return k < n - 1 and ops[k + 1][0] == 0 and ops[k + 1][1] >= self.minFlank

        for k, (flag, length) in enumerate(ops):
            if flag == 1 and self.genIndels:
                if not flanked(k) or readOffset + length > r["rlen"]:       # (beyond the read: undefined in the reference)
                    readOffset += length
                    continue
                inserted = r["seq"][readOffset:readOffset + length]
                if inserted.count(b"N") == 0 and self.genIndels:
                    self.add(new_variant(self.rname, readStart + refOffset - 1, b"", inserted, 1, nFwd, r["mapq"],
                                         r["qual"][readOffset], readOffset + 1))
                readOffset += length
            elif flag == 2:
                if not flanked(k) or readOffset >= r["rlen"]:
                    refOffset += length
                    continue
                deleted = self.ref.seq(self.rname, readStart + refOffset, readStart + refOffset + length)
                if deleted.count(b"N") == 0 and self.genIndels:
                    self.add(new_variant(self.rname, readStart + refOffset - 1, deleted, b"", 1, nFwd, r["mapq"],
                                         r["qual"][readOffset], readOffset + 1))
                refOffset += length
            elif flag == 0:
                if length < self.minFlank:
                    readOffset += length
                    refOffset += length
                    continue
                if self.genSNPs:
                    self.segment(r, readOffset, refOffset, length)
                readOffset += length
                refOffset += length
            elif flag == 3:                                          # both advance, as the reference has it
                readOffset += length
                refOffset += length
            elif flag == 4:
                readOffset += length
            # hard clip, padding, anything else (and an insertion when genIndels is off): nothing moves

    def candidates(self):
        return sorted(self.heap.values(), key=lambda v: v.order())   # stable: ties keep the order of first sight


# ---- candidates -> filtered list ----------------------------------------------------------------------
def standard_format(v, ref):
    """chaplotype.pyx:532-665"""
    nAdded, nRemoved = v.nAdded, v.nRemoved
    if nAdded == nRemoved or v.refPos < 100:
        return v
    window = max(nAdded, nRemoved) + 10000
    windowMin = max(1, v.refPos - window)
    windowMax = min(v.refPos + window, ref.length(v.refName))
    hap = Hap(ref, v.refName, windowMin, windowMax, (v,), 0)
    refSeq, hapSeq = ref.seq(v.refName, windowMin, windowMax), hap.seq
    lenRef, lenHap = len(refSeq), len(hapSeq)
    for index in range(min(lenRef, lenHap)):
        hapIndex, refIndex = (lenHap - index) - 1, (lenRef - index) - 1
        if hapSeq[hapIndex] != refSeq[refIndex]:
            newPos, newAdded, newRemoved = -1, b"", b""
            if nAdded > 0:
                newPos = windowMin + (lenRef - 1) - index
                insStart = newPos - windowMin + 1
                newAdded = hapSeq[insStart:insStart + nAdded]
            if nRemoved > 0:
                newPos = windowMin + (lenRef - 1) - index - nRemoved
                delStart = newPos - windowMin + 1
                newRemoved = refSeq[delStart:delStart + nRemoved]
            effectSize = len(newRemoved) + len(newAdded) + abs(newPos - v.refPos)
            new = new_variant(v.refName, newPos, newRemoved, newAdded, v.nSupportingReads, v.nFwdReads, 0, 0, 0)
            new.maxPos, new.minPos = newPos + effectSize, newPos
            new.sumSq, new.rmsReadPos = v.sumSq, v.rmsReadPos
            new.nFwdReadsWeighted, new.nRevReadsWeighted = v.nFwdReadsWeighted, v.nRevReadsWeighted
            if len(newAdded) != nAdded or len(newRemoved) != nRemoved:
                raise ValueError("Error in variant conversion to standard format")
            return new
    raise ValueError("Error in variant conversion to standard format")


def filter_variants(varList, ref, minSupport, strandFilter, maxDiff):
    """variantFilter.pyx:12-83"""
    def passes(v):
        if v.nSupportingReads < minSupport:
            return False
        if abs(len(v.added) - len(v.removed)) > maxDiff:
            return False
        return not (strandFilter and (v.nFwdReads == 0 or v.nRevReads == 0))

    kept, last = [], None
    for v in varList:
        new = standard_format(v, ref)
        if last is None:
            last = new
        elif new.same() == last.same():
            add_variant(last, new)
        else:
            if passes(last):
                kept.append(last)
            last = new
    if last is not None and passes(last):
        kept.append(last)
    return sorted(kept, key=lambda v: v.order())


def candidates_in_region(chrom, start, end, buffers, ref, opts):
    """variantcaller.pyx:92-108: `buffers` holds every sample's kept reads (buffer_reads)"""
    gen = Generator((chrom, start, end), ref, opts.minMapQual, opts.minFlank, opts.minBaseQual, opts.genSNPs, opts.genIndels)
    for reads in buffers:
        for r in reads:
            gen.read(r)
    kept = filter_variants(gen.candidates(), ref, opts.minReads, opts.strandFilter, opts.maxSize)
    unique = {}
    for v in kept:                                                   # variantSet.update(...), sorted(variantSet)
        unique.setdefault(v.same(), v)
    return sorted(unique.values(), key=lambda v: v.order())
