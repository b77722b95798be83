"""Reference sequence access with the interface the haplotype code uses
(platypus/fastafile.pyx:57-174): ``refs[name].SeqLength``, ``getSequence`` (half-open,
0-based, upper case, end clamped to the sequence length minus one as the reference
does at fastafile.pyx:155-156) and ``getCharacter`` ('-' outside the sequence,
fastafile.pyx:102-103).  Backed by an indexed FASTA file (.fai) or by sequences in
memory (tests, simulations)."""
import collections

SequenceTuple = collections.namedtuple("SequenceTuple", "SeqName SeqLength StartPos LineLength FullLineLength")


class FastaFile(object):
    def __init__(self, fileName=None, indexName=None, parseNCBI=True, sequences=None):
        self.refs = {}
        self._mem = None
        self._file = None
        self._cache = None                                        # (name, begin, end, bytes)
        if sequences is not None:
            self._mem = {}
            for name, seq in sequences.items():
                name = name if isinstance(name, bytes) else name.encode()
                seq = (seq if isinstance(seq, bytes) else seq.encode()).upper()
                self._mem[name] = seq
                self.refs[name] = SequenceTuple(name, len(seq), 0, len(seq), len(seq) + 1)
            return
        self._file = open(fileName, "rb")
        with open(indexName if indexName is not None else fileName + ".fai", "rb") as index:
            for line in index:
                f = line.strip().split(b"\t")
                if len(f) < 5:
                    continue
                name = f[0]
                if name.startswith(b"gi|") and parseNCBI:          # fastafile.pyx:48-51
                    ids = name.split(b"|")
                    if len(ids) >= 4 and ids[2] == b"ref":
                        name = ids[3]
                self.refs[name] = SequenceTuple(f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]))

    def close(self):
        if self._file is not None:
            self._file.close()

    def getTotalSequenceLength(self):
        return sum(t.SeqLength for t in self.refs.values())

    def _offset(self, t, pos):
        return t.StartPos + pos + (t.FullLineLength - t.LineLength) * (pos // t.LineLength)

    def getCharacter(self, seqName, pos):
        t = self.refs[seqName]
        if pos >= t.SeqLength or pos < 0:
            return b"-"
        if self._mem is not None:
            return self._mem[seqName][pos:pos + 1]
        c = self._cache
        if c is not None and c[0] == seqName and c[1] <= pos < c[2]:
            return c[3][pos - c[1]:pos - c[1] + 1]
        self._file.seek(self._offset(t, pos))
        return self._file.read(1).upper()

    def setCacheSequence(self, seqName, beginPos, endPos):
        """fastafile.pyx:112-139: keep [beginPos, endPos) of one sequence in memory; getSequence
        answers from it while the request lies inside (the region caller asks for the region
        +- 25 kb, variantcaller.pyx:214)"""
        t = self.refs[seqName]
        beginPos = max(0, beginPos)
        endPos = min(t.SeqLength - 1, endPos)
        if endPos < beginPos:
            raise IndexError("Cannot have beginPos = %s, endPos = %s" % (beginPos, endPos))
        self._cache = None
        self._cache = (seqName, beginPos, endPos, self.getSequence(seqName, beginPos, endPos))

    def getSequence(self, seqName, beginPos, endPos):
        c = self._cache
        if c is not None and c[0] == seqName and beginPos >= c[1] and endPos < c[2] and endPos >= beginPos:
            return c[3][beginPos - c[1]:endPos - c[1]]               # fastafile.pyx:148-150
        t = self.refs[seqName]
        beginPos = max(0, beginPos)
        endPos = min(t.SeqLength - 1, endPos)
        if endPos < beginPos:
            raise IndexError("Cannot have beginPos = %s, endPos = %s" % (beginPos, endPos))
        if self._mem is not None:
            return self._mem[seqName][beginPos:endPos]
        start, end = self._offset(t, beginPos), self._offset(t, endPos)
        self._file.seek(start)
        return self._file.read(end - start).replace(b"\n", b"").upper()
