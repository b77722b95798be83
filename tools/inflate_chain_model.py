#!/usr/bin/env python
"""What bounds a BGZF block's time in k_bgzf_inflate_wave: a host model of the kernel's passes over the token
stream of real blocks (a BAM file written by this package): tokens per pass, and why chains end.

    python tools/inflate_chain_model.py [file.bam] [blocks]
"""
import collections
import os
import struct
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_X = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
            8193, 12289, 16385, 24577]
DST_X = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class Bits(object):
    def __init__(self, data):
        self.v = int.from_bytes(data, "little")
        self.pos = 0

    def take(self, n):
        x = (self.v >> self.pos) & ((1 << n) - 1)
        self.pos += n
        return x


def table(lengths):
    """canonical Huffman: {(length, code read bit by bit, first bit highest) : symbol}"""
    count = collections.Counter(l for l in lengths if l)
    code, nxt = 0, {}
    for l in range(1, 16):
        code = (code + count.get(l - 1, 0)) << 1
        nxt[l] = code
    out = {}
    for s, l in enumerate(lengths):
        if l:
            out[(l, nxt[l])] = s
            nxt[l] += 1
    return out


def decode(B, tab):
    code = 0
    for l in range(1, 16):
        code = (code << 1) | B.take(1)
        if (l, code) in tab:
            return tab[(l, code)], l
    raise ValueError("bad code")


def tokens(data):
    """[(kind, bits, size, dist, code bits of the literal/length code, code bits of the distance code)] of a raw deflate
    stream; kind 1 literal, 2 match, 0 end of block / header (bits = its length)"""
    B = Bits(data)
    out = []
    while True:
        start = B.pos
        last = B.take(1)
        typ = B.take(2)
        if typ == 0:
            B.pos = (B.pos + 7) & ~7
            n = B.take(16); B.take(16)
            B.pos += 8 * n
            out.append((0, B.pos - start, n, 0, 0, 0))
        else:
            if typ == 1:
                lit = table([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
                dst = table([5] * 32)
            else:
                hlit, hdist, hclen = B.take(5) + 257, B.take(5) + 1, B.take(4) + 4
                order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
                cl = [0] * 19
                for i in range(hclen):
                    cl[order[i]] = B.take(3)
                ct = table(cl)
                lens = []
                while len(lens) < hlit + hdist:
                    s, _ = decode(B, ct)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + B.take(2))
                    elif s == 17:
                        lens += [0] * (3 + B.take(3))
                    else:
                        lens += [0] * (11 + B.take(7))
                lit, dst = table(lens[:hlit]), table(lens[hlit:hlit + hdist])
            out.append((0, B.pos - start, 0, 0, 0, 0))
            while True:
                start = B.pos
                s, l1 = decode(B, lit)
                if s < 256:
                    out.append((1, B.pos - start, 1, 0, l1, 0))
                elif s == 256:
                    out.append((0, B.pos - start, 0, 0, l1, 0))
                    break
                else:
                    s -= 257
                    size = LEN_BASE[s] + B.take(LEN_X[s])
                    d, l2 = decode(B, dst)
                    dist = DST_BASE[d] + B.take(DST_X[d])
                    out.append((2, B.pos - start, size, dist, l1, l2))
        if last:
            return out


def passes(toks, window=64, max_size=32):
    """the kernel's passes: (tokens taken by the chain, reason the chain ended) per pass; a token the chain cannot
    start with is a pass of its own"""
    res = []
    i, n = 0, len(toks)
    while i < n:
        j_bits, P, took, why = 0, 0, 0, None
        while i < n:
            kind, bits, size, dist, l1, l2 = toks[i]
            if j_bits >= window:
                why = "window"; break
            if kind == 0:
                why = "end/header"; break
            if l1 > 10 or l2 > 9:
                why = "long code"; break
            if kind == 2:
                if dist < size + P:
                    why = "overlaps the pass" if dist >= size else "overlaps itself"; break
                if size > max_size:
                    why = "long match"; break
                if dist > 32768 - 2304:
                    why = "far"; break
            took += 1; P += size; j_bits += bits; i += 1
        if took == 0:
            i += 1                                  # the whole wavefront takes this one
            res.append((1, "alone: " + (why or "?")))
        else:
            res.append((took, why or "end of data"))
    return res


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else None
    n_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    if path is None:
        import tempfile
        import bench_e2e
        work = tempfile.mkdtemp(dir="/dev/shm")
        bams, *_ = bench_e2e.make_data("wgs", work, mb=1.0)
        path = bams[0]
    raw = open(path, "rb").read()
    at, k = 0, 0
    tot_t = tot_p = 0
    why = collections.Counter()
    sizes = collections.Counter()
    while at < len(raw) and k < n_blocks + 2:
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        body = raw[at + 18:at + bsize - 8]
        at += bsize
        k += 1
        if k <= 2 or len(body) < 1000:
            continue                                # header blocks
        t = tokens(body)
        p = passes(t)
        tot_t += len(t); tot_p += len(p)
        for took, w in p:
            why[w] += 1
            sizes[min(took, 12)] += 1
        if k <= 5:
            print("block of %d bytes in, %d out: %d tokens (%d literals, %d matches of %.1f bytes on average), %d passes" % (
                len(body), len(zlib.decompress(body, -15)), len(t), sum(x[0] == 1 for x in t), sum(x[0] == 2 for x in t),
                sum(x[2] for x in t if x[0] == 2) / max(1, sum(x[0] == 2 for x in t)), len(p)))
    print("%d tokens in %d passes: %.2f tokens per pass" % (tot_t, tot_p, tot_t / max(1, tot_p)))
    print("passes by tokens taken:", dict(sorted(sizes.items())))
    print("why chains ended:", why.most_common())


if __name__ == "__main__":
    main()
