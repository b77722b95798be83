"""Synthetic windows of haplotypes and reads for tests and benchmarks.

A *window* is what platypus/variantcaller.pyx:116-187 hands to the caller: H
candidate haplotypes over the same reference stretch (the reference haplotype
plus SNP / indel variants of it) and the R reads that overlap it.  The work of
the window is the dense R x H pair matrix (platypus/cpopulation.pyx:698-720).

The packed batch produced here is the layout every backend takes:
concatenated byte strings plus int64 offsets for haplotypes and reads, int32
metadata, and an int32 (hap, read) pair list ordered haplotype-major inside a
window (the order Haplotype.alignReads walks, platypus/chaplotype.pyx:259-268).
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
SEED = 20261111


def _base_sequences(rng, n, length, homopolymers, n_frac):
    seq = ACGT[rng.integers(0, 4, size=(n, length))]
    if homopolymers:
        # 1-3 runs of length 5-40 per sequence
        for _ in range(3):
            use = rng.random(n) < (1.0 if _ == 0 else 0.5)
            start = rng.integers(0, length - 41, size=n)
            runlen = rng.integers(5, 41, size=n)
            base = ACGT[rng.integers(0, 4, size=n)]
            cols = np.arange(length)[None, :]
            mask = use[:, None] & (cols >= start[:, None]) & (cols < (start + runlen)[:, None])
            seq = np.where(mask, base[:, None], seq)
    if n_frac > 0:
        seq = np.where(rng.random(seq.shape) < n_frac, np.uint8(ord("N")), seq)
    return np.ascontiguousarray(seq)


def make_batch(n_windows, haps_per_window=4, reads_per_window=32, read_len=150,
               hap_len=750, indel_hap_frac=0.25, homopolymers=False, n_frac=0.0,
               sub_rate=0.01, read_indel_frac=0.10, seed=SEED, jitter_frac=0.2,
               ragged=False):
    """Build a packed batch of ``n_windows`` windows.

    indel_hap_frac   share of the non-reference haplotypes that carry an indel
                     (their varSeqLen != 0, so their pairs take the 7-mer
                     anchoring path of platypus/calign.pyx:66-77)
    homopolymers     insert long homopolymer runs (the amplicon-panel stress case)
    ragged           draw read lengths from [read_len/2, read_len] instead of a
                     single length
    """
    rng = np.random.default_rng(seed)
    W, H, R, L = n_windows, haps_per_window, reads_per_window, read_len
    base = _base_sequences(rng, W, hap_len, homopolymers, n_frac)

    # ---- haplotypes -------------------------------------------------------
    hap_chunks, hap_lens = [], np.zeros(W * H, dtype=np.int64)
    hap_varlen = np.zeros(W * H, dtype=np.int32)
    hap_delta = np.zeros(W * H, dtype=np.int32)       # length change of the variant
    hap_vpos = np.zeros(W * H, dtype=np.int32)        # variant position in base coords
    is_indel = rng.random(W * H) < indel_hap_frac
    vpos_all = rng.integers(hap_len // 3, 2 * hap_len // 3, size=W * H)
    vlen_all = rng.integers(1, 7, size=W * H)
    vins_all = rng.random(W * H) < 0.5
    # the same haplotypes as (window's reference stretch, variant list): what pg_haps_build takes
    var_hap, var_pos, var_nrem, var_added = [], [], [], []
    for w in range(W):
        b = base[w]
        for h in range(H):
            i = w * H + h
            if h == 0:
                s = b
            elif not is_indel[i]:
                s = b.copy()
                p = vpos_all[i]
                s[p] = ACGT[(np.searchsorted(ACGT, s[p] if s[p] != 78 else 65) + 1) % 4]
                hap_vpos[i] = p
                var_hap.append(i); var_pos.append(int(p)); var_nrem.append(1); var_added.append(s[p:p + 1])
            else:
                p, n = int(vpos_all[i]), int(vlen_all[i])
                if homopolymers and rng.random() < 0.5:
                    # put the indel inside a run if there is one: extend/shrink it
                    eq = np.flatnonzero((b[1:] == b[:-1]))
                    if len(eq):
                        p = int(eq[rng.integers(0, len(eq))])
                        p = min(max(p, 20), hap_len - 40)
                if vins_all[i]:
                    ins = np.repeat(b[p], n) if homopolymers else ACGT[rng.integers(0, 4, size=n)]
                    s = np.concatenate([b[:p + 1], ins, b[p + 1:]])
                    hap_delta[i] = n
                    var_hap.append(i); var_pos.append(p); var_nrem.append(0); var_added.append(ins)
                else:
                    s = np.concatenate([b[:p + 1], b[p + 1 + n:]])
                    hap_delta[i] = -n
                    var_hap.append(i); var_pos.append(p); var_nrem.append(n); var_added.append(ACGT[:0])
                hap_varlen[i] = n
                hap_vpos[i] = p
            hap_chunks.append(s)
            hap_lens[i] = len(s)
    hap_off = np.zeros(W * H + 1, dtype=np.int64)
    np.cumsum(hap_lens, out=hap_off[1:])
    hap_seq = np.concatenate(hap_chunks)
    win_start = (1000 + np.arange(W, dtype=np.int64) * 2000).astype(np.int32)
    hap_start = np.repeat(win_start, H)

    # ---- reads ------------------------------------------------------------
    nR = W * R
    if ragged:
        rlen = rng.integers(max(8, L // 2), L + 1, size=nR)
    else:
        rlen = np.full(nR, L, dtype=np.int64)
    read_off = np.zeros(nR + 1, dtype=np.int64)
    np.cumsum(rlen, out=read_off[1:])
    win_of_read = np.repeat(np.arange(W), R)
    src_h = rng.integers(0, H, size=nR)
    src = win_of_read * H + src_h
    src_len = hap_lens[src]
    # start inside the source haplotype so that every haplotype of the window
    # still has rlen+15 bases after the slice start (calign.pyx:61)
    lo = 16
    hi = np.minimum(src_len, hap_len - 8) - rlen - 40
    o = lo + (rng.random(nR) * np.maximum(hi - lo, 1)).astype(np.int64)
    # mapped position in base coordinates
    d = hap_delta[src].astype(np.int64)
    vp = hap_vpos[src].astype(np.int64)
    o_ref = np.where((d != 0) & (o > vp), np.maximum(o - d, vp), o)
    jitter = np.where(rng.random(nR) < jitter_frac, rng.integers(-6, 7, size=nR), 0)
    read_pos = (win_start[win_of_read] + o_ref + jitter).astype(np.int32)
    read_mapq = rng.integers(20, 61, size=nR).astype(np.int32)

    total = int(read_off[-1])
    read_seq = np.empty(total, dtype=np.uint8)
    read_qual = np.empty(total, dtype=np.uint8)
    maxL = int(rlen.max())
    CH = 65536
    for c0 in range(0, nR, CH):
        c1 = min(nR, c0 + CH)
        n = c1 - c0
        j = np.arange(maxL)[None, :]
        rl = rlen[c0:c1, None]
        valid = j < rl
        idx = o[c0:c1, None] + j
        # one indel per ~10% of reads
        has = rng.random(n) < read_indel_frac
        isins = rng.random(n) < 0.5
        cut = (rng.random(n) * np.maximum(rlen[c0:c1] - 20, 1)).astype(np.int64) + 10
        ilen = rng.integers(1, 7, size=n)
        dele = (has & ~isins)[:, None] & (j >= cut[:, None])
        idx = idx + np.where(dele, ilen[:, None], 0)
        insm = (has & isins)[:, None]
        idx = idx - np.where(insm & (j >= (cut + ilen)[:, None]), ilen[:, None], 0)
        idx = np.minimum(idx, src_len[c0:c1, None] - 1)
        flat = hap_seq[hap_off[src[c0:c1], None] + idx]
        inserted = insm & (j >= cut[:, None]) & (j < (cut + ilen)[:, None])
        rnd = ACGT[rng.integers(0, 4, size=flat.shape)]
        flat = np.where(inserted, rnd, flat)
        sub = rng.random(flat.shape) < sub_rate
        flat = np.where(sub, ACGT[rng.integers(0, 4, size=flat.shape)], flat)
        # qualities: 90% Q30-41, 10% Q2-20, trimmed (zeroed) head/tail
        q = np.where(rng.random(flat.shape) < 0.9,
                     rng.integers(30, 42, size=flat.shape),
                     rng.integers(2, 21, size=flat.shape)).astype(np.uint8)
        head = np.where(rng.random(n) < 0.3, rng.integers(0, 6, size=n), 0)
        tail = np.where(rng.random(n) < 0.5, rng.integers(0, 16, size=n), 0)
        q = np.where((j < head[:, None]) | (j >= rl - tail[:, None]), np.uint8(0), q)
        read_seq[read_off[c0]:read_off[c1]] = flat[valid]
        read_qual[read_off[c0]:read_off[c1]] = q[valid]

    # ---- pairs: per window, haplotype-major ----------------------------------
    hh = np.arange(H, dtype=np.int32)
    rr = np.arange(R, dtype=np.int32)
    pair_hap = (np.arange(W, dtype=np.int32)[:, None, None] * H + hh[None, :, None]
                + 0 * rr[None, None, :]).reshape(-1).astype(np.int32)
    pair_read = (np.arange(W, dtype=np.int32)[:, None, None] * R + 0 * hh[None, :, None]
                 + rr[None, None, :]).reshape(-1).astype(np.int32)

    var_hap = np.asarray(var_hap, dtype=np.int64)
    hap_var_off = np.zeros(W * H + 1, dtype=np.int32)
    np.cumsum(np.bincount(var_hap, minlength=W * H), out=hap_var_off[1:])
    var_added_off = np.zeros(len(var_hap) + 1, dtype=np.int32)
    np.cumsum([len(a) for a in var_added], out=var_added_off[1:])
    built = dict(ref_seq=base.reshape(-1), ref_off=np.arange(W + 1, dtype=np.int64) * hap_len, ref_start=win_start.astype(np.int32),
                 hap_stretch=np.repeat(np.arange(W, dtype=np.int32), H), hap_var_off=hap_var_off,
                 var_pos=(np.asarray(var_pos, dtype=np.int64) + win_start[var_hap // H]).astype(np.int32),
                 var_n_removed=np.asarray(var_nrem, dtype=np.int32), var_added_off=var_added_off,
                 var_added=np.concatenate(var_added) if var_added else ACGT[:0])

    return dict(hap_seq=hap_seq, hap_off=hap_off, hap_start=hap_start.astype(np.int32),
                hap_varlen=hap_varlen, built=built,
                read_seq=read_seq, read_qual=read_qual, read_off=read_off,
                read_pos=read_pos, read_mapq=read_mapq,
                pair_hap=pair_hap, pair_read=pair_read,
                meta=dict(windows=W, haps_per_window=H, reads_per_window=R, read_len=L,
                          hap_len=hap_len, seed=seed))


def oracle_view(batch):
    """The subset of keys the CPU checker's batch driver takes."""
    keys = ("hap_seq", "hap_off", "hap_start", "hap_varlen", "read_seq", "read_qual",
            "read_off", "read_pos", "pair_hap", "pair_read")
    return {k: batch[k] for k in keys}
