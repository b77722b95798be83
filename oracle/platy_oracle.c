/*
 * platy_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A scalar CPU restatement of the read-to-haplotype alignment and likelihood
 * path of Platypus 0.1.5.  It exists to check the HIP kernels: only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load it.
 * Nothing in the product library links, calls or falls back to this file.
 *
 * Parity status:
 *   po_fast_align          pinned: checked bit-for-bit against the reference's
 *                          own C routine compiled from its sources
 *                          (oracle/Makefile target `ref`), and against golden
 *                          vectors generated from that build (tests/golden/).
 *   po_hash_sequence, po_map_read, po_align_no_traceback, po_genotype_loglik
 *                          "parity unpinned": the reference layer they restate
 *                          is Python-2 Cython that cannot be imported here and
 *                          the reference ships no test or fixture for it; they
 *                          follow the cited source text line by line.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "platy_oracle.h"

enum { BAND = 8 };               /* 8 even + 8 odd diagonals               */
#define POS_INF ((int16_t)0x2000) /* platypus/c/align.c:28                  */

static inline int16_t add16(int16_t a, int16_t b) { return (int16_t)(a + b); }
static inline int16_t min16(int16_t a, int16_t b) { return a < b ? a : b; }

/*
 * Per-position gap-open penalties of a haplotype slice.
 * Follows platypus/c/align.c:70-90 (usehomopolgapq branch): scan right to
 * left, count the run of equal characters to the right, an 'N' never extends a
 * run, the count stops growing where the table string ends.
 */
void po_gap_open(const char *seq1, int len1, const char *table, int16_t *out)
{
    int prev = -1, run = 0;
    for (int i = len1 - 1; i >= 0; --i) {
        if (seq1[i] == prev) {
            if (table[run + 1] != 0) run += 1;
        } else {
            run = 0;
        }
        out[i] = (int16_t)(4 * (table[run] - '!'));
        prev = (seq1[i] == 'N') ? 0 : seq1[i];
    }
}

/*
 * Banded affine-gap alignment score.  Follows platypus/c/align.c:18-235 for
 * usehomopolgapq = 1.  Lane k of the "even" vectors is diagonal 2k, lane k of
 * the "odd" vectors is diagonal 2k+1; all arithmetic is 16-bit wrapping with
 * signed minimum, as the 8 x int16 vectors of the reference are.
 */
int po_fast_align(const char *seq1, const char *seq2, const char *qual2,
                  int len1, int len2, int gapextend, int nucprior,
                  const char *table)
{
    const int16_t ge = (int16_t)(gapextend * 4);   /* align.c:29 */
    const int16_t np = (int16_t)(nucprior * 4);    /* align.c:30 */
    const int16_t nscore = 4;                      /* align.c:31 */

    int16_t m1[BAND], i1[BAND], d1[BAND], m2[BAND], i2[BAND], d2[BAND];
    int16_t s1[BAND], s1n[BAND], s2[BAND], q2[BAND], go[BAND];
    int16_t *inith = (int16_t *)malloc(sizeof(int16_t) * (size_t)(len1 > 8 ? len1 : 8));
    int16_t best = POS_INF;

    for (int k = 0; k < BAND; ++k) {
        m1[k] = i1[k] = d1[k] = m2[k] = i2[k] = d2[k] = POS_INF;   /* :53-58 */
        s1[k] = (int16_t)seq1[k];                                  /* :63 */
        s2[k] = POS_INF;                                           /* :64 */
        q2[k] = 64 * 4;                                            /* :65 */
        /* :68 -- (cmpeq + C) & C with C = pos_inf + n_score */
        s1n[k] = (int16_t)(((s1[k] == 'N' ? -1 : 0) + (POS_INF + nscore)) & (POS_INF + nscore));
    }
    po_gap_open(seq1, len1, table, inith);
    for (int k = 0; k < BAND; ++k) go[k] = inith[k];               /* :100 */

    const int iters = len2 + BAND;                                 /* :106 */
    for (int t = 0; t < iters; ++t) {
        /* read window moves up one lane, next read base enters lane 0 (:111-121) */
        for (int k = BAND - 1; k > 0; --k) { s2[k] = s2[k - 1]; q2[k] = q2[k - 1]; }
        if (t < len2) { s2[0] = (int16_t)seq2[t]; q2[0] = (int16_t)(4 * qual2[t]); }
        else          { s2[0] = '0';              q2[0] = 64 * 4; }

        /* free start on lane t while t < 8 (:50,:127-128,:189) */
        if (t < BAND) { m1[t] = 0; m2[t] = 0; }

        /* ---- even half-step (:129-173) ---- */
        for (int k = 0; k < BAND; ++k) m1[k] = min16(m1[k], min16(i1[k], d1[k]));
        if (t >= len2) {
            int16_t sc = m1[t - len2];
            if (sc < best) best = sc;
        }
        int16_t dnew[BAND];
        for (int k = 0; k < BAND; ++k) {
            int16_t mism = (s2[k] == s1[k]) ? 0 : q2[k];
            m1[k] = add16(m1[k], min16(mism, s1n[k]));
            dnew[k] = min16(add16(d2[k], ge), add16(min16(m2[k], i2[k]), go[k]));
            i1[k] = add16(min16(add16(i2[k], ge), add16(m2[k], go[k])), np);
        }
        for (int k = BAND - 1; k > 0; --k) d1[k] = dnew[k - 1];
        d1[0] = POS_INF;

        /* ---- haplotype window, N penalties and gap-open vector advance (:180-187) ---- */
        const char c = (8 + t < len1) ? seq1[8 + t] : 'N';
        for (int k = 0; k < BAND - 1; ++k) { s1[k] = s1[k + 1]; s1n[k] = s1n[k + 1]; }
        s1[BAND - 1] = (int16_t)c;
        s1n[BAND - 1] = (c == 'N') ? nscore : POS_INF;
        for (int k = BAND - 1; k > 0; --k) go[k] = go[k - 1];
        go[0] = inith[(8 + t < len1) ? 8 + t : len1 - 1];

        /* ---- odd half-step (:190-231) ---- */
        for (int k = 0; k < BAND; ++k) m2[k] = min16(m2[k], min16(i2[k], d2[k]));
        if (t >= len2) {
            int16_t sc = m2[t - len2];
            if (sc < best) best = sc;
        }
        int16_t inew[BAND];
        for (int k = 0; k < BAND; ++k) {
            int16_t mism = (s2[k] == s1[k]) ? 0 : q2[k];
            m2[k] = add16(m2[k], min16(mism, s1n[k]));
            d2[k] = min16(add16(d1[k], ge), add16(min16(m1[k], i1[k]), go[k]));
            inew[k] = add16(min16(add16(i1[k], ge), add16(m1[k], go[k])), np);
        }
        for (int k = 0; k < BAND - 1; ++k) i2[k] = inew[k + 1];
        i2[BAND - 1] = POS_INF;
    }
    free(inith);
    return best >> 2;                                              /* :234 */
}

/* 7-mer -> 14-bit code.  Follows platypus/calign.pyx:81-96. */
unsigned po_hash7(const char *s)
{
    unsigned h = 0;
    for (int i = 0; i < PO_HASH_NUCS; ++i) {
        int c = s[i] & 7;
        if (c == 7) c = 2;
        h = (h << 2) + (unsigned)(c & 3);
    }
    return h;
}

/*
 * Unique-7-mer position table of a haplotype: 0 = absent, i+1 = the single
 * start position, -1 = seen more than once.  Follows platypus/calign.pyx:99-123
 * (the loop bound seqlen - hash_nucs leaves out the last 7-mer).
 * `table` has PO_HASH_SIZE entries and is overwritten.
 */
void po_hash_sequence(const char *seq, int seqlen, int *table)
{
    memset(table, 0, sizeof(int) * PO_HASH_SIZE);
    if (seqlen < PO_HASH_NUCS) return;
    for (int i = 0; i < seqlen - PO_HASH_NUCS; ++i) {
        unsigned h = po_hash7(seq + i);
        table[h] = (table[h] != 0) ? -1 : i + 1;
    }
}

/*
 * Vote for the read's start position on the haplotype.  Follows
 * platypus/calign.pyx:126-158: the first position whose running count reaches
 * the final maximum wins; 0 when nothing votes.  May be negative.
 */
int po_map_read(const char *read, int readlen, int seqlen, const int *table)
{
    if (readlen < PO_HASH_NUCS) return 0;
    int *counts = (int *)calloc((size_t)(seqlen + readlen), sizeof(int));
    int maxcount = 0, maxpos = 0;
    for (int i = 0; i < readlen - PO_HASH_NUCS; ++i) {
        int pos = table[po_hash7(read + i)];
        if (pos > 0) {
            pos -= 1 + i;
            int c = ++counts[pos + readlen];
            if (c > maxcount) { maxcount = c; maxpos = pos; }
        }
    }
    free(counts);
    return maxpos;
}

/* score, mapping quality -> read likelihood.  platypus/calign.pyx:55-56,63,78 */
double po_likelihood(int score, int mapq)
{
    const double mLTOT = -0.23025850929940459;     /* calign.pyx:17 */
    double pwrong = exp(mLTOT * mapq);
    double pright = 1.0 - pwrong;
    double v = exp(mLTOT * score) * pright + pwrong;
    return v < 1.0 ? v : 1.0;
}

/*
 * Anchor choice + alignment of one read against one haplotype.  Follows
 * platypus/calign.pyx:36-78.  `hashtab` may be NULL (built here, as the
 * reference does per call) or a table made by po_hash_sequence for `hap`.
 * Returns the score; *anchor receives the slice start.  When the reference's
 * assertion at calign.pyx:61 would fail, returns PO_ASSERT_FAILED (the
 * reference's cdef function then yields likelihood 0.0).
 */
int po_align_no_traceback(const char *read, const char *qual, int rlen, int rpos,
                          const char *hap, int hap_len, int hap_start, int var_seq_len,
                          int gapextend, int nucprior, const char *table,
                          const int *hashtab, int *anchor)
{
    const int len1 = rlen + 15;                    /* calign.pyx:51 */
    const int hap_pos = rpos - hap_start;          /* calign.pyx:52 */
    if (var_seq_len == 0) {
        int a = hap_pos - 8 > 0 ? hap_pos - 8 : 0;
        *anchor = a;
        if (!(len1 + a <= hap_len)) return PO_ASSERT_FAILED;
        return po_fast_align(hap + a, read, qual, len1, rlen, gapextend, nucprior, table);
    }
    int *own = NULL;
    if (!hashtab) {
        own = (int *)malloc(sizeof(int) * PO_HASH_SIZE);
        po_hash_sequence(hap, hap_len, own);
        hashtab = own;
    }
    int mp = po_map_read(read, rlen, hap_len, hashtab);
    int a = mp - 8 > 0 ? mp - 8 : 0;               /* calign.pyx:67 */
    free(own);
    *anchor = a;
    int score;
    if (!(len1 + a <= hap_len)) {                  /* calign.pyx:73-75 */
        char *ext = (char *)malloc((size_t)hap_len + (size_t)rlen + 16);
        memcpy(ext, hap, (size_t)hap_len);
        memset(ext + hap_len, 'N', (size_t)rlen);
        memset(ext + hap_len + rlen, 0, 16);
        score = po_fast_align(ext + a, read, qual, len1, rlen, gapextend, nucprior, table);
        free(ext);
    } else {
        score = po_fast_align(hap + a, read, qual, len1, rlen, gapextend, nucprior, table);
    }
    return score;
}

/*
 * log-likelihood of a diploid genotype: sum over reads, in read order, of
 * log(0.5*(l1+l2)).  Follows platypus/cgenotype.pyx:58-61.
 */
double po_genotype_loglik(const double *l1, const double *l2, int nreads)
{
    double s = 0.0;
    for (int r = 0; r < nreads; ++r) s += log(0.5 * (l1[r] + l2[r]));
    return s;
}

/*
 * Batch driver used by tests and by the CPU-baseline timing: pairs
 * [lo, hi) of a packed batch, optionally through an externally supplied
 * alignment routine with the reference's signature (the reference's own
 * compiled C, loaded by the caller), so the same driver times either.
 */
void po_run_pairs(const po_batch *b, long lo, long hi, po_align_fn ext_align,
                  int *out_score, int *out_anchor)
{
    int *tab = (int *)malloc(sizeof(int) * PO_HASH_SIZE);
    int cached_hap = -1;
    for (long p = lo; p < hi; ++p) {
        const int h = b->pair_hap[p], r = b->pair_read[p];
        const char *hap = b->hap_seq + b->hap_off[h];
        const int hap_len = (int)(b->hap_off[h + 1] - b->hap_off[h]);
        const char *rs = b->read_seq + b->read_off[r];
        const char *rq = b->read_qual + b->read_off[r];
        const int rlen = (int)(b->read_off[r + 1] - b->read_off[r]);
        int a = 0, sc;
        if (ext_align == NULL) {
            const int *ht = NULL;
            if (b->hap_varlen[h] != 0 && b->reuse_hash) {
                if (cached_hap != h) { po_hash_sequence(hap, hap_len, tab); cached_hap = h; }
                ht = tab;
            }
            sc = po_align_no_traceback(rs, rq, rlen, b->read_pos[r], hap, hap_len,
                                       b->hap_start[h], b->hap_varlen[h],
                                       b->gapextend, b->nucprior, b->table, ht, &a);
        } else {
            /* same anchor logic, reference routine for the DP itself */
            const int len1 = rlen + 15;
            if (b->hap_varlen[h] == 0) {
                int hp = b->read_pos[r] - b->hap_start[h];
                a = hp - 8 > 0 ? hp - 8 : 0;
                if (len1 + a <= hap_len)
                    sc = ext_align((char *)hap + a, (char *)rs, (char *)rq, len1, rlen,
                                   b->gapextend, b->nucprior, (char *)b->table, 1);
                else
                    sc = PO_ASSERT_FAILED;
            } else {
                po_hash_sequence(hap, hap_len, tab);   /* per pair, as calign.pyx:66 */
                int mp = po_map_read(rs, rlen, hap_len, tab);
                a = mp - 8 > 0 ? mp - 8 : 0;
                if (len1 + a <= hap_len) {
                    sc = ext_align((char *)hap + a, (char *)rs, (char *)rq, len1, rlen,
                                   b->gapextend, b->nucprior, (char *)b->table, 1);
                } else {
                    char *ext = (char *)malloc((size_t)hap_len + (size_t)rlen + 16);
                    memcpy(ext, hap, (size_t)hap_len);
                    memset(ext + hap_len, 'N', (size_t)rlen);
                    memset(ext + hap_len + rlen, 0, 16);
                    sc = ext_align(ext + a, (char *)rs, (char *)rq, len1, rlen,
                                   b->gapextend, b->nucprior, (char *)b->table, 1);
                    free(ext);
                }
            }
        }
        out_score[p] = sc;
        out_anchor[p] = a;
    }
    free(tab);
}
