/*
 * platy_oracle.h -- TEST INFRASTRUCTURE ONLY (see platy_oracle.c).
 * CPU restatement of the Platypus 0.1.5 alignment/likelihood path, used as
 * the checker for the HIP kernels and as the timed CPU baseline.
 */
#ifndef PLATY_ORACLE_H
#define PLATY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_HASH_SIZE 16384      /* platypus/calign.pyx:12 */
#define PO_HASH_NUCS 7          /* platypus/calign.pyx:13 */
#define PO_ASSERT_FAILED (-1)   /* assertion at platypus/calign.pyx:61 would fail */

/* signature of the reference routine, platypus/c/align.h:6 */
typedef int (*po_align_fn)(char *seq1, char *seq2, char *qual2, int len1, int len2,
                           int gapextend, int nucprior, char *table, char usehomopol);

void po_gap_open(const char *seq1, int len1, const char *table, int16_t *out);
int po_fast_align(const char *seq1, const char *seq2, const char *qual2,
                  int len1, int len2, int gapextend, int nucprior, const char *table);
unsigned po_hash7(const char *s);
void po_hash_sequence(const char *seq, int seqlen, int *table);
int po_map_read(const char *read, int readlen, int seqlen, const int *table);
double po_likelihood(int score, int mapq);
int po_align_no_traceback(const char *read, const char *qual, int rlen, int rpos,
                          const char *hap, int hap_len, int hap_start, int var_seq_len,
                          int gapextend, int nucprior, const char *table,
                          const int *hashtab, int *anchor);
double po_genotype_loglik(const double *l1, const double *l2, int nreads);

/* packed batch: concatenated sequences + offsets, one pair list */
typedef struct {
    const char *hap_seq;  const int64_t *hap_off;  const int *hap_start; const int *hap_varlen;
    const char *read_seq; const char *read_qual;   const int64_t *read_off;
    const int *read_pos;
    const int *pair_hap;  const int *pair_read;
    const char *table;    /* NUL-terminated gap-open string */
    int gapextend, nucprior;
    int reuse_hash;       /* 1: build each haplotype's 7-mer table once per run of equal hap indices */
} po_batch;

void po_run_pairs(const po_batch *b, long lo, long hi, po_align_fn ext_align,
                  int *out_score, int *out_anchor);

#ifdef __cplusplus
}
#endif
#endif
