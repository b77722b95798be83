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

/* ---- genotype likelihood / EM / variant posterior stage (platy_caller_oracle.c) ----
 * window matrix like[h * n_reads + r]; individual i owns reads [ind_read_off[i], ind_read_off[i+1]);
 * genotypes in the order of platypus/cgenotype.pyx:90-96 */
int po_n_genotypes(int n_hap);
void po_genotype_indexes(int n_hap, int32_t *h1, int32_t *h2);
void po_caller_setup(int n_hap, int n_ind, int n_reads, const int32_t *ind_read_off,
                     const double *like, double *post);
int po_caller_call(int n_hap, int n_ind, const int32_t *ind_read_off, const double *post,
                   int max_iters, int literal, double *freq, double *cisr_out,
                   double *max_change_out, int32_t *best_gen, double *best_val);
double po_variant_posterior(int n_hap, int n_ind, const int32_t *ind_read_off, const double *post,
                            const double *freq, const uint8_t *hap_has_var, double prior);

#ifdef __cplusplus
}
#endif
#endif
