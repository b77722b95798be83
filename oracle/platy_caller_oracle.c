/*
 * platy_caller_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the genotype-likelihood / EM / variant-posterior stage of
 * Platypus 0.1.5 (platypus/cpopulation.pyx), the consumer of the read x
 * haplotype likelihood matrix.  It is the checker for the HIP kernels of that
 * stage; nothing in the product links or loads it.
 *
 * PARITY UNPINNED: the reference's Cython layer cannot be imported here and the
 * reference ships no test or fixture for this stage, so these functions follow
 * the source text only (each cites the lines it restates).  All sums run in the
 * reference's order, so on the same libm the results are the reference's doubles.
 *
 * A window's likelihoods are one dense matrix like[h * n_reads + r]; the reads
 * of individual i are r in [ind_read_off[i], ind_read_off[i+1]).  Genotypes are
 * the n_hap (n_hap + 1) / 2 pairs (s <= r) in the order of
 * generateAllGenotypesFromHaplotypeList, platypus/cgenotype.pyx:90-96.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "platy_oracle.h"

int po_n_genotypes(int n_hap) { return n_hap * (n_hap + 1) / 2; }

/* haplotypeIndexes of platypus/cpopulation.pyx:711-715 for the canonical list */
void po_genotype_indexes(int n_hap, int32_t *h1, int32_t *h2)
{
    int j = 0;
    for (int a = 0; a < n_hap; ++a)
        for (int b = a; b < n_hap; ++b) { h1[j] = a; h2[j] = b; ++j; }
}

/* Caller.setup, platypus/cpopulation.pyx:696-734: log-likelihood of every
 * (individual, genotype), one maximum over all of them (start value -1e7),
 * post[i][g] = exp(ll - max).  An individual without reads keeps 0.0 as its
 * log-likelihoods (cpopulation.pyx:717-718) and does not take part in the maximum. */
void po_caller_setup(int n_hap, int n_ind, int n_reads, const int32_t *ind_read_off,
                     const double *like, double *post)
{
    const int n_gen = po_n_genotypes(n_hap);
    double max_ll = -1e7;                                    /* cpopulation.pyx:696 */
    for (int i = 0; i < n_ind; ++i) {
        const int r0 = ind_read_off[i], nr = ind_read_off[i + 1] - r0;
        int g = 0;
        for (int a = 0; a < n_hap; ++a)
            for (int b = a; b < n_hap; ++b, ++g) {
                if (nr == 0) { post[(size_t)i * n_gen + g] = 0.0; continue; }
                const double ll = po_genotype_loglik(like + (size_t)a * n_reads + r0,
                                                     like + (size_t)b * n_reads + r0, nr);
                if (ll > max_ll) max_ll = ll;                /* cpopulation.pyx:722-723 */
                post[(size_t)i * n_gen + g] = ll;
            }
    }
    for (size_t k = 0; k < (size_t)n_ind * n_gen; ++k) post[k] = exp(post[k] - max_ll);   /* :734 */
}

/* Caller.EMiteration, platypus/cpopulation.pyx:807-890.  literal != 0 keeps the
 * reference's loop nest (haplotype outermost, O(n_hap n_ind n_gen)); otherwise
 * every genotype adds to the sums of its two haplotypes in one pass over
 * (individual, genotype) -- each sum then receives the same terms in the same
 * order, so the result is the same double. */
static double em_iteration(int n_hap, int n_ind, int n_gen, const int32_t *ind_read_off,
                           const int32_t *h1, const int32_t *h2, const double *post,
                           double *freq, double *cisr, double *tmp, int literal)
{
    int n_with_data = 0;
    for (int i = 0; i < n_ind; ++i) {
        if (ind_read_off[i + 1] == ind_read_off[i]) continue;          /* :834-835 */
        const double *p = post + (size_t)i * n_gen;
        double *csr = cisr + (size_t)i * n_gen;
        double sum = 0.0;
        ++n_with_data;
        for (int j = 0; j < n_gen; ++j) {
            const int s = h1[j], r = h2[j];
            const double v = p[j] * freq[s] * freq[r] * (1 + (r != s));  /* :847 */
            csr[j] = v;
            sum += v;
        }
        if (sum > 0.0)
            for (int j = 0; j < n_gen; ++j) csr[j] /= sum;             /* :852-854 */
    }
    if (literal) {
        double max_change = 0.0;
        for (int k = 0; k < n_hap; ++k) {
            double t = 0.0;
            for (int i = 0; i < n_ind; ++i) {
                if (ind_read_off[i + 1] == ind_read_off[i]) continue;
                const double *csr = cisr + (size_t)i * n_gen;
                for (int j = 0; j < n_gen; ++j) {
                    const int s = h1[j], r = h2[j];
                    if (k == s || k == r) t += (s == r) ? 2.0 * csr[j] : csr[j];   /* :875-880 */
                }
            }
            const double old = freq[k], nf = t / (2 * n_with_data);    /* :882-884 */
            freq[k] = nf;
            const double ch = fabs(old - nf);
            if (ch > max_change) max_change = ch;
        }
        return max_change;
    }
    for (int k = 0; k < n_hap; ++k) tmp[k] = 0.0;
    for (int i = 0; i < n_ind; ++i) {
        if (ind_read_off[i + 1] == ind_read_off[i]) continue;
        const double *csr = cisr + (size_t)i * n_gen;
        for (int j = 0; j < n_gen; ++j) {
            const int s = h1[j], r = h2[j];
            if (s == r) tmp[s] += 2.0 * csr[j];
            else { tmp[s] += csr[j]; tmp[r] += csr[j]; }
        }
    }
    double max_change = 0.0;
    for (int k = 0; k < n_hap; ++k) {
        const double old = freq[k], nf = tmp[k] / (2 * n_with_data);
        freq[k] = nf;
        const double ch = fabs(old - nf);
        if (ch > max_change) max_change = ch;
    }
    return max_change;
}

/* Caller.call, platypus/cpopulation.pyx:741-805.  Returns the number of
 * iterations.  best_gen[i] = -1 for an individual without reads (its call is
 * None and its value 0.0, cpopulation.pyx:777-780).  cisr may be NULL. */
int po_caller_call(int n_hap, int n_ind, const int32_t *ind_read_off, const double *post,
                   int max_iters, int literal, double *freq, double *cisr_out,
                   double *max_change_out, int32_t *best_gen, double *best_val)
{
    const int n_gen = po_n_genotypes(n_hap);
    int32_t *h1 = malloc(sizeof(int32_t) * n_gen), *h2 = malloc(sizeof(int32_t) * n_gen);
    double *cisr = cisr_out ? cisr_out : calloc((size_t)n_ind * n_gen, sizeof(double));
    double *tmp = malloc(sizeof(double) * n_hap);
    po_genotype_indexes(n_hap, h1, h2);
    if (cisr_out) for (size_t k = 0; k < (size_t)n_ind * n_gen; ++k) cisr[k] = 0.0;   /* calloc, :701 */
    double eps = 1.0 / (n_ind * 2 * 2);                                /* :751 */
    if (1e-3 < eps) eps = 1e-3;
    double max_change = eps + 1;
    for (int k = 0; k < n_hap; ++k) freq[k] = 1.0 / n_hap;              /* :753,763-764 */
    int iters = 0;
    while (max_change > eps && iters < max_iters) {                     /* :766-768 */
        max_change = em_iteration(n_hap, n_ind, n_gen, ind_read_off, h1, h2, post, freq, cisr, tmp, literal);
        ++iters;
    }
    for (int i = 0; i < n_ind; ++i) {                                   /* :773-797 */
        if (ind_read_off[i + 1] == ind_read_off[i]) { best_gen[i] = -1; best_val[i] = 0.0; continue; }
        const double *csr = cisr + (size_t)i * n_gen;
        int best = -1;
        double mx = 0.0;
        for (int j = 0; j < n_gen; ++j)
            if (best == -1 || csr[j] > mx) { mx = csr[j]; best = j; }
        best_gen[i] = best;
        best_val[i] = mx;
    }
    *max_change_out = max_change;
    free(h1); free(h2); free(tmp);
    if (!cisr_out) free(cisr);
    return iters;
}

/* Population.calculatePosterior, platypus/cpopulation.pyx:515-632.
 * hap_has_var[h] != 0: the variant is one of haplotype h's variants. */
double po_variant_posterior(int n_hap, int n_ind, const int32_t *ind_read_off, const double *post,
                            const double *freq, const uint8_t *hap_has_var, double prior)
{
    const int n_gen = po_n_genotypes(n_hap);
    double *fp = malloc(sizeof(double) * n_hap);
    double sum_freqs = 0.0;
    for (int i = 0; i < n_hap; ++i) {                                   /* :554-566 */
        if (!hap_has_var[i]) { fp[i] = freq[i]; sum_freqs += freq[i]; }
        else fp[i] = 0.0;
    }
    if (sum_freqs > 0)                                                  /* :571-580 */
        for (int i = 0; i < n_hap; ++i)
            if (fp[i] != 0.0) fp[i] /= sum_freqs;
    double slp_var = 0.0, slp_novar = 0.0;
    const int log_min_float = -708;
    for (int i = 0; i < n_ind; ++i) {                                   /* :586-624 */
        if (ind_read_off[i + 1] == ind_read_off[i]) continue;
        const double *gl = post + (size_t)i * n_gen;
        double pv = 0.0, pn = 0.0;
        int j = 0;
        for (int r = 0; r < n_hap; ++r)
            for (int s = r; s < n_hap; ++s, ++j) {
                const double factor = r != s ? 2.0 : 1.0;
                pv += (factor * freq[r] * freq[s] * gl[j]);            /* :609 */
                pn += (factor * fp[r] * fp[s] * gl[j]);                /* :614 */
            }
        slp_var += pv > 0 ? log(pv) : log_min_float;
        slp_novar += pn > 0 ? log(pn) : log_min_float;
    }
    free(fp);
    const double ratio = exp(slp_novar - slp_var);                      /* :626-627 */
    return prior / (prior + ratio * (1.0 - prior));
}
