/*
 * sanitize_driver.c -- TEST INFRASTRUCTURE ONLY.
 * Runs both CPU restatements (platy_oracle.c, platy_caller_oracle.c) under
 * AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle asan) on
 *   (1) golden fastAlignmentRoutine calls read from a flat file written by the
 *       test (n, then per call: len2, expected score, seq1[len2+15], seq2[len2], qual[len2]),
 *   (2) random windows through hash_sequence / map_read / alignNoTraceback,
 *   (3) random windows through Caller.setup / Caller.call / calculatePosterior.
 * Exit status 0 = every golden score matched and no sanitizer report (a report aborts).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "platy_oracle.h"

static const char TABLE[] = "NKJHFA=854210/.-,,+**))((('''&&&%%%$$$$#####\"\"\"\"\"";
static unsigned long long rng_state = 88172645463325252ULL;
static unsigned rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 11); }

int main(int argc, char **argv)
{
    long bad = 0, n_golden = 0;
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) { perror(argv[1]); return 2; }
        int32_t n;
        if (fread(&n, 4, 1, f) != 1) return 2;
        for (int i = 0; i < n; ++i) {
            int32_t hdr[2];
            if (fread(hdr, 4, 2, f) != 2) return 2;
            const int len2 = hdr[0], len1 = len2 + 15;
            /* exact-size heap blocks: any read past an end is reported */
            char *s1 = malloc((size_t)len1), *s2 = malloc((size_t)len2 > 0 ? (size_t)len2 : 1), *q = malloc((size_t)len2 > 0 ? (size_t)len2 : 1);
            if (fread(s1, 1, (size_t)len1, f) != (size_t)len1 || fread(s2, 1, (size_t)len2, f) != (size_t)len2 ||
                fread(q, 1, (size_t)len2, f) != (size_t)len2) return 2;
            const int got = po_fast_align(s1, s2, q, len1, len2, 3, 2, TABLE);
            if (got != hdr[1]) { if (bad < 5) fprintf(stderr, "golden call %d: %d != %d\n", i, got, hdr[1]); ++bad; }
            free(s1); free(s2); free(q);
            ++n_golden;
        }
        fclose(f);
    }
    /* (2) windows: haplotypes with and without indels, reads hanging off both ends */
    for (int w = 0; w < 200; ++w) {
        const int hl = 15 + (int)(rnd() % 900), rl = 1 + (int)(rnd() % 200);
        char *hap = malloc((size_t)hl), *read = malloc((size_t)rl + 1), *qual = malloc((size_t)rl);
        for (int i = 0; i < hl; ++i) hap[i] = "ACGTN"[rnd() % 100 == 0 ? 4 : rnd() % 4];
        const int o = (int)(rnd() % (unsigned)hl);
        for (int i = 0; i < rl; ++i) { read[i] = (o + i < hl && rnd() % 20) ? hap[o + i] : "ACGT"[rnd() % 4]; qual[i] = (char)(rnd() % 94); }
        read[rl] = 0;
        int *tab = malloc(sizeof(int) * PO_HASH_SIZE);
        po_hash_sequence(hap, hl, tab);
        (void)po_map_read(read, rl, hl, tab);
        int anchor = 0;
        for (int var = 0; var < 2; ++var)
            (void)po_align_no_traceback(read, qual, rl, 1000 + o + (int)(rnd() % 17) - 8, hap, hl, 1000, var ? 3 : 0,
                                        3, 2, TABLE, var ? tab : NULL, &anchor);
        free(tab); free(hap); free(read); free(qual);
    }
    /* (3) caller stage */
    for (int w = 0; w < 60; ++w) {
        const int nh = 1 + (int)(rnd() % 12), ni = 1 + (int)(rnd() % 5), nr = ni + (int)(rnd() % 40);
        const int ng = po_n_genotypes(nh);
        int32_t *off = malloc(sizeof(int32_t) * (size_t)(ni + 1));
        for (int i = 0; i <= ni; ++i) off[i] = (int32_t)((long)nr * i / ni);
        double *like = malloc(sizeof(double) * (size_t)nh * (size_t)nr), *post = malloc(sizeof(double) * (size_t)ni * (size_t)ng);
        for (int i = 0; i < nh * nr; ++i) like[i] = 1e-6 + (rnd() % 1000000) / 1000001.0;
        po_caller_setup(nh, ni, nr, off, like, post);
        double *freq = malloc(sizeof(double) * (size_t)nh), *cisr = malloc(sizeof(double) * (size_t)ni * (size_t)ng), mc = 0;
        int32_t *bg = malloc(sizeof(int32_t) * (size_t)ni);
        double *bv = malloc(sizeof(double) * (size_t)ni);
        for (int literal = 0; literal < 2; ++literal)
            (void)po_caller_call(nh, ni, off, post, 100, literal, freq, cisr, &mc, bg, bv);
        uint8_t *mask = calloc((size_t)nh, 1);
        mask[rnd() % (unsigned)nh] = 1;
        (void)po_variant_posterior(nh, ni, off, post, freq, mask, 1e-3);
        free(off); free(like); free(post); free(freq); free(cisr); free(bg); free(bv); free(mask);
    }
    printf("sanitize_driver: %ld golden calls, %ld mismatches\n", n_golden, bad);
    return bad ? 1 : 0;
}
