/*
 * platgpu.h -- C ABI of libplatgpu.so: batched read-to-haplotype alignment and
 * likelihood on an AMD MI355X (gfx950), the device replacement for the
 * per-pair CPU path of Platypus 0.1.5.
 *
 * Conventions: every function returns PG_OK (0) or a negative PG_ERR_* code
 * and never throws or longjmps across the boundary; every buffer is owned by
 * the caller; a context belongs to one host thread at a time; all work of a
 * context runs on one HIP stream created with the context.  There is no CPU
 * fallback: without a usable gfx950 device pg_ctx_create fails.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference checkout) is cited on the declaration.
 */
#ifndef PLATGPU_H
#define PLATGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 8

enum {
    PG_OK = 0,
    PG_ERR_NO_DEVICE = -1,     /* no HIP device / wrong architecture                    */
    PG_ERR_INVALID = -2,       /* bad argument (NULL, negative size, index out of range) */
    PG_ERR_HIP = -3,           /* a HIP runtime call failed (see pg_last_error_string)   */
    PG_ERR_UNSUPPORTED = -4,   /* input outside the supported domain (see below)         */
    PG_ERR_STATE = -5          /* call order: nothing uploaded / nothing launched        */
};

/* score value of a pair for which the reference's assertion at
 * platypus/calign.pyx:61 fails (slice would run past the haplotype end on the
 * no-indel path); the reference's cdef function then returns likelihood 0.0. */
#define PG_SCORE_ASSERT_FAILED (-1)

/* largest haplotype the 7-mer anchoring accepts, platypus/calign.pyx:14,112,132 */
#define PG_MAX_HASHED_HAP_LEN 16383

/* kernel ids for pg_kernel_time_ms */
enum { PG_K_ANCHOR = 0, PG_K_ALIGN = 1, PG_K_GENOTYPE = 2, PG_K_CALLER_SETUP = 3, PG_K_CALLER_EM = 4, PG_K_COUNT = 5 };

typedef struct pg_ctx pg_ctx;

/* ABI version of the loaded library (compare with PG_ABI_VERSION). */
int pg_abi_version(void);

/*
 * Number of gfx950 devices this process can open (0 when there is none).  A
 * launcher that starts one worker per GPU binds worker k to device
 * k mod this number.  Replaces: nothing in the reference; its counterpart is
 * the --nCPU option that sizes the worker pool (platypus/runner.py:98,132).
 */
int pg_device_count(int *count);

/*
 * One context per process and GPU.  `max_read_len` bounds the read length the
 * context accepts (device haplotypes are stored with that many trailing 'N's,
 * the extension platypus/calign.pyx:73-75 makes per call); 0 selects 512.
 * Replaces: nothing in the reference (it has no device); it is the state the
 * Cython layer keeps for the lifetime of a worker process (platypus/runner.py:132).
 */
int pg_ctx_create(int device_ordinal, int max_read_len, pg_ctx **out);
int pg_ctx_destroy(pg_ctx *ctx);

/* Text of the last error of this context (or of creation when ctx is NULL). */
const char *pg_last_error_string(const pg_ctx *ctx);

/*
 * Page-locked host memory for the buffers a caller hands to pg_*_upload and
 * pg_results_fetch / pg_caller_fetch: transfers from and to such memory run at
 * the full PCIe rate (pageable memory goes through a staging copy at about a
 * third of it).  Optional: every entry point takes any host memory.  No
 * counterpart in the reference (it has no device).
 */
int pg_host_alloc(size_t bytes, void **out);
int pg_host_free(void *p);

/*
 * Alignment parameters: the arguments platypus/chaplotype.pyx:264 passes down
 * (gapextend 3, nucprior 2, the homopolymer gap-open string of
 * platypus/chaplotype.pyx:37-40, usehomopolgapq = 1).  `table` holds
 * `table_len` phred+33 characters, 1..128 of them (the reference's has 49).
 * Defaults are those values.
 */
int pg_set_params(pg_ctx *ctx, int gapextend, int nucprior, const char *table, int table_len);

/*
 * Kernel-choice overrides for tests and measurements; the library's own choice
 * is always correct, an override only forces one of several equivalent paths.
 * No counterpart in the reference.  Options (value "default" or "" restores the
 * library's choice):
 *   "em_kernel"      small | big | multi   EM path of pg_caller_launch
 *   "em_small"       exact | fast          windows of up to 8 haplotypes (16 while light): EM sums in the reference's own order
 *                                          (the default: its doubles) or regrouped (faster, equal to 1e-9)
 *   "gl_kernel"      cell | tile           genotype-likelihood kernel of pg_caller_launch
 *   "em_stage_max"   doubles of one window the "big" EM kernel may keep in LDS
 *   "em_step"        full | sym            the "multi" path's step reads whole matrices or their upper triangle
 *   "em_step_waves"  1..16 wavefronts per workgroup of the triangle step
 *   "vote_waves"     1..4 wavefronts per workgroup of the anchoring kernel (bit-parallel kernel: 64 pairs each)
 *   "vote_kernel"    table | bits          anchoring by a 16384-entry position table per haplotype in LDS, or by exact
 *                                          bit-parallel counts for candidate positions (the default for reads of up to
 *                                          263 bases; longer reads always take the table)
 *   "align_kernel"   lane | band           the banded DP with one lane per pair, or with eight lanes per pair as the
 *                                          reference's 8 x int16 vectors have it (the default for launches of up to
 *                                          16384 pairs, whose few wavefronts then finish in a third of the time)
 *   "align_lds_pad"  bytes of unused LDS per workgroup of the DP kernel (lowers occupancy)
 * The process environment is never consulted.
 */
int pg_set_option(pg_ctx *ctx, const char *name, const char *value);

/*
 * Haplotypes of a batch: `n` sequences concatenated in `seq`, sequence i is
 * seq[offsets[i] .. offsets[i+1]).  start_pos = Haplotype.startPos,
 * var_seq_len = Haplotype.varSeqLen (platypus/chaplotype.pxd:8-25,
 * platypus/chaplotype.pyx:59-78).  Replaces the per-call arguments hapSeq,
 * hapStart, varSeqLen, hapLen of alignNoTraceback (platypus/calign.pxd:5-17)
 * and the per-call gap-open table build of platypus/c/align.c:70-96, which
 * becomes a per-haplotype run-length array computed once on upload.
 * A haplotype with var_seq_len != 0 must be 15..16383 long (outside that range
 * the reference itself reads out of bounds or fails its own assertions).
 * Sequence bytes (haplotypes and reads) are compared for equality only, as the
 * reference does; bytes 1..31 and 33..126 are taken, any other is refused with
 * PG_ERR_UNSUPPORTED.
 */
int pg_haps_upload(pg_ctx *ctx, int n, const char *seq, const int64_t *offsets,
                   const int32_t *start_pos, const int32_t *var_seq_len);

/*
 * The same batch of haplotypes put together on the device from reference
 * stretches and variant lists: what the reference's constructor does on the host,
 * one haplotype at a time (platypus/chaplotype.pyx:50-86 through
 * getMutatedSequence, platypus/chaplotype.pyx:283-353).  A window's stretch goes up
 * once and every haplotype of it costs its variants only, instead of a whole
 * sequence each.
 *   stretch s   ref_seq[ref_off[s] .. ref_off[s+1]) holds the reference bases of
 *               positions ref_start[s] .. ref_start[s] + length: the padded,
 *               clamped bounds [startPos, endPos) of platypus/chaplotype.pyx:70-71,
 *               as refFile.getSequence returns them
 *   haplotype i stretch hap_stretch[i] with the variants
 *               [hap_var_off[i], hap_var_off[i+1]) applied, which must be in the
 *               sorted order the reference's constructor is given them in
 *   variant v   refPos var_pos[v], nRemoved var_n_removed[v], added sequence
 *               var_added[var_added_off[v] .. var_added_off[v+1]) (nAdded is its
 *               length): substitutions (nAdded == nRemoved), deletions
 *               (nAdded == 0; refPos is the base before the deleted ones) and
 *               insertions (nRemoved == 0; after the base at refPos), anything else
 *               is PG_ERR_UNSUPPORTED (the reference's ValueError,
 *               platypus/chaplotype.pyx:340-342)
 * Afterwards the context holds what pg_haps_upload leaves for the sequences
 * getMutatedSequence returns, start_pos = ref_start[hap_stretch[i]] and
 * var_seq_len = the sum of |nAdded - nRemoved| over the haplotype's variants
 * (platypus/chaplotype.pyx:66-67); the same limits and errors apply.  A variant
 * that reaches outside its stretch is PG_ERR_INVALID.  At most 2^31 - 1 bytes
 * of stretches per call.
 */
int pg_haps_build(pg_ctx *ctx, int n_stretches, const char *ref_seq, const int64_t *ref_off, const int32_t *ref_start,
                  int n_haps, const int32_t *hap_stretch, const int32_t *hap_var_off, const int32_t *var_pos,
                  const int32_t *var_n_removed, const int32_t *var_added_off, const char *var_added);

/*
 * The sequences of the context's haplotype batch back on the host (synchronises):
 * n + 1 offsets starting at 0 and the bytes, Haplotype.haplotypeSequence of
 * platypus/chaplotype.pyx:78 for haplotypes made by pg_haps_build.  Either output
 * may be NULL; ask for the offsets first to size `seq`.
 */
int pg_haps_fetch(pg_ctx *ctx, char *seq, int64_t *offsets);

/*
 * Reads of a batch, the five cAlignedRead fields the path uses
 * (platypus/samtoolsWrapper.pxd:293-315: seq, qual, pos, rlen, mapq), packed:
 * ASCII bases and RAW phred qualities (no +33), read i at [offsets[i],
 * offsets[i+1]).  Qualities must be <= 127 and mapq in 0..255 (BAM's ranges),
 * lengths 1..max_read_len.
 */
int pg_reads_upload(pg_ctx *ctx, int n, const char *seq, const char *qual,
                    const int64_t *offsets, const int32_t *pos, const int32_t *mapq);

/*
 * The same batch with the bases as a BAM record holds them: two per byte in the
 * 4-bit code "=ACMGRSVTWYHKDBN" (bam1_seq / bam_nt16_rev_table of the vendored
 * samtools, platypus/c/samtools/bam.h; what platypus/samtoolsWrapper.pyx turns
 * into the characters of cAlignedRead.seq one read at a time), the high half of a
 * byte first.  Base k of the batch is nibble k of seq4 -- read i's base j is nibble
 * offsets[i] + j, reads follow one another without padding -- so seq4 holds
 * (offsets[n] + 1) / 2 bytes; qualities, offsets, positions and mapping
 * qualities as for pg_reads_upload.  The device expands the codes to the same
 * characters and everything behind that is shared; a batch's bases cost half the
 * bytes on the way up.
 */
int pg_reads_upload_nibbles(pg_ctx *ctx, int n, const uint8_t *seq4, const char *qual,
                            const int64_t *offsets, const int32_t *pos, const int32_t *mapq);

/*
 * The (haplotype, read) pairs to score, indices into the uploaded batches.
 * Replaces the loop of Haplotype.alignReads (platypus/chaplotype.pyx:259-268)
 * over one individual's reads, for as many haplotypes/individuals/windows as
 * the caller batches together.
 */
int pg_pairs_upload(pg_ctx *ctx, int64_t n_pairs, const int32_t *hap_idx, const int32_t *read_idx);

/* Dense block: every haplotype in [hap_lo,hap_hi) x every read in
 * [read_lo,read_hi), haplotype-major (pair = (h-hap_lo)*(read_hi-read_lo) + (r-read_lo)). */
int pg_pairs_block(pg_ctx *ctx, int hap_lo, int hap_hi, int read_lo, int read_hi);

/*
 * Several dense blocks in one pair list, block after block, each haplotype-major
 * like pg_pairs_block: the pairs of all windows collected for one launch (the
 * double loop of platypus/cpopulation.pyx:698-725, window after window of
 * platypus/variantcaller.pyx:116-187).  pair_base (may be NULL) receives the index of
 * every block's first pair, which is the like_base pg_windows_upload takes.
 */
int pg_pairs_blocks(pg_ctx *ctx, int n_blocks, const int32_t *hap_lo, const int32_t *n_haps,
                    const int32_t *read_lo, const int32_t *n_reads, int64_t *pair_base);

/*
 * Enqueue the anchoring and alignment kernels for the uploaded pairs on the
 * context's stream (asynchronous).  Per pair this computes what
 * alignNoTraceback does (platypus/calign.pyx:36-78): anchor =
 * indexOfReadIntoHap (hapPos-8 clamped at 0 on the no-indel path, the 7-mer
 * vote of hash_sequence + map_read, platypus/calign.pyx:99-158, otherwise),
 * score = fastAlignmentRoutine (platypus/c/align.c:18-235) on the slice at
 * that anchor, likelihood = min(1, exp(mLTOT*score)*(1-pW)+pW).
 */
int pg_align_launch(pg_ctx *ctx);

/* Block until everything enqueued on the context has finished. */
int pg_sync(pg_ctx *ctx);

/*
 * Copy results of the last launch to the host (synchronises first).  Any of
 * the three outputs may be NULL.  score: int32 per pair, 0..2048 or
 * PG_SCORE_ASSERT_FAILED; anchor: the slice start; likelihood: double per
 * pair, bit-identical to the reference's libm arithmetic (the two exp()
 * factors are tabulated on the host with the C library).
 */
int pg_results_fetch(pg_ctx *ctx, int32_t *score, int32_t *anchor, double *likelihood);

/*
 * The same copies enqueued on the context's stream without waiting: the outputs
 * are complete after the next pg_sync (or any later synchronising call) on this
 * context.  With outputs in page-locked memory (pg_host_alloc) the transfer runs
 * beside whatever the host does next -- another context's uploads, say; into
 * pageable memory the HIP runtime may wait before it returns.
 */
int pg_results_fetch_async(pg_ctx *ctx, int32_t *score, int32_t *anchor, double *likelihood);

/* upload pairs + launch + sync + fetch in one call */
int pg_align_pairs(pg_ctx *ctx, int64_t n_pairs, const int32_t *hap_idx, const int32_t *read_idx,
                   int32_t *score, int32_t *anchor, double *likelihood);

/*
 * One window in one call: every haplotype against every read, haplotype-major
 * (pair h * n_reads + r), the matrix the reference fills for a window by calling
 * Haplotype.alignReads for each of its haplotypes (platypus/chaplotype.pyx:239-270,
 * from platypus/cgenotype.pyx:55-56), each of which walks the window's reads
 * through alignNoTraceback (platypus/calign.pyx:36-78).  Arguments as for
 * pg_haps_upload and pg_reads_upload, outputs as for pg_results_fetch
 * (n_haps * n_reads elements each, likelihood may be NULL); the same values as
 * pg_haps_upload + pg_reads_upload + pg_pairs_block + pg_align_launch +
 * pg_results_fetch give, and the same errors.  It is the call for a caller that
 * works window by window, as the reference does: the host lays the window out as
 * one block in pinned memory, one copy takes it to the device, one kernel makes
 * every array the anchoring and alignment kernels read, and one copy and one wait
 * bring the results back (the five calls above make some twenty copies and four
 * waits).  Windows of more than 2^20 pairs or 16 Mi sequence elements, and windows
 * with indel-bearing haplotypes whose reads are longer than 263 bases, are passed
 * through those five calls.  Afterwards the context holds no uploaded batch;
 * the results stay on the device for pg_genotype_loglik, the caller stage and
 * pg_results_fetch, as after pg_align_launch (pg_kernel_time_ms has nothing to
 * report for this call: its kernels are not bracketed by events).
 */
int pg_align_window(pg_ctx *ctx, int n_haps, const char *hap_seq, const int64_t *hap_off, const int32_t *hap_start_pos,
                    const int32_t *hap_var_seq_len, int n_reads, const char *read_seq, const char *read_qual,
                    const int64_t *read_off, const int32_t *read_pos, const int32_t *read_mapq,
                    int32_t *score, int32_t *anchor, double *likelihood);

/*
 * pg_align_pairs for callers that cannot allocate device memory: tight serving
 * loops, harnesses that forbid hipMalloc, code between two captured HIP graphs
 * (the call itself cannot be captured: it waits for the stream and reads status
 * words back).  It has no counterpart in the reference.  Same computation and results as
 * pg_align_pairs on the uploaded haplotypes and reads, but
 *   - the pair indices and the three outputs are DEVICE pointers of the caller
 *     (n_pairs elements each; d_likelihood may be NULL),
 *   - every intermediate array (vote records, slot order, wavefront descriptors,
 *     anchors, counters) lives in one fixed device array of about 7 MiB that
 *     each GPU the library runs on holds until the process ends, so the call
 *     makes no hipMalloc / hipFree and no allocating library call,
 *   - at most PG_NOALLOC_MAX_PAIRS pairs per call and at most
 *     PG_NOALLOC_MAX_INDEL_HAPS haplotypes with var_seq_len != 0 in the uploaded
 *     batch; more is refused with PG_ERR_UNSUPPORTED.
 * The call returns after the results are complete (it waits for the context's
 * stream).  All contexts of a GPU share that GPU's array: two calls that overlap
 * on one GPU (two host threads, two contexts) would race on it, so the second
 * one is refused with PG_ERR_STATE and writes nothing; the caller serialises
 * them or tries again.  The results of pg_align_launch and the pair list of pg_pairs_upload
 * are left untouched.  An index outside the uploaded batches gives
 * PG_ERR_INVALID and no output is written.
 */
#define PG_NOALLOC_MAX_PAIRS 262144
#define PG_NOALLOC_MAX_INDEL_HAPS 65536
int pg_align_pairs_noalloc(pg_ctx *ctx, int64_t n_pairs, const int32_t *d_hap_idx, const int32_t *d_read_idx,
                           int32_t *d_score, int32_t *d_anchor, double *d_likelihood);

/*
 * hash_sequence of the reference on its own (platypus/calign.pyx:99-123): the
 * table of unique 7-mer starts of `seq` in the reference's layout -- int32[16384],
 * index my_hash (platypus/calign.pyx:81-96), value i+1, -1 for a 7-mer seen more
 * than once, 0 for an absent one; the last 7-mer is left out as in the reference.
 * len must be below 16384 (platypus/calign.pyx:112); a sequence shorter than 7
 * gives the empty table.  The alignment path does not need this call (its
 * table lives in LDS); it exists for callers of the reference's cdef function.
 */
int pg_hash_sequence(pg_ctx *ctx, const char *seq, int len, int32_t *table);

/*
 * map_read of the reference on its own (platypus/calign.pyx:126-158): the start
 * of `read` in a sequence of length seq_len whose table (as made by
 * pg_hash_sequence) is given; may be negative; 0 if nothing votes or the read
 * is shorter than 7.  A table entry that points outside the sequence (the
 * reference's assertion, platypus/calign.pyx:148) is an error.
 */
int pg_map_read(pg_ctx *ctx, const char *read, int read_len, int seq_len, const int32_t *table, int32_t *position);

/*
 * Device time in milliseconds of kernel group `which` in the last
 * pg_align_launch / pg_genotype_loglik / pg_caller_launch, measured with HIP events on the
 * context's stream.  Synchronises.
 */
int pg_kernel_time_ms(pg_ctx *ctx, int which, float *ms);

/*
 * Counts of the last pg_align_launch's anchoring step (synchronises):
 * counts[0] pairs on indel-bearing haplotypes (the ones platypus/calign.pyx:66-77
 * maps by 7-mer votes); counts[1] how many of them the bit-parallel kernel's
 * candidate positions did not decide, so that a wavefront counted the positions
 * their 7-mers name; counts[2] how many it counted over every position (both 0
 * when the position-table kernel ran).  Measurement only; no counterpart in the
 * reference.
 */
int pg_anchor_stats(pg_ctx *ctx, int64_t counts[3]);

/*
 * Genotype log-likelihoods from the likelihoods of the last launch, which
 * stay on the device: out[g] = sum over r in [0,n_reads) of
 * log(0.5*(L[pair_base1[g]+r] + L[pair_base2[g]+r])) (64 partial sums over reads r, r+64, ...
 * combined in a fixed tree: reproducible, equal to the read-order sum to rounding).
 * Replaces DiploidGenotype.calculateDataLikelihood (platypus/cgenotype.pyx:48-63)
 * for n_gen (genotype, individual) combinations; pair_base* are pair indices of
 * the first read of the two haplotypes' likelihood rows.  Relative error vs the
 * reference's libm log <= 1e-9.
 */
int pg_genotype_loglik(pg_ctx *ctx, int64_t n_gen, const int64_t *pair_base1,
                       const int64_t *pair_base2, const int32_t *n_reads, double *out);

/*
 * The same sum for one genotype whose two likelihood rows are on the host
 * (the rows Haplotype.alignReads returned, platypus/cgenotype.pyx:55-56): the
 * rows are copied to the device and summed there.  The per-call form of
 * DiploidGenotype.calculateDataLikelihood; slow, use the batched entry points
 * where the caller can.
 */
int pg_genotype_loglik_rows(pg_ctx *ctx, int n_reads, const double *row1, const double *row2, double *out);

/*
 * ---- The stage that consumes the likelihood matrix: Caller.setup / Caller.call /
 * ---- Population.calculatePosterior of platypus/cpopulation.pyx, batched over windows.
 *
 * pg_windows_upload describes the windows of the uploaded pair list.  Window w
 * has n_haps[w] haplotypes (1..1000, platypus/cpopulation.pyx:546) and
 * n_inds[w] individuals (the `haplotypes` and `readBuffers` arguments of
 * Caller.setup, platypus/cpopulation.pyx:658); its pairs must form one dense,
 * haplotype-major block of the pair list starting at pair like_base[w]:
 * pair(h, r) = like_base[w] + h * n_reads + r.  ind_read_off holds, window after
 * window, n_inds[w] + 1 ascending read offsets inside that block (first 0, last
 * n_reads): individual i owns reads [off[i], off[i+1]) -- the
 * [windowStart, windowEnd) range of its bamReadBuffer (platypus/cpopulation.pyx:702-704).
 * Genotypes are implied: all pairs (a <= b) of the window's haplotypes in the
 * order of generateAllGenotypesFromHaplotypeList (platypus/cgenotype.pyx:90-96),
 * the only generator the reference uses (platypus/variantcaller.pyx:153,178).
 * A window in which no individual has a read is refused (the reference divides
 * by zero there, platypus/cpopulation.pyx:883).
 */
int pg_windows_upload(pg_ctx *ctx, int n_windows, const int32_t *n_haps, const int32_t *n_inds,
                      const int64_t *like_base, const int32_t *ind_read_off);

/*
 * Enqueue, after pg_align_launch on the same stream, for every window:
 * Caller.setup (platypus/cpopulation.pyx:696-734: genotype log-likelihoods with
 * DiploidGenotype.calculateDataLikelihood, platypus/cgenotype.pyx:48-63, one
 * maximum per window, exp(ll - max)) and Caller.call (platypus/cpopulation.pyx:741-805:
 * EM iterations, platypus/cpopulation.pyx:807-890, until the largest frequency
 * change is <= eps or max_iters is reached -- the reference uses 100 -- then the
 * genotype with the largest csr value per individual).  Likelihoods never leave
 * the device.  Sums over genotypes are regrouped, so results agree with the
 * reference's doubles to rounding (tests: 1e-9 relative), not bit for bit.
 */
int pg_caller_launch(pg_ctx *ctx, int max_iters);

/*
 * Results of the last pg_caller_launch (synchronises); any output may be NULL.
 *   freq            concatenated over windows, n_haps[w] each: Population.frequencies
 *   iters           per window: EM iterations run
 *   max_change      per window: last maximum frequency change
 *   best_genotype   concatenated, n_inds[w] each: index of the called genotype in
 *                   the implied order, -1 for an individual without reads
 *                   (genotypeCalls None, platypus/cpopulation.pyx:777-780)
 *   best_value      its csr value (genotypeLikelihoods, platypus/cpopulation.pyx:797)
 *   genotype_post   concatenated, n_inds[w] x n_gen[w] each: the normalised
 *                   genotype likelihoods (genotypePosteriors, platypus/cpopulation.pyx:734)
 */
int pg_caller_fetch(pg_ctx *ctx, double *freq, int32_t *iters, double *max_change,
                    int32_t *best_genotype, double *best_value, double *genotype_post);

/*
 * Population.calculatePosterior (platypus/cpopulation.pyx:515-632) for n_vars
 * variants against the results of the last pg_caller_launch.  var_window[v] is
 * the variant's window; hap_has_var holds, variant after variant, one byte per
 * haplotype of that window (non-zero: the variant is among the haplotype's
 * variants, platypus/cpopulation.pyx:562); prior[v] is var.calculatePrior().
 */
int pg_variant_posteriors(pg_ctx *ctx, int n_vars, const int32_t *var_window,
                          const uint8_t *hap_has_var, const double *prior, double *out);

/*
 * Measured issue rate of the packed 16-bit integer VALU instructions the
 * alignment kernel is built from (v_pk_add_u16 / v_pk_min_i16), in int16
 * operations per second (one packed instruction on one lane = 2 operations).
 * The denominator of the kernel's roofline fraction.  No counterpart in the
 * reference.  Invalidates the results of the last launch.
 */
int pg_valu_int16_peak(pg_ctx *ctx, double *ops_per_second);

/*
 * Measured issue rate of the instruction forms the alignment kernel executes
 * (v_add_u16 / v_min_u16, VGPR operands), in lane-instructions per second
 * (one instruction of one wavefront = 64).  The denominator of the share of
 * the VALU issue rate the kernel reaches.  No counterpart in the reference.
 * Invalidates the results of the last launch.
 */
int pg_valu_vop2_rate(pg_ctx *ctx, double *lane_instructions_per_second);

/* ------------------------------------------------------------------------
 * BGZF inflation on the device.
 *
 * Replaces, for the stretch of a BAM file a region fetch reads, what the
 * reference gets from its vendored samtools: bgzf_read_block / inflate_block
 * (platypus/c/samtools/bgzf.c), reached through bam_read1 from
 * platypus/samtoolsWrapper.pyx (Samfile.fetch, IteratorRow).  One wavefront
 * inflates one BGZF block; the host finds the block boundaries only.
 *
 * An inflater is independent of any pg_ctx (own stream, device buffers and one
 * page-locked output buffer); one thread at a time may use it, several may
 * exist per process and device.
 * ------------------------------------------------------------------------ */
typedef struct pg_inflater pg_inflater;
int pg_inflater_create(int device_ordinal, pg_inflater **out);
int pg_inflater_destroy(pg_inflater *h);
/* message of the last failed call on h (of the last failed create for NULL) */
const char *pg_inflater_error(const pg_inflater *h);
/* 1 (default): one wavefront inflates one BGZF block; 8, 16 or 32: one lane per block, that many blocks per
 * wavefront (the first version of the kernel, kept for comparison) */
int pg_inflater_set_lanes(pg_inflater *h, int lanes);
/* device time of the last call's kernel, milliseconds */
int pg_inflater_last_ms(pg_inflater *h, float *ms);

/*
 * A page-locked buffer of at least `bytes` bytes, owned by the inflater, for
 * the caller to read file bytes into: pg_bgzf_inflate copies from it to the
 * device without the runtime's staging copy (which callers on several threads
 * would contend for).  Grow-only; growing keeps the content; valid until the
 * next call of this function with a larger size, or the inflater's end.
 */
int pg_inflater_input(pg_inflater *h, size_t bytes, void **buf);

/*
 * Host only: the complete BGZF blocks at the start of raw[0, n_raw) -- bytes
 * they occupy, bytes they inflate to (sum of ISIZE), their number (n_blocks
 * may be NULL).  PG_ERR_UNSUPPORTED for bytes that are not a BGZF block
 * (check_header, c/samtools/bgzf.c).
 */
int pg_bgzf_scan(const void *raw, size_t n_raw, size_t *consumed, size_t *total_out, int *n_blocks);

/*
 * Inflates every complete BGZF block at the start of raw[0, n_raw), one after
 * the other, behind a copy of prefix[0, n_prefix) (what the caller had left
 * over from the previous call; it may point into the buffer the previous call
 * returned).  *out is the inflater's own page-locked buffer, valid until the
 * next call on h or its destruction; *n_out = n_prefix + inflated bytes;
 * *consumed = bytes of raw the complete blocks occupy (a trailing incomplete
 * block is left to the caller).  A block whose deflate stream is corrupt or
 * does not produce ISIZE bytes: PG_ERR_UNSUPPORTED (the CRC is not checked,
 * as in the reference's inflate_block).  At most 1 GiB in, 3 GiB out per call.
 */
int pg_bgzf_inflate(pg_inflater *h, const void *raw, size_t n_raw, const void *prefix, size_t n_prefix,
                    void **out, size_t *n_out, size_t *consumed);

#ifdef __cplusplus
}
#endif
#endif /* PLATGPU_H */
