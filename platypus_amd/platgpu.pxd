# C ABI of libplatgpu.so (include/platgpu.h)
from libc.stdint cimport int32_t, int64_t, uint8_t

cdef extern from "platgpu.h":
    ctypedef struct pg_ctx:
        pass
    enum:
        PG_OK
        PG_SCORE_ASSERT_FAILED
    int pg_ctx_create(int device_ordinal, int max_read_len, pg_ctx** out)
    int pg_ctx_destroy(pg_ctx* ctx)
    const char* pg_last_error_string(const pg_ctx* ctx)
    int pg_set_params(pg_ctx* ctx, int gapextend, int nucprior, const char* table, int table_len)
    int pg_set_option(pg_ctx* ctx, const char* name, const char* value)
    int pg_results_fetch_async(pg_ctx* ctx, int32_t* score, int32_t* anchor, double* likelihood)
    int pg_host_alloc(size_t bytes, void** out)
    int pg_host_free(void* p)
    int pg_align_pairs_noalloc(pg_ctx* ctx, int64_t n_pairs, const int32_t* d_hap_idx, const int32_t* d_read_idx,
                               int32_t* d_score, int32_t* d_anchor, double* d_likelihood)
    int pg_haps_upload(pg_ctx* ctx, int n, const char* seq, const int64_t* offsets,
                       const int32_t* start_pos, const int32_t* var_seq_len)
    int pg_haps_build(pg_ctx* ctx, int n_stretches, const char* ref_seq, const int64_t* ref_off, const int32_t* ref_start,
                      int n_haps, const int32_t* hap_stretch, const int32_t* hap_var_off, const int32_t* var_pos,
                      const int32_t* var_n_removed, const int32_t* var_added_off, const char* var_added)
    int pg_haps_fetch(pg_ctx* ctx, char* seq, int64_t* offsets)
    int pg_reads_upload(pg_ctx* ctx, int n, const char* seq, const char* qual, const int64_t* offsets,
                        const int32_t* pos, const int32_t* mapq)
    int pg_reads_upload_nibbles(pg_ctx* ctx, int n, const unsigned char* seq4, const char* qual, const int64_t* offsets,
                                const int32_t* pos, const int32_t* mapq)
    int pg_pairs_upload(pg_ctx* ctx, int64_t n_pairs, const int32_t* hap_idx, const int32_t* read_idx)
    int pg_pairs_block(pg_ctx* ctx, int hap_lo, int hap_hi, int read_lo, int read_hi)
    int pg_align_launch(pg_ctx* ctx)
    int pg_sync(pg_ctx* ctx)
    int pg_results_fetch(pg_ctx* ctx, int32_t* score, int32_t* anchor, double* likelihood)
    int pg_align_window(pg_ctx* ctx, int n_haps, const char* hap_seq, const int64_t* hap_off, const int32_t* hap_start_pos,
                        const int32_t* hap_var_seq_len, int n_reads, const char* read_seq, const char* read_qual,
                        const int64_t* read_off, const int32_t* read_pos, const int32_t* read_mapq,
                        int32_t* score, int32_t* anchor, double* likelihood)
    int pg_genotype_loglik(pg_ctx* ctx, int64_t n_gen, const int64_t* pair_base1, const int64_t* pair_base2,
                           const int32_t* n_reads, double* out)
    int pg_windows_upload(pg_ctx* ctx, int n_windows, const int32_t* n_haps, const int32_t* n_inds,
                          const int64_t* like_base, const int32_t* ind_read_off)
    int pg_caller_launch(pg_ctx* ctx, int max_iters)
    int pg_caller_fetch(pg_ctx* ctx, double* freq, int32_t* iters, double* max_change,
                        int32_t* best_genotype, double* best_value, double* genotype_post)
    int pg_variant_posteriors(pg_ctx* ctx, int n_vars, const int32_t* var_window,
                              const uint8_t* hap_has_var, const double* prior, double* out)
    int pg_genotype_loglik_rows(pg_ctx* ctx, int n_reads, const double* row1, const double* row2, double* out)
    int pg_pairs_blocks(pg_ctx* ctx, int n_blocks, const int32_t* hap_lo, const int32_t* n_haps,
                        const int32_t* read_lo, const int32_t* n_reads, int64_t* pair_base)
    int pg_hash_sequence(pg_ctx* ctx, const char* seq, int len, int32_t* table)
    int pg_map_read(pg_ctx* ctx, const char* read, int read_len, int seq_len, const int32_t* table, int32_t* position)
    int pg_kernel_time_ms(pg_ctx* ctx, int which, float* ms)
    int pg_anchor_stats(pg_ctx* ctx, int64_t* counts)
