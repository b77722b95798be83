// libplatgpu.so -- host side of the C ABI declared in include/platgpu.h.
// One context = one device, one stream, grow-only device buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include <unistd.h>

#include "../../include/platgpu.h"
#include "pg_kernels.hip.h"
#include "pg_align.hip.h"
#include "pg_prep.hip.h"
#include "pg_caller.hip.h"
#include "pg_inflate.hip.h"
#include "pg_band.hip.h"
#include "pg_window.hip.h"

namespace {

thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// ---- fixed device scratch of pg_align_pairs_noalloc -------------------------------
// The work arrays of one pair-list preparation (pg::PairWork) at their largest supported
// size, laid out at fixed offsets of one device array that every GPU the library runs on
// holds until the process ends.  Host code never uses the array's name as an address: its
// address on a device is asked for once (scratch_on_device) and kept per device.
namespace {
constexpr size_t kNaPairs = PG_NOALLOC_MAX_PAIRS;
constexpr size_t kNaVoteHaps = PG_NOALLOC_MAX_INDEL_HAPS;
constexpr size_t kNaLen = 16384 + 2;                               // read lengths 0..16384 (pg_ctx_create's limit) + total
constexpr size_t kNaWaves = kNaPairs / 64 + 16384 + 2;             // wavefront bound of a ragged list
constexpr size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr size_t kNaOffStatus = 0;
constexpr size_t kNaOffHapCnt = kNaOffStatus + up256(pg::ST_WORDS * 4);
constexpr size_t kNaOffHapCur = kNaOffHapCnt + up256((kNaVoteHaps + 1) * 4);
constexpr size_t kNaOffLen = kNaOffHapCur + up256((kNaVoteHaps + 1) * 4);
constexpr size_t kNaOffOrder = kNaOffLen + up256(4 * kNaLen * 4);
constexpr size_t kNaOffWaves = kNaOffOrder + up256(kNaPairs * 4);
constexpr size_t kNaOffRecs = kNaOffWaves + up256(kNaWaves * sizeof(pg::WaveDesc));
constexpr size_t kNaOffAnchor = kNaOffRecs + up256(kNaPairs * sizeof(pg::VoteRec));
constexpr size_t kNaBytes = kNaOffAnchor + up256(kNaPairs * 4);
constexpr int kMaxDevices = 64;
constexpr int64_t kBandMaxPairs = 16384;                            // launches up to this size take k_align_band
}  // namespace

static __device__ __align__(256) unsigned char scratch[kNaBytes];

namespace {
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
std::atomic<int> g_scratch_busy[kMaxDevices];      // 1 while a call is using the device's array

struct ScratchHold {                                // the device's array for the length of a call; `got` says whether it was free
    int device; bool got;
    explicit ScratchHold(int d) : device(d), got(g_scratch_busy[d].exchange(1) == 0) {}
    ~ScratchHold() { if (got) g_scratch_busy[device].store(0); }
};

// Device address of `scratch` on `device`, which must be the current device.  Every GPU holds its
// own copy of the array at its own address.
hipError_t scratch_on_device(int device, unsigned char** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct pg_ctx {
    int device = 0;
    unsigned char* scratch_ptr = nullptr;   // device address of the fixed scratch array on this context's GPU
    int max_read_len = 512;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2 * PG_K_COUNT] = {};
    hipEvent_t ev_block[2] = {};         // EM launch-pair path: one per block of iterations in flight
    int32_t* h_active = nullptr;         // pinned: windows still running after each block
    bool ev_valid[PG_K_COUNT] = {};
    std::string err;

    // parameters (platypus/chaplotype.pyx:264)
    int gapextend = 3, nucprior = 2;
    std::string table;
    DevBuf<uint8_t> d_tab1;              // table[h] - 33
    DevBuf<pg::SeqCodes> d_codes;        // code tables of the sequence bytes (pg_align.hip.h)
    // kernel-choice overrides of pg_set_option (tests and measurements); 0 / -1 = the library decides
    int vote_waves = 0;                  // "vote_waves": wavefronts per workgroup of the vote kernel
    int align_kernel = -1;               // "align_kernel": 0 one lane per pair (k_align2), 1 eight lanes per pair (k_align_band)
    int vote5_lds = 0;                   // dynamic LDS k_vote5 has been allowed so far
    int vote_kernel = -1;                // "vote_kernel": 0 position table in LDS (k_vote4), 1 bit-parallel counts (k_vote5)
    int lds_pad = 0;                     // "align_lds_pad": extra LDS per workgroup of the DP kernel, lowers its occupancy
    int gl_kernel = -1;                  // "gl_kernel": 0 cell, 1 tile
    int em_kernel = -1;                  // "em_kernel": 0 small, 1 big, 2 multi
    int64_t em_stage_max = 16384;        // "em_stage_max": doubles of a window the big EM kernel may keep in LDS
    int em_step = -1;                    // "em_step": 0 full matrix, 1 upper triangle (launch-pair EM path)
    int em_small_exact = -1;             // "em_small": 0 regrouped sums (k_em_small), 1 / default: the reference's order (k_em_exact)
    int em_step_waves = 0;               // "em_step_waves": wavefronts per workgroup of the triangle step (1..8)
    int32_t* h_status = nullptr;         // pinned: status words of the last pair-list preparation
    DevBuf<double> d_exp_score, d_exp_mapq;

    // haplotypes
    int n_haps = 0;
    int max_hap_len = 0;
    int n_vote_haps = 0;                 // indel-bearing haplotypes of the upload (var_seq_len != 0)
    DevBuf<int32_t> d_hap_vote_ord;      // per haplotype: its number among those, -1 for the others
    DevBuf<uint8_t> d_hap_raw;
    DevBuf<uint16_t> d_hapx;
    DevBuf<uint32_t> d_hap2;            // 2-bit packed bases
    DevBuf<uint32_t> d_hapU;            // indel-bearing haplotypes: starts that own a unique 7-mer, same layout
    DevBuf<uint32_t> d_hapUbits;        // ... and one bit per 7-mer value: owned by a unique start (512 words each)
    DevBuf<int64_t> d_hap2off;
    DevBuf<int64_t> d_hap_soff, d_hap_doff;
    DevBuf<int32_t> d_hap_len, d_hap_start, d_hap_varlen;
    int64_t hap_bytes = 0;               // bytes of all sequences in d_hap_raw
    DevBuf<uint8_t> d_build_ref, d_build_added;   // pg_haps_build: reference stretches, added sequences,
    DevBuf<pg::HapPiece> d_build_pieces;          // the pieces every haplotype is put together from
    DevBuf<int32_t> d_build_off;

    // reads
    int n_reads = 0;
    int max_rlen = 0;
    bool uniform_rlen = true;
    DevBuf<int32_t> d_read_len;
    DevBuf<uint8_t> d_rseq, d_rqual;
    DevBuf<uint8_t> d_rseq4;            // pg_reads_upload_nibbles: the bases as they came, two per byte
    DevBuf<uint16_t> d_readx;
    DevBuf<int64_t> d_roff, d_read_xoff, d_read2off, d_off_in, d_scan_tops;
    DevBuf<uint32_t> d_read2;           // 2-bit packed bases
    DevBuf<int32_t> d_read_pos, d_read_mapq, d_flag;

    // pairs
    int64_t n_pairs = 0;
    DevBuf<int32_t> d_pair_hap, d_pair_read, d_order, d_anchor_in;
    DevBuf<int32_t> d_status, d_hap_cnt, d_hap_cur, d_len_work;
    DevBuf<pg::VoteHap> d_vh;            // one per indel-bearing haplotype, made by pg_haps_upload
    DevBuf<pg::WaveDesc> d_waves;
    DevBuf<pg::VoteRec> d_vh_recs;
    DevBuf<int32_t> d_blk32;             // block descriptors of pg_pairs_blocks
    DevBuf<int64_t> d_blk_base;
    pg::PairWork work;                   // the arrays above, as the kernels take them
    std::vector<int64_t> win_tmp;        // pg_align_window's host scratch
    std::vector<int32_t> win_tmp32;

    // one window per call (pg_align_window): the block the host fills and its copy on the device
    uint8_t* h_win = nullptr;
    size_t h_win_cap = 0;
    DevBuf<uint8_t> d_win;

    // results: where the last launch put them (the arrays below, or pg_align_window's block) and how many pairs they are of
    bool launched = false;
    int64_t n_results = 0;
    const int32_t* r_score = nullptr;
    const int32_t* r_anchor = nullptr;
    const double* r_like = nullptr;
    const int32_t* r_status = nullptr;   // status words of that launch's pair list
    DevBuf<int32_t> d_score, d_anchor;
    DevBuf<double> d_like;

    // windows of the genotype / EM stage
    int n_windows = 0, max_win_haps = 0, max_win_inds = 0;
    int64_t max_win_full = 0;             // largest n_ind * n_hap * ld of a window
    int64_t win_cells = 0, win_full = 0, win_haps_total = 0, win_inds_total = 0;
    bool caller_done = false;
    std::vector<pg::WinDesc> h_wins;
    DevBuf<pg::WinDesc> d_wins;
    DevBuf<int32_t> d_ioff, d_em_iters, d_best_gen, d_var_win;
    DevBuf<double> d_cells, d_full, d_freq, d_em_change, d_best_val, d_var_prior, d_var_out;
    DevBuf<int64_t> d_var_moff;
    DevBuf<unsigned long long> d_win_max;
    DevBuf<double> d_fprev, d_share;
    DevBuf<int32_t> d_em_done;
    DevBuf<int64_t> d_share_base;
    int64_t win_share = 0;
    DevBuf<uint8_t> d_var_mask;

    // genotype scratch
    DevBuf<int64_t> d_g1, d_g2;
    DevBuf<int32_t> d_gn;
    DevBuf<double> d_gout, d_rows;
    DevBuf<uint8_t> d_cmp_bytes;         // stand-alone hash_sequence / map_read
    DevBuf<int32_t> d_cmp_ints;
};

namespace {

int fail(pg_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define PG_HIP(c, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail((c), PG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

const char kDefaultTable[] = "NKJHFA=854210/.-,,+**))((('''&&&%%%$$$$#####\"\"\"\"\"";  // chaplotype.pyx:37-40

int upload_params(pg_ctx* c) {
    uint8_t tab1[256];                            // gap-open cost per run length, align.c:85 without its factor 4
    const int n = (int)c->table.size();
    for (int i = 0; i < 256; ++i) tab1[i] = (uint8_t)((unsigned char)c->table[std::min(i, n - 1)] - '!');
    PG_HIP(c, c->d_tab1.reserve(256));
    PG_HIP(c, hipMemcpyAsync(c->d_tab1.p, tab1, sizeof tab1, hipMemcpyHostToDevice, c->stream));
    // code tables of pg_align.hip.h: the 126 nine-bit words with four bits set, in ascending order, for the
    // bytes 33..126 and 1..31 (a NUL would end the reference's strlen, calign.pyx:134); 'N' columns beyond the first eight of a slice match everything at cost 1
    static pg::SeqCodes codes;
    {
        std::memset(&codes, 0, sizeof codes);
        int next = 0;
        uint16_t pattern[126];
        for (uint32_t w = 0; w < 512 && next < 126; ++w)
            if (__builtin_popcount(w) == 4) pattern[next++] = (uint16_t)w;
        for (int b = 0; b < 256; ++b) {
            const int idx = b >= 33 && b <= 126 ? b - 33 : (b >= 1 && b < 32 ? 94 + b : -1);
            if (idx < 0) continue;
            const uint16_t pat = (uint16_t)(pattern[idx] << 7);
            codes.read[b] = (uint16_t)(pat | (b != 'N'));
            codes.hap0[b] = (uint16_t)(~pat & 0xff80u);
            codes.hap[b] = b == 'N' ? (uint16_t)1 : codes.hap0[b];
        }
    }
    PG_HIP(c, c->d_codes.reserve(1));
    PG_HIP(c, hipMemcpyAsync(c->d_codes.p, &codes, sizeof codes, hipMemcpyHostToDevice, c->stream));
    // likelihood factors with the host C library, so the device result is the
    // same double the reference computes (calign.pyx:17,55,63)
    const double mLTOT = -0.23025850929940459;
    std::vector<double> es(2049), em(256);
    for (int s = 0; s <= 2048; ++s) es[s] = std::exp(mLTOT * s);
    for (int q = 0; q < 256; ++q) em[q] = std::exp(mLTOT * q);
    PG_HIP(c, c->d_exp_score.reserve(es.size()));
    PG_HIP(c, c->d_exp_mapq.reserve(em.size()));
    PG_HIP(c, hipMemcpyAsync(c->d_exp_score.p, es.data(), es.size() * 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_exp_mapq.p, em.data(), em.size() * 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    return PG_OK;
}


// ---- pair lists ---------------------------------------------------------------
// Everything the kernels need beyond the pair indices (vote records grouped by
// indel-bearing haplotype; for ragged batches the slot order by read length and
// the wavefront descriptors) is made on the device by the kernels of
// pg_prep.hip.h, in the arrays of a pg::PairWork.

unsigned prep_grid(int64_t n_pairs) { return (unsigned)std::min<int64_t>((n_pairs + 255) / 256, 8192); }

// wavefronts of the alignment kernel: exact for a uniform batch, an upper bound for a ragged one
// (every read length present may end with a partly filled wavefront)
int wave_bound(bool uniform, int max_rlen, int64_t np) {
    const int64_t full = (np + 63) / 64;
    return (int)(uniform ? full : full + std::min<int64_t>(np, max_rlen));
}
int wave_bound(const pg_ctx* c, int64_t np) { return wave_bound(c->uniform_rlen, c->max_rlen, np); }

// what the anchoring and alignment kernels read of a haplotype batch and a read batch
struct AlignBufs {
    const uint16_t* hapx; const int64_t* hap_doff; const int32_t* hap_len; const int32_t* hap_start; const int32_t* hap_varlen;
    const uint32_t* hap2; const uint32_t* hapU; const uint32_t* hapUbits; const pg::VoteHap* vh;
    const uint16_t* readx; const int64_t* read_xoff; const uint32_t* read2; const int32_t* read_pos; const int32_t* read_mapq;
    const int32_t* read_len;
    int n_vote_haps, max_rlen, max_hap_len;
    bool uniform;
    int n_waves;          // wavefronts of the alignment kernel (ragged: an upper bound, the count is in the status words)
    bool timed;           // record the context's events around the kernels and clear the anchoring counters first
};

AlignBufs ctx_bufs(const pg_ctx* c, int64_t np) {
    AlignBufs B;
    B.hapx = c->d_hapx.p; B.hap_doff = c->d_hap_doff.p; B.hap_len = c->d_hap_len.p; B.hap_start = c->d_hap_start.p;
    B.hap_varlen = c->d_hap_varlen.p; B.hap2 = c->d_hap2.p; B.hapU = c->d_hapU.p; B.hapUbits = c->d_hapUbits.p; B.vh = c->d_vh.p;
    B.readx = c->d_readx.p; B.read_xoff = c->d_read_xoff.p; B.read2 = c->d_read2.p; B.read_pos = c->d_read_pos.p;
    B.read_mapq = c->d_read_mapq.p; B.read_len = c->d_read_len.p;
    B.n_vote_haps = c->n_vote_haps; B.max_rlen = c->max_rlen; B.max_hap_len = c->max_hap_len; B.uniform = c->uniform_rlen;
    B.n_waves = wave_bound(c, np);
    B.timed = true;
    return B;
}

// Enqueue the preparation of `np` pairs (device arrays pair_hap / pair_read) into W.
int enqueue_prep(pg_ctx* c, const pg::PairWork& W, const int32_t* pair_hap, const int32_t* pair_read, int64_t np) {
    pg::PrepArgs A;
    A.pair_hap = pair_hap; A.pair_read = pair_read; A.n_pairs = np;
    A.n_haps = c->n_haps; A.n_reads = c->n_reads; A.n_vote_haps = c->n_vote_haps; A.max_len = c->max_rlen;
    A.ragged = !c->uniform_rlen;
    A.hap_vote_ord = c->d_hap_vote_ord.p; A.read_len = c->d_read_len.p; A.read2off = c->d_read2off.p;
    A.W = W;
    hipLaunchKernelGGL(pg::k_prep_init, dim3(1), dim3(64), 0, c->stream, W.status);
    PG_HIP(c, hipMemsetAsync(W.hap_cnt, 0, ((size_t)c->n_vote_haps + 1) * sizeof(int32_t), c->stream));
    if (A.ragged) PG_HIP(c, hipMemsetAsync(W.len_cnt, 0, ((size_t)c->max_rlen + 2) * sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(pg::k_prep_count, dim3(prep_grid(np)), dim3(256), 0, c->stream, A);
    hipLaunchKernelGGL(pg::k_prep_scan, dim3(1), dim3(1024), 0, c->stream, A);
    hipLaunchKernelGGL(pg::k_prep_scatter, dim3(prep_grid(np)), dim3(256), 0, c->stream, A);
    if (A.ragged) {
        const int wb = wave_bound(c, np);
        hipLaunchKernelGGL(pg::k_prep_waves, dim3((wb + 255) / 256), dim3(256), 0, c->stream, A, wb);
    }
    PG_HIP(c, hipGetLastError());
    return PG_OK;
}

// The pair list (np pairs) is in c->d_pair_hap / d_pair_read, or on its way there on the stream:
// size the context's work arrays, prepare, wait, and refuse a list with an index out of range.
int finish_pairs(pg_ctx* c, int64_t np) {
    const size_t L = (size_t)c->max_rlen + 2;
    const int wb = wave_bound(c, np);
    PG_HIP(c, c->d_status.reserve(pg::ST_WORDS));
    PG_HIP(c, c->d_hap_cnt.reserve((size_t)c->n_vote_haps + 1));
    PG_HIP(c, c->d_hap_cur.reserve((size_t)c->n_vote_haps + 1));
    PG_HIP(c, c->d_vh_recs.reserve(np));
    PG_HIP(c, c->d_anchor_in.reserve(np));
    PG_HIP(c, c->d_score.reserve(np));
    PG_HIP(c, c->d_anchor.reserve(np));
    PG_HIP(c, c->d_like.reserve(np));
    if (!c->uniform_rlen) {
        PG_HIP(c, c->d_len_work.reserve(4 * L));
        PG_HIP(c, c->d_order.reserve(np));
        PG_HIP(c, c->d_waves.reserve((size_t)wb + 1));
    }
    pg::PairWork& W = c->work;
    W.status = c->d_status.p; W.hap_cnt = c->d_hap_cnt.p; W.hap_cur = c->d_hap_cur.p;
    W.len_cnt = c->d_len_work.p; W.len_slot = c->d_len_work.p + L; W.len_wave = c->d_len_work.p + 2 * L;
    W.len_cur = c->d_len_work.p + 3 * L;
    W.order = c->uniform_rlen ? nullptr : c->d_order.p;
    W.waves = c->uniform_rlen ? nullptr : c->d_waves.p;
    W.recs = c->d_vh_recs.p; W.anchor_in = c->d_anchor_in.p;
    int rc = enqueue_prep(c, W, c->d_pair_hap.p, c->d_pair_read.p, np);
    if (rc != PG_OK) return rc;
    PG_HIP(c, hipMemcpyAsync(c->h_status, W.status, pg::ST_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    if (c->h_status[pg::ST_BAD_PAIR] != 0x7fffffff)
        return fail(c, PG_ERR_INVALID, "pair %d refers to a haplotype or read outside the uploaded batches",
                    c->h_status[pg::ST_BAD_PAIR]);
    c->n_pairs = np;
    return PG_OK;
}

// Enqueue anchoring + alignment of a prepared pair list; results go to score / anchor / like (device).
int enqueue_align(pg_ctx* c, const AlignBufs& U, const pg::PairWork& W, const int32_t* pair_hap, const int32_t* pair_read, int64_t np,
                  int32_t* score, int32_t* anchor, double* like) {
    // anchoring of pairs on indel-bearing haplotypes
    if (U.timed) PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_ANCHOR], c->stream));
    // reads of up to 263 bases (sixteen 16-base groups) take the table-free kernel
    const bool bits = c->vote_kernel >= 0 ? (c->vote_kernel == 1 && U.max_rlen <= 263) : U.max_rlen <= 263;
    if (U.n_vote_haps > 0 && bits) {
        if (U.timed) PG_HIP(c, hipMemsetAsync(W.status + pg::ST_N_NAMED, 0, 2 * sizeof(int32_t), c->stream));   // counted per launch
        const int pad = (U.max_rlen + 15) / 16 + 3;
        const int row_cap = (U.max_hap_len + 15) / 16 + 2 + 2 * pad;
        // one lane per pair: a workgroup takes the next 64 * n_wav vote records, whatever haplotypes they are on
        const int n_wav = c->vote_waves > 0 ? std::min(c->vote_waves, 4) : 2;
        const size_t lds = ((size_t)pg::kVoteSlots * (2 * (size_t)row_cap + pg::kCandSlots / 2) + (size_t)n_wav * 20) * 4;
        if (lds > 160 * 1024)
            return fail(c, PG_ERR_UNSUPPORTED, "anchoring rows for haplotype length %d need %zu bytes of LDS", U.max_hap_len, lds);
        if ((int)lds > c->vote5_lds) {
            PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_vote5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            c->vote5_lds = (int)lds;
        }
        const int64_t grid = (np + 64 * n_wav - 1) / (64 * n_wav);          // records <= pairs; workgroups past the last record return
        hipLaunchKernelGGL(pg::k_vote5, dim3((unsigned)grid), dim3(64 * n_wav), lds, c->stream, U.hap2, U.hapU,
                           U.hapUbits, U.read2, U.vh, W.hap_cnt, W.recs, W.status, W.anchor_in, pad, row_cap,
                           U.n_vote_haps);
        PG_HIP(c, hipGetLastError());
    } else if (U.n_vote_haps > 0) {
        int cnt_stride = U.max_hap_len + U.max_rlen + 8;
        cnt_stride += cnt_stride & 1;
        const int row_words = (U.max_rlen + 15) / 16 + 2;
        // four lanes per read, four wavefronts per haplotype
        int n_wav = c->vote_waves > 0 ? std::min(c->vote_waves, 4) : 4;
        auto lds_bytes = [&](int w) { return (size_t)pg::kHashSize * 2 + (size_t)w * (cnt_stride * 2 + row_words * 4); };
        while (n_wav > 1 && lds_bytes(n_wav) > 160 * 1024) n_wav >>= 1;
        const size_t lds = lds_bytes(n_wav);
        if (lds > 160 * 1024)
            return fail(c, PG_ERR_UNSUPPORTED, "vote counters for haplotype length %d need %zu bytes of LDS", U.max_hap_len, lds);
        PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_vote4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(pg::k_vote4, dim3(U.n_vote_haps), dim3(64 * n_wav), lds, c->stream, U.hap2,
                           U.read2, U.vh, W.hap_cnt, W.recs, W.status, W.anchor_in, cnt_stride, row_words);
        PG_HIP(c, hipGetLastError());
    }
    if (U.timed) {
        PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_ANCHOR + 1], c->stream));
        c->ev_valid[PG_K_ANCHOR] = true;
    }

    const int n_waves = U.n_waves;
    pg::Align2Args B;
    B.hapx = U.hapx; B.hap_xoff = U.hap_doff; B.hap_len = U.hap_len;
    B.hap_start = U.hap_start; B.hap_varlen = U.hap_varlen;
    B.readx = U.readx; B.read_xoff = U.read_xoff;
    B.read_pos = U.read_pos; B.read_mapq = U.read_mapq;
    B.pair_hap = pair_hap; B.pair_read = pair_read;
    B.order = W.order; B.waves = W.waves; B.status = W.status;
    B.anchor_in = W.anchor_in; B.tab = c->d_tab1.p; B.codes = c->d_codes.p;
    B.exp_score = c->d_exp_score.p; B.exp_mapq = c->d_exp_mapq.p;
    B.score = score; B.anchor = anchor; B.like = like;
    B.n_waves = n_waves; B.uniform_len = U.max_rlen; B.n_pairs = (int)np;
    B.ge = c->gapextend; B.np = c->nucprior;
    if (c->lds_pad > 0)
        PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_align2), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_pad));
    if (U.timed) PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_ALIGN], c->stream));
    // A launch of few pairs goes to the kernel with eight lanes per pair: its wavefronts finish in a fifth of the time
    // a lone wavefront of k_align2 takes, and there are eight times as many of them to spread over the CUs; from some
    // tens of thousands of pairs on k_align2's wavefronts fill the machine and do the same work in half the instructions.
    const bool band = c->align_kernel >= 0 ? c->align_kernel == 1 : np <= kBandMaxPairs;
    if (band) {
        hipLaunchKernelGGL(pg::k_align_band, dim3((unsigned)((np + 7) / 8)), dim3(64), 0, c->stream, B, U.read_len);
    } else {
        hipLaunchKernelGGL(pg::k_align2, dim3((n_waves + 3) / 4), dim3(256), c->lds_pad, c->stream, B);
    }
    PG_HIP(c, hipGetLastError());
    if (U.timed) {
        PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_ALIGN + 1], c->stream));
        c->ev_valid[PG_K_ALIGN] = true;
    }
    return PG_OK;
}

}  // namespace

extern "C" {

int pg_abi_version(void) { return PG_ABI_VERSION; }

const char* pg_last_error_string(const pg_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int pg_device_count(int* count) {
    if (!count) return PG_ERR_INVALID;
    *count = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return PG_OK;        // no driver, no device: zero
    for (int d = 0; d < ndev; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++*count;
    }
    return PG_OK;
}

int pg_ctx_create(int device_ordinal, int max_read_len, pg_ctx** out) {
    if (!out) return fail(nullptr, PG_ERR_INVALID, "pg_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PG_ERR_NO_DEVICE, "no HIP device visible");
    if (device_ordinal < 0 || device_ordinal >= ndev)
        return fail(nullptr, PG_ERR_INVALID, "device ordinal %d out of range (%d devices)", device_ordinal, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess)
        return fail(nullptr, PG_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PG_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                    device_ordinal, prop.gcnArchName);
    if (max_read_len == 0) max_read_len = 512;
    if (max_read_len < 1 || max_read_len > 16384)
        return fail(nullptr, PG_ERR_INVALID, "max_read_len %d outside 1..16384", max_read_len);
    pg_ctx* c = new pg_ctx;
    c->device = device_ordinal;
    c->max_read_len = max_read_len;
    c->table = kDefaultTable;
    // the context's stream is a high-priority one: it gets a hardware queue that the (low-priority) streams of the
    // BGZF inflaters do not share, so its short kernels never queue behind their long ones
    int prio_low = 0, prio_high = 0;
    if (hipSetDevice(device_ordinal) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_high) != hipSuccess) {
        delete c;
        return fail(nullptr, PG_ERR_HIP, "cannot create a stream on device %d", device_ordinal);
    }
    {
        hipError_t e = scratch_on_device(device_ordinal, &c->scratch_ptr);     // the device is current here
        if (e != hipSuccess) {
            delete c;
            return fail(nullptr, PG_ERR_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device_ordinal, hipGetErrorString(e));
        }
    }
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete c; return fail(nullptr, PG_ERR_HIP, "hipEventCreate failed"); }
    for (auto& e : c->ev_block)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete c; return fail(nullptr, PG_ERR_HIP, "hipEventCreate failed"); }
    if (hipHostMalloc((void**)&c->h_status, pg::ST_WORDS * sizeof(int32_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_active, 2 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
        delete c;
        return fail(nullptr, PG_ERR_HIP, "cannot allocate pinned status words");
    }
    int rc = upload_params(c);
    if (rc != PG_OK) { g_create_error = c->err; delete c; return rc; }
    *out = c;
    return PG_OK;
}

int pg_ctx_destroy(pg_ctx* c) {
    if (!c) return PG_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->d_tab1.release(); c->d_codes.release(); c->d_hapx.release();
    c->d_readx.release(); c->d_read_xoff.release(); c->d_hap2.release(); c->d_hapU.release(); c->d_hapUbits.release(); c->d_hap2off.release();
    c->d_read2.release(); c->d_read2off.release();
    c->d_exp_score.release(); c->d_exp_mapq.release();
    c->d_hap_raw.release(); c->d_build_ref.release(); c->d_build_added.release(); c->d_build_pieces.release(); c->d_build_off.release();
    c->d_hap_soff.release(); c->d_hap_doff.release();
    c->d_hap_len.release(); c->d_hap_start.release(); c->d_hap_varlen.release();
    c->d_rseq.release(); c->d_rseq4.release(); c->d_rqual.release(); c->d_roff.release();
    c->d_read_pos.release(); c->d_read_mapq.release(); c->d_flag.release();
    c->d_pair_hap.release(); c->d_pair_read.release(); c->d_order.release();
    c->d_vh.release(); c->d_anchor_in.release(); c->d_hap_vote_ord.release(); c->d_read_len.release();
    c->d_status.release(); c->d_hap_cnt.release(); c->d_hap_cur.release(); c->d_len_work.release();
    c->d_blk32.release(); c->d_blk_base.release(); c->d_off_in.release(); c->d_scan_tops.release();
    c->d_wins.release(); c->d_ioff.release(); c->d_em_iters.release(); c->d_best_gen.release();
    c->d_var_win.release(); c->d_cells.release(); c->d_full.release(); c->d_freq.release(); c->d_em_change.release();
    c->d_best_val.release(); c->d_var_prior.release(); c->d_var_out.release(); c->d_var_moff.release(); c->d_win_max.release(); c->d_fprev.release(); c->d_share.release(); c->d_em_done.release(); c->d_share_base.release(); c->d_rows.release(); c->d_cmp_bytes.release(); c->d_cmp_ints.release(); c->d_var_mask.release(); c->d_vh_recs.release(); c->d_waves.release(); c->d_score.release(); c->d_anchor.release(); c->d_like.release();
    c->d_g1.release(); c->d_g2.release(); c->d_gn.release(); c->d_gout.release();
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_block) if (e) (void)hipEventDestroy(e);
    c->d_win.release();
    if (c->h_win) (void)hipHostFree(c->h_win);
    if (c->h_active) (void)hipHostFree(c->h_active);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PG_OK;
}

int pg_host_alloc(size_t bytes, void** out) {
    if (!out) return PG_ERR_INVALID;
    *out = nullptr;
    if (bytes == 0) return PG_OK;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(nullptr, PG_ERR_HIP, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    return PG_OK;
}

int pg_host_free(void* p) {
    if (!p) return PG_OK;
    return hipHostFree(p) == hipSuccess ? PG_OK : fail(nullptr, PG_ERR_HIP, "hipHostFree failed");
}

int pg_set_params(pg_ctx* c, int gapextend, int nucprior, const char* table, int table_len) {
    if (!c) return PG_ERR_INVALID;
    if (!table || table_len < 1 || table_len > pg::kMaxRun + 1)
        return fail(c, PG_ERR_INVALID, "gap-open table must hold 1..%d characters", pg::kMaxRun + 1);
    if (gapextend < 0 || gapextend > 64 || nucprior < 0 || nucprior > 64)
        return fail(c, PG_ERR_INVALID, "gapextend/nucprior outside 0..64");
    for (int i = 0; i < table_len; ++i)
        if ((unsigned char)table[i] < '!' || (unsigned char)table[i] > '~')
            return fail(c, PG_ERR_INVALID, "gap-open table character %d is not phred+33", i);
    PG_HIP(c, hipSetDevice(c->device));
    c->gapextend = gapextend;
    c->nucprior = nucprior;
    c->table.assign(table, table + table_len);
    c->n_haps = 0;           // run-length cap depends on the table: haplotypes must be uploaded again
    c->n_pairs = 0;
    c->launched = false;
    return upload_params(c);
}

int pg_set_option(pg_ctx* c, const char* name, const char* value) {
    if (!c) return PG_ERR_INVALID;
    if (!name || !value) return fail(c, PG_ERR_INVALID, "pg_set_option: NULL argument");
    const std::string n = name, v = value;
    const bool dflt = v.empty() || v == "default";
    if (n == "em_kernel") {
        if (dflt) c->em_kernel = -1; else if (v == "small") c->em_kernel = 0; else if (v == "big") c->em_kernel = 1;
        else if (v == "multi") c->em_kernel = 2; else return fail(c, PG_ERR_INVALID, "em_kernel: small, big, multi or default");
    } else if (n == "gl_kernel") {
        if (dflt) c->gl_kernel = -1; else if (v == "cell") c->gl_kernel = 0; else if (v == "tile") c->gl_kernel = 1;
        else return fail(c, PG_ERR_INVALID, "gl_kernel: cell, tile or default");
    } else if (n == "em_stage_max") {
        c->em_stage_max = dflt ? 16384 : std::max<int64_t>(0, std::atoll(value));
    } else if (n == "em_step") {
        if (dflt) c->em_step = -1; else if (v == "full") c->em_step = 0; else if (v == "sym") c->em_step = 1;
        else return fail(c, PG_ERR_INVALID, "em_step: full, sym or default");
    } else if (n == "em_small") {
        if (dflt) c->em_small_exact = -1; else if (v == "fast") c->em_small_exact = 0; else if (v == "exact") c->em_small_exact = 1;
        else return fail(c, PG_ERR_INVALID, "em_small: fast, exact or default");
    } else if (n == "em_step_waves") {
        c->em_step_waves = dflt ? 0 : std::max(1, std::min(16, std::atoi(value)));
    } else if (n == "vote_waves") {
        c->vote_waves = dflt ? 0 : std::max(1, std::min(4, std::atoi(value)));
    } else if (n == "vote_kernel") {
        if (dflt) c->vote_kernel = -1; else if (v == "table") c->vote_kernel = 0; else if (v == "bits") c->vote_kernel = 1;
        else return fail(c, PG_ERR_INVALID, "vote_kernel: table, bits or default");
    } else if (n == "align_kernel") {
        if (dflt) c->align_kernel = -1; else if (v == "lane") c->align_kernel = 0; else if (v == "band") c->align_kernel = 1;
        else return fail(c, PG_ERR_INVALID, "align_kernel: lane, band or default");
    } else if (n == "align_lds_pad") {
        c->lds_pad = dflt ? 0 : std::max(0, std::min(159 * 1024, std::atoi(value)));
    } else {
        return fail(c, PG_ERR_INVALID, "pg_set_option: unknown option '%s'", name);
    }
    return PG_OK;
}

namespace {

// The haplotype batch of a context from `n` sequences whose bytes `fill` puts on the device at d_hap_raw (sequence i at
// soff[i] .. soff[i + 1], soff[0] = 0): the element rows with their run lengths (k_pack_hapx: the gap-open table build of
// platypus/c/align.c:70-96, once per haplotype), the packed bases, and for the indel-bearing ones the rows of unique
// 7-mer starts.
int haps_install(pg_ctx* c, int n, const std::vector<int64_t>& soff, const int32_t* start_pos, const int32_t* var_seq_len,
                 const std::function<int()>& fill) {
    c->n_haps = 0; c->n_pairs = 0; c->launched = false;
    if (n == 0) return PG_OK;
    const int pad = c->max_read_len + 64;
    std::vector<int64_t> doff(n + 1);
    std::vector<int32_t> len(n);
    int maxlen = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t l = soff[i + 1] - soff[i];
        if (l < 0 || l > (1 << 30)) return fail(c, PG_ERR_INVALID, "haplotype %d has length %lld", i, (long long)l);
        if (var_seq_len[i] != 0 && l > PG_MAX_HASHED_HAP_LEN)
            return fail(c, PG_ERR_UNSUPPORTED, "haplotype %d with indels is %lld long; the 7-mer anchoring takes < 16384", i, (long long)l);
        // a shorter one makes the reference read past the 'N' extension it appends (calign.pyx:73-77)
        if (var_seq_len[i] != 0 && l < 15)
            return fail(c, PG_ERR_UNSUPPORTED, "haplotype %d with indels is only %lld long; at least 15 bases are needed", i, (long long)l);
        len[i] = (int32_t)l;
        maxlen = std::max(maxlen, (int)l);
        doff[i] = soff[i] + (int64_t)i * pad;
    }
    const int64_t total = soff[n];
    doff[n] = total + (int64_t)n * pad;
    PG_HIP(c, c->d_hap_raw.reserve((size_t)total + 16));
    PG_HIP(c, c->d_hapx.reserve((size_t)doff[n] + 64));
    PG_HIP(c, c->d_hap_soff.reserve(n + 1));
    PG_HIP(c, c->d_hap_doff.reserve(n + 1));
    PG_HIP(c, c->d_hap_len.reserve(n));
    PG_HIP(c, c->d_hap_start.reserve(n));
    PG_HIP(c, c->d_hap_varlen.reserve(n));
    PG_HIP(c, hipMemcpyAsync(c->d_hap_soff.p, soff.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_hap_doff.p, doff.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_hap_len.p, len.data(), n * 4, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_hap_start.p, start_pos, n * 4, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_hap_varlen.p, var_seq_len, n * 4, hipMemcpyHostToDevice, c->stream));
    {
        const int rc = fill();
        if (rc != PG_OK) return rc;
    }
    const int cap = (int)c->table.size() - 1;
    const int grid = std::min(n, 65536);
    PG_HIP(c, c->d_flag.reserve(64));
    PG_HIP(c, hipMemsetAsync(c->d_flag.p, 0, 64 * sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(pg::k_pack_hapx, dim3(grid), dim3(256), 0, c->stream, c->d_hap_raw.p, c->d_hap_soff.p,
                       c->d_hap_doff.p, c->d_hapx.p, c->d_codes.p, c->d_flag.p, n, pad, cap);
    PG_HIP(c, hipGetLastError());
    std::vector<int64_t> off2(n + 1);
    off2[0] = 0;
    for (int i = 0; i < n; ++i) off2[i + 1] = off2[i] + (len[i] + 15) / 16 + 2;
    PG_HIP(c, c->d_hap2.reserve((size_t)off2[n] + 16));
    PG_HIP(c, c->d_hap2off.reserve(n + 1));
    PG_HIP(c, hipMemcpyAsync(c->d_hap2off.p, off2.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pg::k_pack_codes, dim3(std::min((n + 3) / 4, 65536)), dim3(256), 0, c->stream, c->d_hap_raw.p,
                       c->d_hap_soff.p, c->d_hap2off.p, c->d_hap2.p, n);
    PG_HIP(c, hipGetLastError());
    // the indel-bearing haplotypes: one workgroup of the vote kernel each
    std::vector<int32_t> ord(n);
    std::vector<pg::VoteHap> vh;
    for (int i = 0; i < n; ++i) {
        ord[i] = var_seq_len[i] != 0 ? (int32_t)vh.size() : -1;
        if (var_seq_len[i] != 0) vh.push_back({i, len[i], off2[i]});
    }
    PG_HIP(c, c->d_hap_vote_ord.reserve(n));
    PG_HIP(c, c->d_vh.reserve(vh.size() + 1));
    PG_HIP(c, hipMemcpyAsync(c->d_hap_vote_ord.p, ord.data(), n * 4, hipMemcpyHostToDevice, c->stream));
    if (!vh.empty()) {
        PG_HIP(c, hipMemcpyAsync(c->d_vh.p, vh.data(), vh.size() * sizeof(pg::VoteHap), hipMemcpyHostToDevice, c->stream));
        // which of their starts own a unique 7-mer (the reference's table entries > 0, calign.pyx:114-121): a row per
        // haplotype beside its packed bases, made once here and read by every launch's anchoring kernel
        PG_HIP(c, c->d_hapU.reserve((size_t)off2[n] + 16));
        PG_HIP(c, c->d_hapUbits.reserve(vh.size() * (size_t)(pg::kHashSize / 32)));
        hipLaunchKernelGGL(pg::k_hap_unique, dim3((unsigned)vh.size()), dim3(256), 0, c->stream, c->d_hap2.p, c->d_vh.p, c->d_hapU.p,
                           c->d_hapUbits.p);
        PG_HIP(c, hipGetLastError());
    }
    int32_t bad = 0;
    PG_HIP(c, hipMemcpyAsync(&bad, c->d_flag.p, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));   // host vectors above go out of scope
    if (bad) return fail(c, PG_ERR_UNSUPPORTED, "a haplotype holds a byte outside 1..31, 33..126");
    c->n_vote_haps = (int)vh.size();
    c->max_hap_len = maxlen;
    c->hap_bytes = total;
    c->n_haps = n;
    return PG_OK;
}

}  // namespace

int pg_haps_upload(pg_ctx* c, int n, const char* seq, const int64_t* off,
                   const int32_t* start_pos, const int32_t* var_seq_len) {
    if (!c) return PG_ERR_INVALID;
    if (n < 0 || (n > 0 && (!seq || !off || !start_pos || !var_seq_len)))
        return fail(c, PG_ERR_INVALID, "pg_haps_upload: bad arguments");
    PG_HIP(c, hipSetDevice(c->device));
    c->n_haps = 0; c->n_pairs = 0; c->launched = false;
    if (n == 0) return PG_OK;
    std::vector<int64_t> soff(n + 1);
    for (int i = 0; i <= n; ++i) soff[i] = off[i] - off[0];
    return haps_install(c, n, soff, start_pos, var_seq_len, [&]() -> int {
        PG_HIP(c, hipMemcpyAsync(c->d_hap_raw.p, seq + off[0], (size_t)soff[n], hipMemcpyHostToDevice, c->stream));
        return PG_OK;
    });
}

int pg_haps_build(pg_ctx* c, int n_stretches, const char* ref_seq, const int64_t* ref_off, const int32_t* ref_start,
                  int n, const int32_t* hap_stretch, const int32_t* hap_var_off, const int32_t* var_pos,
                  const int32_t* var_n_removed, const int32_t* var_added_off, const char* var_added) {
    if (!c) return PG_ERR_INVALID;
    if (n < 0 || n_stretches < 0 || (n > 0 && (n_stretches == 0 || !ref_seq || !ref_off || !ref_start || !hap_stretch || !hap_var_off)))
        return fail(c, PG_ERR_INVALID, "pg_haps_build: bad arguments");
    PG_HIP(c, hipSetDevice(c->device));
    c->n_haps = 0; c->n_pairs = 0; c->launched = false;
    if (n == 0) return PG_OK;
    const int64_t ref_total = ref_off[n_stretches] - ref_off[0];
    if (ref_total < 0 || ref_total > 0x7fffffffLL) return fail(c, PG_ERR_INVALID, "pg_haps_build: the stretches hold %lld bytes; at most 2^31 - 1 are taken", (long long)ref_total);
    const int n_vars = hap_var_off[n] - hap_var_off[0];
    if (n_vars < 0 || (n_vars > 0 && (!var_pos || !var_n_removed || !var_added_off)))
        return fail(c, PG_ERR_INVALID, "pg_haps_build: bad arguments");
    const int64_t added_total = n_vars > 0 ? (int64_t)var_added_off[hap_var_off[n]] - var_added_off[hap_var_off[0]] : 0;
    if (added_total < 0 || (added_total > 0 && !var_added)) return fail(c, PG_ERR_INVALID, "pg_haps_build: bad arguments");
    // The walk of getMutatedSequence (platypus/chaplotype.pyx:283-353) per haplotype, in pieces instead of bytes: which
    // bytes of the stretch and of the variants' added sequences follow one another.  The pieces go to the device, which
    // copies the bytes; lengths, varSeqLen (chaplotype.pyx:66-67) and startPos are known here.
    std::vector<pg::HapPiece> pieces;
    std::vector<int32_t> piece_off(n + 1), start(n), varlen(n);
    std::vector<int64_t> soff(n + 1);
    pieces.reserve((size_t)n + 2 * (size_t)n_vars);
    soff[0] = 0;
    const int32_t a0 = n_vars > 0 ? var_added_off[hap_var_off[0]] : 0;
    for (int i = 0; i < n; ++i) {
        const int s = hap_stretch[i];
        if (s < 0 || s >= n_stretches) return fail(c, PG_ERR_INVALID, "haplotype %d names stretch %d of %d", i, s, n_stretches);
        const int64_t slen = ref_off[s + 1] - ref_off[s];
        if (slen < 0) return fail(c, PG_ERR_INVALID, "stretch %d has length %lld", s, (long long)slen);
        const int64_t S = ref_start[s], E = S + slen, base = ref_off[s] - ref_off[0] - S;   // byte of position x: base + x
        int64_t dst = 0;
        bool outside = false;
        auto from_ref = [&](int64_t a, int64_t b) {              // refFile.getSequence(refName, a, b)
            if (a < S || b < a || b > E) { outside = true; return; }
            if (b > a) { pieces.push_back({(int32_t)(base + a), (int32_t)dst, (int32_t)(b - a)}); dst += b - a; }
        };
        const int v0 = hap_var_off[i], v1 = hap_var_off[i + 1];
        if (v1 < v0) return fail(c, PG_ERR_INVALID, "pg_haps_build: variant offsets of haplotype %d descend", i);
        piece_off[i] = (int32_t)pieces.size();
        int vl = 0;
        if (v0 == v1) {
            from_ref(S, E);                                      // chaplotype.pyx:290-291
        } else {
            int64_t cur = var_pos[v0];
            from_ref(S, cur);                                    // chaplotype.pyx:296-297
            for (int v = v0; v < v1; ++v) {
                const int64_t pos = var_pos[v];
                const int nrem = var_n_removed[v], nadd = var_added_off[v + 1] - var_added_off[v];
                if (nrem < 0 || nadd < 0) return fail(c, PG_ERR_INVALID, "variant %d removes %d and adds %d bases", v, nrem, nadd);
                vl += std::abs(nadd - nrem);
                if (pos < S || pos + nrem + (nadd != nrem) > E) { outside = true; break; }   // the bases it names are not in the stretch
                auto added = [&]() {
                    if (nadd > 0) { pieces.push_back({~(int32_t)(var_added_off[v] - a0), (int32_t)dst, nadd}); dst += nadd; }
                };
                if (pos > cur) { from_ref(cur, pos); cur = pos; }                         // chaplotype.pyx:303-305
                if (nadd == nrem) { added(); cur += nadd; }                               // chaplotype.pyx:308-310
                else if (nadd == 0) { if (pos == cur) { from_ref(pos, pos + 1); cur += 1; } cur += nrem; }   // :313-323
                else if (nrem == 0) { if (pos == cur) { from_ref(pos, pos + 1); cur += 1; } added(); }       // :326-338
                else return fail(c, PG_ERR_UNSUPPORTED, "variant %d removes %d and adds %d bases: not a substitution, an insertion or a deletion", v, nrem, nadd);   // chaplotype.pyx:340-342
                if (outside) break;
            }
            if (!outside && cur < E) from_ref(cur, E);           // chaplotype.pyx:345-346
        }
        if (outside) return fail(c, PG_ERR_INVALID, "haplotype %d: a variant lies outside its stretch %d..%d", i, (int)S, (int)E);
        start[i] = (int32_t)S;
        varlen[i] = vl;
        soff[i + 1] = soff[i] + dst;
    }
    piece_off[n] = (int32_t)pieces.size();
    return haps_install(c, n, soff, start.data(), varlen.data(), [&]() -> int {
        PG_HIP(c, c->d_build_ref.reserve((size_t)ref_total + 16));
        PG_HIP(c, c->d_build_added.reserve((size_t)added_total + 16));
        PG_HIP(c, c->d_build_pieces.reserve(pieces.size() + 1));
        PG_HIP(c, c->d_build_off.reserve((size_t)n + 1));
        PG_HIP(c, hipMemcpyAsync(c->d_build_ref.p, ref_seq + ref_off[0], (size_t)ref_total, hipMemcpyHostToDevice, c->stream));
        if (added_total > 0)
            PG_HIP(c, hipMemcpyAsync(c->d_build_added.p, var_added + a0, (size_t)added_total, hipMemcpyHostToDevice, c->stream));
        PG_HIP(c, hipMemcpyAsync(c->d_build_pieces.p, pieces.data(), pieces.size() * sizeof(pg::HapPiece), hipMemcpyHostToDevice, c->stream));
        PG_HIP(c, hipMemcpyAsync(c->d_build_off.p, piece_off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(pg::k_build_haps, dim3(std::min((n + 3) / 4, 65536)), dim3(256), 0, c->stream, c->d_build_ref.p,
                           c->d_build_added.p, c->d_build_pieces.p, c->d_build_off.p, c->d_hap_soff.p, c->d_hap_raw.p, n);
        PG_HIP(c, hipGetLastError());
        return PG_OK;
    });
}

int pg_haps_fetch(pg_ctx* c, char* seq, int64_t* offsets) {
    if (!c) return PG_ERR_INVALID;
    if (c->n_haps == 0) return fail(c, PG_ERR_STATE, "pg_haps_fetch: no haplotypes on the device");
    PG_HIP(c, hipSetDevice(c->device));
    if (offsets) PG_HIP(c, hipMemcpyAsync(offsets, c->d_hap_soff.p, ((size_t)c->n_haps + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (seq && c->hap_bytes > 0) PG_HIP(c, hipMemcpyAsync(seq, c->d_hap_raw.p, (size_t)c->hap_bytes, hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    return PG_OK;
}

namespace {
// the read batch of a context; nibbles: `seq` holds two bases per byte in BAM's 4-bit code (pg_reads_upload_nibbles)
int reads_install(pg_ctx* c, int n, const char* seq, const char* qual, const int64_t* off,
                  const int32_t* pos, const int32_t* mapq, bool nibbles) {
    if (!c) return PG_ERR_INVALID;
    if (n < 0 || (n > 0 && (!seq || !qual || !off || !pos || !mapq)))
        return fail(c, PG_ERR_INVALID, "%s: bad arguments", nibbles ? "pg_reads_upload_nibbles" : "pg_reads_upload");
    PG_HIP(c, hipSetDevice(c->device));
    c->n_reads = 0; c->n_pairs = 0; c->launched = false;
    if (n == 0) return PG_OK;
    const int64_t total = off[n] - off[0];
    if (total < n) return fail(c, PG_ERR_UNSUPPORTED, "%d reads in %lld bases: reads must hold 1..%d bases", n, (long long)total, c->max_read_len);
    // The host touches the two ends of `off` only: lengths, the offsets of the element and packed rows, the
    // longest and shortest read and the validity of lengths and mapping qualities are worked out on the device.
    // Element and packed arrays are sized by what n reads of `total` bases can need at most.
    const size_t x_bound = (size_t)total + (size_t)n * 31 + 64, w_bound = (size_t)total / 16 + (size_t)n * 3 + 16;
    const int nb = (n + pg::kScanBlock - 1) / pg::kScanBlock;
    PG_HIP(c, c->d_flag.reserve(64));
    PG_HIP(c, c->d_rseq.reserve((size_t)total + 64));
    PG_HIP(c, c->d_rqual.reserve((size_t)total + 64));
    PG_HIP(c, c->d_roff.reserve((size_t)n + 1));
    PG_HIP(c, c->d_off_in.reserve((size_t)n + 1));
    PG_HIP(c, c->d_read_pos.reserve(n));
    PG_HIP(c, c->d_read_mapq.reserve(n));
    PG_HIP(c, c->d_read_len.reserve(n));
    PG_HIP(c, c->d_readx.reserve(x_bound));
    PG_HIP(c, c->d_read_xoff.reserve((size_t)n + 1));
    PG_HIP(c, c->d_read2.reserve(w_bound));
    PG_HIP(c, c->d_read2off.reserve((size_t)n + 1));
    PG_HIP(c, c->d_scan_tops.reserve(2 * (size_t)nb + 2));
    static const int32_t kMeta0[8] = {0, 0, 0x7fffffff, 0x7fffffff, 0, 0, 0, 0};   // d_flag[0..7]: pack flags; [8..15]: meta
    PG_HIP(c, hipMemsetAsync(c->d_flag.p, 0, 8 * sizeof(int32_t), c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_flag.p + 8, kMeta0, sizeof kMeta0, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_off_in.p, off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_read_mapq.p, mapq, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_read_pos.p, pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pg::k_read_meta, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_off_in.p, c->d_read_mapq.p, n,
                       c->max_read_len, c->d_roff.p, c->d_read_len.p, c->d_read_xoff.p, c->d_read2off.p, c->d_flag.p + 8);
    for (int which = 0; which < 2; ++which) {          // row sizes -> row offsets
        int64_t* a = (which == 0 ? c->d_read_xoff.p : c->d_read2off.p) + 1;
        int64_t* tops = c->d_scan_tops.p + (size_t)which * (nb + 1);
        hipLaunchKernelGGL(pg::scan64_local, dim3(nb), dim3(1024), 0, c->stream, a, (int64_t)n, tops);
        hipLaunchKernelGGL(pg::scan64_tops, dim3(1), dim3(1024), 0, c->stream, tops, nb);
        hipLaunchKernelGGL(pg::scan64_add, dim3(nb), dim3(1024), 0, c->stream, a, (int64_t)n, tops);
    }
    PG_HIP(c, hipGetLastError());
    if (nibbles) {                                     // base k of the batch is nibble off[0] + k of `seq`, the high one first
        const int64_t first = off[0] >> 1, n_bytes = ((off[n] + 1) >> 1) - first;
        PG_HIP(c, c->d_rseq4.reserve((size_t)n_bytes + 16));
        PG_HIP(c, hipMemcpyAsync(c->d_rseq4.p, seq + first, (size_t)n_bytes, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(pg::k_unpack_nibbles, dim3((unsigned)std::min<int64_t>((total + 511) / 512, 65536)), dim3(256), 0, c->stream,
                           c->d_rseq4.p, (int)(off[0] & 1), total, c->d_rseq.p);
    } else {
        PG_HIP(c, hipMemcpyAsync(c->d_rseq.p, seq + off[0], (size_t)total, hipMemcpyHostToDevice, c->stream));
    }
    PG_HIP(c, hipMemcpyAsync(c->d_rqual.p, qual + off[0], (size_t)total, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pg::k_pack_readx, dim3(std::min((n + 3) / 4, 65536)), dim3(256), 0, c->stream, c->d_rseq.p,
                       c->d_rqual.p, c->d_roff.p, c->d_read_xoff.p, c->d_readx.p, c->d_codes.p, n, c->d_flag.p);
    hipLaunchKernelGGL(pg::k_pack_codes, dim3(std::min((n + 3) / 4, 65536)), dim3(256), 0, c->stream, c->d_rseq.p,
                       c->d_roff.p, c->d_read2off.p, c->d_read2.p, n);
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipMemcpyAsync(c->h_status, c->d_flag.p, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    const int32_t* st = c->h_status;
    if (st[8] & 1) {
        const int i = st[11];
        return fail(c, PG_ERR_UNSUPPORTED, "read %d has length %lld; this context takes 1..%d", i,
                    (long long)(off[i + 1] - off[i]), c->max_read_len);
    }
    if (st[8] & 2) return fail(c, PG_ERR_UNSUPPORTED, "read %d has mapping quality %d outside 0..255", st[11], mapq[st[11]]);
    if (st[0] & 1) return fail(c, PG_ERR_UNSUPPORTED, "a base quality above 127 was found");
    if (st[0] & 2) return fail(c, PG_ERR_UNSUPPORTED, "a read holds a byte outside 1..31, 33..126");
    c->max_rlen = st[9];
    c->uniform_rlen = (st[9] == st[10]);
    c->n_reads = n;
    return PG_OK;
}
}  // namespace

int pg_reads_upload(pg_ctx* c, int n, const char* seq, const char* qual, const int64_t* off,
                    const int32_t* pos, const int32_t* mapq) {
    return reads_install(c, n, seq, qual, off, pos, mapq, false);
}

int pg_reads_upload_nibbles(pg_ctx* c, int n, const uint8_t* seq4, const char* qual, const int64_t* off,
                            const int32_t* pos, const int32_t* mapq) {
    return reads_install(c, n, reinterpret_cast<const char*>(seq4), qual, off, pos, mapq, true);
}

int pg_pairs_upload(pg_ctx* c, int64_t np, const int32_t* hap_idx, const int32_t* read_idx) {
    if (!c) return PG_ERR_INVALID;
    if (np < 0 || np > 0x7fffffffLL || (np > 0 && (!hap_idx || !read_idx)))
        return fail(c, PG_ERR_INVALID, "pg_pairs_upload: bad arguments");
    if (c->n_haps == 0 || c->n_reads == 0)
        return fail(c, PG_ERR_STATE, "upload haplotypes and reads before pairs");
    PG_HIP(c, hipSetDevice(c->device));
    c->n_pairs = 0; c->launched = false;
    if (np == 0) return PG_OK;
    PG_HIP(c, c->d_pair_hap.reserve(np));
    PG_HIP(c, c->d_pair_read.reserve(np));
    PG_HIP(c, hipMemcpyAsync(c->d_pair_hap.p, hap_idx, np * 4, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_pair_read.p, read_idx, np * 4, hipMemcpyHostToDevice, c->stream));
    return finish_pairs(c, np);
}

int pg_pairs_block(pg_ctx* c, int hap_lo, int hap_hi, int read_lo, int read_hi) {
    if (!c) return PG_ERR_INVALID;
    if (hap_lo < 0 || hap_hi < hap_lo || hap_hi > c->n_haps || read_lo < 0 || read_hi < read_lo || read_hi > c->n_reads)
        return fail(c, PG_ERR_INVALID, "pg_pairs_block: range outside the uploaded batches");
    const int32_t nh = hap_hi - hap_lo, nr = read_hi - read_lo;
    return pg_pairs_blocks(c, 1, &hap_lo, &nh, &read_lo, &nr, nullptr);
}

int pg_pairs_blocks(pg_ctx* c, int nb, const int32_t* hap_lo, const int32_t* n_haps,
                    const int32_t* read_lo, const int32_t* n_reads, int64_t* pair_base) {
    if (!c) return PG_ERR_INVALID;
    if (nb < 0 || (nb > 0 && (!hap_lo || !n_haps || !read_lo || !n_reads)))
        return fail(c, PG_ERR_INVALID, "pg_pairs_blocks: bad arguments");
    if (c->n_haps == 0 || c->n_reads == 0)
        return fail(c, PG_ERR_STATE, "upload haplotypes and reads before pairs");
    std::vector<int64_t> base((size_t)nb + 1);
    int64_t total = 0;
    for (int b = 0; b < nb; ++b) {
        if (hap_lo[b] < 0 || n_haps[b] < 0 || (int64_t)hap_lo[b] + n_haps[b] > c->n_haps || read_lo[b] < 0 ||
            n_reads[b] < 0 || (int64_t)read_lo[b] + n_reads[b] > c->n_reads)
            return fail(c, PG_ERR_INVALID, "pg_pairs_blocks: block %d reaches outside the uploaded batches", b);
        base[b] = total;
        if (pair_base) pair_base[b] = total;
        total += (int64_t)n_haps[b] * n_reads[b];
    }
    base[nb] = total;
    if (total > 0x7fffffffLL) return fail(c, PG_ERR_INVALID, "pg_pairs_blocks: %lld pairs", (long long)total);
    PG_HIP(c, hipSetDevice(c->device));
    c->n_pairs = 0; c->launched = false;
    if (total == 0) return PG_OK;
    // the blocks go up as descriptors (12 bytes each + a base); the pair list is written on the device
    PG_HIP(c, c->d_blk32.reserve((size_t)3 * nb));
    PG_HIP(c, c->d_blk_base.reserve((size_t)nb + 1));
    PG_HIP(c, c->d_pair_hap.reserve(total));
    PG_HIP(c, c->d_pair_read.reserve(total));
    PG_HIP(c, hipMemcpyAsync(c->d_blk32.p, hap_lo, 4 * (size_t)nb, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_blk32.p + nb, read_lo, 4 * (size_t)nb, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_blk32.p + 2 * (size_t)nb, n_reads, 4 * (size_t)nb, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_blk_base.p, base.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pg::k_expand_blocks, dim3(prep_grid(total)), dim3(256), 0, c->stream, nb, c->d_blk32.p,
                       c->d_blk32.p + nb, c->d_blk32.p + 2 * (size_t)nb, c->d_blk_base.p, c->d_pair_hap.p,
                       c->d_pair_read.p, total);
    PG_HIP(c, hipGetLastError());
    return finish_pairs(c, total);      // synchronises: `base` and the caller's arrays are free after it
}

int pg_align_launch(pg_ctx* c) {
    if (!c) return PG_ERR_INVALID;
    if (c->n_pairs == 0) return fail(c, PG_ERR_STATE, "no pairs uploaded");
    PG_HIP(c, hipSetDevice(c->device));
    int rc = enqueue_align(c, ctx_bufs(c, c->n_pairs), c->work, c->d_pair_hap.p, c->d_pair_read.p, c->n_pairs,
                           c->d_score.p, c->d_anchor.p, c->d_like.p);
    if (rc != PG_OK) return rc;
    c->launched = true;
    c->n_results = c->n_pairs; c->r_score = c->d_score.p; c->r_anchor = c->d_anchor.p; c->r_like = c->d_like.p;
    c->r_status = c->work.status;
    return PG_OK;
}

int pg_sync(pg_ctx* c) {
    if (!c) return PG_ERR_INVALID;
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    return PG_OK;
}

int pg_results_fetch_async(pg_ctx* c, int32_t* score, int32_t* anchor, double* like) {
    if (!c) return PG_ERR_INVALID;
    if (!c->launched) return fail(c, PG_ERR_STATE, "nothing launched");
    PG_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->n_results;
    if (score) PG_HIP(c, hipMemcpyAsync(score, c->r_score, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (anchor) PG_HIP(c, hipMemcpyAsync(anchor, c->r_anchor, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (like) PG_HIP(c, hipMemcpyAsync(like, c->r_like, n * 8, hipMemcpyDeviceToHost, c->stream));
    return PG_OK;
}

int pg_results_fetch(pg_ctx* c, int32_t* score, int32_t* anchor, double* like) {
    int rc = pg_results_fetch_async(c, score, anchor, like);
    if (rc != PG_OK) return rc;
    PG_HIP(c, hipStreamSynchronize(c->stream));
    return PG_OK;
}

int pg_align_pairs(pg_ctx* c, int64_t np, const int32_t* hap_idx, const int32_t* read_idx,
                   int32_t* score, int32_t* anchor, double* like) {
    int rc = pg_pairs_upload(c, np, hap_idx, read_idx);
    if (rc != PG_OK) return rc;
    if (np == 0) return PG_OK;
    rc = pg_align_launch(c);
    if (rc != PG_OK) return rc;
    return pg_results_fetch(c, score, anchor, like);
}

int pg_align_window(pg_ctx* c, int n_haps, const char* hap_seq, const int64_t* hap_off, const int32_t* start_pos,
                    const int32_t* var_seq_len, int n_reads, const char* seq, const char* qual, const int64_t* off,
                    const int32_t* pos, const int32_t* mapq, int32_t* score, int32_t* anchor, double* like) {
    if (!c) return PG_ERR_INVALID;
    if (n_haps < 0 || n_reads < 0 || (n_haps > 0 && (!hap_seq || !hap_off || !start_pos || !var_seq_len)) ||
        (n_reads > 0 && (!seq || !qual || !off || !pos || !mapq)))
        return fail(c, PG_ERR_INVALID, "pg_align_window: bad arguments");
    c->n_haps = 0; c->n_reads = 0; c->n_pairs = 0; c->launched = false;      // whatever was uploaded is given up
    if (n_haps == 0 || n_reads == 0) return PG_OK;
    const int64_t np = (int64_t)n_haps * n_reads;
    // ---- the window as the host sees it: the same checks, in the same order, as the upload calls make ----
    const int pad = c->max_read_len + 64;
    int max_hap_len = 0, n_vote = 0;
    for (int i = 0; i < n_haps; ++i) {
        const int64_t l = hap_off[i + 1] - hap_off[i];
        if (l < 0 || l > (1 << 30)) return fail(c, PG_ERR_INVALID, "haplotype %d has length %lld", i, (long long)l);
        if (var_seq_len[i] != 0 && l > PG_MAX_HASHED_HAP_LEN)
            return fail(c, PG_ERR_UNSUPPORTED, "haplotype %d with indels is %lld long; the 7-mer anchoring takes < 16384", i, (long long)l);
        if (var_seq_len[i] != 0 && l < 15)
            return fail(c, PG_ERR_UNSUPPORTED, "haplotype %d with indels is only %lld long; at least 15 bases are needed", i, (long long)l);
        max_hap_len = std::max(max_hap_len, (int)l);
        n_vote += var_seq_len[i] != 0;
    }
    const int64_t hap_total = hap_off[n_haps] - hap_off[0], read_total = off[n_reads] - off[0];
    if (read_total < n_reads)
        return fail(c, PG_ERR_UNSUPPORTED, "%d reads in %lld bases: reads must hold 1..%d bases", n_reads, (long long)read_total, c->max_read_len);
    int max_rlen = 0, min_rlen = 0x7fffffff;
    for (int i = 0; i < n_reads; ++i) {
        const int64_t l = off[i + 1] - off[i];
        if (l < 1 || l > c->max_read_len)
            return fail(c, PG_ERR_UNSUPPORTED, "read %d has length %lld; this context takes 1..%d", i, (long long)l, c->max_read_len);
        max_rlen = std::max(max_rlen, (int)l); min_rlen = std::min(min_rlen, (int)l);
    }
    for (int i = 0; i < n_reads; ++i)
        if (mapq[i] < 0 || mapq[i] > 255) return fail(c, PG_ERR_UNSUPPORTED, "read %d has mapping quality %d outside 0..255", i, mapq[i]);
    const bool uniform = max_rlen == min_rlen;
    // a window too large for one block, or one the table-free anchoring kernel does not take, goes the long way
    const bool bits = c->vote_kernel >= 0 ? (c->vote_kernel == 1 && max_rlen <= 263) : max_rlen <= 263;
    const int64_t x_hap = hap_total + (int64_t)n_haps * pad + 64, x_read = read_total + (int64_t)n_reads * 31 + 64;
    if (np > (1 << 20) || (n_vote > 0 && !bits) || x_hap + x_read > (16 << 20)) {
        int rc = pg_haps_upload(c, n_haps, hap_seq, hap_off, start_pos, var_seq_len);
        if (rc == PG_OK) rc = pg_reads_upload(c, n_reads, seq, qual, off, pos, mapq);
        if (rc == PG_OK) rc = pg_pairs_block(c, 0, n_haps, 0, n_reads);
        if (rc == PG_OK) rc = pg_align_launch(c);
        if (rc == PG_OK) rc = pg_results_fetch(c, score, anchor, like);
        c->n_haps = 0; c->n_reads = 0; c->n_pairs = 0;     // the results stay (n_results)
        return rc;
    }
    PG_HIP(c, hipSetDevice(c->device));
    // ---- the block: the results, the flags, what the host fills, what the kernels make ----
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 15) & ~(size_t)15; return o; };
    const int n_waves = wave_bound(uniform, max_rlen, np);
    // results first, then the flags, then what the host fills: one copy takes [flags .. end of the host's part] up (the
    // flags cleared), one copy brings [results .. flags] back
    const size_t o_res = at;
    const size_t o_score = take(4 * (size_t)np), o_anchor = take(4 * (size_t)np), o_like = take(8 * (size_t)np), o_flags = take(64);
    const size_t res_bytes = at - o_res;
    const size_t o_soff = take(8 * ((size_t)n_haps + 1)), o_doff = take(8 * ((size_t)n_haps + 1)), o_h2off = take(8 * ((size_t)n_haps + 1));
    const size_t o_hlen = take(4 * (size_t)n_haps), o_hstart = take(4 * (size_t)n_haps), o_hvar = take(4 * (size_t)n_haps), o_hord = take(4 * (size_t)n_haps);
    const size_t o_vh = take(sizeof(pg::VoteHap) * ((size_t)n_vote + 1)), o_hcnt = take(4 * ((size_t)n_vote + 1));
    const size_t o_roff = take(8 * ((size_t)n_reads + 1)), o_xoff = take(8 * ((size_t)n_reads + 1)), o_r2off = take(8 * ((size_t)n_reads + 1));
    const size_t o_rlen = take(4 * (size_t)n_reads), o_rpos = take(4 * (size_t)n_reads), o_rmq = take(4 * (size_t)n_reads);
    const size_t o_status = take(4 * pg::ST_WORDS);
    const size_t o_order = take(uniform ? 0 : 4 * (size_t)np), o_waves = take(uniform ? 0 : sizeof(pg::WaveDesc) * ((size_t)n_waves + 1));
    const size_t o_hraw = take((size_t)hap_total + 16), o_rseq = take((size_t)read_total + 64), o_rqual = take((size_t)read_total + 64);
    const size_t in_end = at;
    // row offsets (the same as pg_haps_upload and k_read_meta make)
    std::vector<int64_t>& h2 = c->win_tmp;
    h2.assign((size_t)n_haps + n_reads + 2, 0);
    int64_t* hap2off = h2.data(); int64_t* read2off = h2.data() + n_haps + 1;
    for (int i = 0; i < n_haps; ++i) hap2off[i + 1] = hap2off[i] + ((hap_off[i + 1] - hap_off[i]) + 15) / 16 + 2;
    int64_t xo = 0;
    for (int i = 0; i < n_reads; ++i) read2off[i + 1] = read2off[i] + ((off[i + 1] - off[i]) + 15) / 16 + 2;
    const size_t o_hapx = take(2 * (size_t)x_hap), o_hap2 = take(4 * ((size_t)hap2off[n_haps] + 16)), o_hapU = take(4 * ((size_t)hap2off[n_haps] + 16));
    const size_t o_ubits = take(4 * (size_t)(n_vote + 1) * (pg::kHashSize / 32));
    const size_t o_readx = take(2 * (size_t)x_read), o_read2 = take(4 * ((size_t)read2off[n_reads] + 16));
    const size_t o_ph = take(4 * (size_t)np), o_pr = take(4 * (size_t)np), o_recs = take(sizeof(pg::VoteRec) * ((size_t)n_vote * n_reads + 1));
    const size_t o_ain = take(4 * (size_t)np);
    const size_t total_bytes = at;
    if (c->h_win_cap < in_end) {
        if (c->h_win) { PG_HIP(c, hipHostFree(c->h_win)); c->h_win = nullptr; c->h_win_cap = 0; }
        const size_t want = in_end * 2 + 4096;
        PG_HIP(c, hipHostMalloc((void**)&c->h_win, want, hipHostMallocDefault));
        c->h_win_cap = want;
    }
    PG_HIP(c, c->d_win.reserve(total_bytes + 256));
    uint8_t* H = c->h_win;
    uint8_t* D = c->d_win.p;
    {
        int64_t* soff = (int64_t*)(H + o_soff); int64_t* doff = (int64_t*)(H + o_doff); int64_t* h2o = (int64_t*)(H + o_h2off);
        int32_t* hlen = (int32_t*)(H + o_hlen); int32_t* hst = (int32_t*)(H + o_hstart); int32_t* hvar = (int32_t*)(H + o_hvar);
        int32_t* hord = (int32_t*)(H + o_hord); pg::VoteHap* vh = (pg::VoteHap*)(H + o_vh); int32_t* hcnt = (int32_t*)(H + o_hcnt);
        int nv = 0;
        for (int i = 0; i < n_haps; ++i) {
            soff[i] = hap_off[i] - hap_off[0];
            doff[i] = soff[i] + (int64_t)i * pad;
            h2o[i] = hap2off[i];
            hlen[i] = (int32_t)(hap_off[i + 1] - hap_off[i]);
            hst[i] = start_pos[i]; hvar[i] = var_seq_len[i];
            hord[i] = var_seq_len[i] != 0 ? nv : -1;
            if (var_seq_len[i] != 0) { vh[nv] = pg::VoteHap{i, hlen[i], hap2off[i]}; hcnt[nv] = nv * n_reads; ++nv; }
        }
        soff[n_haps] = hap_total; doff[n_haps] = hap_total + (int64_t)n_haps * pad; h2o[n_haps] = hap2off[n_haps];
        hcnt[nv] = nv * n_reads;
        int64_t* roff = (int64_t*)(H + o_roff); int64_t* xoff = (int64_t*)(H + o_xoff); int64_t* r2o = (int64_t*)(H + o_r2off);
        int32_t* rlen = (int32_t*)(H + o_rlen);
        for (int i = 0; i < n_reads; ++i) {
            const int len = (int)(off[i + 1] - off[i]);
            roff[i] = off[i] - off[0]; xoff[i] = xo; r2o[i] = read2off[i]; rlen[i] = len;
            xo += ((len + 7) / 8) * 8 + 24;
        }
        roff[n_reads] = read_total; xoff[n_reads] = xo; r2o[n_reads] = read2off[n_reads];
        std::memcpy(H + o_rpos, pos, 4 * (size_t)n_reads);
        std::memcpy(H + o_rmq, mapq, 4 * (size_t)n_reads);
        int32_t* st = (int32_t*)(H + o_status);
        std::memset(st, 0, 4 * pg::ST_WORDS);
        st[pg::ST_BAD_PAIR] = 0x7fffffff; st[pg::ST_N_VOTE] = nv * n_reads;
        if (!uniform) {
            // slots grouped by read length, wavefronts of up to 64 slots of one length; a pair is haplotype * n_reads + read
            std::vector<int32_t>& by_len = c->win_tmp32;
            by_len.assign((size_t)max_rlen + 2, 0);
            for (int i = 0; i < n_reads; ++i) ++by_len[rlen[i] + 1];
            int32_t* order = (int32_t*)(H + o_order); pg::WaveDesc* waves = (pg::WaveDesc*)(H + o_waves);
            int nw = 0; int64_t slot = 0;
            for (int L = 0; L <= max_rlen; ++L) {             // by_len[L + 1] reads of length L: n_haps pairs each
                const int64_t cnt = (int64_t)by_len[L + 1] * n_haps;
                for (int64_t k = 0; k < cnt; k += 64) waves[nw++] = pg::WaveDesc{(int32_t)(slot + k), (int32_t)std::min<int64_t>(64, cnt - k), L, 0};
                by_len[L + 1] = (int32_t)slot;               // from here on: next free slot of the length
                slot += cnt;
            }
            for (int h = 0; h < n_haps; ++h)
                for (int r = 0; r < n_reads; ++r) order[by_len[rlen[r] + 1]++] = h * n_reads + r;
            st[pg::ST_N_WAVES] = nw;
        }
        std::memcpy(H + o_hraw, hap_seq + hap_off[0], (size_t)hap_total);
        std::memcpy(H + o_rseq, seq + off[0], (size_t)read_total);
        std::memcpy(H + o_rqual, qual + off[0], (size_t)read_total);
    }
    if (x_read < xo + 64) return fail(c, PG_ERR_HIP, "pg_align_window: element rows larger than their bound");
    std::memset(H + o_flags, 0, 64);
    PG_HIP(c, hipMemcpyAsync(D + o_flags, H + o_flags, in_end - o_flags, hipMemcpyHostToDevice, c->stream));
    pg::WindowArgs A;
    A.n_haps = n_haps; A.n_reads = n_reads; A.n_vote_haps = n_vote; A.pad = pad; A.cap = (int)c->table.size() - 1;
    A.hap_raw = D + o_hraw; A.hap_soff = (const int64_t*)(D + o_soff); A.hap_doff = (const int64_t*)(D + o_doff);
    A.hap2off = (const int64_t*)(D + o_h2off); A.hap_vote_ord = (const int32_t*)(D + o_hord);
    A.rseq = D + o_rseq; A.rqual = D + o_rqual; A.roff = (const int64_t*)(D + o_roff); A.read_xoff = (const int64_t*)(D + o_xoff);
    A.read2off = (const int64_t*)(D + o_r2off); A.read_len = (const int32_t*)(D + o_rlen); A.codes = c->d_codes.p;
    A.hapx = (uint16_t*)(D + o_hapx); A.hap2 = (uint32_t*)(D + o_hap2); A.hapU = (uint32_t*)(D + o_hapU); A.hapUbits = (uint32_t*)(D + o_ubits);
    A.readx = (uint16_t*)(D + o_readx); A.read2 = (uint32_t*)(D + o_read2);
    A.pair_hap = (int32_t*)(D + o_ph); A.pair_read = (int32_t*)(D + o_pr); A.recs = (pg::VoteRec*)(D + o_recs);
    A.flags = (int32_t*)(D + o_flags);
    const int pair_blocks = (int)std::min<int64_t>((np + 255) / 256, 1024);
    hipLaunchKernelGGL(pg::k_window_pack, dim3((unsigned)(n_haps + (n_reads + 3) / 4 + pair_blocks)), dim3(256), 0, c->stream, A);
    PG_HIP(c, hipGetLastError());
    AlignBufs U;
    U.hapx = A.hapx; U.hap_doff = A.hap_doff; U.hap_len = (const int32_t*)(D + o_hlen); U.hap_start = (const int32_t*)(D + o_hstart);
    U.hap_varlen = (const int32_t*)(D + o_hvar); U.hap2 = A.hap2; U.hapU = A.hapU; U.hapUbits = A.hapUbits;
    U.vh = (const pg::VoteHap*)(D + o_vh);
    U.readx = A.readx; U.read_xoff = A.read_xoff; U.read2 = A.read2; U.read_pos = (const int32_t*)(D + o_rpos);
    U.read_mapq = (const int32_t*)(D + o_rmq); U.read_len = (const int32_t*)(D + o_rlen);
    U.n_vote_haps = n_vote; U.max_rlen = max_rlen; U.max_hap_len = max_hap_len; U.uniform = uniform; U.n_waves = n_waves;
    U.timed = false;
    pg::PairWork W{};
    W.status = (int32_t*)(D + o_status); W.hap_cnt = (int32_t*)(D + o_hcnt);
    W.order = uniform ? nullptr : (int32_t*)(D + o_order); W.waves = uniform ? nullptr : (pg::WaveDesc*)(D + o_waves);
    W.recs = A.recs; W.anchor_in = (int32_t*)(D + o_ain);
    int rc = enqueue_align(c, U, W, A.pair_hap, A.pair_read, np, (int32_t*)(D + o_score), (int32_t*)(D + o_anchor), (double*)(D + o_like));
    if (rc != PG_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    uint8_t* R = H + o_res;                               // the results' place in the host block
    PG_HIP(c, hipMemcpyAsync(R, D + o_res, res_bytes, hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    const int32_t flags = *(const int32_t*)(R + (o_flags - o_res));
    if (flags & 1) return fail(c, PG_ERR_UNSUPPORTED, "a base quality above 127 was found");
    if (flags & 2) return fail(c, PG_ERR_UNSUPPORTED, "a haplotype or a read holds a byte outside 1..31, 33..126");
    if (score) std::memcpy(score, R + (o_score - o_res), 4 * (size_t)np);
    if (anchor) std::memcpy(anchor, R + (o_anchor - o_res), 4 * (size_t)np);
    if (like) std::memcpy(like, R + (o_like - o_res), 8 * (size_t)np);
    // the results stay on the device for pg_genotype_loglik, the caller stage and pg_results_fetch, as after pg_align_launch
    c->launched = true;
    c->n_results = np; c->r_score = (const int32_t*)(D + o_score); c->r_anchor = (const int32_t*)(D + o_anchor);
    c->r_like = (const double*)(D + o_like); c->r_status = (const int32_t*)(D + o_status);
    c->ev_valid[PG_K_ANCHOR] = c->ev_valid[PG_K_ALIGN] = false;      // this call's kernels are not timed
    return PG_OK;
}

int pg_align_pairs_noalloc(pg_ctx* c, int64_t np, const int32_t* d_hap_idx, const int32_t* d_read_idx,
                           int32_t* d_score, int32_t* d_anchor, double* d_like) {
    if (!c) return PG_ERR_INVALID;
    if (np < 0 || (np > 0 && (!d_hap_idx || !d_read_idx || !d_score || !d_anchor)))
        return fail(c, PG_ERR_INVALID, "pg_align_pairs_noalloc: bad arguments");
    if (np > (int64_t)kNaPairs)
        return fail(c, PG_ERR_UNSUPPORTED, "pg_align_pairs_noalloc takes at most %lld pairs per call, not %lld",
                    (long long)kNaPairs, (long long)np);
    if (c->n_haps == 0 || c->n_reads == 0)
        return fail(c, PG_ERR_STATE, "upload haplotypes and reads before pairs");
    if ((size_t)c->n_vote_haps > kNaVoteHaps)
        return fail(c, PG_ERR_UNSUPPORTED, "pg_align_pairs_noalloc takes at most %lld indel-bearing haplotypes in the "
                    "uploaded batch, not %d", (long long)kNaVoteHaps, c->n_vote_haps);
    if ((size_t)c->max_rlen + 2 > kNaLen)
        return fail(c, PG_ERR_UNSUPPORTED, "pg_align_pairs_noalloc: reads of %d bases", c->max_rlen);
    if (np == 0) return PG_OK;
    ScratchHold hold(c->device);
    if (!hold.got)
        return fail(c, PG_ERR_STATE, "pg_align_pairs_noalloc: another call is using the fixed device array of GPU %d", c->device);
    PG_HIP(c, hipSetDevice(c->device));
    unsigned char* base = c->scratch_ptr;              // never the array's name: that is no device address on the host
    pg::PairWork W;
    W.status = reinterpret_cast<int32_t*>(base + kNaOffStatus);
    W.hap_cnt = reinterpret_cast<int32_t*>(base + kNaOffHapCnt);
    W.hap_cur = reinterpret_cast<int32_t*>(base + kNaOffHapCur);
    W.len_cnt = reinterpret_cast<int32_t*>(base + kNaOffLen);
    W.len_slot = W.len_cnt + kNaLen; W.len_wave = W.len_cnt + 2 * kNaLen; W.len_cur = W.len_cnt + 3 * kNaLen;
    W.order = c->uniform_rlen ? nullptr : reinterpret_cast<int32_t*>(base + kNaOffOrder);
    W.waves = c->uniform_rlen ? nullptr : reinterpret_cast<pg::WaveDesc*>(base + kNaOffWaves);
    W.recs = reinterpret_cast<pg::VoteRec*>(base + kNaOffRecs);
    W.anchor_in = reinterpret_cast<int32_t*>(base + kNaOffAnchor);
    // the array keeps the last call's contents: enqueue_prep clears the counters it adds into and
    // rewrites the status words; every other array is written before it is read
    int rc = enqueue_prep(c, W, d_hap_idx, d_read_idx, np);
    if (rc != PG_OK) return rc;
    rc = enqueue_align(c, ctx_bufs(c, np), W, d_hap_idx, d_read_idx, np, d_score, d_anchor, d_like);
    if (rc != PG_OK) return rc;
    PG_HIP(c, hipMemcpyAsync(c->h_status, W.status, pg::ST_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    if (c->h_status[pg::ST_BAD_PAIR] != 0x7fffffff)      // the kernels wrote nothing in that case
        return fail(c, PG_ERR_INVALID, "pair %d refers to a haplotype or read outside the uploaded batches",
                    c->h_status[pg::ST_BAD_PAIR]);
    return PG_OK;
}

int pg_anchor_stats(pg_ctx* c, int64_t* counts) {
    if (!c || !counts) return PG_ERR_INVALID;
    if (!c->launched) return fail(c, PG_ERR_STATE, "nothing launched");
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, hipMemcpyAsync(c->h_status, c->r_status, pg::ST_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    counts[0] = c->h_status[pg::ST_N_VOTE];
    counts[1] = c->h_status[pg::ST_N_NAMED];
    counts[2] = c->h_status[pg::ST_N_FULL];
    return PG_OK;
}

int pg_kernel_time_ms(pg_ctx* c, int which, float* ms) {
    if (!c || !ms || which < 0 || which >= PG_K_COUNT) return PG_ERR_INVALID;
    if (!c->ev_valid[which]) return fail(c, PG_ERR_STATE, "kernel group %d has not run", which);
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, hipEventSynchronize(c->ev[2 * which + 1]));
    PG_HIP(c, hipEventElapsedTime(ms, c->ev[2 * which], c->ev[2 * which + 1]));
    return PG_OK;
}

int pg_genotype_loglik(pg_ctx* c, int64_t ng, const int64_t* b1, const int64_t* b2,
                       const int32_t* nreads, double* out) {
    if (!c) return PG_ERR_INVALID;
    if (ng < 0 || (ng > 0 && (!b1 || !b2 || !nreads || !out)))
        return fail(c, PG_ERR_INVALID, "pg_genotype_loglik: bad arguments");
    if (!c->launched) return fail(c, PG_ERR_STATE, "no likelihoods on the device: launch an alignment first");
    if (ng == 0) return PG_OK;
    for (int64_t g = 0; g < ng; ++g)
        if (nreads[g] < 0 || b1[g] < 0 || b2[g] < 0 || b1[g] + nreads[g] > c->n_results || b2[g] + nreads[g] > c->n_results)
            return fail(c, PG_ERR_INVALID, "genotype row %lld reaches outside the pair list", (long long)g);
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, c->d_g1.reserve(ng)); PG_HIP(c, c->d_g2.reserve(ng));
    PG_HIP(c, c->d_gn.reserve(ng)); PG_HIP(c, c->d_gout.reserve(ng));
    PG_HIP(c, hipMemcpyAsync(c->d_g1.p, b1, ng * 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_g2.p, b2, ng * 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_gn.p, nreads, ng * 4, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_GENOTYPE], c->stream));
    hipLaunchKernelGGL(pg::k_genotype, dim3((unsigned)((ng * 64 + 255) / 256)), dim3(256), 0, c->stream,
                       c->r_like, c->d_g1.p, c->d_g2.p, c->d_gn.p, c->d_gout.p, ng);
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_GENOTYPE + 1], c->stream));
    c->ev_valid[PG_K_GENOTYPE] = true;
    PG_HIP(c, hipMemcpyAsync(out, c->d_gout.p, ng * 8, hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    return PG_OK;
}

int pg_hash_sequence(pg_ctx* c, const char* seq, int len, int32_t* table) {
    if (!c) return PG_ERR_INVALID;
    if (!seq || !table || len < 0) return fail(c, PG_ERR_INVALID, "pg_hash_sequence: bad arguments");
    if (len >= pg::kHashSize) return fail(c, PG_ERR_UNSUPPORTED, "sequence of %d bases (limit 16383, calign.pyx:112)", len);
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, c->d_cmp_bytes.reserve((size_t)len + 16));
    PG_HIP(c, c->d_cmp_ints.reserve(pg::kHashSize));
    if (len > 0) PG_HIP(c, hipMemcpyAsync(c->d_cmp_bytes.p, seq, len, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pg::k_hash_sequence, dim3(1), dim3(1024), 0, c->stream, c->d_cmp_bytes.p,
                       len < pg::kHashNucs ? 0 : len, c->d_cmp_ints.p);
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipMemcpyAsync(table, c->d_cmp_ints.p, pg::kHashSize * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    return PG_OK;
}

int pg_map_read(pg_ctx* c, const char* read, int rlen, int seq_len, const int32_t* table, int32_t* position) {
    if (!c) return PG_ERR_INVALID;
    if (!read || !table || !position || rlen < 0 || seq_len < 0)
        return fail(c, PG_ERR_INVALID, "pg_map_read: bad arguments");
    if (seq_len >= pg::kHashSize) return fail(c, PG_ERR_UNSUPPORTED, "sequence of %d bases (limit 16383, calign.pyx:132)", seq_len);
    *position = 0;
    if (rlen < pg::kHashNucs) return PG_OK;                       // calign.pyx:140-141
    PG_HIP(c, hipSetDevice(c->device));
    const size_t n_counts = (size_t)seq_len + rlen;
    PG_HIP(c, c->d_cmp_bytes.reserve((size_t)rlen + 16));
    PG_HIP(c, c->d_cmp_ints.reserve(pg::kHashSize + n_counts + 2));
    int32_t* d_table = c->d_cmp_ints.p;
    int32_t* d_counts = d_table + pg::kHashSize;
    int32_t* d_out = d_counts + n_counts;
    PG_HIP(c, hipMemcpyAsync(c->d_cmp_bytes.p, read, rlen, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(d_table, table, pg::kHashSize * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemsetAsync(d_counts, 0, (n_counts + 2) * sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(pg::k_map_read, dim3(1), dim3(64), 0, c->stream, c->d_cmp_bytes.p, rlen, seq_len, d_table, d_counts, d_out);
    PG_HIP(c, hipGetLastError());
    int32_t out[2] = {0, 0};
    PG_HIP(c, hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    if (out[1]) return fail(c, PG_ERR_INVALID, "the table points outside the sequence (assertion of calign.pyx:148)");
    *position = out[0];
    return PG_OK;
}

int pg_genotype_loglik_rows(pg_ctx* c, int n, const double* row1, const double* row2, double* out) {
    if (!c) return PG_ERR_INVALID;
    if (n < 0 || !out || (n > 0 && (!row1 || !row2)))
        return fail(c, PG_ERR_INVALID, "pg_genotype_loglik_rows: bad arguments");
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, c->d_rows.reserve((size_t)2 * n + 2));
    PG_HIP(c, c->d_g1.reserve(1)); PG_HIP(c, c->d_g2.reserve(1));
    PG_HIP(c, c->d_gn.reserve(1)); PG_HIP(c, c->d_gout.reserve(1));
    const int64_t b1 = 0, b2 = n;
    const int32_t nr = n;
    if (n > 0) {
        PG_HIP(c, hipMemcpyAsync(c->d_rows.p, row1, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        PG_HIP(c, hipMemcpyAsync(c->d_rows.p + n, row2, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    }
    PG_HIP(c, hipMemcpyAsync(c->d_g1.p, &b1, 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_g2.p, &b2, 8, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_gn.p, &nr, 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pg::k_genotype, dim3(1), dim3(256), 0, c->stream,
                       c->d_rows.p, c->d_g1.p, c->d_g2.p, c->d_gn.p, c->d_gout.p, (int64_t)1);
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipMemcpyAsync(out, c->d_gout.p, 8, hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));   // b1, b2, nr live until here
    return PG_OK;
}

int pg_windows_upload(pg_ctx* c, int nw, const int32_t* n_haps, const int32_t* n_inds,
                      const int64_t* like_base, const int32_t* ind_read_off) {
    if (!c) return PG_ERR_INVALID;
    if (nw < 0 || (nw > 0 && (!n_haps || !n_inds || !like_base || !ind_read_off)))
        return fail(c, PG_ERR_INVALID, "pg_windows_upload: bad arguments");
    c->caller_done = false;
    std::vector<pg::WinDesc> wins((size_t)nw);
    int64_t ioff = 0, cells = 0, full = 0, haps = 0, inds = 0, share = 0;
    std::vector<int64_t> share_base((size_t)nw + 1, 0);
    int max_h = 0, max_i = 0;
    int64_t max_full = 0;
    for (int w = 0; w < nw; ++w) {
        const int nh = n_haps[w], ni = n_inds[w];
        if (nh < 1 || nh > 1000)
            return fail(c, PG_ERR_UNSUPPORTED, "window %d has %d haplotypes (1..1000, cpopulation.pyx:546)", w, nh);
        if (ni < 1) return fail(c, PG_ERR_INVALID, "window %d has %d individuals", w, ni);
        const int32_t* o = ind_read_off + ioff;
        if (o[0] != 0) return fail(c, PG_ERR_INVALID, "window %d: read offsets must start at 0", w);
        for (int i = 0; i < ni; ++i)
            if (o[i + 1] < o[i]) return fail(c, PG_ERR_INVALID, "window %d: read offsets must ascend", w);
        const int nr = o[ni];
        if (nr == 0)
            return fail(c, PG_ERR_UNSUPPORTED, "window %d has no reads (the reference divides by zero, cpopulation.pyx:883)", w);
        if (like_base[w] < 0 || like_base[w] + (int64_t)nh * nr > (c->launched ? c->n_results : c->n_pairs))
            return fail(c, PG_ERR_INVALID, "window %d reaches outside the pair list", w);
        pg::WinDesc& W = wins[w];
        W.n_hap = nh; W.n_ind = ni; W.n_reads = nr; W.n_gen = nh * (nh + 1) / 2;
        W.ld = nh > 64 ? (nh + 15) & ~15 : nh; W.reserved = 0;
        W.like_base = like_base[w]; W.ioff_base = ioff; W.cell_base = cells; W.full_base = full;
        W.hap_base = haps; W.ind_base = inds;
        share_base[w] = share; share += (int64_t)ni * nh;
        ioff += ni + 1; cells += (int64_t)ni * W.n_gen; full += (int64_t)ni * nh * W.ld; haps += nh; inds += ni;
        max_h = std::max(max_h, nh); max_i = std::max(max_i, ni);
        max_full = std::max(max_full, (int64_t)ni * nh * W.ld);
    }
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, c->d_wins.reserve((size_t)nw + 1)); PG_HIP(c, c->d_ioff.reserve((size_t)ioff + 1));
    PG_HIP(c, c->d_cells.reserve((size_t)cells + 1));
    // the triangle step reads up to sym_slack doubles past the last matrix, and they must be finite
    PG_HIP(c, c->d_full.reserve((size_t)(full + pg::sym_slack(max_h)) + 1));
    PG_HIP(c, hipMemsetAsync(c->d_full.p + full, 0, sizeof(double) * (size_t)pg::sym_slack(max_h), c->stream));
    PG_HIP(c, c->d_freq.reserve((size_t)haps + 1)); PG_HIP(c, c->d_em_iters.reserve((size_t)nw + 1));
    PG_HIP(c, c->d_em_change.reserve((size_t)nw + 1)); PG_HIP(c, c->d_best_gen.reserve((size_t)inds + 1));
    PG_HIP(c, c->d_best_val.reserve((size_t)inds + 1));
    PG_HIP(c, c->d_share_base.reserve((size_t)nw + 1));
    if (nw > 0) {
        PG_HIP(c, hipMemcpyAsync(c->d_share_base.p, share_base.data(), 8 * (size_t)nw, hipMemcpyHostToDevice, c->stream));
        PG_HIP(c, hipMemcpyAsync(c->d_wins.p, wins.data(), sizeof(pg::WinDesc) * nw, hipMemcpyHostToDevice, c->stream));
        PG_HIP(c, hipMemcpyAsync(c->d_ioff.p, ind_read_off, sizeof(int32_t) * ioff, hipMemcpyHostToDevice, c->stream));
        PG_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->h_wins.swap(wins);
    c->n_windows = nw; c->max_win_haps = max_h; c->max_win_inds = max_i; c->max_win_full = max_full;
    c->win_share = share;
    c->win_cells = cells; c->win_full = full; c->win_haps_total = haps; c->win_inds_total = inds;
    return PG_OK;
}

int pg_caller_launch(pg_ctx* c, int max_iters) {
    if (!c) return PG_ERR_INVALID;
    if (!c->launched) return fail(c, PG_ERR_STATE, "no likelihoods on the device: launch an alignment first");
    if (c->n_windows == 0) return fail(c, PG_ERR_STATE, "no windows uploaded");
    for (const pg::WinDesc& W : c->h_wins)
        if (W.like_base + (int64_t)W.n_hap * W.n_reads > c->n_results)
            return fail(c, PG_ERR_STATE, "the windows do not fit the current pair list");
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_CALLER_SETUP], c->stream));
    PG_HIP(c, c->d_win_max.reserve((size_t)c->n_windows));
    hipLaunchKernelGGL(pg::k_fill_u64, dim3((c->n_windows + 255) / 256), dim3(256), 0, c->stream,
                       c->d_win_max.p, c->n_windows, -1e7);
    const bool tile = c->gl_kernel >= 0 ? c->gl_kernel == 1 : c->max_win_haps >= 24;   // >= 300 genotypes: a workgroup has work
    if (tile) {
        const int lds_doubles = std::max(6 * 1024, c->max_win_haps);  // 48 KiB, at least one read per chunk
        PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_gl_tile), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(lds_doubles * sizeof(double))));
        hipLaunchKernelGGL(pg::k_gl_tile, dim3((unsigned)c->win_inds_total), dim3(256), lds_doubles * sizeof(double), c->stream,
                           c->d_wins.p, c->n_windows, c->d_ioff.p, c->r_like, c->d_cells.p, c->d_win_max.p, lds_doubles);
    } else {
        hipLaunchKernelGGL(pg::k_gl, dim3((unsigned)((c->win_cells + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_wins.p, c->n_windows, c->d_ioff.p, c->r_like, c->d_cells.p, c->d_win_max.p, c->win_cells);
    }
    PG_HIP(c, hipGetLastError());
    if (tile) {
        const int T = (c->max_win_haps + 31) / 32;
        hipLaunchKernelGGL(pg::k_norm_tile, dim3((unsigned)c->win_inds_total, (unsigned)(T * (T + 1) / 2)), dim3(256), 0,
                           c->stream, c->d_wins.p, c->n_windows, c->d_win_max.p, c->d_cells.p, c->d_full.p);
    } else {
        hipLaunchKernelGGL(pg::k_norm, dim3((unsigned)((c->win_cells + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_wins.p, c->n_windows, c->d_win_max.p, c->d_cells.p, c->d_full.p, c->win_cells);
    }
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_CALLER_SETUP + 1], c->stream));
    c->ev_valid[PG_K_CALLER_SETUP] = true;
    // one wavefront per individual, as many as the LDS rows allow: (3 + 2 W) n_max doubles
    const int n_max = c->max_win_haps;
    int n_wav = std::max(1, std::min(16, c->max_win_inds));
    while (n_wav > 1 && (size_t)(3 + 2 * n_wav) * n_max * sizeof(double) > 150 * 1024) --n_wav;
    size_t lds = (size_t)(3 + 2 * n_wav) * n_max * sizeof(double);
    // windows whose matrices fit in what is left of the LDS (at most 128 KiB) keep them there: every
    // iteration reads them again, and a window's wavefronts wait on each other's memory latency
    int stage = 0;
    {
        const int64_t stage_max = c->em_stage_max;
        const int64_t room = ((int64_t)152 * 1024 - (int64_t)lds) / (int64_t)sizeof(double);   // static LDS and rounding need the rest
        const int64_t cap = std::min(stage_max, room);
        if (cap > 0) { stage = (int)std::min<int64_t>(c->max_win_full, cap); lds += (size_t)stage * sizeof(double); }
    }
    PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_em), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_CALLER_EM], c->stream));
    // Kernel choice: windows of at most 8 haplotypes take one wavefront each; if every window's matrices fit
    // the LDS staging area, one workgroup per window iterates on them there; otherwise every iteration is a
    // launch pair over (window, individual) workgroups, so that the whole chip streams the matrices.
    // pg_set_option("em_kernel", small | big | multi) forces one of them (tests, measurements).
    // Windows of up to eight haplotypes add their EM sums in the reference's own order (k_em_exact) where their values fit
    // a wavefront's share of LDS: four wavefronts per workgroup within 64 KiB, else one within 160 KiB.  In a batch that
    // also holds wider windows the two kinds go to their own kernels (split): the wider ones' kernels return at once for
    // the small ones and the other way round.
    int64_t slice = 0;
    int n_small = 0;
    for (const pg::WinDesc& W : c->h_wins)
        if (pg::em_is_exact(W.n_hap, W.n_ind, W.n_gen)) { slice = std::max<int64_t>(slice, (int64_t)W.n_ind * (W.n_gen + 1) + pg::kExactPad); ++n_small; }
    const int exact_waves = n_small == 0 ? 0 : (slice * 8 * 4 <= 64 * 1024 ? 4 : (slice * 8 <= 160 * 1024 ? 1 : 0));
    const bool exact = c->em_small_exact != 0 && exact_waves > 0;
    const int split = exact && n_small < c->n_windows ? 1 : 0;              // both kinds in this batch
    auto launch_exact = [&]() -> int {
        const size_t lds_exact = (size_t)slice * 8 * exact_waves;
        PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_em_exact), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_exact));
        hipLaunchKernelGGL(pg::k_em_exact, dim3((c->n_windows + exact_waves - 1) / exact_waves), dim3(64 * exact_waves), lds_exact,
                           c->stream, c->d_wins.p, c->n_windows, c->d_ioff.p, c->d_full.p, max_iters, (int)slice,
                           c->d_freq.p, c->d_em_iters.p, c->d_em_change.p, c->d_best_gen.p, c->d_best_val.p);
        return PG_OK;
    };
    const bool small_only = n_max <= 8 || (exact && n_small == c->n_windows);   // nothing for the wider windows' kernels
    int path = small_only ? 0 : (c->max_win_full > (int64_t)stage && c->max_win_inds <= 65535 ? 2 : 1);
    if (c->em_kernel == 1) path = 1;
    if (c->em_kernel == 2) path = 2;
    if (c->em_kernel == 0 && small_only) path = 0;
    if (path == 2 && c->max_win_inds > 65535)
        return fail(c, PG_ERR_UNSUPPORTED, "the launch-pair EM path takes at most 65535 individuals per window");
    if (path == 0) {
        if (exact) {
            const int rc = launch_exact();
            if (rc != PG_OK) return rc;
        } else {
            hipLaunchKernelGGL(pg::k_em_small, dim3((c->n_windows + 3) / 4), dim3(256), 0, c->stream,
                               c->d_wins.p, c->n_windows, c->d_ioff.p, c->d_full.p, max_iters,
                               c->d_freq.p, c->d_em_iters.p, c->d_em_change.p, c->d_best_gen.p, c->d_best_val.p);
        }
    } else if (path == 1) {
        hipLaunchKernelGGL(pg::k_em, dim3(c->n_windows), dim3(64 * n_wav), lds, c->stream,
                           c->d_wins.p, c->d_ioff.p, c->d_full.p, max_iters, n_max, stage, split,
                           c->d_freq.p, c->d_em_iters.p, c->d_em_change.p, c->d_best_gen.p, c->d_best_val.p);
        if (split) {
            const int rc = launch_exact();
            if (rc != PG_OK) return rc;
        }
    } else {
        PG_HIP(c, c->d_fprev.reserve((size_t)c->win_haps_total + 1));
        PG_HIP(c, c->d_share.reserve((size_t)c->win_share + 1));
        PG_HIP(c, c->d_em_done.reserve(2 * (size_t)c->n_windows + 1)); // [n_windows] = windows still running, then one arrival counter per window
        pg::EmState S;
        S.f = c->d_freq.p; S.f_prev = c->d_fprev.p; S.share = c->d_share.p; S.iters = c->d_em_iters.p;
        S.change = c->d_em_change.p; S.done = c->d_em_done.p; S.active = c->d_em_done.p + c->n_windows; S.arrived = S.active + 1;
        S.share_base = c->d_share_base.p;
        S.csr = c->d_best_val.p;                            // k_em_calls replaces the sums by the calls' values
        hipLaunchKernelGGL(pg::k_em_init, dim3(c->n_windows), dim3(256), 0, c->stream, c->d_wins.p, c->n_windows, S, split,
                           c->n_windows - n_small);
        int32_t active = c->n_windows - (split ? n_small : 0);
        // wavefronts per individual (the rows of its matrix are split between them): enough to keep loads in
        // flight for a large matrix, one for a small one
        const int step_threads = n_max <= 64 ? 64 : (n_max <= 128 ? 256 : 512);
        const size_t lds_step = (size_t)(1 + step_threads / 64) * n_max * sizeof(double);
        PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_em_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        // all but narrow matrices are stepped from their upper triangle alone (half the bytes; measured better from
        // about 24 haplotypes on)
        const bool sym = c->em_step >= 0 ? c->em_step == 1 : n_max > 24;
        int n_pad = 8;
        for (const pg::WinDesc& W : c->h_wins) n_pad = std::max(n_pad, pg::sym_pad(W.n_hap));
        auto sym_lds = [&](int waves) { return ((size_t)(1 + waves) * n_pad + (size_t)waves * pg::kSymTile) * sizeof(double); };
        // Wavefronts per (window, individual): about ten tiles of the widest matrix each.  Fewer tiles per wavefront
        // and the workgroup's start and end (staging the frequencies, adding the copies of y) weigh more than the
        // walk; more, and the last workgroups of a launch run long on their own (measured at 30 to 257 haplotypes).
        int sym_tiles = 0;
        {
            const int lim = n_max - pg::sym_edge(n_max), n_cb = (lim + 63) >> 6;
            for (int a0 = 0; a0 < lim; a0 += pg::kSymRows) sym_tiles += n_cb - (a0 >> 6);
        }
        int sym_waves = c->em_step_waves > 0 ? c->em_step_waves : (sym_tiles < 12 ? 1 : sym_tiles < 22 ? 2 : sym_tiles < 60 ? 4 : 8);
        while (sym_waves > 1 && sym_lds(sym_waves) > 160 * 1024) sym_waves >>= 1;
        const int sym_threads = 64 * sym_waves;
        const size_t lds_sym = std::max(sym_lds(sym_waves), (size_t)((c->max_win_inds + 15) & ~15));
        if (sym) {
            if (lds_sym > 160 * 1024) return fail(c, PG_ERR_UNSUPPORTED, "EM step of %d haplotypes needs %zu bytes of LDS", n_max, lds_sym);
            PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_em_step_sym), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sym));
        }
        // Blocks of eight iterations.  After every block the number of windows still running is copied to
        // pinned host memory; the host reads the answer of block b only after it has queued block b + 1, so the
        // device never waits for the host (a converged window's workgroups return at once: the one block that
        // may be queued in vain costs a few empty launches).
        c->h_active[0] = c->h_active[1] = active;
        const size_t lds_update = (size_t)((c->max_win_inds + 15) & ~15);
        PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_em_update), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_update));
        const int n_blocks = (max_iters + 7) / 8;
        for (int b = 0; b < n_blocks; ++b) {
            if (b >= 2) {                                       // the answer of block b - 2
                PG_HIP(c, hipEventSynchronize(c->ev_block[b & 1]));
                if (c->h_active[b & 1] <= 0) break;
            }
            for (int it = 8 * b; it < std::min(max_iters, 8 * b + 8); ++it) {
                if (sym) {            // the last workgroup of a window to finish does the window's update
                    hipLaunchKernelGGL(pg::k_em_step_sym, dim3(c->n_windows, c->max_win_inds), dim3(sym_threads), lds_sym, c->stream, c->d_wins.p, c->d_ioff.p, c->d_full.p, S, n_pad, max_iters);
                } else {
                    hipLaunchKernelGGL(pg::k_em_step, dim3(c->n_windows, c->max_win_inds), dim3(step_threads), lds_step, c->stream, c->d_wins.p, c->d_ioff.p, c->d_full.p, S, n_max);
                    hipLaunchKernelGGL(pg::k_em_update, dim3(c->n_windows), dim3(256), lds_update, c->stream, c->d_wins.p, c->d_ioff.p, S, max_iters);
                }
            }
            PG_HIP(c, hipMemcpyAsync(&c->h_active[b & 1], S.active, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            PG_HIP(c, hipEventRecord(c->ev_block[b & 1], c->stream));
        }
        PG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_em_calls), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        hipLaunchKernelGGL(pg::k_em_calls, dim3(c->n_windows, c->max_win_inds), dim3(step_threads), lds_step, c->stream,
                           c->d_wins.p, c->d_ioff.p, c->d_full.p, S, n_max, split, c->d_best_gen.p, c->d_best_val.p);
        if (split) {                                        // after the wider windows' kernels: it writes the small windows' outputs
            const int rc = launch_exact();
            if (rc != PG_OK) return rc;
        }
    }
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipEventRecord(c->ev[2 * PG_K_CALLER_EM + 1], c->stream));
    c->ev_valid[PG_K_CALLER_EM] = true;
    c->caller_done = true;
    return PG_OK;
}

int pg_caller_fetch(pg_ctx* c, double* freq, int32_t* iters, double* max_change,
                    int32_t* best_genotype, double* best_value, double* genotype_post) {
    if (!c) return PG_ERR_INVALID;
    if (!c->caller_done) return fail(c, PG_ERR_STATE, "pg_caller_fetch before pg_caller_launch");
    PG_HIP(c, hipSetDevice(c->device));
    if (freq) PG_HIP(c, hipMemcpyAsync(freq, c->d_freq.p, 8 * c->win_haps_total, hipMemcpyDeviceToHost, c->stream));
    if (iters) PG_HIP(c, hipMemcpyAsync(iters, c->d_em_iters.p, 4 * (size_t)c->n_windows, hipMemcpyDeviceToHost, c->stream));
    if (max_change) PG_HIP(c, hipMemcpyAsync(max_change, c->d_em_change.p, 8 * (size_t)c->n_windows, hipMemcpyDeviceToHost, c->stream));
    if (best_genotype) PG_HIP(c, hipMemcpyAsync(best_genotype, c->d_best_gen.p, 4 * c->win_inds_total, hipMemcpyDeviceToHost, c->stream));
    if (best_value) PG_HIP(c, hipMemcpyAsync(best_value, c->d_best_val.p, 8 * c->win_inds_total, hipMemcpyDeviceToHost, c->stream));
    if (genotype_post) PG_HIP(c, hipMemcpyAsync(genotype_post, c->d_cells.p, 8 * c->win_cells, hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));
    return PG_OK;
}

int pg_variant_posteriors(pg_ctx* c, int nv, const int32_t* var_window, const uint8_t* hap_has_var,
                          const double* prior, double* out) {
    if (!c) return PG_ERR_INVALID;
    if (nv < 0 || (nv > 0 && (!var_window || !hap_has_var || !prior || !out)))
        return fail(c, PG_ERR_INVALID, "pg_variant_posteriors: bad arguments");
    if (!c->caller_done) return fail(c, PG_ERR_STATE, "pg_variant_posteriors before pg_caller_launch");
    if (nv == 0) return PG_OK;
    std::vector<int64_t> moff((size_t)nv + 1, 0);
    for (int v = 0; v < nv; ++v) {
        if (var_window[v] < 0 || var_window[v] >= c->n_windows)
            return fail(c, PG_ERR_INVALID, "variant %d names window %d of %d", v, var_window[v], c->n_windows);
        moff[v + 1] = moff[v] + c->h_wins[var_window[v]].n_hap;
    }
    PG_HIP(c, hipSetDevice(c->device));
    PG_HIP(c, c->d_var_win.reserve(nv)); PG_HIP(c, c->d_var_moff.reserve((size_t)nv + 1));
    PG_HIP(c, c->d_var_mask.reserve((size_t)moff[nv] + 1)); PG_HIP(c, c->d_var_prior.reserve(nv));
    PG_HIP(c, c->d_var_out.reserve(nv));
    PG_HIP(c, hipMemcpyAsync(c->d_var_win.p, var_window, 4 * (size_t)nv, hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_var_moff.p, moff.data(), 8 * ((size_t)nv + 1), hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_var_mask.p, hap_has_var, (size_t)moff[nv], hipMemcpyHostToDevice, c->stream));
    PG_HIP(c, hipMemcpyAsync(c->d_var_prior.p, prior, 8 * (size_t)nv, hipMemcpyHostToDevice, c->stream));
    const size_t lds = (size_t)2 * c->max_win_haps * sizeof(double);
    hipLaunchKernelGGL(pg::k_var_post, dim3(nv), dim3(64), lds, c->stream,
                       c->d_wins.p, c->d_ioff.p, c->d_full.p, c->d_freq.p, c->d_var_win.p, c->d_var_moff.p,
                       c->d_var_mask.p, c->d_var_prior.p, c->d_var_out.p, c->max_win_haps);
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipMemcpyAsync(out, c->d_var_out.p, 8 * (size_t)nv, hipMemcpyDeviceToHost, c->stream));
    PG_HIP(c, hipStreamSynchronize(c->stream));   // also keeps moff alive until the copies are done
    return PG_OK;
}

}  // extern "C"

namespace {
// best of three timed launches of an issue-rate probe (the first, untimed, warms up); instructions per lane
// and launch = iters * 8 * 16 * 2
template <typename K>
int time_rate_probe(pg_ctx* c, K kernel, double* lane_instr_per_second) {
    PG_HIP(c, hipSetDevice(c->device));
    const int blocks = 256 * 8, iters = 4096;      // 8 workgroups of 4 wavefronts per CU
    PG_HIP(c, c->d_score.reserve((size_t)blocks * 256));
    uint32_t* out = reinterpret_cast<uint32_t*>(c->d_score.p);
    hipEvent_t e0, e1;
    PG_HIP(c, hipEventCreate(&e0));
    PG_HIP(c, hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        PG_HIP(c, hipEventRecord(e0, c->stream));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, c->stream, out, iters, 0x00030005u, 0x7ff07ff1u);
        PG_HIP(c, hipEventRecord(e1, c->stream));
        PG_HIP(c, hipEventSynchronize(e1));
        float ms = 0;
        PG_HIP(c, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->launched = false;                           // the probe used the score buffer
    *lane_instr_per_second = (double)blocks * 256.0 * iters * 8.0 * 16.0 * 2.0 / (best * 1e-3);
    return PG_OK;
}
}  // namespace

extern "C" {

int pg_valu_int16_peak(pg_ctx* c, double* ops_per_second) {
    if (!c || !ops_per_second) return PG_ERR_INVALID;
    double rate = 0;
    int rc = time_rate_probe(c, pg::k_pk_rate, &rate);
    *ops_per_second = 2.0 * rate;                  // one packed instruction on one lane = 2 int16 operations
    return rc;
}

int pg_valu_vop2_rate(pg_ctx* c, double* lane_instructions_per_second) {
    if (!c || !lane_instructions_per_second) return PG_ERR_INVALID;
    return time_rate_probe(c, pg::k_vop2_rate, lane_instructions_per_second);
}

}  // extern "C"

// ---- BGZF inflation (pg_inflate.hip.h) ----------------------------------------------------
// An inflater is independent of any pg_ctx: a stream, grow-only device buffers and one page-locked output
// buffer of its own, so that helper threads that read BAM files can each keep one beside the thread that calls
// variants.  One thread at a time per inflater.
struct pg_inflater {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;                    // blocking-sync event: the waiting thread sleeps
    DevBuf<unsigned char> d_raw, d_out;
    DevBuf<pg::InflBlock> d_blocks;
    DevBuf<int32_t> d_status;
    unsigned char* h_out = nullptr;               // page-locked; what pg_bgzf_inflate hands out
    size_t h_out_cap = 0;
    unsigned char* h_in = nullptr;                // page-locked; what pg_inflater_input hands out
    size_t h_in_cap = 0;
    std::vector<pg::InflBlock> blocks;
    int32_t* h_status = nullptr;                  // page-locked as well: a copy into pageable memory would make the call spin
    size_t h_status_cap = 0;
    int lanes = 1;                                // 1: one wavefront per BGZF block; 8, 16, 32: one lane per block, that many per wavefront
    float last_ms = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    std::string err;
};

namespace {
int ifail(pg_inflater* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}
#define PG_IHIP(h, call)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return ifail((h), PG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// Block boundaries of the complete BGZF blocks at the start of raw (header layout: c/samtools/bgzf.c,
// check_header and the BSIZE field).  Returns 0, or -1 for bytes that are not a BGZF block.
int bgzf_scan(const unsigned char* raw, size_t n, std::vector<pg::InflBlock>* blocks, size_t* consumed, size_t* total) {
    size_t p = 0, out = 0;
    while (p + 18 <= n) {
        if (raw[p] != 0x1f || raw[p + 1] != 0x8b || raw[p + 2] != 8 || !(raw[p + 3] & 4)) return -1;
        const size_t xlen = raw[p + 10] | (raw[p + 11] << 8);
        if (p + 12 + xlen > n) break;
        long bsize = -1;
        size_t k = p + 12;
        while (k + 4 <= p + 12 + xlen) {
            const size_t slen = raw[k + 2] | (raw[k + 3] << 8);
            if (raw[k] == 66 && raw[k + 1] == 67 && slen == 2 && k + 6 <= p + 12 + xlen) bsize = raw[k + 4] | (raw[k + 5] << 8);
            k += 4 + slen;
        }
        if (bsize < 0 || (size_t)bsize + 1 < 12 + xlen + 8) return -1;
        if (p + bsize + 1 > n) break;
        uint32_t isize;
        memcpy(&isize, raw + p + bsize + 1 - 4, 4);
        if (isize > 65536) return -1;
        if (blocks) {
            pg::InflBlock b;
            b.in_off = (uint32_t)(p + 12 + xlen);
            b.in_len = (uint32_t)(bsize + 1 - 12 - xlen - 8);
            b.out_off = (uint32_t)out;
            b.out_len = isize;
            blocks->push_back(b);
        }
        out += isize;
        p += bsize + 1;
    }
    *consumed = p;
    *total = out;
    return 0;
}

template <int AL>
hipError_t launch_inflate(pg_inflater* h, int n_blocks) {
    const size_t lds = (size_t)pg::IT_U16 * 2 * AL;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_bgzf_inflate<AL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pg::k_bgzf_inflate<AL>, dim3((n_blocks + AL - 1) / AL), dim3(64), lds, h->stream, h->d_raw.p, h->d_blocks.p,
                       n_blocks, h->d_out.p, h->d_status.p);
    return hipGetLastError();
}
}  // namespace

extern "C" {

int pg_inflater_create(int device_ordinal, pg_inflater** out) {
    if (!out) return ifail(nullptr, PG_ERR_INVALID, "pg_inflater_create: out is NULL");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_ordinal < 0 || device_ordinal >= n_dev)
        return ifail(nullptr, PG_ERR_NO_DEVICE, "no HIP device %d", device_ordinal);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ifail(nullptr, PG_ERR_NO_DEVICE, "device %d is not gfx950", device_ordinal);
    pg_inflater* h = new pg_inflater();
    h->device = device_ordinal;
    hipError_t e = hipSetDevice(device_ordinal);
    int prio_low = 0, prio_high = 0;
    if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_low);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->done, hipEventBlockingSync | hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&h->t0);
    if (e == hipSuccess) e = hipEventCreate(&h->t1);
    if (e != hipSuccess) {
        ifail(nullptr, PG_ERR_HIP, "pg_inflater_create: %s", hipGetErrorString(e));
        delete h;
        return PG_ERR_HIP;
    }
    *out = h;
    return PG_OK;
}

int pg_inflater_destroy(pg_inflater* h) {
    if (!h) return PG_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->d_raw.release(); h->d_out.release(); h->d_blocks.release(); h->d_status.release();
    if (h->h_out) (void)hipHostFree(h->h_out);
    if (h->h_status) (void)hipHostFree(h->h_status);
    if (h->h_in) (void)hipHostFree(h->h_in);
    if (h->done) (void)hipEventDestroy(h->done);
    if (h->t0) (void)hipEventDestroy(h->t0);
    if (h->t1) (void)hipEventDestroy(h->t1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PG_OK;
}

const char* pg_inflater_error(const pg_inflater* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pg_inflater_input(pg_inflater* h, size_t bytes, void** buf) {
    if (!h || !buf) return PG_ERR_INVALID;
    if (bytes > h->h_in_cap) {
        PG_IHIP(h, hipSetDevice(h->device));
        unsigned char* fresh = nullptr;
        const size_t cap = bytes + bytes / 4 + 4096;
        PG_IHIP(h, hipHostMalloc(reinterpret_cast<void**>(&fresh), cap, hipHostMallocDefault));
        if (h->h_in) {
            memcpy(fresh, h->h_in, h->h_in_cap);
            (void)hipHostFree(h->h_in);
        }
        h->h_in = fresh;
        h->h_in_cap = cap;
    }
    *buf = h->h_in;
    return PG_OK;
}

int pg_inflater_set_lanes(pg_inflater* h, int lanes) {
    if (!h || (lanes != 1 && lanes != 8 && lanes != 16 && lanes != 32)) return PG_ERR_INVALID;
    h->lanes = lanes;
    return PG_OK;
}

int pg_inflater_last_ms(pg_inflater* h, float* ms) {
    if (!h || !ms) return PG_ERR_INVALID;
    *ms = h->last_ms;
    return PG_OK;
}

int pg_bgzf_scan(const void* raw, size_t n_raw, size_t* consumed, size_t* total_out, int* n_blocks) {
    if ((!raw && n_raw) || !consumed || !total_out) return PG_ERR_INVALID;
    std::vector<pg::InflBlock> blocks;
    if (bgzf_scan(static_cast<const unsigned char*>(raw), n_raw, n_blocks ? &blocks : nullptr, consumed, total_out) != 0)
        return PG_ERR_UNSUPPORTED;
    if (n_blocks) *n_blocks = (int)blocks.size();
    return PG_OK;
}

int pg_bgzf_inflate(pg_inflater* h, const void* raw_, size_t n_raw, const void* prefix, size_t n_prefix,
                    void** out, size_t* n_out, size_t* consumed) {
    if (!h) return PG_ERR_INVALID;
    if ((!raw_ && n_raw) || (!prefix && n_prefix) || !out || !n_out || !consumed) return ifail(h, PG_ERR_INVALID, "pg_bgzf_inflate: NULL argument");
    if (n_raw > (size_t)1 << 30) return ifail(h, PG_ERR_INVALID, "pg_bgzf_inflate: at most 1 GiB of compressed bytes per call");
    const unsigned char* raw = static_cast<const unsigned char*>(raw_);
    h->blocks.clear();
    size_t eaten = 0, total = 0;
    if (bgzf_scan(raw, n_raw, &h->blocks, &eaten, &total) != 0) return ifail(h, PG_ERR_UNSUPPORTED, "not a BGZF block");
    if (total > (size_t)3 << 30) return ifail(h, PG_ERR_INVALID, "pg_bgzf_inflate: more than 3 GiB of output in one call");
    PG_IHIP(h, hipSetDevice(h->device));
    const size_t need = n_prefix + total;
    if (need > h->h_out_cap) {
        // the prefix may lie in the buffer that is about to be replaced
        unsigned char* fresh = nullptr;
        const size_t cap = need + need / 4 + 4096;
        PG_IHIP(h, hipHostMalloc(reinterpret_cast<void**>(&fresh), cap, hipHostMallocDefault));
        if (n_prefix) memcpy(fresh, prefix, n_prefix);
        if (h->h_out) (void)hipHostFree(h->h_out);
        h->h_out = fresh;
        h->h_out_cap = cap;
    } else if (n_prefix && prefix != h->h_out) {
        memmove(h->h_out, prefix, n_prefix);
    }
    *out = h->h_out;
    *n_out = need;
    *consumed = eaten;
    h->last_ms = 0;
    const int nb = (int)h->blocks.size();
    if (nb == 0 || total == 0) return PG_OK;
    PG_IHIP(h, h->d_raw.reserve(eaten + 2048));    // the kernels read whole words and, ahead of the decoder, up to 540 bytes past a stream
    PG_IHIP(h, h->d_out.reserve(total + 64));
    PG_IHIP(h, h->d_blocks.reserve(nb));
    PG_IHIP(h, h->d_status.reserve(nb));
    if ((size_t)nb > h->h_status_cap) {
        if (h->h_status) (void)hipHostFree(h->h_status);
        h->h_status = nullptr;
        h->h_status_cap = 0;
        PG_IHIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_status), ((size_t)nb + 1024) * sizeof(int32_t), hipHostMallocDefault));
        h->h_status_cap = (size_t)nb + 1024;
    }
    PG_IHIP(h, hipMemcpyAsync(h->d_raw.p, raw, eaten, hipMemcpyHostToDevice, h->stream));
    PG_IHIP(h, hipMemcpyAsync(h->d_blocks.p, h->blocks.data(), (size_t)nb * sizeof(pg::InflBlock), hipMemcpyHostToDevice, h->stream));
    PG_IHIP(h, hipEventRecord(h->t0, h->stream));
    if (h->lanes == 1) {
        PG_IHIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pg::k_bgzf_inflate_wave), hipFuncAttributeMaxDynamicSharedMemorySize, pg::IW_LDS));
        hipLaunchKernelGGL(pg::k_bgzf_inflate_wave, dim3(nb), dim3(64), pg::IW_LDS, h->stream, h->d_raw.p, h->d_blocks.p, nb, h->d_out.p, h->d_status.p);
        PG_IHIP(h, hipGetLastError());
    }
    else if (h->lanes == 8) PG_IHIP(h, launch_inflate<8>(h, nb));
    else if (h->lanes == 16) PG_IHIP(h, launch_inflate<16>(h, nb));
    else PG_IHIP(h, launch_inflate<32>(h, nb));
    PG_IHIP(h, hipEventRecord(h->t1, h->stream));
    PG_IHIP(h, hipEventRecord(h->done, h->stream));
    // The kernel runs for tens of milliseconds whatever the number of blocks.  The calling thread sleeps in short
    // steps instead of spinning on the event (helper threads share the host's cores with the thread that calls
    // variants), and the copies back are queued only once the kernel has ended: a copy that waits for a kernel
    // inside a copy engine's queue holds up the copies of every other stream behind it.
    {   // a lane needs about 0.4 us per byte of its block: sleep through most of the largest block's time, then poll
        uint32_t longest = 0;
        for (const pg::InflBlock& b : h->blocks) longest = std::max(longest, b.out_len);
        const double first_us = (h->lanes == 1 ? 0.02 : 0.3) * (double)longest;
        if (first_us > 500.0) usleep((useconds_t)first_us);
    }
    for (;;) {
        const hipError_t q = hipEventQuery(h->done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return ifail(h, PG_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        usleep(500);
    }
    PG_IHIP(h, hipMemcpyAsync(h->h_out + n_prefix, h->d_out.p, total, hipMemcpyDeviceToHost, h->stream));
    PG_IHIP(h, hipMemcpyAsync(h->h_status, h->d_status.p, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PG_IHIP(h, hipStreamSynchronize(h->stream));
    (void)hipEventElapsedTime(&h->last_ms, h->t0, h->t1);
    for (int b = 0; b < nb; ++b)
        if (h->h_status[b] != pg::INF_OK)
            return ifail(h, PG_ERR_UNSUPPORTED, "corrupt BGZF block %d of the stretch (deflate error %d)", b, h->h_status[b]);
    return PG_OK;
}

}  // extern "C"
