// Device-side preparation of a pair list for the anchoring and alignment
// kernels: everything the launch needs beyond the (haplotype, read) indices is
// derived here, so the host never walks the pairs.
//
//   k_expand_blocks  dense (haplotype range x read range) blocks -> pair list
//   k_prep_count     validates the indices; counts the pairs of every
//                    indel-bearing haplotype and (ragged batches) of every
//                    read length
//   k_prep_scan      one workgroup: counts -> first slots, wavefront totals
//   k_prep_scatter   vote records grouped by indel-bearing haplotype; slot order
//                    grouped by read length
//   k_prep_waves     wavefront descriptors of a ragged batch
//
// All of them work on a PairWork: a set of scratch arrays that is either owned
// by the context (grown on demand) or carved out of the library's fixed
// device array (the entry point that allocates nothing).
//
// The grouping uses atomics, so the order of the records inside a group and of
// the slots inside a read length differs from run to run.  Results do not: every
// pair's score, anchor and likelihood depend on the pair alone and are written
// to the pair's own index.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_kernels.hip.h"

namespace pg {

// status words of a PairWork
enum { ST_BAD_PAIR = 0,   // smallest index of a pair that names a haplotype / read outside the batch, else INT32_MAX
       ST_N_WAVES = 1,    // wavefronts of the alignment kernel (ragged batches)
       ST_N_VOTE = 2,     // pairs on indel-bearing haplotypes
       ST_N_NAMED = 3,    // of those, pairs the anchoring kernel's candidates did not decide: counted over the positions their 7-mers name
       ST_N_FULL = 4,     //           ... and pairs counted over every position (both per launch)
       ST_WORDS = 16 };

struct PairWork {
    int32_t* status;      // [ST_WORDS]
    int32_t* hap_cnt;     // [n_vote_haps + 1]  counts, then first record of every indel-bearing haplotype
    int32_t* hap_cur;     // [n_vote_haps]      next free record while scattering
    int32_t* len_cnt;     // [max_len + 2]      pairs per read length            (ragged batches only, else unused)
    int32_t* len_slot;    // [max_len + 2]      first slot of the length
    int32_t* len_wave;    // [max_len + 2]      first wavefront of the length
    int32_t* len_cur;     // [max_len + 2]      next free slot while scattering
    int32_t* order;       // [n_pairs]          slot -> pair                      (ragged batches only)
    WaveDesc* waves;      // [n_pairs / 64 + max_len + 2]                         (ragged batches only)
    VoteRec* recs;        // [n_pairs]
    int32_t* anchor_in;   // [n_pairs]
};

struct PrepArgs {
    const int32_t* pair_hap;
    const int32_t* pair_read;
    int64_t n_pairs;
    int n_haps, n_reads, n_vote_haps, max_len;
    bool ragged;
    const int32_t* hap_vote_ord;   // per haplotype: its number among the indel-bearing ones, -1 for the others
    const int32_t* read_len;
    const int64_t* read2off;
    PairWork W;
};

// Adds one to counters[key] for every lane with valid set, one atomic per
// distinct key of the wavefront; returns the value the lane's own increment
// started from (old counter + rank among the lanes with the same key).
__device__ __forceinline__ int wave_count_by_key(int32_t* counters, int key, bool valid) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    int mine = 0;
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int k = __builtin_amdgcn_readlane(key, lead);
        const bool same = valid && key == k;
        const unsigned long long m = __ballot(same);
        int base = 0;
        if (lane == lead) base = atomicAdd(&counters[k], (int)__popcll(m));
        base = __builtin_amdgcn_readlane(base, lead);
        if (same) mine = base + (int)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    return mine;
}

__global__ void k_prep_init(int32_t* status)
{
    if (threadIdx.x < ST_WORDS) status[threadIdx.x] = threadIdx.x == ST_BAD_PAIR ? 0x7fffffff : 0;
}

__global__ __launch_bounds__(256) void k_expand_blocks(int n_blocks, const int32_t* __restrict__ hap_lo,
                                                       const int32_t* __restrict__ read_lo,
                                                       const int32_t* __restrict__ n_reads,
                                                       const int64_t* __restrict__ pair_base,   // [n_blocks + 1]
                                                       int32_t* __restrict__ pair_hap, int32_t* __restrict__ pair_read,
                                                       int64_t n_pairs)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = n_blocks;                   // last block with pair_base <= p (empty blocks share a base)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pair_base[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t local = p - pair_base[lo];
        const int nr = n_reads[lo];
        const int h = (int)(local / nr);
        pair_hap[p] = hap_lo[lo] + h;
        pair_read[p] = read_lo[lo] + (int)(local - (int64_t)h * nr);
    }
}

__global__ __launch_bounds__(256) void k_prep_count(const PrepArgs A)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    // whole wavefronts walk the list together (the ballots of wave_count_by_key need all lanes there)
    const int64_t n_round = (A.n_pairs + 63) & ~(int64_t)63;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_round; p += stride) {
        const bool in = p < A.n_pairs;
        const int h = in ? A.pair_hap[p] : 0, r = in ? A.pair_read[p] : 0;
        const bool ok = in && (uint32_t)h < (uint32_t)A.n_haps && (uint32_t)r < (uint32_t)A.n_reads;
        if (in && !ok) atomicMin(&A.W.status[ST_BAD_PAIR], (int32_t)p);
        const int ord = ok ? A.hap_vote_ord[h] : -1;
        (void)wave_count_by_key(A.W.hap_cnt, ord, ord >= 0);
        if (A.ragged) (void)wave_count_by_key(A.W.len_cnt, ok ? A.read_len[r] : 0, ok);
    }
}

// exclusive prefix sums by one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_prep_scan(const PrepArgs A)
{
    __shared__ int32_t s_a[1024], s_b[1024];
    const int tid = threadIdx.x;
    auto scan_lds = [&](int32_t* s) {               // inclusive, in place
        for (int off = 1; off < 1024; off <<= 1) {
            const int32_t v = tid >= off ? s[tid - off] : 0;
            __syncthreads();
            s[tid] += v;
            __syncthreads();
        }
    };
    {   // records of the indel-bearing haplotypes
        const int n = A.n_vote_haps;
        const int chunk = (n + 1023) / 1024, lo = min(n, tid * chunk), hi = min(n, lo + chunk);
        int32_t sum = 0;
        for (int i = lo; i < hi; ++i) sum += A.W.hap_cnt[i];
        s_a[tid] = sum;
        __syncthreads();
        scan_lds(s_a);
        int32_t run = s_a[tid] - sum;
        for (int i = lo; i < hi; ++i) {
            const int32_t c = A.W.hap_cnt[i];
            A.W.hap_cnt[i] = run; A.W.hap_cur[i] = run;
            run += c;
        }
        if (tid == 1023) { A.W.hap_cnt[n] = s_a[1023]; A.W.status[ST_N_VOTE] = s_a[1023]; }
        __syncthreads();
    }
    if (A.ragged) {   // slots and wavefronts of every read length
        const int n = A.max_len + 1;
        const int chunk = (n + 1023) / 1024, lo = min(n, tid * chunk), hi = min(n, lo + chunk);
        int32_t sum = 0, wsum = 0;
        for (int i = lo; i < hi; ++i) { const int32_t c = A.W.len_cnt[i]; sum += c; wsum += (c + 63) >> 6; }
        s_a[tid] = sum; s_b[tid] = wsum;
        __syncthreads();
        scan_lds(s_a);
        scan_lds(s_b);
        int32_t run = s_a[tid] - sum, wrun = s_b[tid] - wsum;
        for (int i = lo; i < hi; ++i) {
            const int32_t c = A.W.len_cnt[i];
            A.W.len_slot[i] = run; A.W.len_cur[i] = run; A.W.len_wave[i] = wrun;
            run += c; wrun += (c + 63) >> 6;
        }
        if (tid == 1023) { A.W.len_slot[n] = s_a[1023]; A.W.len_wave[n] = s_b[1023]; A.W.status[ST_N_WAVES] = s_b[1023]; }
    }
}

__global__ __launch_bounds__(256) void k_prep_scatter(const PrepArgs A)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t n_round = (A.n_pairs + 63) & ~(int64_t)63;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_round; p += stride) {
        const bool in = p < A.n_pairs;
        const int h = in ? A.pair_hap[p] : 0, r = in ? A.pair_read[p] : 0;
        const bool ok = in && (uint32_t)h < (uint32_t)A.n_haps && (uint32_t)r < (uint32_t)A.n_reads;
        const int ord = ok ? A.hap_vote_ord[h] : -1;
        const int rl = ok ? A.read_len[r] : 0;
        const int slot = wave_count_by_key(A.W.hap_cur, ord, ord >= 0);
        if (ord >= 0) A.W.recs[slot] = VoteRec{(int32_t)p, rl, A.read2off[r]};
        if (A.ragged) {
            const int s = wave_count_by_key(A.W.len_cur, rl, ok);
            if (ok) A.W.order[s] = (int32_t)p;
        }
    }
}

__global__ __launch_bounds__(256) void k_prep_waves(const PrepArgs A, int wave_bound)
{
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= wave_bound || w >= A.W.status[ST_N_WAVES]) return;
    int lo = 0, hi = A.max_len + 1;                  // last length whose first wavefront is <= w: it owns w
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.W.len_wave[mid] <= w) lo = mid; else hi = mid;
    }
    const int k = w - A.W.len_wave[lo];
    const int c = A.W.len_cnt[lo];
    A.W.waves[w] = WaveDesc{A.W.len_slot[lo] + 64 * k, min(64, c - 64 * k), lo, 0};
}

// ---------------------------------------------------------------------------
// Upload bookkeeping on the device: the offset arrays of a read batch are derived
// from the caller's offsets by kernels, so the host never walks the reads either.
//   k_read_meta   per read: length, offset relative to the first read, the sizes of its
//                 element row and of its packed row; validates length and mapping quality;
//                 longest and shortest read
//   scan64_*      inclusive prefix sums of an int64 array in place (three passes)
// meta words: [0] flags (1 a length outside 1..max, 2 a mapping quality outside 0..255),
// [1] longest read, [2] shortest read, [3] first offending read
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_read_meta(const int64_t* __restrict__ off, const int32_t* __restrict__ mapq, int n,
                                                   int max_read_len, int64_t* __restrict__ roff, int32_t* __restrict__ rlen,
                                                   int64_t* __restrict__ xoff, int64_t* __restrict__ off2,
                                                   int32_t* __restrict__ meta)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int64_t o0 = off[0];
    int lmax = 0, lmin = 0x7fffffff;
    if (i == 0) { xoff[0] = 0; off2[0] = 0; roff[n] = off[n] - o0; }
    if (i < n) {
        const int64_t l = off[i + 1] - off[i];
        const bool bad_len = l < 1 || l > max_read_len, bad_mq = mapq[i] < 0 || mapq[i] > 255;
        if (bad_len || bad_mq) { atomicOr(&meta[0], bad_len ? 1 : 2); atomicMin(&meta[3], i); }
        const int len = bad_len ? 1 : (int)l;
        roff[i] = off[i] - o0;
        rlen[i] = len;
        xoff[i + 1] = ((len + 7) / 8) * 8 + 24;        // 8-element aligned rows with >= 24 padding elements
        off2[i + 1] = (len + 15) / 16 + 2;             // packed 2-bit rows end with two zero words
        lmax = lmin = len;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lmax = max(lmax, __shfl_xor(lmax, o)); lmin = min(lmin, __shfl_xor(lmin, o)); }
    if ((threadIdx.x & 63) == 0 && lmax > 0) { atomicMax(&meta[1], lmax); atomicMin(&meta[2], lmin); }
}

constexpr int kScanBlock = 4096;                        // elements per workgroup of the scan passes

__global__ __launch_bounds__(1024) void scan64_local(int64_t* __restrict__ a, int64_t n, int64_t* __restrict__ tops)
{
    __shared__ int64_t s[1024];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kScanBlock + 4 * tid;
    int64_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? a[base + k] : 0; sum += v[k]; v[k] = sum; }
    s[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t t = tid >= o ? s[tid - o] : 0;
        __syncthreads();
        s[tid] += t;
        __syncthreads();
    }
    const int64_t before = s[tid] - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) a[base + k] = before + v[k];
    if (tid == 1023) tops[blockIdx.x] = s[1023];
}

__global__ __launch_bounds__(1024) void scan64_tops(int64_t* __restrict__ tops, int nb)   // exclusive, one workgroup
{
    __shared__ int64_t s[1024];
    const int tid = threadIdx.x;
    const int chunk = (nb + 1023) / 1024, lo = min(nb, tid * chunk), hi = min(nb, lo + chunk);
    int64_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += tops[i];
    s[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t t = tid >= o ? s[tid - o] : 0;
        __syncthreads();
        s[tid] += t;
        __syncthreads();
    }
    int64_t run = s[tid] - sum;
    for (int i = lo; i < hi; ++i) { const int64_t t = tops[i]; tops[i] = run; run += t; }
}

__global__ __launch_bounds__(1024) void scan64_add(int64_t* __restrict__ a, int64_t n, const int64_t* __restrict__ tops)
{
    const int64_t add = tops[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanBlock + 4 * threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) a[base + k] += add;
}

}  // namespace pg
