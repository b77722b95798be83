// Device kernels of libplatgpu (gfx950 only), all but the banded DP (pg_align.hip.h).
//
//   k_pack_codes  upload pass: 2-bit packed bases of haplotypes and reads
//   k_vote4       7-mer anchoring (hash_sequence + map_read of the reference)
//   k_genotype    per-genotype log-likelihood sums
//   k_pk_rate     issue-rate probe of the packed int16 VALU instructions
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

constexpr int kHashSize = 16384;     // platypus/calign.pyx:12
constexpr int kHashNucs = 7;         // platypus/calign.pyx:13
constexpr uint32_t kPosInf = 0x2000; // platypus/c/align.c:28
constexpr uint32_t kPosInf2 = 0x20002000u;

struct WaveDesc {   // one wavefront of the alignment kernel: up to 64 pairs of equal read length
    int32_t first;  // first slot in the (length-sorted) pair order
    int32_t count;  // 1..64
    int32_t len;    // read length shared by the wave
    int32_t pad;
};

// ---------------------------------------------------------------------------
// packed int16 helpers (two 16-bit lanes per 32-bit register)
// ---------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_i(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// 7-mer code of platypus/calign.pyx:81-96: per base c = ch & 7, 7 -> 2, two low bits kept.
__device__ __forceinline__ uint32_t nuc_code(uint32_t ch) {
    uint32_t c = ch & 7u;
    c = (c == 7u) ? 2u : c;
    return c & 3u;
}
// ---------------------------------------------------------------------------
// 2-bit packing of bases for the 7-mer hash: code(p) of platypus/calign.pyx:89-94
// sits at bits [2(p%16), 2(p%16)+2) of word p/16.  Row w of sequence s starts at
// word off2[s]; rows end with two zero words, so the 64-bit window of any 7-mer
// start can be read without a bounds test.
// ---------------------------------------------------------------------------
// one row: words first, first + step, ... of the nw words of a sequence of len bytes at seq
__device__ __forceinline__ void pack_codes_row(const uint8_t* __restrict__ seq, int len, uint32_t* __restrict__ dst, int nw,
                                               int first, int step)
{
    for (int w = first; w < nw; w += step) {
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = w * 16 + j;
            if (p < len) bits |= nuc_code(seq[p]) << (2 * j);
        }
        dst[w] = bits;
    }
}

__global__ __launch_bounds__(256) void k_pack_codes(const uint8_t* __restrict__ seq,
                                                    const int64_t* __restrict__ soff,
                                                    const int64_t* __restrict__ off2,
                                                    uint32_t* __restrict__ dst, int n_seq)
{
    const int wave = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int n_waves = (int)(gridDim.x * 4u);
    for (int s = wave; s < n_seq; s += n_waves) {
        const int64_t s0 = soff[s];
        const int64_t d0 = off2[s];
        pack_codes_row(seq + s0, (int)(soff[s + 1] - s0), dst + d0, (int)(off2[s + 1] - d0), lane, 64);
    }
}

// Bases in BAM's 4-bit code ("=ACMGRSVTWYHKDBN", bam.h: bam_nt16_rev_table; the high half of a byte first) to one byte
// each: dst[k] = base of nibble first + k of src.  Two bases per thread and pass.
__global__ __launch_bounds__(256) void k_unpack_nibbles(const uint8_t* __restrict__ src, int first, int64_t n, uint8_t* __restrict__ dst)
{
    const uint64_t table_lo = 0x565352474d43413dull, table_hi = 0x4e42444b48595754ull;   // "=ACMGRSV" "TWYHKDBN", first letter in the low byte
    for (int64_t k = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x); k < n; k += 2 * (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (k + j >= n) break;
            const int64_t nib = first + k + j;
            const uint32_t b = src[nib >> 1];
            const uint32_t code = (nib & 1) ? (b & 15u) : (b >> 4);
            dst[k + j] = (uint8_t)((code < 8 ? table_lo >> (8 * code) : table_hi >> (8 * (code - 8))) & 0xffu);
        }
    }
}

// One piece of a haplotype put together on the device (platypus/chaplotype.pyx:283-353, getMutatedSequence): `len`
// bytes to offset `dst` of the haplotype, from byte `src` of the reference stretches, or -- src < 0 -- from byte ~src
// of the variants' added sequences.
struct HapPiece { int32_t src, dst, len; };

// haplotype i = its pieces [piece_off[i], piece_off[i + 1]) one after the other, at raw + soff[i]; one wavefront each
__global__ __launch_bounds__(256) void k_build_haps(const uint8_t* __restrict__ ref, const uint8_t* __restrict__ added,
                                                    const HapPiece* __restrict__ pieces, const int32_t* __restrict__ piece_off,
                                                    const int64_t* __restrict__ soff, uint8_t* __restrict__ raw, int n_haps)
{
    const int wave = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int n_waves = (int)(gridDim.x * 4u);
    for (int h = wave; h < n_haps; h += n_waves) {
        uint8_t* out = raw + soff[h];
        const int p1 = piece_off[h + 1];
        for (int p = piece_off[h]; p < p1; ++p) {
            const HapPiece pc = pieces[p];
            const uint8_t* from = pc.src >= 0 ? ref + pc.src : added + ~pc.src;
            for (int j = lane; j < pc.len; j += 64) out[pc.dst + j] = from[j];
        }
    }
}

// the 14 bits of the 7-mer starting at base i of a packed row: a relabelling of
// the hash of platypus/calign.pyx:81-96 (first base in the low bits instead of
// the high ones), used for both the table and the look-ups
__device__ __forceinline__ uint32_t hash7_packed(const uint32_t* __restrict__ row, int i) {
    const uint32_t w0 = row[i >> 4], w1 = row[(i >> 4) + 1];
    const unsigned long long v = ((unsigned long long)w1 << 32) | w0;
    return (uint32_t)(v >> (2 * (i & 15))) & 0x3fffu;
}

struct VoteHap {      // one indel-bearing haplotype of k_vote4 (made once per haplotype upload)
    int32_t hap, len;
    int64_t hap2off;  // word offset of the haplotype's packed row
};

struct VoteRec {      // one (read, indel haplotype) pair of k_vote4, built by k_prep_scatter
    int32_t pair;     // index into the pair list
    int32_t rlen;     // read length
    int64_t r2off;    // word offset of the read's packed row
};

// ---------------------------------------------------------------------------
// vote_read: one wavefront maps one read against the unique-7-mer table of a
// haplotype in LDS (16384 x u16: i+1 unique start, 0xffff absent or repeated (0 also reads as absent);
// platypus/calign.pyx:99-158) in three sweeps over its votes, none of which
// depends on the order in which lanes reach LDS:
//   A  atomic +1 on the vote counter of every hit position
//   B  maximum final count
//   C  backwards over the votes: a position holding the maximum is met first at
//      its LAST vote -- the vote that brought it to the maximum; the reference's
//      winner, the first position to reach the final maximum, is therefore the
//      one met last.  Counters are cleared on the way.
// The packed read sits in a private LDS row of the wavefront.  k_vote4 uses it
// for the reads its majority test cannot decide.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash7_lds(const uint32_t* row, int i) {
    const uint32_t w0 = row[i >> 4], w1 = row[(i >> 4) + 1];
    return __builtin_amdgcn_alignbit(w1, w0, 2 * (i & 15)) & 0x3fffu;
}

// maximum of non-negative values over the 64 lanes, in registers (DPP), result in every lane
__device__ __forceinline__ int wave_max_nonneg(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));   // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true));   // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true));   // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

// The three sweeps for one read.  KEEP: the read has at most 4 x 64 votes and
// their positions stay in registers between the sweeps; otherwise every sweep
// looks its votes up again.
template <bool KEEP>
__device__ __forceinline__ int vote_read(const uint16_t* tab, const uint32_t* row, uint16_t* cnt16,
                                         uint32_t* cnt32, int rl, int nv, int n_chunks, int lane)
{
    // vote of read offset i: position pos (calign.pyx:144-147) and counter slot idx
    auto vote = [&](int i, int& pos, int& idx) -> bool {
        pos = 0; idx = 0;
        if (i >= nv) return false;
        const int pv = (int16_t)tab[hash7_lds(row, i)];
        pos = pv - 1 - i;
        idx = pos + rl;
        return pv > 0;
    };
    int kpos[4], kidx[4];
    bool kval[4];
    if (KEEP) {
#pragma unroll
        for (int c = 0; c < 4; ++c) kval[c] = c < n_chunks && vote(c * 64 + lane, kpos[c], kidx[c]);
    }
    if (KEEP) {
        // Shortcut without counters: a position named by more than half of all votes is the
        // only one that can hold the maximum, so it wins whatever the order of the votes.
        // Tried for the positions of the first and of the last vote of the read.
        unsigned long long vm[4];
        int total = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) { vm[c] = __ballot(kval[c]); total += __popcll(vm[c]); }
        if (total == 0) return 0;                // calign.pyx:137: nothing votes
        int cf = 0, cl = 0;
#pragma unroll
        for (int c = 3; c >= 0; --c) if (vm[c]) cf = c;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (vm[c]) cl = c;
        const unsigned long long mf = cf == 0 ? vm[0] : cf == 1 ? vm[1] : cf == 2 ? vm[2] : vm[3];
        const unsigned long long ml = cl == 0 ? vm[0] : cl == 1 ? vm[1] : cl == 2 ? vm[2] : vm[3];
        const int lf = __ffsll((long long)mf) - 1, ll = 63 - __clzll((long long)ml);
        const int posf = cf == 0 ? kpos[0] : cf == 1 ? kpos[1] : cf == 2 ? kpos[2] : kpos[3];
        const int posl = cl == 0 ? kpos[0] : cl == 1 ? kpos[1] : cl == 2 ? kpos[2] : kpos[3];
        const int cand[2] = {__builtin_amdgcn_readlane(posf, lf), __builtin_amdgcn_readlane(posl, ll)};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int n = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) n += __popcll(__ballot(kval[c] && kpos[c] == cand[k]));
            if (2 * n > total) return cand[k];
        }
    }
    // A: count.  Most votes of a chunk name the same position: the lanes that agree with
    // the chunk's first vote are added as one number, only the others go one by one.
    auto sweep_a = [&](bool valid, int pos, int idx) {
        const unsigned long long any = __ballot(valid);
        if (any) {
            const int lead = __ffsll((long long)any) - 1;
            const int p0 = __builtin_amdgcn_readlane(pos, lead);
            const bool same = valid && pos == p0;
            const unsigned long long m = __ballot(same);
            if (lane == lead) atomicAdd(&cnt32[idx >> 1], (uint32_t)__popcll(m) << ((idx & 1) * 16));
            if (valid && !same) atomicAdd(&cnt32[idx >> 1], 1u << ((idx & 1) * 16));
        }
    };
    if (KEEP) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < n_chunks) sweep_a(kval[c], kpos[c], kidx[c]);
    } else {
        for (int c = 0; c < n_chunks; ++c) {
            int pos, idx;
            const bool v = vote(c * 64 + lane, pos, idx);
            sweep_a(v, pos, idx);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // B: maximum final count
    int cmax = 0;
    if (KEEP) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < n_chunks && kval[c]) cmax = max(cmax, (int)cnt16[kidx[c]]);
    } else {
        for (int c = 0; c < n_chunks; ++c) {
            int pos, idx;
            if (vote(c * 64 + lane, pos, idx)) cmax = max(cmax, (int)cnt16[idx]);
        }
    }
    cmax = wave_max_nonneg(cmax);
    // C: backwards; the position met last among those at the maximum wins
    int winner = 0;                              // calign.pyx:137
    auto sweep_c = [&](bool valid, int pos, int idx) {
        const bool at_max = valid && (int)cnt16[idx] == cmax;
        unsigned long long mask = __ballot(at_max);
        while (mask) {
            const int top = 63 - __clzll((long long)mask);
            winner = __builtin_amdgcn_readlane(pos, top);
            const bool same = at_max && pos == winner;
            if (same) cnt16[idx] = 0;            // earlier votes of this position no longer show the maximum
            mask &= ~__ballot(same);
        }
        __builtin_amdgcn_wave_barrier();
        if (valid) cnt16[idx] = 0;
        __builtin_amdgcn_wave_barrier();
    };
    if (KEEP) {
#pragma unroll
        for (int c = 3; c >= 0; --c) if (c < n_chunks) sweep_c(kval[c], kpos[c], kidx[c]);
    } else {
        for (int c = n_chunks - 1; c >= 0; --c) {
            int pos, idx;
            const bool valid = vote(c * 64 + lane, pos, idx);
            sweep_c(valid, pos, idx);
        }
    }
    __builtin_amdgcn_wave_barrier();
    return winner;
}

// ---------------------------------------------------------------------------
// k_vote4: the same anchoring with four LANES per read.  The reference's
// winner, the first position to reach the final maximum count
// (platypus/calign.pyx:148-155), is decided without counters from the counts of
// three candidate positions: the one named by the first vote among the read's
// first eight 7-mers, by the last vote among its last eight, and by the first
// vote among eight 7-mers from its middle (a read that carries an indel relative
// to this haplotype splits its votes between two positions, one with an indel of
// its own on top between three).  With n_k the candidates' counts and rest the
// votes for any other position (an upper bound of every other count):
//   max n_k > rest, one candidate holds it    that position is the only maximum and wins
//                                             whatever the order of the votes
//   max n_k > rest, several hold it           each reaches the final maximum with its last
//                                             vote; the reference keeps the position that
//                                             gets there first (its update needs count >
//                                             maxcount): the one whose last vote comes earliest
// Anything else (about one read in a thousand) is mapped by the whole wavefront with
// the counting sweeps of vote_read.  The four lanes of a read take its 16-base groups
// in turn -- per 7-mer a shift, a mask, one LDS look-up, a comparison for "votes at all"
// and one per candidate -- and combine their counts with quad shuffles.
// Global memory is off the critical path: the words of round r + 1's reads and the
// records of round r + 2 are requested before round r is counted, and the first two
// rounds' words are requested before the table is built.
// LDS: table (32 KiB) | per wavefront: counters u16[cnt_stride] | packed read u32[row_words].
// ---------------------------------------------------------------------------
__device__ __forceinline__ int quad_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ int quad_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
    return v;
}

// first 7-mer of the window in which lane `part` of a read's quad looks for its candidate
__device__ __forceinline__ int vote_window(int part, int nv) {
    const int s = part == 0 ? 0 : part == 1 ? nv - 8 : nv >> 1;
    return s > 0 ? s : 0;
}

// what one lane of a read's quad fetches for a round: the word pairs of its groups part, part + 4,
// part + 8, part + 12 (reads of up to 263 bases need no more) and the three words that hold the
// eight 7-mers of its candidate window
struct VoteWords {
    uint2 g[4];
    uint32_t c[3];
};

__device__ __forceinline__ void vote_fetch(VoteWords& Q, const uint32_t* __restrict__ rw, int part, int nv, int n_groups)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        Q.g[j] = make_uint2(0, 0);
        if (part + 4 * j < n_groups) __builtin_memcpy(&Q.g[j], rw + part + 4 * j, 8);
    }
    Q.c[0] = Q.c[1] = Q.c[2] = 0;
    if (part < 3 && nv > 0) __builtin_memcpy(Q.c, rw + (vote_window(part, nv) >> 4), 12);
}

#ifndef PG_VOTE_ABLATE
#define PG_VOTE_ABLATE 0      // measurement builds only: 1 stop after the table build, 2 no counting sweeps for undecided reads
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_vote4(const uint32_t* __restrict__ hap2,
                                               const uint32_t* __restrict__ read2,
                                               const VoteHap* __restrict__ vh,
                                               const int32_t* __restrict__ rec_start,
                                               const VoteRec* __restrict__ vh_recs,
                                               const int32_t* __restrict__ status,
                                               int32_t* __restrict__ anchor, int cnt_stride, int row_words)
{
    // records [rec_start[b], rec_start[b + 1]) belong to indel-bearing haplotype b; a haplotype without
    // pairs in this list has nothing to do, and nothing runs on a list with an index out of range
    const int64_t q0 = rec_start[blockIdx.x], q_end = rec_start[blockIdx.x + 1];
    if (q0 == q_end || status[0] != 0x7fffffff) return;
    extern __shared__ uint32_t lds[];
    uint16_t* tab = reinterpret_cast<uint16_t*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_thr = blockDim.x, n_wav = n_thr >> 6;
    const int part = tid & 3, slot = tid >> 2, n_slots = n_thr >> 2;      // four lanes per read
    uint16_t* cnt16 = tab + kHashSize + wave * cnt_stride;
    uint32_t* cnt32 = reinterpret_cast<uint32_t*>(cnt16);
    uint32_t* row = lds + (kHashSize + n_wav * cnt_stride) / 2 + wave * row_words;
    const VoteHap H = vh[blockIdx.x];                 // one 16-byte descriptor: no chain of dependent loads
    const uint32_t* hw = hap2 + H.hap2off;
    const int L = H.len;

    // records of the first two rounds and this thread's first four 7-mers of the haplotype
    const VoteRec none = {0, 0, 0};
    VoteRec rec = q0 + slot < q_end ? vh_recs[q0 + slot] : none;
    VoteRec rec_next = q0 + n_slots + slot < q_end ? vh_recs[q0 + n_slots + slot] : none;
    const int nk = L - kHashNucs;               // last 7-mer is left out, calign.pyx:114
    uint32_t hk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { const int i = tid + m * n_thr; hk[m] = i < nk ? hash7_packed(hw, i) : 0u; }
    auto votes_of = [&](const VoteRec& r) { return r.rlen > kHashNucs ? r.rlen - kHashNucs : 0; };   // calign.pyx:143
    // the words of both rounds are on their way while the table is built
    VoteWords Q, Q_next;
    vote_fetch(Q, read2 + rec.r2off, part, votes_of(rec), (votes_of(rec) + 15) >> 4);
    vote_fetch(Q_next, read2 + rec_next.r2off, part, votes_of(rec_next), (votes_of(rec_next) + 15) >> 4);

    {   // the table starts empty (platypus/calign.pyx:105; "no start" is 0xffff here, as is "repeated")
        uint4* z = reinterpret_cast<uint4*>(lds);
        for (int i = tid; i < (kHashSize / 8); i += n_thr) z[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    bool counters_ready = false;        // a wavefront's vote counters are zeroed when it first needs them
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) { const int i = tid + m * n_thr; if (i < nk) tab[hk[m]] = (uint16_t)(i + 1); }
    for (int i = tid + 4 * n_thr; i < nk; i += n_thr) tab[hash7_packed(hw, i)] = (uint16_t)(i + 1);
    __syncthreads();
    // some other start owns the slot: the 7-mer is repeated
#pragma unroll
    for (int m = 0; m < 4; ++m) { const int i = tid + m * n_thr; if (i < nk && tab[hk[m]] != (uint16_t)(i + 1)) tab[hk[m]] = 0xffffu; }
    for (int i = tid + 4 * n_thr; i < nk; i += n_thr) {
        const uint32_t k = hash7_packed(hw, i);
        if (tab[k] != (uint16_t)(i + 1)) tab[k] = 0xffffu;
    }
    __syncthreads();
    if (PG_VOTE_ABLATE == 1) { if (Q.g[0].x == 0x12345u && Q_next.g[0].x == 0x54321u) anchor[0] = 0; return; }

    const uint8_t* tab_bytes = reinterpret_cast<const uint8_t*>(tab);
    for (int64_t qb = q0; qb < q_end; qb += n_slots) {
        const bool mine = qb + slot < q_end;
        const int nv = mine ? votes_of(rec) : 0;
        const uint32_t* rw = read2 + rec.r2off;
        const int n_groups = (nv + 15) >> 4;
        // request for the round after the next
        const VoteRec rec_after = qb + 2 * (int64_t)n_slots + slot < q_end ? vh_recs[qb + 2 * (int64_t)n_slots + slot] : none;
        const VoteWords Qc = Q;
        Q = Q_next;

        // candidates: lanes 0, 1, 2 of the quad look at eight 7-mers from the start, the end and the middle
        int c = 0x40000000 + part;                                               // matches no position
        {
            const int s0 = vote_window(part, nv), w0 = s0 >> 4;
            uint32_t pvs[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int bit = 2 * (s0 + t - 16 * w0);                          // 0 .. 44
                const uint32_t lo = bit < 32 ? Qc.c[0] : Qc.c[1], hi = bit < 32 ? Qc.c[1] : Qc.c[2];
                const uint32_t off = (__builtin_amdgcn_alignbit(hi, lo, bit & 31) << 1) & 0x7ffeu;
                pvs[t] = *reinterpret_cast<const uint16_t*>(tab_bytes + off);
            }
            int first = c, last = c;
            bool found = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool valid = pvs[t] != 0xffffu && s0 + t < nv && part < 3;
                const int pos = (int)pvs[t] - 1 - (s0 + t);
                first = valid && !found ? pos : first;
                last = valid ? pos : last;
                found |= valid;
            }
            c = part == 1 ? last : first;
        }
        const int c1 = __builtin_amdgcn_update_dpp(0, c, 0x00, 0xf, 0xf, false);   // quad_perm [0,0,0,0]
        int c2 = __builtin_amdgcn_update_dpp(0, c, 0x55, 0xf, 0xf, false);         // quad_perm [1,1,1,1]
        int c3 = __builtin_amdgcn_update_dpp(0, c, 0xAA, 0xf, 0xf, false);         // quad_perm [2,2,2,2]
        if (c2 == c1) c2 = 0x40000001;                                             // every position counted once
        if (c3 == c1 || c3 == c2) c3 = 0x40000002;

        // Counts over this lane's 16-base groups: part, part + 4, ...  A vote of 7-mer i for position c is a
        // table value of c + 1 + i (calign.pyx:144-147).  (An absent or repeated 7-mer reads 0xffff, which
        // is no c + 1 + i: positions are below 16384.)
        uint32_t total = 0, n1 = 0, n2 = 0, n3 = 0;
        const uint32_t e1 = (uint32_t)c1 + 1u, e2 = (uint32_t)c2 + 1u, e3 = (uint32_t)c3 + 1u;
        auto count_group = [&](const uint2 w, int g, bool partial) {
            uint32_t pv[16];                          // all 16 look-ups first: they are independent
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t off = (j == 0 ? w.x << 1 : __builtin_amdgcn_alignbit(w.y, w.x, 2 * j - 1)) & 0x7ffeu;
                pv[j] = *reinterpret_cast<const uint16_t*>(tab_bytes + off);
            }
            const uint32_t g1 = e1 + 16u * (uint32_t)g, g2 = e2 + 16u * (uint32_t)g, g3 = e3 + 16u * (uint32_t)g;
            const int jmax = nv - 16 * g;             // 7-mers of this group that exist
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t v = pv[j];
                if (partial && j >= jmax) v = 0xffffu;
                const uint32_t d = v - (uint32_t)j;
                total += v != 0xffffu;
                n1 += d == g1;
                n2 += d == g2;
                n3 += d == g3;
            }
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // all groups of a step are complete for reads of one length, except in the last step
            const int g = part + 4 * j;
            const bool have = g < n_groups, partial = have && 16 * g + 16 > nv;
            if (__any(partial)) {
                if (have) count_group(Qc.g[j], g, true);
            } else if (have) {
                count_group(Qc.g[j], g, false);
            }
        }
        for (int g = part + 16; g < n_groups; g += 4) {               // reads beyond 263 bases
            uint2 w;
            __builtin_memcpy(&w, rw + g, 8);
            count_group(w, g, true);
        }
        total = (uint32_t)quad_sum((int)total); n1 = (uint32_t)quad_sum((int)n1);
        n2 = (uint32_t)quad_sum((int)n2); n3 = (uint32_t)quad_sum((int)n3);
        const uint32_t rest = total - n1 - n2 - n3;
        const uint32_t top = max(n1, max(n2, n3));
        int winner = 0;                                               // calign.pyx:137: nothing votes
        bool undecided = false, tie = false;
        if (total > 0) {
            if (top > rest) {
                const int holders = (n1 == top) + (n2 == top) + (n3 == top);
                winner = n1 == top ? c1 : n2 == top ? c2 : c3;
                tie = holders > 1;
                if (tie && n_groups > 16) { tie = false; undecided = true; }
            } else {
                undecided = true;
            }
        }
        if (__any(tie)) {
            // several candidates reach the final maximum: the one whose last vote comes first got there first
            int last1 = -1, last2 = -1, last3 = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = part + 4 * j;
                if (tie && g < n_groups) {
                    const uint2 w = Qc.g[j];
                    for (int k = 0; k < 16 && 16 * g + k < nv; ++k) {
                        const uint32_t off = (k == 0 ? w.x << 1 : __builtin_amdgcn_alignbit(w.y, w.x, 2 * k - 1)) & 0x7ffeu;
                        const uint32_t v = *reinterpret_cast<const uint16_t*>(tab_bytes + off);
                        const uint32_t i = 16u * (uint32_t)g + (uint32_t)k;
                        if (v == e1 + i) last1 = (int)i;
                        if (v == e2 + i) last2 = (int)i;
                        if (v == e3 + i) last3 = (int)i;
                    }
                }
            }
            last1 = quad_max(last1); last2 = quad_max(last2); last3 = quad_max(last3);
            if (tie) {
                int best = 0x7fffffff;
                if (n3 == top) { best = last3; winner = c3; }
                if (n2 == top && last2 < best) { best = last2; winner = c2; }
                if (n1 == top && last1 < best) { best = last1; winner = c1; }
            }
        }
        if (mine && part == 0 && !undecided) anchor[rec.pair] = winner - 8 > 0 ? winner - 8 : 0;   // calign.pyx:67

        // reads these counts do not decide: the wavefront maps them one at a time
        unsigned long long todo = __ballot(undecided && part == 0);
        if (PG_VOTE_ABLATE == 2) todo = 0;
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int pair = __builtin_amdgcn_readlane(rec.pair, j);
            const int rlj = __builtin_amdgcn_readlane(rec.rlen, j);
            const int nvj = rlj - kHashNucs;
            if (!counters_ready) {
                for (int i = lane; i < cnt_stride / 2; i += 64) cnt32[i] = 0;      // cnt_stride is even
                counters_ready = true;
            }
            if (nvj <= 256) {
                // the read's words are in the registers of its four lanes
                if ((lane >> 2) == (j >> 2)) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (part + 4 * k < n_groups) { row[part + 4 * k] = Qc.g[k].x; row[part + 4 * k + 1] = Qc.g[k].y; }
                }
            } else {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rec.r2off, j);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)rec.r2off >> 32), j);
                const int64_t r2 = (int64_t)(((uint64_t)hi << 32) | lo);
                const int nw = (rlj + 15) / 16 + 1;
                for (int w = lane; w < nw; w += 64) row[w] = read2[r2 + w];
            }
            __builtin_amdgcn_wave_barrier();
            const int n_chunks = nvj > 0 ? (nvj + 63) >> 6 : 0;
            int win;
            if (n_chunks <= 4) win = vote_read<true>(tab, row, cnt16, cnt32, rlj, nvj, n_chunks, lane);
            else               win = vote_read<false>(tab, row, cnt16, cnt32, rlj, nvj, n_chunks, lane);
            if (lane == 0) anchor[pair] = win - 8 > 0 ? win - 8 : 0;
            __builtin_amdgcn_wave_barrier();
        }
        // rotate: the next round's record is known, its words were requested a round ago; request the words
        // of the round after it
        rec = rec_next;
        rec_next = rec_after;
        vote_fetch(Q_next, read2 + rec_next.r2off, part, votes_of(rec_next), (votes_of(rec_next) + 15) >> 4);
    }
}

// ---------------------------------------------------------------------------
// k_hap_unique (upload pass, once per haplotype): which starts of an indel-bearing
// haplotype own a 7-mer that occurs exactly once among the starts the reference
// hashes (0 .. len - 8, platypus/calign.pyx:114-121) -- the entries of the
// reference's table that are > 0.  Written as a row beside the haplotype's 2-bit
// packed bases, in the same layout: bit 2 (p % 16) of word p / 16.
// One workgroup per haplotype; the 16384-entry table of the reference in LDS.
// ---------------------------------------------------------------------------
// one haplotype by one workgroup of 256 threads: hw its packed row, U / Ubits the rows to write
__device__ __forceinline__ void hap_unique_one(const uint32_t* __restrict__ hw, int len, uint32_t* __restrict__ U,
                                               uint32_t* __restrict__ Ubits)
{
    __shared__ uint16_t tab[kHashSize];
    const int tid = threadIdx.x;
    const int nk = len - kHashNucs;
    {
        uint4* z = reinterpret_cast<uint4*>(tab);
        for (int i = tid; i < kHashSize / 8; i += 256) z[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    __syncthreads();
    for (int i = tid; i < nk; i += 256) tab[hash7_packed(hw, i)] = (uint16_t)(i + 1);
    __syncthreads();
    for (int i = tid; i < nk; i += 256) {               // some other start owns the slot: the 7-mer is repeated
        const uint32_t k = hash7_packed(hw, i);
        if (tab[k] != (uint16_t)(i + 1)) tab[k] = 0xffffu;
    }
    __syncthreads();
    const int nw = (len + 15) / 16 + 2;
    for (int w = tid; w < nw; w += 256) {
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = 16 * w + k;
            if (i < nk && tab[hash7_packed(hw, i)] == (uint16_t)(i + 1)) bits |= 1u << (2 * k);
        }
        U[w] = bits;
    }
    // ... and which 7-mer VALUES are owned by a unique start: one bit per value, 512 words per haplotype
    for (int w = tid; w < kHashSize / 32; w += 256) {
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) bits |= (uint32_t)(tab[32 * w + k] != 0xffffu) << k;
        Ubits[w] = bits;
    }
}

__global__ __launch_bounds__(256) void k_hap_unique(const uint32_t* __restrict__ hap2, const VoteHap* __restrict__ vh,
                                                    uint32_t* __restrict__ hapU, uint32_t* __restrict__ hapUbits)
{
    const VoteHap H = vh[blockIdx.x];
    hap_unique_one(hap2 + H.hap2off, H.len, hapU + H.hap2off, hapUbits + (size_t)blockIdx.x * (kHashSize / 32));
}

// ---------------------------------------------------------------------------
// k_vote5: the anchoring of platypus/calign.pyx:99-158 without the position table.
//
// A vote of read offset i for haplotype position c exists exactly when the read's
// 7-mer at i equals the haplotype's 7-mer at c + i and that start owns a unique
// 7-mer (k_hap_unique).  For a GIVEN c the votes of sixteen read offsets are
// therefore one line of bit arithmetic on the 2-bit packed rows: xor of the read's
// words with the haplotype's words shifted to c, "equal" bits, runs of seven
// (three and-with-shift steps), and with the uniqueness row at the same shift,
// popcount.  Exact counts n_k for a few candidate positions decide a read as the
// reference does whenever the best of them beats every vote that is not theirs:
//   top = max n_k  >  rest  >=  any other position's count
// (ties between candidates: the one whose last vote comes first reached the final
// maximum first, platypus/calign.pyx:150-153).  rest is bounded two ways: an
// offset whose 7-mer EQUALS the haplotype's at a candidate's place votes there or,
// if that 7-mer is repeated, nowhere -- so only the other offsets can vote
// elsewhere; and only 7-mers owned by a unique start vote at all (one bit per
// 7-mer value, made at upload).  Candidates only have to be GOOD, not exact: they
// come from a small direct-mapped table of the haplotype's 7-mer starts (2048 x
// u16, last writer wins, never cleared; every hit is verified against the
// haplotype's rows), looked up for four 7-mers each from the read's start, middle
// and end (a read with an indel of its own against an indel-bearing haplotype has
// three stretches).
//
// ONE LANE PER PAIR.  A workgroup takes the next blockDim.x vote records (they
// are grouped by haplotype) and, four haplotypes at a time, puts their rows and
// candidate tables side by side in LDS; every lane whose record is on one of them
// then works alone: twelve look-ups, three exact counts over its read's groups.
// (Four lanes per pair with the groups dealt between them cost two to three times
// the instructions per pair: per-lane overheads, uneven shares, combining steps.)
// What stays undecided is taken by the whole wavefront, one read at a time:
//   a  every lane counts the positions its own offsets name; a position nobody
//      could name (its table slot was taken by another 7-mer) holds at most
//      `lost` votes, so the best count decides when it is above that
//   b  the count for EVERY position: largest count, earliest last vote
// Reads of up to 263 bases (sixteen 16-base groups, in registers); the host
// sends launches with longer reads to k_vote4.
// ---------------------------------------------------------------------------
constexpr int kCandSlots = 2048;
constexpr int kNoCand = 0x40000000;
constexpr int kVoteSlots = 4;        // haplotypes side by side in LDS

__device__ __forceinline__ uint32_t cand_slot(uint32_t key) { return (key ^ (key >> 6)) & (uint32_t)(kCandSlots - 1); }

// votes of read offsets 16 g .. 16 g + 15 (read words rl, rh = offsets 16 g .. 16 g + 31) for the position whose
// offset-16g base sits at padded row index s; result: bit 2 k set = offset 16 g + k votes (vm: offsets that exist).
// `matches`: offsets whose 7-mer equals the haplotype's there, unique or not.
__device__ __forceinline__ uint32_t vote_bits(const uint2* hu, uint32_t rl, uint32_t rh, int s, uint32_t vm, uint32_t& matches)
{
    const int wi = s >> 4;
    const uint32_t sh = 2u * ((uint32_t)s & 15u);
    const uint2 a = hu[wi], b = hu[wi + 1], c = hu[wi + 2];
    const uint32_t hl = __builtin_amdgcn_alignbit(b.x, a.x, sh), hh = __builtin_amdgcn_alignbit(c.x, b.x, sh);
    const uint32_t u = __builtin_amdgcn_alignbit(b.y, a.y, sh);
    const uint32_t xl = rl ^ hl, xh = rh ^ hh;
    const uint32_t el = ~(xl | (xl >> 1)), eh = ~(xh | (xh >> 1));           // even bits: bases equal
    const uint32_t a1l = el & __builtin_amdgcn_alignbit(eh, el, 2), a1h = eh & (eh >> 2);          // runs of 2
    const uint32_t a2l = a1l & __builtin_amdgcn_alignbit(a1h, a1l, 4), a2h = a1h & (a1h >> 4);    // runs of 4
    matches = a2l & __builtin_amdgcn_alignbit(a2h, a2l, 6) & vm;                                    // runs of 7
    return matches & u;
}

__device__ __forceinline__ uint32_t group_mask(int nv, int g) {           // offsets 16 g + k < nv
    const int left = nv - 16 * g;
    return left >= 16 ? 0x55555555u : (left <= 0 ? 0u : (0x55555555u & ((1u << (2 * left)) - 1u)));
}

// start of the haplotype's 7-mer `key` according to the candidate table; true if the haplotype really holds the
// 7-mer there and that start owns it alone
__device__ __forceinline__ bool cand_lookup(const uint2* hu, const uint16_t* ctab, int pad, int nk, uint32_t key, int& j)
{
    j = min((int)ctab[cand_slot(key)], nk - 1);                              // (a slot nobody wrote holds anything)
    const uint2 a = hu[pad + (j >> 4)], b = hu[pad + (j >> 4) + 1];
    const uint32_t sh = 2u * ((uint32_t)j & 15u);
    const uint32_t hk = __builtin_amdgcn_alignbit(b.x, a.x, sh) & 0x3fffu;
    return (hk == key) & (bool)((a.y >> sh) & 1u);
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_vote5(const uint32_t* __restrict__ hap2, const uint32_t* __restrict__ hapU,
                                               const uint32_t* __restrict__ hapUbits,
                                               const uint32_t* __restrict__ read2, const VoteHap* __restrict__ vh,
                                               const int32_t* __restrict__ rec_start, const VoteRec* __restrict__ vh_recs,
                                               int32_t* __restrict__ status, int32_t* __restrict__ anchor,
                                               int pad, int row_cap, int n_vote_haps)
{
    if (status[0] != 0x7fffffff) return;             // nothing runs on a list with an index out of range
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_thr = blockDim.x;
    const int64_t n_recs = rec_start[n_vote_haps];
    const int64_t r0 = (int64_t)blockIdx.x * n_thr, r1 = min(r0 + (int64_t)n_thr, n_recs);
    if (r0 >= n_recs) return;
    extern __shared__ uint32_t lds[];
    const int slot_words = 2 * row_cap + kCandSlots / 2;                     // a haplotype's rows {codes, unique} and its table
    uint32_t* rrow = lds + kVoteSlots * slot_words + wave * 20;              // [20] packed read of an undecided pair
    const int base = 16 * pad;                                               // padded row index of haplotype position 0

    // my record and my read
    const int64_t my = r0 + tid;
    const bool have = my < r1;
    const VoteRec rec = have ? vh_recs[my] : VoteRec{0, 0, 0};
    const int nv = rec.rlen > kHashNucs ? rec.rlen - kHashNucs : 0;          // calign.pyx:143
    const int n_groups = (nv + 15) >> 4;
    uint32_t rw[17];
    // the three words that hold the 7-mers looked at near the read's end (offsets nv - 16 .. nv - 1) and in its middle
    // (nv / 2 .. + 15): fetched on their own, a register array cannot be indexed by a lane's own number
    const int e0 = nv > 16 ? nv - 16 : 0, m0 = nv >> 1, we = e0 >> 4, wm = m0 >> 4;
    uint32_t ew[3] = {0u, 0u, 0u}, mw[3] = {0u, 0u, 0u};
    {
        const uint32_t* src = read2 + rec.r2off;
#pragma unroll
        for (int k = 0; k < 17; ++k) rw[k] = (have && k <= n_groups && nv > 0) ? src[k] : 0u;
        if (have && nv > 0) {                          // (rows end with two zero words: we + 2 <= n_groups + 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) { ew[k] = src[min(we + k, n_groups + 1)]; mw[k] = src[min(wm + k, n_groups + 1)]; }
        }
    }
    const int ng_max = wave_max_nonneg(n_groups);

    // first haplotype with a record at or behind r0: the last h with rec_start[h] <= r0
    int h0;
    {
        int lo = 0, hi = n_vote_haps;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rec_start[mid] <= r0) lo = mid; else hi = mid;
        }
        h0 = lo;
    }
    bool done = !have;
    for (int hb = h0; hb < n_vote_haps && rec_start[hb] < r1; hb += kVoteSlots) {
        if (hb != h0) __syncthreads();               // everybody has left the slots
        // ---- the next haplotypes' rows and candidate tables, a share of the workgroup's threads on each ----
        int n_live = 0;
#pragma unroll
        for (int k = 0; k < kVoteSlots; ++k)
            n_live += (int)(hb + k < n_vote_haps && rec_start[hb + k] < r1 && rec_start[hb + k + 1] > rec_start[hb + k] &&
                            rec_start[hb + k + 1] > r0);
        // (the haplotypes with records of mine are the first n_live or fewer of the four: one without records in
        // between takes a share of the threads for nothing, no more)
        const int per = n_thr / max(1, min(kVoteSlots, n_live == kVoteSlots ? kVoteSlots : n_live)), sl0 = tid / per;
        const int st = tid - sl0 * per;
        int sl = -1;                                   // the sl0-th live slot
        {
            int seen = 0;
#pragma unroll
            for (int k = 0; k < kVoteSlots; ++k) {
                const bool lv = hb + k < n_vote_haps && rec_start[hb + k] < r1 && rec_start[hb + k + 1] > rec_start[hb + k] &&
                                rec_start[hb + k + 1] > r0;
                if (lv && seen == sl0 && sl < 0) sl = k;
                seen += (int)lv;
            }
        }
        const bool live = sl >= 0;
        if (sl < 0) sl = 0;
        const int hs = hb + sl;
        {
            uint2* hu = reinterpret_cast<uint2*>(lds + sl * slot_words);
            if (live) {
                const VoteHap H = vh[hs];
                const uint32_t* hw = hap2 + H.hap2off;
                const uint32_t* uw = hapU + H.hap2off;
                const int nw = (H.len + 15) / 16 + 2;
                for (int w = st; w < row_cap; w += per) {
                    const int k = w - pad;
                    hu[w] = (k >= 0 && k < nw) ? make_uint2(hw[k], uw[k]) : make_uint2(0u, 0u);
                }
            }
        }
        __syncthreads();
        if (live) {
            const uint2* hu = reinterpret_cast<const uint2*>(lds + sl * slot_words);
            uint16_t* ctab = reinterpret_cast<uint16_t*>(lds + sl * slot_words + 2 * row_cap);
            const int nk = vh[hs].len - kHashNucs;
            for (int i = st; i < nk; i += per) {
                const uint2 a = hu[pad + (i >> 4)], b = hu[pad + (i >> 4) + 1];
                const uint32_t key = __builtin_amdgcn_alignbit(b.x, a.x, 2u * ((uint32_t)i & 15u)) & 0x3fffu;
                ctab[cand_slot(key)] = (uint16_t)i;
            }
        }
        __syncthreads();

        // ---- which of these haplotypes is mine? ----
        int myslot = -1;
#pragma unroll
        for (int k = 0; k < kVoteSlots; ++k) {
            const int h = hb + k;
            if (h < n_vote_haps && !done && my >= rec_start[h] && my < rec_start[h + 1]) myslot = k;
        }
        const bool active = myslot >= 0;
        if (!__any(active)) continue;
        const int slot_i = active ? myslot : 0, hm = hb + slot_i;
        const uint2* hu = reinterpret_cast<const uint2*>(lds + slot_i * slot_words);
        const uint16_t* ctab = reinterpret_cast<const uint16_t*>(lds + slot_i * slot_words + 2 * row_cap);
        const int nk = vh[min(hm, n_vote_haps - 1)].len - kHashNucs;
        if (active) done = true;

        // ---- candidates: four 7-mers each from the read's start, middle and end, five bases apart (a wrong base
        // spoils at most two of a window's four) ----
        int c1 = kNoCand, c2 = kNoCand, c3 = kNoCand;
        {
            auto key3 = [&](uint32_t lo, uint32_t hi, uint32_t top, int bit) -> uint32_t {     // bit: 0 .. 60
                const uint32_t x = bit < 32 ? lo : hi, y = bit < 32 ? hi : top;
                return __builtin_amdgcn_alignbit(y, x, bit & 31) & 0x3fffu;
            };
#pragma unroll
            for (int t = 3; t >= 0; --t) {                 // start: offsets 0, 5, 10, 15; the smallest that names a position
                int j;
                const bool ok = cand_lookup(hu, ctab, pad, nk, key3(rw[0], rw[1], rw[2], 10 * t), j) & (5 * t < nv);
                c1 = ok ? j - 5 * t : c1;
            }
            const uint32_t e_lo = ew[0], e_hi = ew[1], e_top = ew[2], m_lo = mw[0], m_hi = mw[1], m_top = mw[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {                  // end: offsets nv - 16, - 11, - 6, - 1; the largest that names one
                const int i = e0 + 5 * t;
                int j;
                const bool ok = cand_lookup(hu, ctab, pad, nk, key3(e_lo, e_hi, e_top, 2 * (i - 16 * we)), j) & (i < nv);
                c2 = ok ? j - i : c2;
            }
#pragma unroll
            for (int t = 3; t >= 0; --t) {                 // middle: nv / 2 + 0, 5, 10, 15; the smallest
                const int i = m0 + 5 * t;
                int j;
                const bool ok = cand_lookup(hu, ctab, pad, nk, key3(m_lo, m_hi, m_top, 2 * (i - 16 * wm)), j) & (i < nv);
                c3 = ok ? j - i : c3;
            }
        }
        const bool has1 = active && c1 != kNoCand;
        const bool has2 = active && c2 != kNoCand && c2 != c1;
        const bool has3 = active && c3 != kNoCand && c3 != c1 && c3 != c2;
        const int s1 = base + (has1 ? c1 : 0), s2 = base + (has2 ? c2 : 0), s3 = base + (has3 ? c3 : 0);
        const bool any2 = __any(has2), any3 = __any(has3);

        // ---- exact counts of the candidates over the read's 16-offset groups.  Group g of a candidate needs the
        // haplotype words g, g + 1, g + 2 from its place on (same shift for every group) and the "equal" bits of two
        // read words: one new LDS word and one new xor per group and candidate, the rest slides ----
        int n1 = 0, n2 = 0, n3 = 0, cover = 0;
        {
            const uint2* hp1 = hu + (s1 >> 4); const uint2* hp2 = hu + (s2 >> 4); const uint2* hp3 = hu + (s3 >> 4);
            const uint32_t sh1 = 2u * ((uint32_t)s1 & 15u), sh2 = 2u * ((uint32_t)s2 & 15u), sh3 = 2u * ((uint32_t)s3 & 15u);
            uint2 A1 = hp1[0], B1 = hp1[1], A2 = hp2[0], B2 = hp2[1], A3 = hp3[0], B3 = hp3[1];
            auto eq = [](uint32_t x) -> uint32_t { return ~(x | (x >> 1)); };          // even bits: bases equal
            uint32_t el1 = eq(rw[0] ^ __builtin_amdgcn_alignbit(B1.x, A1.x, sh1));
            uint32_t el2 = eq(rw[0] ^ __builtin_amdgcn_alignbit(B2.x, A2.x, sh2));
            uint32_t el3 = eq(rw[0] ^ __builtin_amdgcn_alignbit(B3.x, A3.x, sh3));
            auto runs7 = [](uint32_t el, uint32_t eh) -> uint32_t {
                const uint32_t a1l = el & __builtin_amdgcn_alignbit(eh, el, 2), a1h = eh & (eh >> 2);          // runs of 2
                const uint32_t a2l = a1l & __builtin_amdgcn_alignbit(a1h, a1l, 4), a2h = a1h & (a1h >> 4);    // runs of 4
                return a2l & __builtin_amdgcn_alignbit(a2h, a2l, 6);                                          // runs of 7
            };
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (g < ng_max) {
                    const uint32_t vm = group_mask(nv, g);
                    uint32_t mm;
                    {
                        const uint2 C = hp1[g + 2];
                        const uint32_t eh = eq(rw[g + 1] ^ __builtin_amdgcn_alignbit(C.x, B1.x, sh1));
                        const uint32_t m = runs7(el1, eh) & (has1 ? vm : 0u);
                        n1 += __popc(m & __builtin_amdgcn_alignbit(B1.y, A1.y, sh1));
                        mm = m;
                        A1 = B1; B1 = C; el1 = eh;
                    }
                    if (any2) {
                        const uint2 C = hp2[g + 2];
                        const uint32_t eh = eq(rw[g + 1] ^ __builtin_amdgcn_alignbit(C.x, B2.x, sh2));
                        const uint32_t m = runs7(el2, eh) & (has2 ? vm : 0u);
                        n2 += __popc(m & __builtin_amdgcn_alignbit(B2.y, A2.y, sh2));
                        mm |= m;
                        A2 = B2; B2 = C; el2 = eh;
                    }
                    if (any3) {
                        const uint2 C = hp3[g + 2];
                        const uint32_t eh = eq(rw[g + 1] ^ __builtin_amdgcn_alignbit(C.x, B3.x, sh3));
                        const uint32_t m = runs7(el3, eh) & (has3 ? vm : 0u);
                        n3 += __popc(m & __builtin_amdgcn_alignbit(B3.y, A3.y, sh3));
                        mm |= m;
                        A3 = B3; B3 = C; el3 = eh;
                    }
                    cover += __popc(mm);
                }
            }
        }
        int rest = nv - cover;                                                // no other position can hold more votes
        const int top = max(n1, max(n2, n3));
        bool undecided = active && nv > 0;                                    // (nv == 0: nothing votes, calign.pyx:137,140)

        int winner = 0;
        bool tie = false;
        if (top > rest) {
            undecided = false;
            const int holders = (int)(has1 && n1 == top) + (int)(has2 && n2 == top) + (int)(has3 && n3 == top);
            winner = (has1 && n1 == top) ? c1 : (has2 && n2 == top) ? c2 : c3;
            tie = holders > 1;
        }
        if (__any(tie)) {
            // several candidates reach the final maximum: the one whose last vote comes first got there first
            int last1 = -1, last2 = -1, last3 = -1;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (g < ng_max) {
                    const uint32_t vm = tie ? group_mask(nv, g) : 0u;
                    uint32_t mm;
                    const uint32_t v1 = vote_bits(hu, rw[g], rw[g + 1], s1 + 16 * g, has1 ? vm : 0u, mm);
                    const uint32_t v2 = vote_bits(hu, rw[g], rw[g + 1], s2 + 16 * g, has2 ? vm : 0u, mm);
                    const uint32_t v3 = vote_bits(hu, rw[g], rw[g + 1], s3 + 16 * g, has3 ? vm : 0u, mm);
                    if (v1) last1 = 16 * g + ((31 - __clz((int)v1)) >> 1);
                    if (v2) last2 = 16 * g + ((31 - __clz((int)v2)) >> 1);
                    if (v3) last3 = 16 * g + ((31 - __clz((int)v3)) >> 1);
                }
            }
            if (tie) {
                int best = 0x7fffffff;
                if (has3 && n3 == top) { best = last3; winner = c3; }
                if (has2 && n2 == top && last2 < best) { best = last2; winner = c2; }
                if (has1 && n1 == top && last1 < best) { best = last1; winner = c1; }
            }
        }
        if (active && !undecided) anchor[rec.pair] = winner - 8 > 0 ? winner - 8 : 0;   // calign.pyx:67

        // ---- the wavefront takes the reads the candidates do not decide, one at a time ----
        unsigned long long todo = __ballot(undecided);
        while (todo) {
            const int jl = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int pair = __builtin_amdgcn_readlane(rec.pair, jl);
            const int nvj = __builtin_amdgcn_readlane(nv, jl);
            const int slj = __builtin_amdgcn_readlane(slot_i, jl);
            const int nkj = __builtin_amdgcn_readlane(nk, jl);
            const int hmj = __builtin_amdgcn_readlane(hm, jl);
            const int ngj = (nvj + 15) >> 4;
            if (lane == jl) {
#pragma unroll
                for (int k = 0; k < 17; ++k) rrow[k] = rw[k];
            }
            __builtin_amdgcn_wave_barrier();
            const uint2* huj = reinterpret_cast<const uint2*>(lds + slj * slot_words);
            const uint16_t* ctj = reinterpret_cast<const uint16_t*>(lds + slj * slot_words + 2 * row_cap);
            const uint32_t* ub = hapUbits + (size_t)hmj * (kHashSize / 32);
            // exact count and last vote of one position; key = count << 16 | (0xffff - last vote): the largest key is
            // the largest count and, among equal counts, the earliest last vote
            auto key_of = [&](int cc) -> uint32_t {
                int n = 0, last = -1;
                for (int g = 0; g < ngj; ++g) {
                    uint32_t mm;
                    const uint32_t v = vote_bits(huj, rrow[g], rrow[g + 1], base + cc + 16 * g, group_mask(nvj, g), mm);
                    n += __popc(v);
                    if (v) last = 16 * g + ((31 - __clz((int)v)) >> 1);
                }
                return n > 0 ? ((uint32_t)n << 16) | (uint32_t)(0xffff - last) : 0u;
            };
            // first the other bound on every other position's count: only 7-mers owned by a unique start vote at all
            {
                int tot = 0;
                for (int i = lane; i < nvj; i += 64) {
                    const uint32_t key = hash7_lds(rrow, i);
                    tot += (int)((ub[key >> 5] >> (key & 31u)) & 1u);
                }
                tot = wave_sum(tot);
                const int topj = __builtin_amdgcn_readlane(top, jl), sumj = __builtin_amdgcn_readlane(n1 + n2 + n3, jl);
                const int holders = (int)(has1 && n1 == top) + (int)(has2 && n2 == top) + (int)(has3 && n3 == top);
                const int holdj = __builtin_amdgcn_readlane(holders, jl);
                const int winj = __builtin_amdgcn_readlane((has1 && n1 == top) ? c1 : (has2 && n2 == top) ? c2 : c3, jl);
                if (tot == 0 || (topj > tot - sumj && holdj == 1)) {           // nothing votes: position 0 (calign.pyx:137)
                    const int w0 = tot == 0 ? 0 : winj;
                    if (lane == 0) anchor[pair] = w0 - 8 > 0 ? w0 - 8 : 0;
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
            }
            uint32_t best_key = 0;
            int best_c = 0, lost = 0;
            for (int i = lane; i < nvj; i += 64) {                             // a: the positions this lane's offsets name
                const uint32_t key = hash7_lds(rrow, i);
                if ((ub[key >> 5] >> (key & 31u)) & 1u) {
                    int jpos;
                    if (cand_lookup(huj, ctj, pad, nkj, key, jpos)) {
                        const uint32_t k2 = key_of(jpos - i);
                        if (k2 > best_key) { best_key = k2; best_c = jpos - i; }
                    } else {
                        ++lost;                                                // it votes, but for a position this table cannot name
                    }
                }
            }
            lost = wave_sum(lost);
            uint32_t wmax = (uint32_t)wave_max_nonneg((int)best_key);          // counts are below 2^15
            const bool full = (int)(wmax >> 16) <= lost && !(wmax == 0 && lost == 0);
            if (full) {                                                        // b: positions -(nvj - 1) .. nk - 1, all of them
                best_key = 0;
                best_c = 0;
                for (int cc = -(nvj - 1) + lane; cc < nkj; cc += 64) {
                    const uint32_t k2 = key_of(cc);
                    if (k2 > best_key) { best_key = k2; best_c = cc; }
                }
                wmax = (uint32_t)wave_max_nonneg((int)best_key);
            }
            const unsigned long long who = __ballot(best_key == wmax);
            const int win = wmax == 0 ? 0 : __builtin_amdgcn_readlane(best_c, __ffsll((long long)who) - 1);
            if (lane == 0) {
                anchor[pair] = win - 8 > 0 ? win - 8 : 0;
                atomicAdd(&status[full ? 4 : 3], 1);                           // ST_N_FULL / ST_N_NAMED of pg_prep.hip.h
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------
// Stand-alone forms of hash_sequence and map_read (platypus/calign.pyx:99-158)
// for callers that use them outside alignNoTraceback: the table in the
// reference's own layout (int32[16384], index = my_hash with the first base in
// the high bits, value i+1 / -1 / 0) and one read mapped against such a table.
// Compatibility paths, one small launch each; the alignment path itself uses
// k_vote4.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t my_hash_bytes(const uint8_t* p) {       // calign.pyx:81-96
    uint32_t h = 0;
#pragma unroll
    for (int i = 0; i < kHashNucs; ++i) h = (h << 2) + nuc_code(p[i]);
    return h;
}

__global__ __launch_bounds__(1024) void k_hash_sequence(const uint8_t* __restrict__ seq, int len, int32_t* __restrict__ table)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < kHashSize; i += 1024) table[i] = 0;               // calloc, calign.pyx:105
    __syncthreads();
    const int nk = len - kHashNucs;                                          // calign.pyx:114
    for (int i = tid; i < nk; i += 1024) table[my_hash_bytes(seq + i)] = i + 1;
    __syncthreads();
    for (int i = tid; i < nk; i += 1024) {
        const uint32_t k = my_hash_bytes(seq + i);
        if (table[k] != i + 1) table[k] = -1;                                // seen more than once, calign.pyx:118-119
    }
}

// The reference's loop as it stands (calign.pyx:143-155), one thread: the vote order decides ties.
__global__ void k_map_read(const uint8_t* __restrict__ read, int rlen, int seq_len,
                           const int32_t* __restrict__ table, int32_t* __restrict__ counts,
                           int32_t* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int maxcount = 0, maxpos = 0, bad = 0;
    for (int i = 0; i < rlen - kHashNucs; ++i) {
        int pos = table[my_hash_bytes(read + i)];
        if (pos > 0) {
            pos -= 1 + i;
            if (pos >= seq_len || pos + rlen < 0) { bad = 1; continue; }     // assertion at calign.pyx:148
            const int count = counts[pos + rlen] + 1;
            counts[pos + rlen] = count;
            if (count > maxcount) { maxcount = count; maxpos = pos; }
        }
    }
    out[0] = maxpos;
    out[1] = bad;
}

// ---------------------------------------------------------------------------
// k_genotype: out[g] = sum_r log(0.5*(L[b1+r] + L[b2+r])), read order
// (platypus/cgenotype.pyx:58-61).  One wavefront per genotype row: lanes take
// reads r = lane, lane+64, ...; partial sums are combined in a fixed tree, so
// the result is reproducible run to run (it differs from the strictly
// sequential sum by rounding only).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_genotype(const double* __restrict__ like,
                                                  const int64_t* __restrict__ b1,
                                                  const int64_t* __restrict__ b2,
                                                  const int32_t* __restrict__ nreads,
                                                  double* __restrict__ out, int64_t n_gen)
{
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (g >= n_gen) return;
    const int lane = threadIdx.x & 63;
    const double* l1 = like + b1[g];
    const double* l2 = like + b2[g];
    const int n = nreads[g];
    double s = 0.0;
    for (int r = lane; r < n; r += 64) s += log(0.5 * (l1[r] + l2[r]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[g] = s;
}

// ---------------------------------------------------------------------------
// k_pk_rate: issue-rate probe for the packed int16 VALU instructions the DP is
// made of (v_pk_add_u16 / v_pk_min_i16).  16 independent dependency chains per
// lane, 2 instructions per chain step; nothing touches memory inside the loop.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pk_rate(uint32_t* __restrict__ out, int iters, uint32_t b, uint32_t c)
{
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0x00010003u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = pk_min_i(pk_add(a[i], b), c);
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

// ---------------------------------------------------------------------------
// k_vop2_rate: issue-rate probe for the instruction forms the DP kernel executes,
// the 16-bit VOP2 instructions v_add_u16 / v_min_u16 with VGPR operands only.
// 16 independent chains per lane; the 16 additions are issued before the 16 minima
// that consume them, so no instruction reads a result still in flight.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vop2_rate(uint32_t* __restrict__ out, int iters, uint32_t b, uint32_t c)
{
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 3u + i;
    uint32_t vb = b, vc = c;
    asm volatile("" : "+v"(vb), "+v"(vc));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(vb));
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(vc));
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

}  // namespace pg
