// k_align_band: the banded affine-gap DP of platypus/c/align.c:18-235 with EIGHT LANES PER PAIR -- lane k holds
// the cells of diagonals 2k and 2k+1, as lane k of the reference's 8 x int16 vectors does -- for launches too
// small to fill the machine with k_align2's one lane per pair.  There a lone wavefront walks 2 x (len2 + 8)
// half-steps of eight cells each for its 64 pairs, one after the other (about 58 us for 150-base reads, however few
// pairs it holds); here the eight cells of a half-step are eight lanes, a wavefront holds eight pairs, and a
// window's pairs spread over many wavefronts on many CUs.  Same inputs (the element arrays of k_pack_hapx /
// k_pack_readx, the anchors of the vote kernel), same outputs, the same integers.
//
// Arithmetic as in k_align2: costs unscaled (the reference's are all multiples of 4), pos_inf 0x800, no value
// near 2^15, so 32-bit registers give what the reference's wrapping 16-bit lanes give.
// The reference's lane shifts (align.c:168,224-231) and its sliding windows (align.c:111-112,182-187) are DPP
// row shifts by one lane; what enters a window at its end lane comes from a "chunk" register that holds the next
// eight entries, one per lane, and is shifted towards that lane once per iteration.  Row shifts cross the border
// between two pairs' lane groups only into lanes that are overwritten (the entering lane) or into chunk lanes that
// are reloaded before they are used.
// One wavefront per workgroup; LDS holds the gap-open table only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_align.hip.h"

namespace pg {

__device__ __forceinline__ uint32_t lane_up(uint32_t old, uint32_t v) {      // lane k takes lane k - 1's v (row_shr:1)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x111, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t lane_down(uint32_t old, uint32_t v) {    // lane k takes lane k + 1's v (row_shl:1)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x101, 0xf, 0xf, false);
}

struct BandLane {
    uint32_t m1, i1, d1, m2, i2, d2;
    uint32_t e2;        // read window: code | quality << 16 of read position t - k
    uint32_t h1;        // haplotype window: code of slice position t + k (even half-step), t + 1 + k (odd)
    uint32_t g;         // gap-open window (align.c:100,185-187: it moves the other way)
    uint32_t cr, ch, cg;   // chunks: the next eight entries of the three windows
    uint32_t best;
};

// cost of align.c:158-162 with the 'N' limit of align.c:68,183, on the codes of pg_align.hip.h (SeqCodes): the read
// window holds code | quality << 16, the haplotype window the column's code; code & code is 0 for equal bases, at least
// 128 -- more than any quality -- for different ones and 1 for an 'N' column past the first eight under a read base
// that is no 'N'
__device__ __forceinline__ uint32_t band_cost(uint32_t e2, uint32_t q2, uint32_t h1) {
    return min(e2 & h1, q2);
}

// iteration t of align.c:106-232 for lane k of a pair
template <bool EDGE>
__device__ __forceinline__ void band_iter(BandLane& L, int t, int k, int len2, uint32_t ge, uint32_t np)
{
    const bool first = k == 0, last = k == 7;
    // every row shift is made by all lanes, then the entering lane takes its new entry
    // align.c:111-121: the read window takes position t
    const uint32_t e2_up = lane_up(L.e2, L.e2);
    L.e2 = first ? L.cr : e2_up;
    L.cr = lane_down(L.cr, L.cr);
    const uint32_t q2 = L.e2 >> 16;
    if (EDGE) { L.m1 = k == t ? 0u : L.m1; L.m2 = k == t ? 0u : L.m2; }   // align.c:127-128
    // ---- even half-step, align.c:129-173 ----
    L.m1 = min(L.m1, min(L.i1, L.d1));
    if (EDGE) L.best = t - len2 == k ? min(L.best, L.m1) : L.best;        // align.c:134-156
    L.m1 += band_cost(L.e2, q2, L.h1);
    const uint32_t dnew = min(L.d2 + ge, min(L.m2, L.i2) + L.g);
    L.i1 = min(L.i2 + ge, L.m2 + L.g) + np;
    const uint32_t d_up = lane_up(dnew, dnew);
    L.d1 = first ? kInf : d_up;                                           // align.c:168
    // ---- align.c:180-187: the haplotype and gap-open windows take slice position t + 8 ----
    const uint32_t h_down = lane_down(L.h1, L.h1);
    L.h1 = last ? L.ch : h_down;
    L.ch = lane_up(L.ch, L.ch);
    const uint32_t g_up = lane_up(L.g, L.g);
    L.g = first ? L.cg : g_up;
    L.cg = lane_down(L.cg, L.cg);
    // ---- odd half-step, align.c:190-231 ----
    L.m2 = min(L.m2, min(L.i2, L.d2));
    if (EDGE) L.best = t - len2 == k ? min(L.best, L.m2) : L.best;        // align.c:195-217
    L.m2 += band_cost(L.e2, q2, L.h1);
    L.d2 = min(L.d1 + ge, min(L.m1, L.i1) + L.g);
    const uint32_t inew = min(L.i1 + ge, L.m1 + L.g) + np;
    const uint32_t i_down = lane_down(inew, inew);
    L.i2 = last ? kInf : i_down;                                          // align.c:224-231
}

__global__ __launch_bounds__(64) void k_align_band(const Align2Args A, const int32_t* __restrict__ read_len)
{
    __shared__ uint8_t s_tab[256];
    __shared__ uint16_t s_read[256], s_hap[256], s_hap0[256];
    if (A.status[0] != 0x7fffffff) return;                 // the pair list holds an index out of range: nothing is written
    const int lane = threadIdx.x, k = lane & 7, grp = lane >> 3;
    for (int i = lane; i < 256; i += 64) {
        s_tab[i] = A.tab[i];
        s_read[i] = A.codes->read[i]; s_hap[i] = A.codes->hap[i]; s_hap0[i] = A.codes->hap0[i];
    }
    const int first_pair = blockIdx.x * 8;
    const bool active = first_pair + grp < A.n_pairs;
    const int pair = active ? first_pair + grp : A.n_pairs - 1;
    const int h = A.pair_hap[pair], r = A.pair_read[pair];
    const int hl = A.hap_len[h];
    const int len2 = read_len[r];
    const int len1 = len2 + 15, e = len1 - 1;

    // anchor: platypus/calign.pyx:60-67 (as in align_wave)
    int a, a_report;
    bool failed = false;
    if (A.hap_varlen[h] == 0) {
        const long long hp = (long long)A.read_pos[r] - (long long)A.hap_start[h];
        const long long a64 = hp - 8 > 0 ? hp - 8 : 0;
        failed = a64 + len1 > hl;                     // assertion at calign.pyx:61
        a_report = (int)(a64 > 0x7fffffffLL ? 0x7fffffffLL : a64);
        a = failed ? 0 : a_report;                    // a failed pair still runs, on a slice that exists
    } else {
        a = a_report = A.anchor_in[pair];
    }
    const uint16_t* hx = A.hapx + A.hap_xoff[h] + a;
    const uint16_t* rx = A.readx + A.read_xoff[r];

    __syncthreads();                                                                  // s_tab
    const uint32_t ge = (uint32_t)A.ge, np = (uint32_t)A.np;
    // wave-uniform loop bounds: the shortest and the longest read of the wavefront's pairs
    int lmin = len2, lmax = len2;
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) { lmin = min(lmin, __shfl_xor(lmin, o)); lmax = max(lmax, __shfl_xor(lmax, o)); }
    lmin = __builtin_amdgcn_readfirstlane(lmin); lmax = __builtin_amdgcn_readfirstlane(lmax);
    const int iters = lmax + 8;                                                       // align.c:106
    const int main_end = lmin & ~7;                                                   // whole blocks before any pair's pick-up

    // The windows' entries come straight from the pair's rows, two blocks of eight iterations ahead of their use (a
    // lone wavefront has nothing else to hide the latency with).  Positions the pair's own iterations cannot reach (a
    // longer read in the same wavefront keeps the loop going) are clamped to the last one they can: the rows' padding
    // covers that one, as it does for k_align2.
    const int r_last = len2 + 23, h_last = len2 + 26;
    struct Raw { uint32_t r, hc, hg; };
    auto fetch = [&](int t0) {         // entries t0 .. t0 + 7 of the read window, slice positions t0 + 8 .. t0 + 15 of the other two
        Raw x;
        x.r = rx[min(t0 + k, r_last)];
        x.hc = hx[min(8 + t0 + 7 - k, h_last)];
        x.hg = hx[min(8 + t0 + k, h_last)];
        return x;
    };
    auto gap_open = [&](uint32_t he, int pos) {                                       // align.c:85: the run is cut at the slice end
        const int room = e - pos;
        return (uint32_t)s_tab[min(he >> 9, (uint32_t)(room > 0 ? room : 0))];
    };
    auto read_entry = [&](uint32_t re) {                                              // code | quality << 16
        return (uint32_t)s_read[(re >> 1) & 0xffu] | ((re >> 9) << 16);
    };
    BandLane L;
    L.m1 = L.i1 = L.d1 = L.m2 = L.i2 = L.d2 = kInf;                                   // align.c:53-58
    L.e2 = kS2Init | (64u << 16);                                                     // align.c:64-65
    {
        const uint32_t he = hx[k];
        L.h1 = s_hap0[(he >> 1) & 0xffu];                                             // align.c:63,68: no 'N' limit in the first eight columns
        L.g = gap_open(he, k);                                                        // align.c:100
    }
    L.best = kInf;                                                                    // align.c:34
    Raw n0 = fetch(0), n1 = fetch(8);
    // what the table look-ups make of a block's raw entries, one block ahead of its use
    uint32_t g_next = gap_open(n0.hg, 8 + k), r_next = read_entry(n0.r), h_next = s_hap[(n0.hc >> 1) & 0xffu];
    auto take = [&](int t0) {          // block t0 starts: its chunks are ready, the next block's are looked up, the one after is asked for
        L.cr = r_next;
        L.ch = h_next;
        L.cg = g_next;
        n0 = n1;
        g_next = gap_open(n0.hg, 8 + t0 + 8 + k);
        r_next = read_entry(n0.r);
        h_next = s_hap[(n0.hc >> 1) & 0xffu];
        n1 = fetch(t0 + 16);
    };
    int t0 = 0;
    take(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) band_iter<true>(L, j, k, len2, ge, np);
    for (t0 = 8; t0 < main_end; t0 += 8) {
        take(t0);
#pragma unroll
        for (int j = 0; j < 8; ++j) band_iter<false>(L, t0 + j, k, len2, ge, np);
    }
    for (; t0 < iters; t0 += 8) {
        take(t0);
#pragma unroll
        for (int j = 0; j < 8; ++j) band_iter<true>(L, t0 + j, k, len2, ge, np);
    }
    // the pair's minimum over its eight lanes
    uint32_t best = L.best;
    best = min(best, (uint32_t)__shfl_xor((int)best, 1));
    best = min(best, (uint32_t)__shfl_xor((int)best, 2));
    best = min(best, (uint32_t)__shfl_xor((int)best, 4));
    if (active && k == 0) {
        const int sc = failed ? -1 : (int)best;                                       // align.c:234
        A.score[pair] = sc;
        A.anchor[pair] = a_report;
        if (A.like) {
            double lk = 0.0;                          // swallowed assertion: calign.pyx:61
            if (!failed) {
                const double pw = A.exp_mapq[A.read_mapq[r]];
                const double pr = __dsub_rn(1.0, pw);                                 // calign.pyx:55-56
                lk = __dadd_rn(__dmul_rn(A.exp_score[sc], pr), pw);                   // calign.pyx:63,78
                lk = lk < 1.0 ? lk : 1.0;
            }
            A.like[pair] = lk;
        }
    }
}

}  // namespace pg
