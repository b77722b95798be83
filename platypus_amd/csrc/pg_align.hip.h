// k_align2: banded affine-gap DP of platypus/c/align.c:18-235, one lane per
// (haplotype, read) pair, one DP cell per VGPR.
//
// Why this shape (measured on MI355X with tools/ubench_valu.hip): the 16-bit
// VOP2 instructions v_add_u16 / v_min_u16 / v_xor_b32 issue at 2 cycles per
// wavefront, every packed (v_pk_*), three-operand, DPP, SDWA or SGPR-operand
// form at 4.  With one cell per register the two lane shifts per iteration of
// the reference (align.c:168,224-231) and its five sliding windows
// (align.c:111-112,182-187) cost nothing: the DP state is never moved (a shift
// is only a different operand index), and the windows are rings of 8 registers
// whose slot numbers are compile-time constants once the loop is unrolled by 8.
//
// Arithmetic: the reference scales every cost by 4 and returns minscore >> 2;
// all its terms are multiples of 4 and no value can reach 2^15 (each D lane is
// bounded by pos_inf + 24*lane), so the same recurrence is run here on the
// unscaled costs: pos_inf 0x800, gap extend ge, nucleotide prior np, base
// quality q, 64 for the padding past the read end, 1 for an 'N' column.
// Bases are kept as (char << 8); a mismatch then gives (s1 ^ s2) >= 256 > q,
// so  min_u16(s1 ^ s2, q)  is the cost of align.c:158-162 in two instructions.
//
// State carried between iterations (8 lanes each, lane k = diagonal 2k / 2k+1):
//   u1 = min(M1,I1), D1, M2, I2, D2, u2 = min(M2,I2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

constexpr uint32_t kInf = 0x800;      // pos_inf / 4
constexpr uint32_t kS2Init = 0x00ff;  // read window before the read starts: equals no (char << 8)
constexpr uint32_t kPadElem = ('0' << 8) | 64;   // align.c:119-120, packed as (char << 8) | quality

// 16-bit VOP2 forms, forced: the compiler would otherwise promote to the
// half-rate 32-bit minimum or pick an SGPR operand.
typedef uint16_t u16;
__device__ __forceinline__ u16 add16(u16 a, u16 b) { return (u16)(a + b); }
__device__ __forceinline__ u16 min16(u16 a, u16 b) { return a < b ? a : b; }
// a uniform value the compiler must keep in a VGPR
__device__ __forceinline__ u16 in_vgpr(uint32_t x) {
    uint32_t d;
    asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "s"(x));
    return (u16)d;
}

struct Align2Args {
    const uint16_t* hapx;      // per haplotype position: (char << 8) | run length, padded with 'N'
    const int64_t* hap_xoff;   // element offset of each haplotype in hapx
    const int32_t* hap_len;
    const int32_t* hap_start;
    const int32_t* hap_varlen;
    const uint8_t* hap_has_n;  // 1: the haplotype contains an 'N'
    const uint16_t* readx;     // per read position: (char << 8) | quality, padded with kPadElem
    const int64_t* read_xoff;  // element offset of each read in readx (multiple of 8)
    const int32_t* read_pos;
    const int32_t* read_mapq;
    const int32_t* pair_hap;
    const int32_t* pair_read;
    const int32_t* order;
    const WaveDesc* waves;
    const int32_t* anchor_in;
    const uint8_t* tab;        // table[h] - 33, 256 entries
    const double* exp_score;
    const double* exp_mapq;
    int32_t* score;
    int32_t* anchor;
    double* like;
    int n_waves;
    int ge, np;
};

struct Lane2 {
    u16 u1[8], d1[8], m2[8], i2[8], d2[8], u2[8];
    u16 s1[8], s2[8], q[8], g[8], gn[8], nq[8];
    u16 best;
};

// min(best, v[j]) for a wave-uniform j, written so that no lane of v is ever
// addressed dynamically (the compiler would park the array in scratch): every
// lane but j is pushed to 0xffff before the minimum.
__device__ __forceinline__ u16 take_min(u16 best, const u16 (&v)[8], int j) {
#pragma unroll
    for (int k = 0; k < 8; ++k) best = min16(best, (u16)(v[k] | (j == k ? (u16)0 : (u16)0xffff)));
    return best;
}

// One iteration t (t & 7 == J) of the loop at align.c:106-232.
//   re  read element of position t, he  haplotype element of slice position t+8
template <int J, bool HASN>
__device__ __forceinline__ void dp_iter(Lane2& L, int t, int len2, int e, uint32_t re, uint32_t he,
                                        const uint8_t* s_tab, u16 ge, u16 genp, u16 np, u16 inf)
{
    // align.c:111-121: the read window takes position t
    L.s2[J] = (u16)(re & 0xff00u);
    L.q[J] = (u16)(re & 0x00ffu);
    if (t < 8) {                      // align.c:127-128: lane t starts at cost 0
        L.u1[J] = 0; L.m2[J] = 0; L.u2[J] = 0;
    }
    // gap-open of slice position t+8 (align.c:85,186): run length cut at the slice end
    const int room = e - 8 - t;
    const u16 hrun = min16((u16)(he & 0x00ffu), in_vgpr((uint32_t)(room > 0 ? room : 0)));
    const u16 gnew = s_tab[hrun];

    // ---- even half-step, align.c:129-173 ----
    u16 m1[8], i1[8], d1n[8], u1n[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m1[k] = min16(L.u1[k], L.d1[k]);
    if (t >= len2) L.best = take_min(L.best, m1, t - len2);     // align.c:134-156
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        u16 c = min16((u16)(L.s2[(J - k) & 7] ^ L.s1[(J + k) & 7]), L.q[(J - k) & 7]);
        if (HASN) c = min16(c, L.nq[(J + k) & 7]);
        m1[k] = add16(m1[k], c);
        i1[k] = min16(add16(L.i2[k], genp), add16(L.m2[k], L.gn[(J + 7 - k) & 7]));
        u1n[k] = min16(m1[k], i1[k]);
    }
    d1n[0] = inf;                     // align.c:168
#pragma unroll
    for (int k = 1; k < 8; ++k)
        d1n[k] = min16(add16(L.d2[k - 1], ge), add16(L.u2[k - 1], L.g[(J + 8 - k) & 7]));

    // ---- align.c:180-187: haplotype window, N limits and gap-open vector take position t+8 ----
    L.s1[J] = (u16)(he & 0xff00u);
    if (HASN) L.nq[J] = ((he & 0xff00u) == ((uint32_t)'N' << 8)) ? (u16)1 : inf;
    L.g[J] = gnew;
    L.gn[J] = add16(gnew, np);

    // ---- odd half-step, align.c:190-231 ----
    u16 m2n[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m2n[k] = min16(L.u2[k], L.d2[k]);
    if (t >= len2) L.best = take_min(L.best, m2n, t - len2);    // align.c:195-217
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        u16 c = min16((u16)(L.s2[(J - k) & 7] ^ L.s1[(J + 1 + k) & 7]), L.q[(J - k) & 7]);
        if (HASN) c = min16(c, L.nq[(J + 1 + k) & 7]);
        L.m2[k] = add16(m2n[k], c);
        L.d2[k] = min16(add16(d1n[k], ge), add16(u1n[k], L.g[(J + 8 - k) & 7]));
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)
        L.i2[k] = min16(add16(i1[k + 1], genp), add16(m1[k + 1], L.gn[(J + 7 - k) & 7]));
    L.i2[7] = inf;                    // align.c:230
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        L.u2[k] = min16(L.m2[k], L.i2[k]);
        L.u1[k] = u1n[k];
        L.d1[k] = d1n[k];
    }
}

__device__ __forceinline__ uint32_t elem16(const uint4& b, int j) {
    const uint32_t w = (j >> 1) == 0 ? b.x : (j >> 1) == 1 ? b.y : (j >> 1) == 2 ? b.z : b.w;
    return (j & 1) ? (w >> 16) : (w & 0xffffu);
}
__device__ __forceinline__ uint4 load16_unaligned(const uint16_t* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

template <bool HASN>
__device__ __forceinline__ u16 align_pair(const uint16_t* hx, const uint16_t* rx, int len2,
                                               const uint8_t* s_tab, uint32_t ge_, uint32_t np_)
{
    const int len1 = len2 + 15, e = len1 - 1;
    const u16 ge = in_vgpr(ge_), np = in_vgpr(np_), genp = in_vgpr(ge_ + np_), inf = in_vgpr(kInf);
    Lane2 L;
    const uint4 h0 = load16_unaligned(hx);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        L.u1[k] = L.d1[k] = L.m2[k] = L.i2[k] = L.d2[k] = L.u2[k] = inf;   // align.c:53-58
        L.s2[k] = kS2Init;                                                   // align.c:64
        L.q[k] = 64;                                                         // align.c:65
        const uint32_t he = elem16(h0, k);
        L.s1[k] = (u16)(he & 0xff00u);                                       // align.c:63
        L.nq[k] = inf;                // align.c:68: no 'N' discount inside the first 8 columns
        // align.c:100 with the window written as a ring: slot j holds the entry that
        // reaches lane k at iteration t when (t + 7 - k) & 7 == j, i.e. position 7-j now
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pos = 7 - j;
        const uint32_t run = elem16(h0, pos) & 0xffu;
        const uint32_t hr = run < (uint32_t)(e - pos) ? run : (uint32_t)(e - pos);
        L.g[j] = s_tab[hr];
        L.gn[j] = (u16)(L.g[j] + np_);
    }
    L.best = inf;                                                            // align.c:34

    const int iters = len2 + 8;                                              // align.c:106
    uint4 rb = *reinterpret_cast<const uint4*>(rx);
    uint4 hb = load16_unaligned(hx + 8);
    for (int tb = 0; tb < iters; tb += 8) {
        const uint4 rb_next = *reinterpret_cast<const uint4*>(rx + tb + 8);
        const uint4 hb_next = load16_unaligned(hx + 16 + tb);
#define PG_IT(J)                                                                                  \
        if (tb + J < iters)                                                                       \
            dp_iter<J, HASN>(L, tb + J, len2, e, elem16(rb, J), elem16(hb, J), s_tab, ge, genp, np, inf);
        PG_IT(0) PG_IT(1) PG_IT(2) PG_IT(3) PG_IT(4) PG_IT(5) PG_IT(6) PG_IT(7)
#undef PG_IT
        rb = rb_next;
        hb = hb_next;
    }
    return L.best;
}

__global__ __launch_bounds__(256) void k_align2(const Align2Args A)
{
    __shared__ uint8_t s_tab[256];
    s_tab[threadIdx.x] = A.tab[threadIdx.x];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
    if (wave >= A.n_waves) return;
    const int lane = threadIdx.x & 63;
    const int w_first = __builtin_amdgcn_readfirstlane(A.waves[wave].first);
    const int w_count = __builtin_amdgcn_readfirstlane(A.waves[wave].count);
    const int len2 = __builtin_amdgcn_readfirstlane(A.waves[wave].len), len1 = len2 + 15;
    const bool active = lane < w_count;
    const int slot = w_first + (active ? lane : w_count - 1);
    const int pair = A.order ? A.order[slot] : slot;
    const int h = A.pair_hap[pair], r = A.pair_read[pair];
    const int hl = A.hap_len[h];

    // anchor: platypus/calign.pyx:60-67
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
    // an 'N' can only be met if the haplotype has one or the slice reaches the 'N' padding
    const bool need_n = A.hap_has_n[h] != 0 || a + len1 + 1 > hl;
    u16 best;
    if (__any(need_n)) best = align_pair<true>(hx, rx, len2, s_tab, (uint32_t)A.ge, (uint32_t)A.np);
    else               best = align_pair<false>(hx, rx, len2, s_tab, (uint32_t)A.ge, (uint32_t)A.np);

    if (active) {
        const int sc = failed ? -1 : (int)best;                  // align.c:234
        A.score[pair] = sc;
        A.anchor[pair] = a_report;
        if (A.like) {
            double lk = 0.0;                          // swallowed assertion: calign.pyx:61
            if (!failed) {
                const double pw = A.exp_mapq[A.read_mapq[r]];
                const double pr = __dsub_rn(1.0, pw);                        // calign.pyx:55-56
                lk = __dadd_rn(__dmul_rn(A.exp_score[sc], pr), pw);          // calign.pyx:63,78
                lk = lk < 1.0 ? lk : 1.0;
            }
            A.like[pair] = lk;
        }
    }
}

// ---------------------------------------------------------------------------
// Upload passes that build the element arrays k_align2 streams.
// ---------------------------------------------------------------------------

// hapx[xoff[h] + p] = (char << 8) | run for p < len, ('N' << 8) for the padding.
// run = number of following equal bases, an 'N' never extends a run, capped at
// `cap` (table length - 1): the count of align.c:73-90 before its cut at the
// slice end.  has_n[h] = 1 if any base is 'N'.
__global__ __launch_bounds__(256) void k_pack_hapx(const uint8_t* __restrict__ src,
                                                   const int64_t* __restrict__ soff,
                                                   const int64_t* __restrict__ xoff,
                                                   uint16_t* __restrict__ dst, uint8_t* __restrict__ has_n,
                                                   int n_haps, int pad, int cap)
{
    __shared__ int s_n;
    for (int h = blockIdx.x; h < n_haps; h += gridDim.x) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const int64_t s0 = soff[h];
        const int len = (int)(soff[h + 1] - s0);
        const int64_t d0 = xoff[h];
        bool any_n = false;
        for (int p = threadIdx.x; p < len + pad; p += blockDim.x) {
            uint32_t c = 'N', r = 0;
            if (p < len) {
                c = src[s0 + p];
                if (c != 'N') {
                    while ((int)r < cap && p + (int)r + 1 < len && src[s0 + p + r + 1] == c) ++r;
                } else {
                    any_n = true;
                }
            }
            dst[d0 + p] = (uint16_t)((c << 8) | r);
        }
        if (any_n) s_n = 1;
        __syncthreads();
        if (threadIdx.x == 0) has_n[h] = (uint8_t)s_n;
        __syncthreads();
    }
}

// readx[xoff[r] + j] = (char << 8) | quality for j < len, kPadElem up to xoff[r+1].
__global__ __launch_bounds__(256) void k_pack_readx(const uint8_t* __restrict__ seq,
                                                    const uint8_t* __restrict__ qual,
                                                    const int64_t* __restrict__ roff,
                                                    const int64_t* __restrict__ xoff,
                                                    uint16_t* __restrict__ dst, int n_reads)
{
    const int wave = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int n_waves = (int)(gridDim.x * 4u);
    for (int r = wave; r < n_reads; r += n_waves) {
        const int64_t s0 = roff[r];
        const int len = (int)(roff[r + 1] - s0);
        const int64_t d0 = xoff[r];
        const int tot = (int)(xoff[r + 1] - d0);
        for (int j = lane; j < tot; j += 64)
            dst[d0 + j] = j < len ? (uint16_t)(((uint32_t)seq[s0 + j] << 8) | qual[s0 + j]) : (uint16_t)kPadElem;
    }
}

}  // namespace pg
