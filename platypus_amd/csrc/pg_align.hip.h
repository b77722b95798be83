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
// Bases are kept as codes (see SeqCodes): a read base as a 9-bit pattern of constant
// weight in bits 7..15 plus bit 0 set for everything but 'N', a haplotype column as
// the complement of its base's pattern -- or, for an 'N' column beyond the first
// eight, as the single bit 0.  Then  s2 & s1  is 0 for equal bases, >= 128 > q for
// different ones and 1 for "read base is no 'N', column is", so
// min_u16(s2 & s1, q)  is the cost of align.c:158-162 AND the 'N' limit of
// align.c:68,183 in two instructions; haplotypes with 'N's need no other code path.
//
// State carried between iterations (8 lanes each, lane k = diagonal 2k / 2k+1):
//   u1 = min(M1,I1), D1, M2, I2, D2, u2 = min(M2,I2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

constexpr uint32_t kInf = 0x800;      // pos_inf / 4
constexpr uint32_t kS2Init = 0xff81;  // read window before the read starts: a code no column matches
// Elements of the read and haplotype arrays: (quality or run length) << 9 | char << 1, so that
// `element & 0x1fe` is the byte offset of the char's entry in a table of 16-bit codes.
constexpr uint32_t kPadElem = (64u << 9) | ('0' << 1);   // align.c:119-120: char '0', cost 256 / 4
constexpr uint32_t kNElem = 'N' << 1;                    // haplotype padding: 'N', run 0
constexpr int kMaxRun = 127;                             // run lengths (gap-open table index) kept per element

// Code tables, 256 entries each, indexed by the sequence byte:
//   read[b]  pattern(b) << 7 | (b != 'N')      pattern = the i-th 9-bit word with four bits set
//   hap[b]   ~pattern(b) << 7 & 0xff80, but 0x0001 for 'N'   (columns 8.. of a slice, align.c:183)
//   hap0[b]  ~pattern(b) << 7 & 0xff80 for every b           (columns 0..7: no 'N' discount, align.c:68)
// 126 patterns exist: bytes 33..126 and 1..31 have one, any other byte is outside the domain (entry 0).
struct SeqCodes {
    uint16_t read[256], hap[256], hap0[256];
};

// 16-bit VOP2 forms, forced: the compiler would otherwise promote to the
// half-rate 32-bit minimum or pick an SGPR operand.
typedef uint16_t u16;
__device__ __forceinline__ u16 add16(u16 a, u16 b) { return (u16)(a + b); }
// The empty asm hides the result from the instruction selector, which would
// otherwise merge two chained minima into v_min3_u16.  That form -- like every
// SDWA form, which the build switches off -- does not just issue slowly on
// gfx950: a few of them in a loop of 16-bit VOP2 instructions halve the rate of
// the whole loop (measured with a replay of this kernel's own instruction stream).
__device__ __forceinline__ u16 min16(u16 a, u16 b) {
    u16 r = a < b ? a : b;
    asm("" : "+v"(r));
    return r;
}
// Hide how a value was made, so that the compiler cannot fold its producer into
// the consumer as a three-operand instruction (v_bitop3_b16, v_min3_u16: half /
// quarter rate on gfx950).  Emits nothing.
__device__ __forceinline__ u16 opaque(u16 x) {
    uint32_t v = x;
    asm volatile("" : "+v"(v));
    return (u16)v;
}
// a uniform value the compiler must keep in a VGPR
__device__ __forceinline__ u16 in_vgpr(uint32_t x) {
    uint32_t d;
    asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "s"(x));
    return (u16)d;
}

struct Align2Args {
    const uint16_t* hapx;      // per haplotype position: run length << 9 | char << 1, padded with 'N'
    const int64_t* hap_xoff;   // element offset of each haplotype in hapx
    const int32_t* hap_len;
    const int32_t* hap_start;
    const int32_t* hap_varlen;
    const uint16_t* readx;     // per read position: quality << 9 | char << 1, padded with kPadElem
    const int64_t* read_xoff;  // element offset of each read in readx (multiple of 8)
    const int32_t* read_pos;
    const int32_t* read_mapq;
    const int32_t* pair_hap;
    const int32_t* pair_read;
    const int32_t* order;      // slot -> pair of a ragged batch, NULL when every read has the same length
    const WaveDesc* waves;     // wavefront descriptors of a ragged batch, NULL otherwise
    const int32_t* status;     // status words of the pair list's preparation (pg_prep.hip.h)
    const int32_t* anchor_in;
    const uint8_t* tab;        // table[h] - 33, 256 entries
    const SeqCodes* codes;
    const double* exp_score;
    const double* exp_mapq;
    int32_t* score;
    int32_t* anchor;
    double* like;
    int n_waves;               // uniform batch: wavefronts; ragged batch: an upper bound, the count is in status
    int uniform_len;           // uniform batch: the read length
    int n_pairs;
    int ge, np;
};

// number of wavefronts with work; 0 if the pair list holds an index out of range
__device__ __forceinline__ int align_wave_count(const Align2Args& A) {
    if (A.status[0] != 0x7fffffff) return 0;
    return A.waves ? min(A.n_waves, A.status[1]) : A.n_waves;
}

// Per-lane state.  Every DP value is kept with an offset of P*d added, where
// d is the diagonal of the cell (even lane k: d = 2k, odd lane k: d = 2k+1) and
// P = gapextend + nucprior.  A move along I steps one diagonal down, so its
// "+ P" (align.c:169-173: gap_extend, then nuc_prior) cancels against the
// offset and  I' = min(I + ge, M + G) + np  becomes  I' = min(I, M + (G - ge)):
// one instruction less per cell.  A move along D steps one diagonal up and pays
// the offset once more: D' = min(D + (ge + P), u + (G + P)).  M stays on its
// diagonal.  Offsets are removed where scores are picked up.  All of this is
// exact integer arithmetic (no value comes near 2^16).
struct Lane2 {
    u16 m1g[8];                      // min(M1, I1, D1) of the even lanes, ready for the next even half-step
    u16 m2[8], i2[8], d2[8], u2[8];  // odd lanes; u2 = min(m2, i2)
    u16 s1[8], s2[8], q[8];          // rings: haplotype column codes, read base codes, read qualities
    u16 gP[8], gM[8];                // rings: gap-open + P (for D), gap-open - ge (for I)
    u16 best;
};

// min(best, v[j] - off(j)) for a wave-uniform j, written so that no lane of v is
// ever addressed dynamically (the compiler would park the array in scratch):
// every lane but j is pushed to 0xffff before the minimum.
template <int ODD>
__device__ __forceinline__ u16 take_min(u16 best, const u16 (&v)[8], int j, uint32_t P) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u16 cand = (u16)(v[k] - (u16)((2 * k + ODD) * P));
        best = min16(best, (u16)(cand | (j == k ? (u16)0 : (u16)0xffff)));
    }
    return best;
}

struct Consts2 {
    u16 geP;      // ge + P
    u16 P;        // ge + np
    u16 mge;      // -ge (mod 2^16)
    u16 inf;      // pos_inf
    u16 inf15;    // pos_inf + 15 P
    uint32_t Ps;  // P again, as a scalar
};

// LDS tables of a workgroup: gap-open cost per run length and the code tables
struct AlignLds {
    uint8_t tab[256];
    uint16_t read[256], hap[256], hap0[256];
};
__device__ __forceinline__ u16 code_of(const uint16_t* table, uint32_t elem) {   // table[char of the element]
    return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(table) + (elem & 0x1feu));
}

// One iteration t (t & 7 == J) of the loop at align.c:106-232.
//   re  read element of position t, rn  of position t + 1, he  haplotype element of slice position t+8
// EDGE iterations carry the start-up (t < 8, align.c:127-128) and the score
// pick-up (t >= len2, align.c:134-156,195-217); the others need neither.
// The code of read position t is already in its ring slot: it was looked up during
// iteration t - 1, as the code of position t + 1 is looked up here, so that no
// instruction waits for LDS.
template <int J, bool EDGE>
__device__ __forceinline__ void dp_iter(Lane2& L, int t, int len2, int e, uint32_t re, uint32_t rn, uint32_t he,
                                        const AlignLds& S, const Consts2& C)
{
    // align.c:111-121: the read window takes position t
    L.q[J] = opaque((u16)(re >> 9));
    if (EDGE && t < 8) {              // align.c:127-128: lane t (== J here) starts at cost 0
        L.m1g[J] = (u16)(2 * J * C.Ps);
        L.m2[J] = L.u2[J] = (u16)((2 * J + 1) * C.Ps);
    }
    // gap-open of slice position t+8 (align.c:85,186): run length cut at the slice end
    const int room = e - 8 - t;
    const u16 hrun = min16((u16)(he >> 9), in_vgpr((uint32_t)(room > 0 ? room : 0)));
    const u16 gnew = S.tab[hrun];

    // The half-steps are written as stages of eight independent instructions with
    // a scheduling barrier between stages.  On gfx950 a VALU instruction that
    // reads a result still in flight stalls its wavefront for about nine cycles
    // (two issue slots); with every dependent pair a whole stage apart no such
    // stall occurs and two wavefronts per SIMD keep the VALU busy.
#define PG_STAGE() __builtin_amdgcn_sched_barrier(0)
    // ---- even half-step, align.c:129-173 ----
    if (EDGE && t >= len2) L.best = take_min<0>(L.best, L.m1g, t - len2, C.Ps);   // align.c:134-156
    u16 x[8], y[8], m1[8], i1[8], u1[8], d1n[8], m2g[8];
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (u16)(L.s2[(J - k) & 7] & L.s1[(J + k) & 7]);
    PG_STAGE();
    // align.c:180-187: the haplotype window takes position t+8; its slot was read for the last time
    // in the stage above, and the odd half-step is far enough away for the look-up to land
    L.s1[J] = code_of(S.hap, he);
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = add16(L.m2[k], L.gM[(J + 7 - k) & 7]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = min16(x[k], L.q[(J - k) & 7]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) i1[k] = min16(L.i2[k], y[k]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) m1[k] = add16(L.m1g[k], x[k]);
    PG_STAGE();
#pragma unroll
    for (int k = 1; k < 8; ++k) x[k] = add16(L.d2[k - 1], C.geP);
    PG_STAGE();
#pragma unroll
    for (int k = 1; k < 8; ++k) y[k] = add16(L.u2[k - 1], L.gP[(J + 8 - k) & 7]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) u1[k] = min16(m1[k], i1[k]);
    PG_STAGE();
    d1n[0] = C.inf;                   // align.c:168
#pragma unroll
    for (int k = 1; k < 8; ++k) d1n[k] = min16(x[k], y[k]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) m2g[k] = min16(L.u2[k], L.d2[k]);                // align.c:190
    PG_STAGE();

    // ---- align.c:185-187: the gap-open vector takes position t+8 ----
    L.gP[J] = add16(gnew, C.P);
    L.gM[J] = add16(gnew, C.mge);

    // ---- odd half-step, align.c:190-231 ----
    if (EDGE && t >= len2) L.best = take_min<1>(L.best, m2g, t - len2, C.Ps);     // align.c:195-217
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (u16)(L.s2[(J - k) & 7] & L.s1[(J + 1 + k) & 7]);
    PG_STAGE();
    // the slot of read position t - 7 is free now: the code of position t + 1 goes there
    L.s2[(J + 1) & 7] = code_of(S.read, rn);
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = add16(d1n[k], C.geP);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = min16(x[k], L.q[(J - k) & 7]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) L.d2[k] = add16(u1[k], L.gP[(J + 8 - k) & 7]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) L.m2[k] = add16(m2g[k], x[k]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) L.d2[k] = min16(y[k], L.d2[k]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = add16(m1[k + 1], L.gM[(J + 7 - k) & 7]);
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) L.m1g[k] = min16(u1[k], d1n[k]);                  // align.c:129, next iteration
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 7; ++k) L.i2[k] = min16(i1[k + 1], x[k]);
    L.i2[7] = C.inf15;                // align.c:230
    PG_STAGE();
#pragma unroll
    for (int k = 0; k < 8; ++k) L.u2[k] = min16(L.m2[k], L.i2[k]);
    PG_STAGE();
#undef PG_STAGE
}

__device__ __forceinline__ uint32_t elem4(const uint2& b, int j) {   // element j (0..3) of four packed u16
    const uint32_t w = (j >> 1) == 0 ? b.x : b.y;
    return (j & 1) ? (w >> 16) : (w & 0xffffu);
}
__device__ __forceinline__ uint2 load8_unaligned(const uint16_t* p) {
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Eight iterations tb .. tb+7.  The element buffers roll: the low halves hold
// positions tb..tb+3 on entry, the high halves are fetched here for tb+4..tb+7,
// and the low halves are refetched for the next block while the high halves
// are in use -- every load has four iterations to land.
template <bool EDGE>
__device__ __forceinline__ void dp_block(Lane2& L, int tb, int iters, int len2, int e,
                                         const uint16_t* rx, const uint16_t* hx8,
                                         uint2& rlo, uint2& hlo, const AlignLds& S, const Consts2& C)
{
    const uint2 rhi = *reinterpret_cast<const uint2*>(rx + tb + 4);
    const uint2 hhi = load8_unaligned(hx8 + tb + 4);
#define PG_IT(J, RB, RN, HB)                                                                              \
    if (!EDGE || tb + J < iters)                                                                          \
        dp_iter<J, EDGE>(L, tb + J, len2, e, elem4(RB, J & 3), elem4(RN, (J + 1) & 3), elem4(HB, J & 3), S, C);
    PG_IT(0, rlo, rlo, hlo) PG_IT(1, rlo, rlo, hlo) PG_IT(2, rlo, rlo, hlo) PG_IT(3, rlo, rhi, hlo)
    rlo = *reinterpret_cast<const uint2*>(rx + tb + 8);
    hlo = load8_unaligned(hx8 + tb + 8);
    PG_IT(4, rhi, rhi, hhi) PG_IT(5, rhi, rhi, hhi) PG_IT(6, rhi, rhi, hhi) PG_IT(7, rhi, rlo, hhi)
#undef PG_IT
}

__device__ __forceinline__ u16 align_pair(const uint16_t* hx, const uint16_t* rx, int len2,
                                          const AlignLds& S, uint32_t ge_, uint32_t np_)
{
    const int len1 = len2 + 15, e = len1 - 1;
    const uint32_t P = ge_ + np_;
    Consts2 C;
    C.geP = in_vgpr(ge_ + P); C.P = in_vgpr(P); C.mge = in_vgpr(0u - ge_);
    C.inf = in_vgpr(kInf); C.inf15 = in_vgpr(kInf + 15 * P); C.Ps = P;
    Lane2 L;
    const uint2 h0 = load8_unaligned(hx), h1 = load8_unaligned(hx + 4);
    uint2 rlo = *reinterpret_cast<const uint2*>(rx);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        L.m1g[k] = (u16)(kInf + 2 * k * P);                                  // align.c:53-58
        L.m2[k] = L.i2[k] = L.d2[k] = L.u2[k] = (u16)(kInf + (2 * k + 1) * P);
        L.s2[k] = kS2Init;                                                   // align.c:64
        L.q[k] = 64;                                                         // align.c:65
        const uint32_t he = k < 4 ? elem4(h0, k & 3) : elem4(h1, k & 3);
        L.s1[k] = code_of(S.hap0, he);                // align.c:63,68: no 'N' discount inside the first 8 columns
    }
    L.s2[0] = code_of(S.read, elem4(rlo, 0));         // read position 0, for iteration 0
    // align.c:100 with the window written as a ring: slot j holds the entry that
    // reaches lane k at iteration t when (t + 7 - k) & 7 == j, i.e. slice position 7-j now
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pos = 7 - j;
        const uint32_t run = (pos < 4 ? elem4(h0, pos & 3) : elem4(h1, pos & 3)) >> 9;
        const uint32_t hr = run < (uint32_t)(e - pos) ? run : (uint32_t)(e - pos);
        const uint32_t g = S.tab[hr];
        L.gP[j] = (u16)(g + P);
        L.gM[j] = (u16)(g - ge_);
    }
    L.best = kInf;                                                           // align.c:34

    const int iters = len2 + 8;                                              // align.c:106
    const uint16_t* hx8 = hx + 8;
    uint2 hlo = load8_unaligned(hx8);
    int tb = 0;
    dp_block<true>(L, tb, iters, len2, e, rx, hx8, rlo, hlo, S, C);
    for (tb = 8; tb + 8 <= len2; tb += 8)
        dp_block<false>(L, tb, iters, len2, e, rx, hx8, rlo, hlo, S, C);
    for (; tb < iters; tb += 8)
        dp_block<true>(L, tb, iters, len2, e, rx, hx8, rlo, hlo, S, C);
    return L.best;
}

__device__ __forceinline__ void align_wave(const Align2Args& A, int wave, const AlignLds& S)
{
    const int lane = threadIdx.x & 63;
    int w_first = wave * 64, w_count = min(64, A.n_pairs - wave * 64), len2 = A.uniform_len;
    if (A.waves) {
        w_first = __builtin_amdgcn_readfirstlane(A.waves[wave].first);
        w_count = __builtin_amdgcn_readfirstlane(A.waves[wave].count);
        len2 = __builtin_amdgcn_readfirstlane(A.waves[wave].len);
    }
    const int len1 = len2 + 15;
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
    const u16 best = align_pair(hx, rx, len2, S, (uint32_t)A.ge, (uint32_t)A.np);

    if (active) {
        const int sc = failed ? -1 : (int)best;                              // align.c:234
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

// 128 registers: four wavefronts per SIMD
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_align2(const Align2Args A)
{
    __shared__ AlignLds S;
    S.tab[threadIdx.x] = A.tab[threadIdx.x];
    S.read[threadIdx.x] = A.codes->read[threadIdx.x];
    S.hap[threadIdx.x] = A.codes->hap[threadIdx.x];
    S.hap0[threadIdx.x] = A.codes->hap0[threadIdx.x];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
    if (wave >= align_wave_count(A)) return;
    align_wave(A, wave, S);
}

// ---------------------------------------------------------------------------
// Upload passes that build the element arrays k_align2 streams.
// ---------------------------------------------------------------------------

// hapx[xoff[h] + p] = run << 9 | char << 1 for p < len, 'N' with run 0 for the padding.
// run = number of following equal bases, an 'N' never extends a run, capped at
// `cap` (table length - 1): the count of align.c:73-90 before its cut at the
// slice end.  *bad_byte is set if a base has no code (SeqCodes).
// one haplotype by one workgroup; returns whether this thread met a base without a code
__device__ __forceinline__ bool pack_hapx_one(const uint8_t* __restrict__ src, int len, uint16_t* __restrict__ dst,
                                              const SeqCodes* __restrict__ codes, int pad, int cap)
{
    bool bad = false;
    for (int p = threadIdx.x; p < len + pad; p += blockDim.x) {
        uint32_t c = 'N', r = 0;
        if (p < len) {
            c = src[p];
            bad |= codes->read[c] == 0;
            if (c != 'N')
                while ((int)r < cap && p + (int)r + 1 < len && src[p + r + 1] == c) ++r;
        }
        dst[p] = (uint16_t)((r << 9) | (c << 1));
    }
    return bad;
}

__global__ __launch_bounds__(256) void k_pack_hapx(const uint8_t* __restrict__ src,
                                                   const int64_t* __restrict__ soff,
                                                   const int64_t* __restrict__ xoff,
                                                   uint16_t* __restrict__ dst, const SeqCodes* __restrict__ codes,
                                                   int32_t* __restrict__ bad_byte, int n_haps, int pad, int cap)
{
    for (int h = blockIdx.x; h < n_haps; h += gridDim.x) {
        const int64_t s0 = soff[h];
        if (pack_hapx_one(src + s0, (int)(soff[h + 1] - s0), dst + xoff[h], codes, pad, cap)) atomicOr(bad_byte, 2);
    }
}

// readx[xoff[r] + j] = quality << 9 | char << 1 for j < len, kPadElem up to xoff[r+1].
// *bad is set to 1 if any quality byte has its top bit set, to 2 if a base has no code.
// one read by one wavefront: `tot` elements; returns the flags this lane saw (1 a quality above 127, 2 a base without a code)
__device__ __forceinline__ int pack_readx_one(const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, int len,
                                              uint16_t* __restrict__ dst, int tot, const SeqCodes* __restrict__ codes, int lane)
{
    uint32_t seen = 0;
    bool no_code = false;
    for (int j = lane; j < tot; j += 64) {
        const uint32_t q = j < len ? qual[j] : 0u;
        const uint32_t c = j < len ? seq[j] : (uint32_t)'A';
        seen |= q;
        no_code |= codes->read[c] == 0;
        dst[j] = j < len ? (uint16_t)(((q & 0x7fu) << 9) | (c << 1)) : (uint16_t)kPadElem;
    }
    // qualities above 127 are outside the domain (the reference reads them as signed char)
    return ((seen & 0x80u) ? 1 : 0) | (no_code ? 2 : 0);
}

__global__ __launch_bounds__(256) void k_pack_readx(const uint8_t* __restrict__ seq,
                                                    const uint8_t* __restrict__ qual,
                                                    const int64_t* __restrict__ roff,
                                                    const int64_t* __restrict__ xoff,
                                                    uint16_t* __restrict__ dst, const SeqCodes* __restrict__ codes,
                                                    int n_reads, int32_t* __restrict__ bad)
{
    const int wave = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int n_waves = (int)(gridDim.x * 4u);
    for (int r = wave; r < n_reads; r += n_waves) {
        const int64_t s0 = roff[r];
        const int64_t d0 = xoff[r];
        const int f = pack_readx_one(seq + s0, qual + s0, (int)(roff[r + 1] - s0), dst + d0, (int)(xoff[r + 1] - d0), codes, lane);
        if (f) atomicOr(bad, f);
    }
}

}  // namespace pg
