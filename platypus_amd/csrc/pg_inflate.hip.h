// BGZF block inflation on the device (RFC 1951 raw deflate streams of at most 64 KiB each, independent of one
// another).  Replaces inflate_block of the reference's vendored samtools (platypus/c/samtools/bgzf.c, called through
// bgzf_read_block) for the stretch of a BAM file a region fetch reads; the host only finds the block boundaries.
//
// k_bgzf_inflate_wave: ONE WAVEFRONT PER BLOCK.  What is serial in a deflate stream is finding where the next code
// starts; what is not is moving bytes.  Lane 0 decodes up to 64 tokens (a literal, or a length/distance pair, each with
// its place in the output) from Huffman tables and a ring of the compressed input in LDS; then the wavefront writes
// them in stream order: the literals between two matches at once, a match 64 bytes per step, out of a ring of the last 32 KiB of output in
// LDS (the deflate window: no match ever waits for global memory, which is what bound one lane per block at 0.4 us per
// byte) and to global memory at the same time.  Four workgroups of one wavefront fit a CU: one per SIMD.
//
// k_bgzf_inflate: one LANE per block, 8 to 32 blocks per wavefront (the first version, kept for comparison).  Every
// lane keeps its tables in LDS, word-interleaved ([word][lane]): whatever index a lane reads, it stays in its own bank.
//
// Tables are canonical: per code length the left-aligned upper end of the codes of up to that length (thr) and the
// offset of the length's first symbol in the list of symbols sorted by code (base).  A symbol is decoded from 15 peeked
// bits, reversed: its length is the number of thresholds the value has reached -- sixteen independent compares on
// eight LDS words instead of a bit-by-bit walk.  Codes of up to ten (literal/length) or nine (distance) bits, nearly
// all, skip even that: a direct table indexed by the next bits holds symbol and length.  A set of code lengths is
// accepted exactly where zlib's inflate_table accepts it (over-subscribed: never; incomplete: only a single code of
// one bit, and never for the code length code).
//
// The table code and the one-lane decoder are one source for device and host (LANES = 64 / 1); the host instantiation
// exists for the tests, which run it against zlib on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

struct InflBlock {          // one BGZF block of a stretch
    uint32_t in_off;        // first byte of the deflate stream in the raw stretch
    uint32_t in_len;        // its length
    uint32_t out_off;       // where its output goes
    uint32_t out_len;       // ISIZE of the block
};

enum {                      // status per block
    INF_OK = 0, INF_BAD_BLOCK_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_LENGTHS = 3, INF_BAD_CODE = 4,
    INF_INPUT_OVERRUN = 5, INF_OUTPUT_OVERRUN = 6, INF_BAD_DISTANCE = 7, INF_SIZE_MISMATCH = 8
};

// per-lane table space, in 16-bit units (every region starts on a 32-bit word)
constexpr int IT_LIT_SYM = 0, IT_LIT_THR = 288, IT_LIT_BASE = 304;
constexpr int IT_DST_SYM = 320, IT_DST_THR = 352, IT_DST_BASE = 368;
constexpr int IT_CLC_SYM = 384, IT_CLC_THR = 404, IT_CLC_BASE = 420;
constexpr int IT_POS = 436;
constexpr int IT_LENS = 452;                   // 320 bytes of code lengths
constexpr int IT_FAST = 452 + 160;             // direct table of the literal/length codes of up to IT_FAST_BITS bits
constexpr int IT_FAST_BITS = 10;               // entry = symbol | length << 9, 0 = longer code (thresholds decide)
constexpr int IT_DFAST = IT_FAST + (1 << IT_FAST_BITS);  // direct table of the distance codes of up to IT_DFAST_BITS bits
constexpr int IT_DFAST_BITS = 9;                         // entry = symbol | length << 9, 0 = longer code
constexpr int IT_U16 = IT_DFAST + (1 << IT_DFAST_BITS);  // 2148 16-bit units per lane
enum { IK_CODES = 0, IK_LENS = 1, IK_DISTS = 2 };        // what a table is for (zlib's CODES / LENS / DISTS)

template <int LANES>
struct InflTab {
    unsigned short* t;      // the lane's word 0 (already offset by the lane)
    __host__ __device__ unsigned short& u16(int region, int i) const {
        const int k = region + i;
        return t[(k >> 1) * (2 * LANES) + (k & 1)];
    }
    __host__ __device__ unsigned char& u8(int region, int i) const {
        unsigned char* b = reinterpret_cast<unsigned char*>(t);
        return b[(region >> 1) * (4 * LANES) + (i >> 2) * (4 * LANES) + (i & 3)];
    }
    __host__ __device__ uint32_t word(int region, int w) const {
        return reinterpret_cast<const uint32_t*>(t)[((region >> 1) + w) * LANES];
    }
};

struct InflBits {
    const uint32_t* words;  // the raw stretch as aligned words (indices, not pointers made from integers: the
                            // compiler keeps the loads in the global address space)
    uint32_t wi;            // index of the word after `ahead`
    uint32_t stop;          // a word index this far past the stream means the stream is corrupt
    uint64_t bb;
    uint32_t ahead;         // the next word, loaded one refill before it is needed
    int bc;
    __host__ __device__ void start(uint32_t byte_off) {
        wi = byte_off >> 2;
        const int skip = (int)(byte_off & 3);
        bb = (uint64_t)words[wi++] >> (8 * skip);
        bc = 32 - 8 * skip;
        ahead = words[wi++];
    }
    __host__ __device__ void refill() {
        if (bc < 32) {
            bb |= (uint64_t)ahead << bc;
            bc += 32;
            ahead = words[wi++];
        }
    }
    __host__ __device__ uint32_t take(int n) {  // n <= 16, bc >= n
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1);
        bb >>= n;
        bc -= n;
        return v;
    }
    // offset of the first byte the bit buffer has not consumed (bc is a multiple of 8 here)
    __host__ __device__ uint32_t byte_pos() const { return 4 * wi - 4 - (uint32_t)(bc >> 3); }
    __host__ __device__ bool overrun() const { return wi > stop; }
};

__host__ __device__ inline uint32_t rev_bits(uint32_t x, int n) {   // the low n bits of x, reversed (n >= 1)
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x) >> (32 - n);
#else
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((x >> i) & 1u) << (n - 1 - i);
    return r;
#endif
}

__host__ __device__ inline uint32_t rev15(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x) >> 17;
#else
    uint32_t r = 0;
    for (int i = 0; i < 15; ++i) r |= ((x >> i) & 1u) << (14 - i);
    return r;
#endif
}

// Canonical table of n symbols whose code lengths (0 = unused, at most 15) are u8(lens_region, lens_at + s).
// Returns false for a set of lengths zlib's inflate_table refuses: an over-subscribed one, or an incomplete one
// unless it is a single code of one bit in a literal/length or distance table (no code at all is accepted:
// decoding then finds no symbol).  fast_region >= 0: also the direct table of the codes of up to fast_bits bits.
template <int LANES>
__host__ __device__ inline bool infl_build(const InflTab<LANES>& T, int lens_region, int lens_at, int n, int sym, int thr, int base,
                                           int kind, int fast_region = -1, int fast_bits = 0) {
    for (int l = 0; l < 16; ++l) T.u16(thr, l) = 0;
    for (int s = 0; s < n; ++s) T.u16(thr, T.u8(lens_region, lens_at + s)) += 1;      // counts, for now
    uint32_t code = 0, off = 0;
    int left = 1, max_len = 0;
    bool ok = true;
    T.u16(thr, 0) = 0;
    T.u16(base, 0) = 0;
    for (int l = 1; l < 16; ++l) {
        const uint32_t c = T.u16(thr, l);
        if (c) max_len = l;
        left = 2 * left - (int)c;
        ok = ok && left >= 0;
        if (left < 0) left = 0;
        T.u16(thr, l) = (unsigned short)(((code + c) << (15 - l)) > 0xffffu ? 0xffffu : ((code + c) << (15 - l)));
        T.u16(base, l) = (unsigned short)(off - code);
        T.u16(IT_POS, l) = (unsigned short)off;
        code = (code + c) << 1;
        off += c;
    }
    if (!ok) return false;
    if (max_len > 0 && left > 0 && (kind == IK_CODES || max_len != 1)) return false;       // incomplete set
    if (fast_region >= 0) {
        uint32_t* words = reinterpret_cast<uint32_t*>(T.t);
        for (int w = 0; w < (1 << fast_bits) / 2; ++w) words[((fast_region >> 1) + w) * LANES] = 0;
    }
    for (int s = 0; s < n; ++s) {
        const int l = T.u8(lens_region, lens_at + s);
        if (l) {
            const int p = T.u16(IT_POS, l);
            T.u16(sym, p) = (unsigned short)s;
            T.u16(IT_POS, l) = (unsigned short)(p + 1);
            if (fast_region >= 0 && l <= fast_bits) {
                // the symbol's code, as the stream presents it (first bit lowest), and every table index that ends in it
                const uint32_t code = (uint32_t)(p - T.u16(base, l)) & 0xffffu;
                const unsigned short entry = (unsigned short)(s | (l << 9));
                for (uint32_t e = rev_bits(code, l); e < (1u << fast_bits); e += 1u << l) T.u16(fast_region, (int)e) = entry;
            }
        }
    }
    return true;
}

// One symbol: returns it, or -1 for a code no symbol has.  The caller has refilled (bc >= 15 is not needed:
// missing bits read as zero and a stream that ends inside a code is caught by the input bound).
template <int LANES, class Bits>
__host__ __device__ inline int infl_decode(const InflTab<LANES>& T, Bits& B, int sym, int thr, int base, int n_sym) {
    const uint32_t v = rev15((uint32_t)B.bb & 0x7fffu);
    uint32_t t[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) t[w] = T.word(thr, w);      // eight independent reads, one wait
    int len = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) len += (v >= (t[w] & 0xffffu)) + (v >= (t[w] >> 16));
    // thr[0] = 0 always counts; a value that has reached all fifteen others belongs to no code
    if (len > 15) return -1;
    const uint32_t idx = (T.u16(base, len) + (v >> (15 - len))) & 0xffffu;
    if (idx >= (uint32_t)n_sym) return -1;
    B.bb >>= len;
    B.bc -= len;
    return T.u16(sym, (int)idx);
}

template <int LANES>
__host__ __device__ inline int infl_block(const InflTab<LANES>& T, const unsigned char* raw, const InflBlock& K, unsigned char* out_base) {
    // input: aligned words from the word that holds the stream's first byte; up to three words past the stream's
    // last are read (never used): the buffer is padded
    InflBits B;
    B.words = reinterpret_cast<const uint32_t*>(raw);        // raw is 4-byte aligned
    B.start(K.in_off);
    B.stop = ((K.in_off + K.in_len + 3) >> 2) + 3;
    unsigned char* const obeg = out_base + K.out_off;
    unsigned char* const oend = obeg + K.out_len;
    unsigned char* op = obeg;
    int n_lit = 0, n_dst = 0, dst_at = 0;
    for (;;) {                                               // deflate blocks of the stream
        B.refill();
        if (B.overrun()) return INF_INPUT_OVERRUN;
        const uint32_t last = B.take(1);
        const uint32_t type = B.take(2);
        if (type == 0) {                                     // stored
            B.take(B.bc & 7);
            B.refill();
            const uint32_t len = B.take(16), nlen = B.take(16);
            if ((len ^ nlen) != 0xffffu) return INF_BAD_STORED;
            // the bytes that follow: give back what the bit buffer holds
            const uint32_t at = B.byte_pos();
            if (at + len > K.in_off + K.in_len) return INF_INPUT_OVERRUN;
            if (op + len > oend) return INF_OUTPUT_OVERRUN;
            for (uint32_t i = 0; i < len; ++i) op[i] = raw[at + i];
            op += len;
            B.start(at + len);
        } else if (type == 3) {
            return INF_BAD_BLOCK_TYPE;
        } else {
            if (type == 1) {                                 // fixed codes (RFC 1951, 3.2.6)
                for (int s = 0; s < 288; ++s) T.u8(IT_LENS, s) = (unsigned char)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
                for (int s = 0; s < 32; ++s) T.u8(IT_LENS, 288 + s) = 5;
                n_lit = 288;
                n_dst = 32;
                dst_at = 288;
            } else {                                         // dynamic codes (3.2.7)
                B.refill();
                n_lit = (int)B.take(5) + 257;
                n_dst = (int)B.take(5) + 1;
                const int n_clc = (int)B.take(4) + 4;
                if (n_lit > 286 || n_dst > 30) return INF_BAD_LENGTHS;
                for (int s = 0; s < 19; ++s) T.u8(IT_LENS, s) = 0;
                for (int i = 0; i < n_clc; ++i) {
                    B.refill();
                    // order of the code length code's lengths (3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15,
                    // five bits each
                    const int s = (int)((i < 12 ? 0x22caa324e804a30ull >> (5 * i) : 0x3c2e1346cull >> (5 * (i - 12))) & 31);
                    T.u8(IT_LENS, s) = (unsigned char)B.take(3);
                }
                if (!infl_build(T, IT_LENS, 0, 19, IT_CLC_SYM, IT_CLC_THR, IT_CLC_BASE, IK_CODES)) return INF_BAD_LENGTHS;
                // the code length code's own lengths are not needed once its table stands: the lengths of the
                // literal/length codes go to bytes 0.., those of the distance codes follow them
                int at = 0;
                const int total = n_lit + n_dst;
                int prev = 0;
                while (at < total) {
                    B.refill();
                    if (B.overrun()) return INF_INPUT_OVERRUN;
                    const int s = infl_decode(T, B, IT_CLC_SYM, IT_CLC_THR, IT_CLC_BASE, 19);
                    if (s < 0) return INF_BAD_CODE;
                    if (s < 16) {
                        T.u8(IT_LENS, at) = (unsigned char)s;
                        prev = s;
                        ++at;
                    } else {
                        int rep, val = 0;
                        if (s == 16) {
                            if (at == 0) return INF_BAD_LENGTHS;
                            val = prev;
                            rep = 3 + (int)B.take(2);
                        } else if (s == 17) {
                            rep = 3 + (int)B.take(3);
                        } else {
                            rep = 11 + (int)B.take(7);
                        }
                        if (at + rep > total) return INF_BAD_LENGTHS;
                        for (int k = 0; k < rep; ++k) T.u8(IT_LENS, at + k) = (unsigned char)val;
                        at += rep;
                        prev = val;
                    }
                }
                if (T.u8(IT_LENS, 256) == 0) return INF_BAD_LENGTHS;               // no end-of-block code
                dst_at = n_lit;
            }
            if (!infl_build(T, IT_LENS, 0, n_lit, IT_LIT_SYM, IT_LIT_THR, IT_LIT_BASE, IK_LENS, IT_FAST, IT_FAST_BITS)) return INF_BAD_LENGTHS;
            if (!infl_build(T, IT_LENS, dst_at, n_dst, IT_DST_SYM, IT_DST_THR, IT_DST_BASE, IK_DISTS)) return INF_BAD_LENGTHS;
            for (;;) {                                       // symbols of the block: every pass consumes at least one bit
                B.refill();
                if (B.overrun()) return INF_INPUT_OVERRUN;
                int s;
                const uint32_t quick = T.u16(IT_FAST, (int)((uint32_t)B.bb & ((1u << IT_FAST_BITS) - 1)));
                if (quick) {                                 // a code of up to ten bits: one look-up
                    s = (int)(quick & 511u);
                    B.bb >>= quick >> 9;
                    B.bc -= (int)(quick >> 9);
                } else {
                    s = infl_decode(T, B, IT_LIT_SYM, IT_LIT_THR, IT_LIT_BASE, n_lit);
                    if (s < 0) return INF_BAD_CODE;
                }
                if (s < 256) {
                    if (op >= oend) return INF_OUTPUT_OVERRUN;
                    *op++ = (unsigned char)s;
                    continue;
                }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) return INF_BAD_CODE;
                int len;
                if (s < 8) len = 3 + s;
                else if (s == 28) len = 258;
                else {
                    const int e = (s - 4) >> 2;
                    len = 3 + ((4 + (s & 3)) << e) + (int)B.take(e);
                }
                B.refill();
                const int d = infl_decode(T, B, IT_DST_SYM, IT_DST_THR, IT_DST_BASE, n_dst);
                if (d < 0 || d >= 30) return INF_BAD_CODE;
                int dist;
                if (d < 4) dist = 1 + d;
                else {
                    const int e = (d >> 1) - 1;
                    dist = 1 + ((2 + (d & 1)) << e) + (int)B.take(e);
                }
                if (dist > op - obeg) return INF_BAD_DISTANCE;
                if (len > oend - op) return INF_OUTPUT_OVERRUN;
                const unsigned char* from = op - dist;
                if (dist >= 4) {                             // four bytes at a time never read what this copy has yet to write
                    while (len >= 4) {
                        uint32_t w;
                        __builtin_memcpy(&w, from, 4);
                        __builtin_memcpy(op, &w, 4);
                        from += 4;
                        op += 4;
                        len -= 4;
                    }
                }
                for (int i = 0; i < len; ++i) op[i] = from[i];
                op += len;
            }
        }
        if (last) break;
    }
    return op == oend ? INF_OK : INF_SIZE_MISMATCH;
}

// grid: one wavefront per workgroup whose first AL lanes take one BGZF block each (fewer blocks per wavefront:
// fewer lanes that wait while one of them builds tables); dynamic LDS IT_U16 * 2 * AL bytes
template <int AL>
__global__ __launch_bounds__(64) void k_bgzf_inflate(const unsigned char* __restrict__ raw, const InflBlock* __restrict__ blocks,
                                                     int n_blocks, unsigned char* __restrict__ out, int32_t* __restrict__ status) {
    extern __shared__ __align__(16) unsigned char infl_lds[];
    const int b = blockIdx.x * AL + threadIdx.x;
    if ((int)threadIdx.x >= AL || b >= n_blocks) return;
    InflTab<AL> T;
    T.t = reinterpret_cast<unsigned short*>(infl_lds) + 2 * threadIdx.x;
    const InflBlock K = blocks[b];
    status[b] = K.out_len ? infl_block<AL>(T, raw, K, out) : INF_OK;
}


// ---------------------------------------------------------------------------------------------------------
// One wavefront per BGZF block.
//
// What is serial in a deflate stream is finding where the next token starts.  The wavefront finds 64 starts' worth at
// once: lane i decodes the whole token that WOULD start at bit i of the next 64 bits -- a literal, or a length code,
// its extra bits, the distance code and its extra bits, through the direct tables (all lanes use the same tables) --
// and the true starts are the chain 0 -> 0 + bits[0] -> ... followed through the lanes' answers: a register read per
// token instead of two table look-ups and their waits.  The tokens of the chain whose sources lie before the pass's
// own output do not depend on each other: their lanes write them side by side, from a ring of the last 32 KiB of
// output in LDS (the deflate window: no match ever waits for global memory, which is what bound one lane per block
// at 0.4 us per byte) into the ring and into global memory at once.  A token that ends the chain -- a match that
// overlaps the pass's output or its own, a long one, the end-of-block code, a code beyond the direct tables -- is
// decoded the ordinary way and copied by the whole wavefront, 64 bytes per step.
// LDS: tables (one lane's layout) | output ring | 2 KiB of compressed input (ring of words).  Four workgroups of
// one wavefront fit a CU: one per SIMD.
// ---------------------------------------------------------------------------------------------------------
constexpr int IW_OUT = 32768;                    // output ring: the deflate window
constexpr int IW_IN_WORDS = 512;                 // input ring, 32-bit words
constexpr int IW_LDS = IT_U16 * 2 + IW_OUT + IW_IN_WORDS * 4 + 64;

struct InflRingBits {                            // lane 0's bit reader over the input ring (block headers)
    const uint32_t* ring;
    uint32_t wi;                                 // next word to take, counted from the aligned start of the stream
    uint64_t bb;
    int bc;
    __device__ void start_bits(uint32_t bitpos) {
        wi = bitpos >> 5;
        const int skip = (int)(bitpos & 31u);
        bb = (uint64_t)ring[wi & (IW_IN_WORDS - 1)] >> skip;
        ++wi;
        bc = 32 - skip;
    }
    __device__ void refill() {
        if (bc < 32) {
            bb |= (uint64_t)ring[wi & (IW_IN_WORDS - 1)] << bc;
            bc += 32;
            ++wi;
        }
    }
    __device__ uint32_t take(int n) {            // n <= 16, bc >= n
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1);
        bb >>= n;
        bc -= n;
        return v;
    }
    __device__ uint32_t bit_pos() const { return 32u * wi - (uint32_t)bc; }   // first bit not consumed
};

struct InflPeek {                                // 64 bits of the stream from a bit position on (every lane the same)
    uint64_t bb;
    int bc;
};

__global__ __launch_bounds__(64) void k_bgzf_inflate_wave(const unsigned char* __restrict__ raw, const InflBlock* __restrict__ blocks,
                                                          int n_blocks, unsigned char* __restrict__ out, int32_t* __restrict__ status)
{
    extern __shared__ __align__(16) unsigned char infl_lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_blocks) return;
    const InflBlock K = blocks[b];
    if (K.out_len == 0) { if (lane == 0) status[b] = INF_OK; return; }
    InflTab<1> T;
    T.t = reinterpret_cast<unsigned short*>(infl_lds);
    unsigned char* oring = infl_lds + IT_U16 * 2;
    uint32_t* iring = reinterpret_cast<uint32_t*>(oring + IW_OUT);
    unsigned char* const gout = out + K.out_off;
    const uint32_t* const words = reinterpret_cast<const uint32_t*>(raw) + (K.in_off >> 2);   // raw is 4-byte aligned and padded
    const uint32_t lead = K.in_off & 3u;                                      // the stream starts `lead` bytes into words[0]
    const uint32_t n_words = (lead + K.in_len + 3) >> 2;                     // words that hold the stream
    const uint32_t end_bit = 8u * (lead + K.in_len);
    uint32_t loaded = 0;                                                      // words in the ring so far

    // keeps the ring full: 128 words at a time while that many are free (consumed: words nobody needs any more)
    auto feed = [&](uint32_t consumed) {
        while (loaded < n_words + 8 && loaded + 128 <= consumed + IW_IN_WORDS) {
            const uint32_t w = loaded + 2 * lane;
            iring[w & (IW_IN_WORDS - 1)] = words[w];                          // (up to 135 words past the stream: padding)
            iring[(w + 1) & (IW_IN_WORDS - 1)] = words[w + 1];
            loaded += 128;
        }
        __builtin_amdgcn_wave_barrier();
    };
    feed(0);
    auto peek64 = [&](uint32_t bitpos) -> InflPeek {
        const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
        const uint32_t a = iring[w & (IW_IN_WORDS - 1)], c = iring[(w + 1) & (IW_IN_WORDS - 1)], d = iring[(w + 2) & (IW_IN_WORDS - 1)];
        InflPeek P;
        P.bb = ((uint64_t)__builtin_amdgcn_alignbit(d, c, sh) << 32) | __builtin_amdgcn_alignbit(c, a, sh);
        P.bc = 64;
        return P;
    };

    // One token decoded the ordinary way at a bit position (every lane the same; the results are scalars): what the
    // direct tables do not hold, and what the chain could not take.
    struct Tok { int kind; uint32_t size, dist, lit; int bits; int err; };   // kind 1 literal, 2 match, 3 end of block
    auto slow_token = [&](uint32_t bp, int n_lit, int n_dst) -> Tok {
        Tok t{0, 0u, 0u, 0u, 0, INF_OK};
        InflPeek P = peek64(bp);
        const uint32_t ej = T.u16(IT_FAST, (int)((uint32_t)P.bb & ((1u << IT_FAST_BITS) - 1)));
        int s;
        if (ej != 0) {
            s = (int)(ej & 511u);
            P.bb >>= ej >> 9;
            P.bc -= (int)(ej >> 9);
        } else {
            s = infl_decode(T, P, IT_LIT_SYM, IT_LIT_THR, IT_LIT_BASE, n_lit);
        }
        s = __builtin_amdgcn_readfirstlane(s);
        if (s < 0) { t.err = INF_BAD_CODE; return t; }
        if (s < 256) { t.kind = 1; t.size = 1; t.lit = (uint32_t)s; }
        else if (s == 256) { t.kind = 3; }
        else {
            s -= 257;
            if (s >= 29) { t.err = INF_BAD_CODE; return t; }
            uint32_t len;
            if (s < 8) len = 3u + (uint32_t)s;
            else if (s == 28) len = 258u;
            else {
                const int x = (s - 4) >> 2;
                len = 3u + ((4u + ((uint32_t)s & 3u)) << x) + ((uint32_t)P.bb & ((1u << x) - 1u));
                P.bb >>= x;
                P.bc -= x;
            }
            int d;
            const uint32_t dq = T.u16(IT_DFAST, (int)((uint32_t)P.bb & ((1u << IT_DFAST_BITS) - 1)));
            if (dq) {
                d = (int)(dq & 511u);
                P.bb >>= dq >> 9;
                P.bc -= (int)(dq >> 9);
            } else {
                d = infl_decode(T, P, IT_DST_SYM, IT_DST_THR, IT_DST_BASE, n_dst);
            }
            d = __builtin_amdgcn_readfirstlane(d);
            if (d < 0 || d >= 30) { t.err = INF_BAD_CODE; return t; }
            uint32_t dist;
            if (d < 4) dist = 1u + (uint32_t)d;
            else {
                const int x = (d >> 1) - 1;
                dist = 1u + ((2u + ((uint32_t)d & 1u)) << x) + ((uint32_t)P.bb & ((1u << x) - 1u));
                P.bb >>= x;
                P.bc -= x;
            }
            t.kind = 2;
            t.size = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
            t.dist = (uint32_t)__builtin_amdgcn_readfirstlane((int)dist);
        }
        t.bits = __builtin_amdgcn_readfirstlane(64 - P.bc);
        return t;
    };

    uint32_t bitpos = 8u * lead;                                              // everything below is the same in every lane
    uint32_t op = 0;                                                          // bytes of output so far
    int n_lit = 0, n_dst = 0;
    int err = INF_OK;
    bool in_block = false, last = false, done = false;

    for (int guard = 0; guard < (1 << 22) && !done && !err; ++guard) {        // every pass consumes input or ends
        if (bitpos > end_bit + 64u) { err = INF_INPUT_OVERRUN; break; }
        if (!in_block) {
            // ---- a block header: lane 0 reads it and builds the tables ----
            uint32_t stored_at = 0, stored_len = 0;
            int e0 = INF_OK;
            if (lane == 0) {
                InflRingBits B;
                B.ring = iring;
                B.start_bits(bitpos);
                B.refill();
                last = B.take(1) != 0;
                const uint32_t type = B.take(2);
                if (type == 0) {                                              // stored
                    B.take(B.bc & 7);
                    B.refill();
                    const uint32_t len = B.take(16);
                    B.refill();
                    const uint32_t nlen = B.take(16);
                    if ((len ^ nlen) != 0xffffu) e0 = INF_BAD_STORED;
                    stored_at = B.bit_pos() >> 3;                             // byte-aligned here
                    stored_len = len;
                    if (!e0 && stored_at + len > lead + K.in_len) e0 = INF_INPUT_OVERRUN;
                    if (!e0 && op + len > K.out_len) e0 = INF_OUTPUT_OVERRUN;
                    if (!e0) stored_len |= 0x80000000u;                        // "copy this" (even an empty one: the reader moves on)
                } else if (type == 3) {
                    e0 = INF_BAD_BLOCK_TYPE;
                } else {
                    int dst_at = 0;
                    if (type == 1) {                                          // fixed codes (RFC 1951, 3.2.6)
                        for (int s = 0; s < 288; ++s) T.u8(IT_LENS, s) = (unsigned char)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
                        for (int s = 0; s < 32; ++s) T.u8(IT_LENS, 288 + s) = 5;
                        n_lit = 288; n_dst = 32; dst_at = 288;
                    } else {                                                  // dynamic codes (3.2.7)
                        B.refill();
                        n_lit = (int)B.take(5) + 257;
                        n_dst = (int)B.take(5) + 1;
                        const int n_clc = (int)B.take(4) + 4;
                        if (n_lit > 286 || n_dst > 30) e0 = INF_BAD_LENGTHS;
                        for (int s = 0; s < 19; ++s) T.u8(IT_LENS, s) = 0;
                        for (int i = 0; i < n_clc && !e0; ++i) {
                            B.refill();
                            // order of the code length code's lengths (3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                            const int s = (int)((i < 12 ? 0x22caa324e804a30ull >> (5 * i) : 0x3c2e1346cull >> (5 * (i - 12))) & 31);
                            T.u8(IT_LENS, s) = (unsigned char)B.take(3);
                        }
                        if (!e0 && !infl_build(T, IT_LENS, 0, 19, IT_CLC_SYM, IT_CLC_THR, IT_CLC_BASE, IK_CODES)) e0 = INF_BAD_LENGTHS;
                        int at = 0, prev = 0;
                        const int total = n_lit + n_dst;
                        while (!e0 && at < total) {                           // (at most 316 lengths of at most 14 bits: the ring holds them)
                            B.refill();
                            if (B.wi > n_words + 4) { e0 = INF_INPUT_OVERRUN; break; }
                            const int s = infl_decode(T, B, IT_CLC_SYM, IT_CLC_THR, IT_CLC_BASE, 19);
                            if (s < 0) { e0 = INF_BAD_CODE; break; }
                            if (s < 16) {
                                T.u8(IT_LENS, at) = (unsigned char)s;
                                prev = s;
                                ++at;
                            } else {
                                int rep, val = 0;
                                if (s == 16) {
                                    if (at == 0) { e0 = INF_BAD_LENGTHS; break; }
                                    val = prev;
                                    rep = 3 + (int)B.take(2);
                                } else if (s == 17) {
                                    rep = 3 + (int)B.take(3);
                                } else {
                                    rep = 11 + (int)B.take(7);
                                }
                                if (at + rep > total) { e0 = INF_BAD_LENGTHS; break; }
                                for (int k = 0; k < rep; ++k) T.u8(IT_LENS, at + k) = (unsigned char)val;
                                at += rep;
                                prev = val;
                            }
                        }
                        if (!e0 && T.u8(IT_LENS, 256) == 0) e0 = INF_BAD_LENGTHS;        // no end-of-block code
                        dst_at = n_lit;
                    }
                    if (!e0 && !infl_build(T, IT_LENS, 0, n_lit, IT_LIT_SYM, IT_LIT_THR, IT_LIT_BASE, IK_LENS, IT_FAST, IT_FAST_BITS)) e0 = INF_BAD_LENGTHS;
                    if (!e0 && !infl_build(T, IT_LENS, dst_at, n_dst, IT_DST_SYM, IT_DST_THR, IT_DST_BASE, IK_DISTS, IT_DFAST, IT_DFAST_BITS)) e0 = INF_BAD_LENGTHS;
                    in_block = true;
                }
                bitpos = B.bit_pos();
            }
            err = __builtin_amdgcn_readfirstlane(e0);
            if (err) break;
            bitpos = (uint32_t)__builtin_amdgcn_readfirstlane((int)bitpos);
            in_block = __builtin_amdgcn_readfirstlane((int)in_block) != 0;
            last = __builtin_amdgcn_readfirstlane((int)last) != 0;
            n_lit = __builtin_amdgcn_readfirstlane(n_lit);
            n_dst = __builtin_amdgcn_readfirstlane(n_dst);
            stored_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)stored_len);
            __builtin_amdgcn_wave_barrier();                                  // the tables are there for everybody
            if (stored_len & 0x80000000u) {
                // a stored block: its bytes follow in the input; copied by the wavefront, then the reader goes on behind them
                stored_len &= 0x7fffffffu;
                stored_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)stored_at);
                const unsigned char* src = reinterpret_cast<const unsigned char*>(words) + stored_at;
                for (uint32_t k = lane; k < stored_len; k += 64) {
                    const unsigned char v = src[k];
                    oring[(op + k) & (IW_OUT - 1)] = v;
                    gout[op + k] = v;
                }
                __builtin_amdgcn_wave_barrier();
                op += stored_len;
                if (last) { done = true; continue; }
                const uint32_t next = stored_at + stored_len;
                bitpos = 8u * next;
                loaded = (next >> 2) & ~127u;                                 // the ring starts over at the reader's new place
                feed(next >> 2);                                              // (slots are word index mod IW_IN_WORDS; `loaded` stays a multiple of 128)
                continue;
            }
            feed(bitpos >> 5);
            continue;
        }
        // ---- inside a block: the whole TOKEN that would start at each of the next 64 bits (a literal, or a length
        // code with its extra bits, distance code and extra bits: at most 37 bits through the direct tables) ----
        uint32_t kind = 0, tlen = 0, size = 0, dist = 0, lit = 0;             // kind: 0 stop here, 1 literal, 2 match
        {
            const uint32_t p = bitpos + (uint32_t)lane, w = p >> 5, sh = p & 31u;
            const uint32_t a = iring[w & (IW_IN_WORDS - 1)], c = iring[(w + 1) & (IW_IN_WORDS - 1)];
            uint32_t d = iring[(w + 2) & (IW_IN_WORDS - 1)];
            asm volatile("" : "+v"(d));                                       // all three words are asked for at once (one wait, not two)
            uint64_t v = ((uint64_t)__builtin_amdgcn_alignbit(d, c, sh) << 32) | __builtin_amdgcn_alignbit(c, a, sh);
            const uint32_t e = T.u16(IT_FAST, (int)((uint32_t)v & ((1u << IT_FAST_BITS) - 1)));
            const uint32_t sym = e & 511u;
            uint32_t used = e >> 9;
            if (e != 0 && sym < 256u) {
                kind = 1; tlen = used; size = 1; lit = sym;
            } else if (e != 0 && sym > 256u && sym < 257u + 29u) {
                const uint32_t s = sym - 257u;
                v >>= used;
                uint32_t x = s < 8u || s == 28u ? 0u : (s - 4u) >> 2;
                const uint32_t len = s < 8u ? 3u + s : (s == 28u ? 258u : 3u + ((4u + (s & 3u)) << x) + ((uint32_t)v & ((1u << x) - 1u)));
                v >>= x;
                used += x;
                const uint32_t dq = T.u16(IT_DFAST, (int)((uint32_t)v & ((1u << IT_DFAST_BITS) - 1)));
                const uint32_t dsym = dq & 511u;
                if (dq != 0 && dsym < 30u) {
                    v >>= dq >> 9;
                    used += dq >> 9;
                    x = dsym < 4u ? 0u : (dsym >> 1) - 1u;
                    dist = dsym < 4u ? 1u + dsym : 1u + ((2u + (dsym & 1u)) << x) + ((uint32_t)v & ((1u << x) - 1u));
                    used += x;
                    kind = 2; tlen = used; size = len;
                }
            }
        }
        // The chain of true starts from bit 0: token after token while each one's source lies wholly before this pass's
        // output (then the tokens of a pass do not depend on each other and their lanes write them side by side);
        // a match that does not (a short distance, a long one that reaches the ring's far end, one of more than 32
        // bytes) ends the chain and is taken on its own below, as are the end-of-block code and codes the direct
        // tables do not hold.
        unsigned long long starts = 0;
        int j = 0;
        uint32_t P = 0, mypos = 0;                                            // bytes this pass writes; where this lane's token goes
        // (scalar copies: the walk below then runs on the scalar unit, its lane reads apart)
        const uint32_t op_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)op);
        const uint32_t room_s = K.out_len - op_s;
        const uint32_t bit_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)bitpos);
        uint32_t n_max = 0;                                                   // the longest match of the pass
        while (j < 64) {
            uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)kind, j);
            if (kj == 0) {
                // a code beyond the direct tables (a tenth of the tokens' chains would end here): decoded the ordinary way,
                // handed to lane j, and the chain goes on; the end-of-block code and a bad code are left to the pass below
                const Tok t = slow_token(bit_s + (uint32_t)j, n_lit, n_dst);
                if (t.err || t.kind == 3) break;
                kj = (uint32_t)t.kind;
                kind = lane == j ? kj : kind;
                size = lane == j ? t.size : size;
                dist = lane == j ? t.dist : dist;
                lit = lane == j ? t.lit : lit;
                tlen = lane == j ? (uint32_t)t.bits : tlen;
            }
            const uint32_t sz = (uint32_t)__builtin_amdgcn_readlane((int)size, j);
            if (sz > room_s - P) break;
            if (kj == 2) {
                const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)dist, j);
                if (dj > op_s + P || dj < sz + P || sz > 32u || dj > (uint32_t)IW_OUT - 2304u) break;
            }
            mypos = lane == j ? op_s + P : mypos;
            starts |= 1ull << j;
            if (kj == 2) n_max = max(n_max, sz);
            P += sz;
            j += (int)__builtin_amdgcn_readlane((int)tlen, j);
        }
        if (starts) {
            const bool mine = (starts >> lane) & 1ull;
            if (mine && kind == 1) {
                oring[mypos & (IW_OUT - 1)] = (unsigned char)lit;
                gout[mypos] = (unsigned char)lit;
            }
            // matches: every lane its own, four bytes read, then written (sources lie before the pass: no order among them;
            // at most 2 KiB are written in a pass, and no source is nearer than that to the ring's far end)
            const uint32_t n_mine = mine && kind == 2 ? size : 0u;
            for (uint32_t k0 = 0; k0 < n_max; k0 += 4) {
                unsigned char x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = k0 + q < n_mine ? oring[(mypos - dist + k0 + q) & (IW_OUT - 1)] : (unsigned char)0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k0 + q < n_mine) { oring[(mypos + k0 + q) & (IW_OUT - 1)] = x[q]; gout[mypos + k0 + q] = x[q]; }
            }
            op += P;
            __builtin_amdgcn_wave_barrier();
        }
        bitpos += (uint32_t)j;
        if (!starts) {
            // ---- a token the chain could not take (every lane the same) ----
            const Tok t = slow_token(bitpos, n_lit, n_dst);
            if (t.err) { err = t.err; break; }
            if (t.kind == 1) {
                if (op >= K.out_len) { err = INF_OUTPUT_OVERRUN; break; }
                if (lane == 0) { oring[op & (IW_OUT - 1)] = (unsigned char)t.lit; gout[op] = (unsigned char)t.lit; }
                op += 1;
                __builtin_amdgcn_wave_barrier();
            } else if (t.kind == 3) {
                in_block = false;
                if (last) done = true;
            } else {
                const uint32_t len = t.size, dist = t.dist;
                if (dist > op) { err = INF_BAD_DISTANCE; break; }
                if (len > K.out_len - op) { err = INF_OUTPUT_OVERRUN; break; }
                if (dist >= 64) {
                    // 64 bytes per step: a step's source lies wholly before what the step writes
                    for (uint32_t k = lane; k < len; k += 64) {
                        const unsigned char x = oring[(op - dist + k) & (IW_OUT - 1)];
                        oring[(op + k) & (IW_OUT - 1)] = x;
                        gout[op + k] = x;
                    }
                } else {
                    // a period shorter than the wavefront: every byte is one of the `dist` bytes before the match
                    const uint32_t m = (uint32_t)lane % dist;
                    const unsigned char x = oring[(op - dist + m) & (IW_OUT - 1)];
                    const uint32_t span = dist * (64u / dist);                // bytes per step: whole periods
                    if ((uint32_t)lane < span) {
                        for (uint32_t k = lane; k < len; k += span) {
                            oring[(op + k) & (IW_OUT - 1)] = x;
                            gout[op + k] = x;
                        }
                    }
                }
                op += len;
                __builtin_amdgcn_wave_barrier();
            }
            bitpos += (uint32_t)t.bits;
        }
        feed(bitpos >> 5);
    }
    if (!err && !done) err = INF_INPUT_OVERRUN;                               // the pass limit: no valid stream gets here
    if (!err && op != K.out_len) err = INF_SIZE_MISMATCH;
    if (lane == 0) status[b] = err;
}

}  // namespace pg
