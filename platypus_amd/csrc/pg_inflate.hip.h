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
// LDS: tables (one lane's layout) | last 32 KiB of output (ring) | 2 KiB of compressed input (ring of words) |
// tokens of a batch.
// ---------------------------------------------------------------------------------------------------------
constexpr int IW_OUT = 32768;                    // output ring: the deflate window
constexpr int IW_IN_WORDS = 512;                 // input ring, 32-bit words
constexpr int IW_TOK = 64;                       // tokens per batch
constexpr int IW_LDS = IT_U16 * 2 + IW_OUT + IW_IN_WORDS * 4 + IW_TOK * 4 + IW_TOK * 2 + 64;

struct InflRingBits {                            // lane 0's bit reader over the input ring
    const uint32_t* ring;
    uint32_t wi;                                 // next word to take, counted from the aligned start of the stream
    uint64_t bb;
    int bc;
    __device__ void start(uint32_t byte_off) {   // byte_off from the aligned start
        wi = byte_off >> 2;
        const int skip = (int)(byte_off & 3);
        bb = (uint64_t)ring[wi & (IW_IN_WORDS - 1)] >> (8 * skip);
        ++wi;
        bc = 32 - 8 * skip;
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
    __device__ uint32_t byte_pos() const { return 4 * wi - (uint32_t)(bc >> 3); }   // first byte not consumed (bc multiple of 8)
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
    uint32_t* tok_val = iring + IW_IN_WORDS;                                  // literal: 0x80000000 | byte; match: len << 16 | dist
    unsigned short* tok_pos = reinterpret_cast<unsigned short*>(tok_val + IW_TOK);   // where it starts in the block's output
    unsigned char* const gout = out + K.out_off;
    const uint32_t* const words = reinterpret_cast<const uint32_t*>(raw) + (K.in_off >> 2);   // raw is 4-byte aligned and padded
    const uint32_t lead = K.in_off & 3u;                                      // the stream starts `lead` bytes into words[0]
    const uint32_t n_words = (lead + K.in_len + 3) >> 2;                     // words that hold the stream
    uint32_t loaded = 0;                                                      // words in the ring so far (whole wavefront knows)

    // keeps the ring full: 128 words at a time while that many are free (consumed: words lane 0 has taken)
    auto feed = [&](uint32_t consumed) {
        while (loaded < n_words + 8 && loaded + 128 <= consumed + IW_IN_WORDS) {
            const uint32_t w = loaded + 2 * lane;
            iring[w & (IW_IN_WORDS - 1)] = words[w];                          // (up to 8 words past the stream: padding)
            iring[(w + 1) & (IW_IN_WORDS - 1)] = words[w + 1];
            loaded += 128;
        }
        __builtin_amdgcn_wave_barrier();
    };
    feed(0);

    // lane 0's state
    InflRingBits B;
    B.ring = iring;
    B.bb = 0; B.bc = 0; B.wi = 0;
    if (lane == 0) B.start(lead);
    uint32_t op = 0;                                                          // bytes of output so far
    int n_lit = 0, n_dst = 0;
    int err = INF_OK;
    bool in_block = false, last = false, done = false;

    for (int guard = 0; guard < (1 << 20) && !done; ++guard) {                // every pass consumes input or ends
        // ---- lane 0: a block header, or up to IW_TOK tokens ----
        int n_tok = 0;
        uint32_t stored_at = 0, stored_len = 0;                               // a stored block for the wavefront to copy
        if (lane == 0) {
            if (!in_block) {
                B.refill();
                last = B.take(1) != 0;
                const uint32_t type = B.take(2);
                if (type == 0) {                                              // stored
                    B.take(B.bc & 7);
                    B.refill();
                    const uint32_t len = B.take(16);
                    B.refill();
                    const uint32_t nlen = B.take(16);
                    if ((len ^ nlen) != 0xffffu) err = INF_BAD_STORED;
                    stored_at = B.byte_pos();
                    stored_len = len;
                    if (!err && stored_at + len > lead + K.in_len) err = INF_INPUT_OVERRUN;
                    if (!err && op + len > K.out_len) err = INF_OUTPUT_OVERRUN;
                    if (!err) {
                        stored_len |= 0x80000000u;                             // "copy this" (even an empty one: the reader moves on)
                        if (last) done = true;
                    }
                } else if (type == 3) {
                    err = INF_BAD_BLOCK_TYPE;
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
                        if (n_lit > 286 || n_dst > 30) err = INF_BAD_LENGTHS;
                        for (int s = 0; s < 19; ++s) T.u8(IT_LENS, s) = 0;
                        for (int i = 0; i < n_clc && !err; ++i) {
                            B.refill();
                            // order of the code length code's lengths (3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                            const int s = (int)((i < 12 ? 0x22caa324e804a30ull >> (5 * i) : 0x3c2e1346cull >> (5 * (i - 12))) & 31);
                            T.u8(IT_LENS, s) = (unsigned char)B.take(3);
                        }
                        if (!err && !infl_build(T, IT_LENS, 0, 19, IT_CLC_SYM, IT_CLC_THR, IT_CLC_BASE, IK_CODES)) err = INF_BAD_LENGTHS;
                        int at = 0, prev = 0;
                        const int total = n_lit + n_dst;
                        while (!err && at < total) {                          // (at most 316 lengths of at most 14 bits: the ring holds them)
                            B.refill();
                            if (B.wi > n_words + 4) { err = INF_INPUT_OVERRUN; break; }
                            const int s = infl_decode(T, B, IT_CLC_SYM, IT_CLC_THR, IT_CLC_BASE, 19);
                            if (s < 0) { err = INF_BAD_CODE; break; }
                            if (s < 16) {
                                T.u8(IT_LENS, at) = (unsigned char)s;
                                prev = s;
                                ++at;
                            } else {
                                int rep, val = 0;
                                if (s == 16) {
                                    if (at == 0) { err = INF_BAD_LENGTHS; break; }
                                    val = prev;
                                    rep = 3 + (int)B.take(2);
                                } else if (s == 17) {
                                    rep = 3 + (int)B.take(3);
                                } else {
                                    rep = 11 + (int)B.take(7);
                                }
                                if (at + rep > total) { err = INF_BAD_LENGTHS; break; }
                                for (int k = 0; k < rep; ++k) T.u8(IT_LENS, at + k) = (unsigned char)val;
                                at += rep;
                                prev = val;
                            }
                        }
                        if (!err && T.u8(IT_LENS, 256) == 0) err = INF_BAD_LENGTHS;       // no end-of-block code
                        dst_at = n_lit;
                    }
                    if (!err && !infl_build(T, IT_LENS, 0, n_lit, IT_LIT_SYM, IT_LIT_THR, IT_LIT_BASE, IK_LENS, IT_FAST, IT_FAST_BITS)) err = INF_BAD_LENGTHS;
                    if (!err && !infl_build(T, IT_LENS, dst_at, n_dst, IT_DST_SYM, IT_DST_THR, IT_DST_BASE, IK_DISTS, IT_DFAST, IT_DFAST_BITS)) err = INF_BAD_LENGTHS;
                    in_block = true;
                }
            } else {
                // tokens: while the batch has room and the ring holds the next one whole (at most 48 bits)
                if (B.wi + 3 > loaded && loaded >= n_words + 8) err = INF_INPUT_OVERRUN;   // nothing more will come
                while (!err && n_tok < IW_TOK && B.wi + 3 <= loaded) {
                    B.refill();
                    int s;
                    const uint32_t quick = T.u16(IT_FAST, (int)((uint32_t)B.bb & ((1u << IT_FAST_BITS) - 1)));
                    if (quick) {                                              // a code of up to ten bits: one look-up
                        s = (int)(quick & 511u);
                        B.bb >>= quick >> 9;
                        B.bc -= (int)(quick >> 9);
                    } else {
                        s = infl_decode(T, B, IT_LIT_SYM, IT_LIT_THR, IT_LIT_BASE, n_lit);
                        if (s < 0) { err = INF_BAD_CODE; break; }
                    }
                    if (s < 256) {
                        if (op >= K.out_len) { err = INF_OUTPUT_OVERRUN; break; }
                        tok_val[n_tok] = 0x80000000u | (uint32_t)s;
                        tok_pos[n_tok] = (unsigned short)op;
                        ++n_tok;
                        ++op;
                        continue;
                    }
                    if (s == 256) { in_block = false; if (last) done = true; break; }
                    s -= 257;
                    if (s >= 29) { err = INF_BAD_CODE; break; }
                    int len;
                    if (s < 8) len = 3 + s;
                    else if (s == 28) len = 258;
                    else {
                        const int e = (s - 4) >> 2;
                        len = 3 + ((4 + (s & 3)) << e) + (int)B.take(e);
                    }
                    B.refill();
                    int d;
                    const uint32_t dq = T.u16(IT_DFAST, (int)((uint32_t)B.bb & ((1u << IT_DFAST_BITS) - 1)));
                    if (dq) {
                        d = (int)(dq & 511u);
                        B.bb >>= dq >> 9;
                        B.bc -= (int)(dq >> 9);
                    } else {
                        d = infl_decode(T, B, IT_DST_SYM, IT_DST_THR, IT_DST_BASE, n_dst);
                    }
                    if (d < 0 || d >= 30) { err = INF_BAD_CODE; break; }
                    int dist;
                    if (d < 4) dist = 1 + d;
                    else {
                        const int e = (d >> 1) - 1;
                        dist = 1 + ((2 + (d & 1)) << e) + (int)B.take(e);
                    }
                    if ((uint32_t)dist > op) { err = INF_BAD_DISTANCE; break; }
                    if ((uint32_t)len > K.out_len - op) { err = INF_OUTPUT_OVERRUN; break; }
                    tok_val[n_tok] = ((uint32_t)len << 16) | (uint32_t)dist;
                    tok_pos[n_tok] = (unsigned short)op;
                    ++n_tok;
                    op += (uint32_t)len;
                }
                if (!err && B.wi > n_words + 4) err = INF_INPUT_OVERRUN;   // the stream ran out inside a block
            }
        }
        // ---- the wavefront: what lane 0 found ----
        err = __builtin_amdgcn_readfirstlane(err);
        if (err) break;
        n_tok = __builtin_amdgcn_readfirstlane(n_tok);
        done = __builtin_amdgcn_readfirstlane((int)done) != 0;
        stored_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)stored_len);
        __builtin_amdgcn_wave_barrier();
        if (stored_len & 0x80000000u) {
            // a stored block: its bytes follow in the input; copied by the wavefront, then the reader starts behind them
            stored_len &= 0x7fffffffu;
            stored_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)stored_at);
            const uint32_t op0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)op);
            const unsigned char* src = reinterpret_cast<const unsigned char*>(words) + stored_at;
            for (uint32_t k = lane; k < stored_len; k += 64) {
                const unsigned char v = src[k];
                oring[(op0 + k) & (IW_OUT - 1)] = v;
                gout[op0 + k] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) op += stored_len;
            if (done) continue;                                               // it was the stream's last block
            const uint32_t next = stored_at + stored_len;
            loaded = (next >> 2) & ~127u;                                     // the ring starts over at the reader's new place
            // (ring slots are word index mod IW_IN_WORDS, and `loaded` stays a multiple of 128)
            feed(next >> 2);
            if (lane == 0) B.start(next);
            continue;
        }
        if (n_tok > 0) {
            const bool mine = lane < n_tok;
            const uint32_t v = mine ? tok_val[lane] : 0x80000000u;
            const uint32_t p = mine ? tok_pos[lane] : 0u;
            // Everything goes into the ring in stream order -- the literals between two matches at once, then the match:
            // a byte written at place q takes the slot of q - 32768, which nothing from q on can refer to (a write out
            // of order could take a slot that an earlier match of the batch still has to read).
            const bool lit = mine && (v & 0x80000000u);
            int prev = 0;
            unsigned long long matches = __ballot(mine && !(v & 0x80000000u));
            while (matches) {                                                 // the matches in their order
                const int t = __ffsll((long long)matches) - 1;
                matches &= matches - 1;
                if (lit && lane >= prev && lane < t) {
                    oring[p & (IW_OUT - 1)] = (unsigned char)v;
                    gout[p] = (unsigned char)v;
                }
                prev = t + 1;
                __builtin_amdgcn_wave_barrier();
                const uint32_t mv = (uint32_t)__builtin_amdgcn_readlane((int)v, t);
                const uint32_t mp = (uint32_t)__builtin_amdgcn_readlane((int)p, t);
                const uint32_t len = mv >> 16, dist = mv & 0xffffu;
                if (dist >= 64) {
                    // 64 bytes per step: a step's source lies wholly before what the step writes
                    for (uint32_t k = lane; k < len; k += 64) {
                        const unsigned char x = oring[(mp - dist + k) & (IW_OUT - 1)];
                        oring[(mp + k) & (IW_OUT - 1)] = x;
                        gout[mp + k] = x;
                    }
                } else {
                    // a period shorter than the wavefront: every byte is one of the `dist` bytes before the match
                    const uint32_t m = (uint32_t)lane % dist;
                    const unsigned char x = oring[(mp - dist + m) & (IW_OUT - 1)];
                    const uint32_t span = dist * (64u / dist);                // bytes per step: whole periods
                    for (uint32_t k = lane; k < len; k += span) {
                        if ((uint32_t)lane < span) {
                            oring[(mp + k) & (IW_OUT - 1)] = x;
                            gout[mp + k] = x;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (lit && lane >= prev) {                                        // the literals behind the last match
                oring[p & (IW_OUT - 1)] = (unsigned char)v;
                gout[p] = (unsigned char)v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        feed((uint32_t)__builtin_amdgcn_readfirstlane((int)B.wi));
    }
    if (!err && !done) err = INF_INPUT_OVERRUN;                               // the pass limit: no valid stream gets here
    if (!err && (uint32_t)__builtin_amdgcn_readfirstlane((int)op) != K.out_len) err = INF_SIZE_MISMATCH;
    if (lane == 0) status[b] = err;
}

}  // namespace pg
