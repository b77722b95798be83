// BGZF block inflation on the device: one lane per BGZF block (RFC 1951 raw deflate streams of at most 64 KiB
// each, independent of one another), 64 blocks per wavefront.  Replaces inflate_block of the reference's vendored
// samtools (platypus/c/samtools/bgzf.c, called through bgzf_read_block) for the stretch of a BAM file a region
// fetch reads; the host only finds the block boundaries.
//
// Every lane keeps its Huffman tables in LDS, word-interleaved ([word][lane]): whatever index a lane reads, it
// stays in its own bank.  A table is canonical: per code length the left-aligned upper end of the codes of up to
// that length (thr) and the offset of the length's first symbol in the list of symbols sorted by code (base).
// A symbol is decoded from 15 peeked bits, reversed: its length is the number of thresholds the value has
// reached -- sixteen independent compares on eight LDS words instead of a bit-by-bit walk.  Literal/length codes of
// up to ten bits (nearly all) skip even that: a direct table indexed by the next ten bits holds symbol and length.
//
// The decoder is one function for device and host (LANES = 64 / 1); the host instantiation exists for the
// tests, which run it against zlib on the CPU.
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
constexpr int IT_U16 = IT_FAST + (1 << IT_FAST_BITS);   // 1636 16-bit units = 818 words per lane

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
// Returns false for an over-subscribed set of lengths.
template <int LANES>
__host__ __device__ inline bool infl_build(const InflTab<LANES>& T, int lens_region, int lens_at, int n, int sym, int thr, int base,
                                           bool fast = false) {
    for (int l = 0; l < 16; ++l) T.u16(thr, l) = 0;
    for (int s = 0; s < n; ++s) T.u16(thr, T.u8(lens_region, lens_at + s)) += 1;      // counts, for now
    uint32_t code = 0, off = 0;
    bool ok = true;
    T.u16(thr, 0) = 0;
    T.u16(base, 0) = 0;
    for (int l = 1; l < 16; ++l) {
        const uint32_t c = T.u16(thr, l);
        ok = ok && code + c <= (1u << l);
        T.u16(thr, l) = (unsigned short)(((code + c) << (15 - l)) > 0xffffu ? 0xffffu : ((code + c) << (15 - l)));
        T.u16(base, l) = (unsigned short)(off - code);
        T.u16(IT_POS, l) = (unsigned short)off;
        code = (code + c) << 1;
        off += c;
    }
    if (!ok) return false;
    if (fast) {
        uint32_t* words = reinterpret_cast<uint32_t*>(T.t);
        for (int w = 0; w < (1 << IT_FAST_BITS) / 2; ++w) words[((IT_FAST >> 1) + w) * LANES] = 0;
    }
    for (int s = 0; s < n; ++s) {
        const int l = T.u8(lens_region, lens_at + s);
        if (l) {
            const int p = T.u16(IT_POS, l);
            T.u16(sym, p) = (unsigned short)s;
            T.u16(IT_POS, l) = (unsigned short)(p + 1);
            if (fast && l <= IT_FAST_BITS) {
                // the symbol's code, as the stream presents it (first bit lowest), and every table index that ends in it
                const uint32_t code = (uint32_t)(p - T.u16(base, l)) & 0xffffu;
                const unsigned short entry = (unsigned short)(s | (l << 9));
                for (uint32_t e = rev_bits(code, l); e < (1u << IT_FAST_BITS); e += 1u << l) T.u16(IT_FAST, (int)e) = entry;
            }
        }
    }
    return true;
}

// One symbol: returns it, or -1 for a code no symbol has.  The caller has refilled (bc >= 15 is not needed:
// missing bits read as zero and a stream that ends inside a code is caught by the input bound).
template <int LANES>
__host__ __device__ inline int infl_decode(const InflTab<LANES>& T, InflBits& B, int sym, int thr, int base, int n_sym) {
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
                if (!infl_build(T, IT_LENS, 0, 19, IT_CLC_SYM, IT_CLC_THR, IT_CLC_BASE)) return INF_BAD_LENGTHS;
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
            if (!infl_build(T, IT_LENS, 0, n_lit, IT_LIT_SYM, IT_LIT_THR, IT_LIT_BASE, true)) return INF_BAD_LENGTHS;
            if (!infl_build(T, IT_LENS, dst_at, n_dst, IT_DST_SYM, IT_DST_THR, IT_DST_BASE)) return INF_BAD_LENGTHS;
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

}  // namespace pg
