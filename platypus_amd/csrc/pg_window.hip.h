// One window per call: everything pg_haps_upload, pg_reads_upload and the pair-list
// preparation do with a dozen kernels and three waits, for ONE small dense window
// (every haplotype x every read, the matrix Haplotype.alignReads walks for each
// haplotype of a window, platypus/chaplotype.pyx:239-270), as ONE kernel.  The host
// has worked out every offset and count (the window is small) and sent them with the
// sequence bytes as one block; k_window_pack turns that block into the arrays the
// anchoring and alignment kernels take.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_align.hip.h"
#include "pg_kernels.hip.h"
#include "pg_prep.hip.h"

namespace pg {

struct WindowArgs {
    int n_haps, n_reads, n_vote_haps, pad, cap;
    // from the host
    const uint8_t* hap_raw; const int64_t* hap_soff; const int64_t* hap_doff; const int64_t* hap2off;
    const int32_t* hap_vote_ord;
    const uint8_t* rseq; const uint8_t* rqual; const int64_t* roff; const int64_t* read_xoff; const int64_t* read2off;
    const int32_t* read_len;
    const SeqCodes* codes;
    // made here
    uint16_t* hapx; uint32_t* hap2; uint32_t* hapU; uint32_t* hapUbits;
    uint16_t* readx; uint32_t* read2;
    int32_t* pair_hap; int32_t* pair_read; VoteRec* recs;
    int32_t* flags;         // [0] |= 1 a quality above 127, 2 a base without a code
};

// Workgroups 0 .. n_haps - 1: one haplotype each (element row, packed row and, for an indel-bearing one, its
// uniqueness rows); the next (n_reads + 3) / 4: one read per wavefront; the rest: the pair list (haplotype-major)
// and the vote records (grouped by indel-bearing haplotype, reads in order).
__global__ __launch_bounds__(256) void k_window_pack(const WindowArgs A)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int read_blocks = (A.n_reads + 3) / 4;
    if (b < A.n_haps) {
        const int64_t s0 = A.hap_soff[b];
        const int len = (int)(A.hap_soff[b + 1] - s0);
        if (pack_hapx_one(A.hap_raw + s0, len, A.hapx + A.hap_doff[b], A.codes, A.pad, A.cap)) atomicOr(A.flags, 2);
        uint32_t* row = A.hap2 + A.hap2off[b];
        pack_codes_row(A.hap_raw + s0, len, row, (int)(A.hap2off[b + 1] - A.hap2off[b]), tid, 256);
        const int ord = A.hap_vote_ord[b];
        if (ord >= 0) {                                   // wave-uniform
            __threadfence_block();                        // the packed row is read back below
            __syncthreads();
            hap_unique_one(row, len, A.hapU + A.hap2off[b], A.hapUbits + (size_t)ord * (kHashSize / 32));
        }
        return;
    }
    if (b < A.n_haps + read_blocks) {
        const int r = (b - A.n_haps) * 4 + (tid >> 6), lane = tid & 63;
        if (r >= A.n_reads) return;
        const int64_t s0 = A.roff[r];
        const int len = (int)(A.roff[r + 1] - s0);
        const int64_t d0 = A.read_xoff[r];
        const int f = pack_readx_one(A.rseq + s0, A.rqual + s0, len, A.readx + d0, (int)(A.read_xoff[r + 1] - d0), A.codes, lane);
        if (f) atomicOr(A.flags, f);
        pack_codes_row(A.rseq + s0, len, A.read2 + A.read2off[r], (int)(A.read2off[r + 1] - A.read2off[r]), lane, 64);
        return;
    }
    const int first = (b - A.n_haps - read_blocks) * 256 + tid;
    const int step = (int)(gridDim.x - A.n_haps - read_blocks) * 256;
    const int np = A.n_haps * A.n_reads;
    for (int p = first; p < np; p += step) {
        const int h = p / A.n_reads, r = p - h * A.n_reads;
        A.pair_hap[p] = h;
        A.pair_read[p] = r;
        const int ord = A.hap_vote_ord[h];
        if (ord >= 0) A.recs[ord * A.n_reads + r] = VoteRec{p, A.read_len[r], A.read2off[r]};
    }
}

}  // namespace pg
