from platypus_amd.samtoolsWrapper cimport cAlignedRead
from platypus_amd.platgpu cimport pg_ctx

# platypus/calign.pxd:5; the table is malloc'd and the caller frees it (platypus/calign.pyx:70)
cdef int* hash_sequence(char* sequence) except NULL
cdef int map_read(char* read, int sequence_length, int* sequence_hash) except? -2147483647

# same signature as the reference, platypus/calign.pxd:7-17
cdef double alignNoTraceback(
        cAlignedRead* read,
        char* hapSeq,
        int hapStart,
        int varSeqLen,
        char* homopolgapq_or_localgapopen,
        int gapextend,
        int nucprior,
        int hapLen,
        int useAsHomopolGapQ
) except? -1.0

# batch form: nHaps haplotypes x the reads [start, end) in ONE launch;
# likelihoods[h * nReads + r], scores/anchors optional (may be NULL)
cdef int alignBatch(
        int nHaps,
        char** hapSeqs,
        int* hapStarts,
        int* varSeqLens,
        int* hapLens,
        cAlignedRead** start,
        cAlignedRead** end,
        char* homopolgapq,
        int gapextend,
        int nucprior,
        double* likelihoods,
        int* scores,
        int* anchors
) except -1

cdef pg_ctx* deviceContext() except NULL
cdef int ensureReadLength(int n) except -1
cdef int setParams(char* table, int gapextend, int nucprior) except -1
