# the numeric part of platypus/cpopulation.pxd:4-47 (VCF formatting is not on the path)
cdef class Population:
    cdef public list genotypeCalls
    cdef public list genotypeLikelihoods
    cdef public list haplotypes
    cdef double** genotypePosteriors
    cdef double* frequencies
    cdef int** haplotypeIndexes
    cdef int* nReads
    cdef int nHaplotypes
    cdef int nIndividuals
    cdef int nGenotypes
    cdef int verbosity
    cdef list readBuffers
    cdef public int nIterations
    cdef public double maxChange
    cdef public int windowIndex
    cdef long generation
    cdef double calculatePosterior(self, object var) except? -1.0


cdef class Caller:
    cdef int nHaplotypes
    cdef int nIndividuals
    cdef int nGenotypes
    cdef int ploidy
    cdef int verbosity
    cdef list haplotypes
    cdef list genotypes
    cdef list readBuffers
    cdef int* nReads
    cdef object options
    cdef Population call(self, int maxIters=*)
    cdef setup(self, list haplotypes, list genotypes, int nInd, int ploidy, int verbosity, list readBuffers)


# collect -> run -> consume over many windows (the rework of the window loop of
# platypus/variantcaller.pyx:116-187 into device batches)
cdef class WindowBatch:
    cdef list windows
    cdef public list populations
    cdef public dict timings
