# Declarations of the caller layer.  Names follow the numeric part of platypus/cpopulation.pxd:4-47;
# the VCF-formatting members of the reference's Population live in platypus_amd/pipeline.py.

cdef class Population:
    # results of one window, copied from the device
    cdef double* frequencies                       # per haplotype
    cdef double** genotypePosteriors               # [individual][genotype]
    cdef int** haplotypeIndexes                    # [genotype] -> two haplotype indexes
    cdef int* nReads                               # per individual
    cdef int nHaplotypes, nIndividuals, nGenotypes, verbosity
    cdef public list genotypeCalls, genotypeLikelihoods, haplotypes
    cdef list readBuffers
    # bookkeeping of this implementation
    cdef public int nIterations, windowIndex
    cdef public double maxChange
    cdef long generation
    cdef double calculatePosterior(self, object var) except? -1.0


cdef class Caller:
    cdef int nHaplotypes, nIndividuals, nGenotypes, ploidy, verbosity
    cdef int* nReads
    cdef list haplotypes, genotypes, readBuffers
    cdef object options
    cdef setup(self, list haplotypes, list genotypes, int nInd, int ploidy, int verbosity, list readBuffers)
    cdef Population call(self, int maxIters=*)


# collect -> run -> consume over many windows (the rework of the window loop of
# platypus/variantcaller.pyx:116-187 into device batches)
cdef class WindowBatch:
    cdef list windows
    cdef list pairBase
    cdef public list populations
    cdef public dict timings
