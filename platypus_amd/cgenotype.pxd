# Declarations of the genotype layer; attribute and method names follow platypus/cgenotype.pxd:5-20
# so that code written against the reference compiles against this module.
from platypus_amd.samtoolsWrapper cimport cAlignedRead
from platypus_amd.chaplotype cimport Haplotype

cdef list generateAllGenotypesFromHaplotypeList(int ploidy, list haplotypes)


cdef class DiploidGenotype:
    cdef Haplotype hap1, hap2
    cdef public list haplotypes      # [hap1, hap2]
    # sum over the reads [start, end) of log(0.5 (L1 + L2)), computed on the device
    cdef double calculateDataLikelihood(DiploidGenotype self, cAlignedRead** start, cAlignedRead** end,
                                        int individualIndex, int nReads, int nIndividuals) except? -1.0
