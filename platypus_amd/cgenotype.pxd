from platypus_amd.chaplotype cimport Haplotype
from platypus_amd.samtoolsWrapper cimport cAlignedRead

# platypus/cgenotype.pxd:5-17
cdef class DiploidGenotype:
    cdef public list haplotypes
    cdef Haplotype hap1
    cdef Haplotype hap2
    cdef double calculateDataLikelihood(
            DiploidGenotype self,
            cAlignedRead** start,
            cAlignedRead** end,
            int individualIndex,
            int nReads,
            int nIndividuals
    ) except? -1.0

cdef list generateAllGenotypesFromHaplotypeList(int ploidy, list haplotypes)
