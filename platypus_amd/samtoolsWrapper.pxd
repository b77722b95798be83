# The read record the alignment path consumes.  Field for field the struct of the
# reference (platypus/samtoolsWrapper.pxd:293-315), so code written against the
# reference's cAlignedRead compiles against this one unchanged.  The path reads
# seq, qual (RAW phred, no +33), pos, rlen, end and mapq only.
ctypedef struct cAlignedRead:
    char* seq
    char* qual
    int* cigarOps
    int* cigarLens
    int cigarLen
    int pos
    int rlen
    int end
    int mapq
    int isReverse
    int isPaired
    int isDuplicate
    int isUnMapped
    int insertSize
    int mateIsUnmapped
    int mateIsReverse
    int matePos
    int isReadOne
    int nQualsBelow5
    int nQualsBelow10
    int nQualsBelow15
    int nQualsBelow20
