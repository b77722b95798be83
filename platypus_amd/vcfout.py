"""VCF text for the calls of platypus_amd.pipeline: the line layout of the reference's writer
(platypus/vcf.py:710-739 with the field order of format_formatdata, :297-328), REF/ALT by
the rule of platypus/variantcaller.pyx:442-502, INFO / FILTER / FORMAT fields of
platypus/cpopulation.pyx:46-79."""
import datetime

INFO = [("FR", ".", "Float", "Estimated population frequency"),
        ("PP", ".", "Float", "Posterior probability (phred scaled) that this variant segregates"),
        ("TC", "1", "Integer", "Total coverage at this locus"),
        ("TCR", "1", "Integer", "Total reverse strand coverage at this locus"),
        ("TCF", "1", "Integer", "Total forward strand coverage at this locus"),
        ("TR", ".", "Integer", "Total number of reads containing this variant"),
        ("NF", ".", "Integer", "Total number of forward reads containing this variant"),
        ("NR", ".", "Integer", "Total number of reverse reads containing this variant"),
        ("SC", "1", "String", "Genomic sequence 10 bases either side of variant position"),
        ("RMP", ".", "Float", "RMS Position in reads of Variant"),
        ("HP", "1", "Integer", "Homopolmer run length"),
        ("FPV", ".", "Float", "Forward strand p-value"),
        ("RPV", ".", "Float", "Reverse strand p-value"),
        ("ABPV", ".", "Float", "Allele-bias p-value. Testing for low variant coverage"),
        ("MMLQ", ".", "Float", "Median minimum base quality for bases around variant")]
FILTER = [("ab", "Variant fails allele-bias filter"), ("sb", "Variant fails strand-bias filter"),
          ("badReads", "Variant supported only by reads with low quality bases close to variant position, and not present "
                       "on both strands."),
          ("hp10", "Flanking sequence contains homopolymer of length 10 or greater")]   # declared, never raised (cpopulation.pyx:70)
FORMAT = [("GT", "1", "String", "Unphased genotypes"),
          ("GL", ".", "Float", "Genotype log-likelihoods (log10) for AA,AB and BB genotypes, where A = ref and B = variant. "
                               "Only applicable for bi-allelic sites"),
          ("GQ", "1", "Integer", "Genotype quality, as Phred score"),
          ("NR", "1", "Integer", "Number of reads covering variant in this sample")]


def header(samples, options_text=""):
    lines = ["##fileformat=VCFv4.0", "##fileDate=%s" % datetime.date.today(),
             "##source=platypus_amd (Platypus 0.1.5 calling model on MI355X)"]
    if options_text:
        lines.append("##platypusOptions=%s" % options_text)
    for k, n, t, d in INFO:
        lines.append('##INFO=<ID=%s,Number=%s,Type=%s,Description="%s">' % (k, n, t, d))
    for k, d in FILTER:
        lines.append('##FILTER=<ID=%s,Description="%s">' % (k, d))
    for k, n, t, d in FORMAT:
        lines.append('##FORMAT=<ID=%s,Number=%s,Type=%s,Description="%s">' % (k, n, t, d))
    lines.append("\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + list(samples)))
    return lines


def ref_and_alt(chrom, pos, variants, refFile):
    """variantcaller.pyx:442-502"""
    non_snp = any(v.nRemoved != 1 or v.nAdded != 1 for v in variants)
    indel = any(v.nRemoved != v.nAdded for v in variants)
    if not non_snp:
        return refFile.getCharacter(chrom, pos), [v.added for v in variants]
    rlen = max(v.nRemoved for v in variants)
    ref = refFile.getSequence(chrom, pos, pos + rlen + (1 if indel else 0))
    alts = []
    for v in variants:
        seq = bytearray(ref)
        if v.nRemoved == v.nAdded:
            seq[0:len(v.added)] = v.added
        else:
            seq[1:1 + v.nRemoved] = v.added
        alts.append(bytes(seq))
    return ref, alts


def _num(x):
    s = str(x)
    return s[:-2] if s.endswith(".0") and isinstance(x, float) and False else s


def record(call, refFile):
    """the text of one body line (vcf.py:710-739): INFO keys in sorted order, one sample column per individual"""
    ref, alts = ref_and_alt(call.chrom, call.pos, call.variants, refFile)
    join = ",".join
    info_text = "ABPV=%s;FPV=%s;FR=%s;HP=%s;%sNF=%s;NR=%s;PP=%s;RMP=%s;RPV=%s;SC=%s;TC=%s;TCF=%s;TCR=%s;TR=%s" % (
        call.ABPV, call.FPV, join(call.FR), call.HP, "" if call.MMLQ is None else "MMLQ=%s;" % (call.MMLQ,),
        join(map(str, call.NF)), join(map(str, call.NR)), join(call.PP), call.RMP, call.RPV, call.SC.decode(),
        call.TC, call.TCF, call.TCR, join(map(str, call.TR)))
    cols = [call.chrom.decode(), str(call.pos + 1), ".", ref.decode(), join([a.decode() for a in alts]),
            str(call.qual), ";".join(call.filters) if call.filters else "PASS", info_text, "GT:GL:GQ:NR"]
    GT, GL, GQ, NR = call.GT, call.GL, call.GQ, call.nReads
    for i in range(len(GT)):
        gl = GL[i]
        cols.append("%s:%s,%s,%s:%s:%s" % (GT[i], gl[0], gl[1], gl[2], GQ[i], NR[i]))   # a GQ of -1 is written as it is
    return "\t".join(cols)
