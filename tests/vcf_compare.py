"""Shared by tests/test_gpu_vcf_parity.py and tests/fuzz_vcf.py: call one simulated region with the product
(platypus_amd.pipeline + vcfout, device kernels) and with the CPU restatement of the reference's procedure
(oracle/vcf_oracle.py) from the same buffered reads and the same candidate list, and compare the VCF body
lines as text."""
import numpy as np

FIELDS = ("CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT")


def oracle_inputs(bufs, cands, genome, chrom):
    from oracle import vcf_oracle as vo
    samples = [vo.Reads(*b.readArrays()) for b in bufs]
    vs = [vo.V(refName=v.refName, refPos=v.refPos, removed=v.removed, added=v.added, nAdded=v.nAdded, nRemoved=v.nRemoved,
               minPos=v.minPos, maxPos=v.maxPos, nSupportingReads=v.nSupportingReads, nFwdReads=v.nFwdReads,
               nRevReads=v.nRevReads, nFwdReadsWeighted=v.nFwdReadsWeighted, nRevReadsWeighted=v.nRevReadsWeighted,
               rmsReadPos=v.rmsReadPos) for v in cands]
    return samples, vs, vo.Ref({chrom: genome})


def call_both(chrom, start, end, bufs, genome, options):
    """returns (product lines, restatement lines, windows the restatement skipped)"""
    from oracle import vcf_oracle as vo
    from platypus_amd import pipeline, vcfout
    from platypus_amd.fastafile import FastaFile
    fa = FastaFile(sequences={chrom: genome})
    cands = pipeline.candidatesInRegion(chrom, start, end, bufs, fa, options)
    calls = pipeline.callRegion(chrom, start, end, bufs, fa, options, variants=cands)
    got = [vcfout.record(c, fa) for c in calls]
    samples, vs, ref = oracle_inputs(bufs, cands, genome, chrom)
    skipped = []
    want = vo.call_region(chrom, start, end, vs, samples, ref, options, skipped=skipped)
    return got, want, skipped


def classify(got, want):
    """differences between two lists of VCF body lines, by kind: {kind: [(position, detail), ...]}"""
    out = {}
    g = {tuple(l.split("\t")[:2]) + (l.split("\t")[4],): l for l in got}
    w = {tuple(l.split("\t")[:2]) + (l.split("\t")[4],): l for l in want}
    for k in sorted(set(g) | set(w), key=lambda k: (k[0], int(k[1]), k[2])):
        if k not in w:
            out.setdefault("only in product", []).append((k, g[k]))
        elif k not in g:
            out.setdefault("only in restatement", []).append((k, w[k]))
        elif g[k] != w[k]:
            a, b = g[k].split("\t"), w[k].split("\t")
            for i, (x, y) in enumerate(zip(a, b)):
                if x == y:
                    continue
                if i == 7:
                    ia = dict(f.split("=", 1) for f in x.split(";")); ib = dict(f.split("=", 1) for f in y.split(";"))
                    for key in sorted(set(ia) | set(ib)):
                        if ia.get(key) != ib.get(key):
                            out.setdefault("INFO " + key, []).append((k, "%s vs %s" % (ia.get(key), ib.get(key))))
                elif i >= 9:
                    names = ("GT", "GL", "GQ", "NR")
                    for nm, p, q in zip(names, x.split(":"), y.split(":")):
                        if p != q:
                            out.setdefault("sample " + nm, []).append((k, "%s vs %s" % (p, q)))
                else:
                    out.setdefault(FIELDS[i], []).append((k, "%s vs %s" % (x, y)))
    return out


def simulated_region(seed, G=12000, n_var=40, coverage=30, read_len=100, n_samples=1, error_rate=0.005, paired=False,
                     spacing=120, indel_frac=0.3, half_covered=False):
    """half_covered: the last sample has reads over the first half of the region only, so that it is an
    individual without reads in the windows of the second half (genotype None, variantcaller.pyx:385-387)"""
    from platypus_amd import simulate
    rng = np.random.default_rng(seed)
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, n_var, spacing=spacing, indel_frac=indel_frac)
    bufs = []
    for s in range(n_samples):
        mine = [t for t in truth if rng.random() < 0.7] if n_samples > 1 else truth
        kw = dict(paired=True, insert=(300, 40)) if paired else {}
        g = genome
        if half_covered and s == n_samples - 1 and n_samples > 1:
            g = genome[:G // 2]
            mine = [t for t in mine if t[0] < G // 2 - 2 * read_len]
        reads = simulate.simulate_reads(rng, g, mine, coverage=coverage, read_len=read_len, error_rate=error_rate, **kw)
        bufs.append(simulate.buffer_of(reads, 0, G))
    return genome, truth, bufs
