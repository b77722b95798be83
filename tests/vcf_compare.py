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


def files_both(work, seed, inflate="device", G=90000, unit=15000, rlen=100, n_var=110, spacing=300, indel_frac=0.4,
               coverage=18, paired=True, error_rate=0.008, n_samples=2, chunk=45000, homopolymers=30, options=(),
               harden=False):
    """BAM + FASTA files written under `work` and called to a VCF file by platypus_amd.runner (device kernels; BGZF
    blocks inflated as `inflate` says), and the same files through the CPU restatements chained from the records on
    (oracle/candidates_oracle.py, oracle/vcf_oracle.py) over the same chunks and read buffers.  `options`: pairs of
    (runner argument name, value) that both sides get.  harden: the simulated records get duplicates, unmapped and
    low-quality records, short fragments, bad heads and tails, 'N's, soft and hard clips and '='/'X' operations
    (tests/test_candidates_oracle.py) before they are written.  Returns (product body lines, restatement lines, windows the
    restatement skipped)."""
    import os
    from oracle import candidates_oracle as co, vcf_oracle as vo
    from platypus_amd import bamio, pipeline, runner, shard, simulate
    rng = np.random.default_rng(seed)
    genome = simulate.random_genome(rng, G, homopolymers=homopolymers)
    truth = simulate.plant_variants(rng, genome, n_var, spacing=spacing, indel_frac=indel_frac)
    fa = os.path.join(work, "ref.fa")
    with open(fa, "wb") as f:
        f.write(b">sim\n" + b"\n".join(genome[i:i + 60] for i in range(0, G, 60)) + b"\n")
    with open(fa + ".fai", "wb") as f:
        f.write(b"sim\t%d\t5\t60\t61\n" % G)
    bams = []
    for k in range(n_samples):
        mine = [t for t in truth if rng.random() < 0.75]
        kw = dict(paired=True, insert=(int(2.8 * rlen), 40)) if paired else {}
        reads = simulate.simulate_reads(rng, genome, mine, coverage=coverage, read_len=rlen, error_rate=error_rate, **kw)
        if harden:
            from tests.test_candidates_oracle import _clip_some, _harden
            reads = _clip_some(rng, _harden(rng, reads, G))
        path = os.path.join(work, "s%d.bam" % k)
        bamio.write_bam(path, [(b"sim", G)], reads, sample="S%d" % k)
        bams.append(path)
    out = os.path.join(work, "calls.vcf")
    start, end = 1000, G - 1000
    extra = [x for name, value in options for x in ("--" + name, str(value))]
    rc = runner.main(["callVariants", "--bamFiles", ",".join(bams), "--refFile", fa, "--regions", "sim:%d-%d" % (start, end),
                      "--output", out, "--processRegionSize", str(chunk), "--bufferSize", str(unit), "--maxReadLength", str(rlen),
                      "--inflate", inflate] + extra)
    assert rc == 0
    got = [l for l in open(out).read().splitlines() if not l.startswith("#")]
    opts = pipeline.Options(rlen=rlen, **dict(options))
    ref = vo.Ref({b"sim": genome})
    want, skipped = [], []
    readers = [bamio.BamReader(b) for b in bams]
    for ch in shard.split_regions([("sim", start, end)], chunk):
        for s in range(ch[1], ch[2], unit):
            e = min(s + unit, ch[2])
            kept = [co.buffer_reads(co.create_reads(r.fetch(b"sim", s, e)), opts.minMapQual, int(opts.maxReads),
                                    opts.minGoodQualBases, opts.maxBadQualBases) for r in readers]
            cands = co.candidates_in_region(b"sim", s, e, kept, ref, opts)
            samples = [vo.Reads([r["pos"] for r in k], [r["end"] for r in k], [r["mapq"] for r in k], [r["isReverse"] for r in k],
                                [r["seq"] for r in k], [bytes(r["qual"]) for r in k]) for k in kept]
            want.extend(vo.call_region(b"sim", s, e, cands, samples, ref, opts, skipped=skipped))
    for r in readers:
        r.close()
    return got, want, skipped
