"""A whole region call on simulated data: reads with true CIGARs -> candidates -> windows -> haplotypes ->
ONE alignment launch + ONE genotype/EM launch -> calls.  The truth is known by construction."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_planted_variants_are_called_with_their_genotypes():
    from platypus_amd import calign, pipeline, simulate
    from platypus_amd.fastafile import FastaFile
    rng = np.random.default_rng(7)
    G = 24000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 70)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=30, read_len=100, error_rate=0.005)
    fa = FastaFile(sequences={b"sim": genome})
    buf = simulate.buffer_of(reads, 0, G)
    stats = {}
    calls = pipeline.callRegion(b"sim", 0, G, [buf], fa, pipeline.Options(rlen=100), stats=stats)
    assert stats["called_windows"] >= len(truth) - 2 and stats["windows"] >= stats["called_windows"]
    called = {}
    for c in calls:
        for vi, v in enumerate(c.variants):
            called[(v.refPos, v.removed, v.added)] = (c, vi)
    missed = [t for t in truth if (t[0], t[1], t[2]) not in called]
    assert len(missed) <= 1, missed
    wrong_gt = 0
    for pos, rem, add, gt in truth:
        if (pos, rem, add) not in called:
            continue
        c, vi = called[(pos, rem, add)]
        assert int(c.PP[vi]) >= 20
        alleles = sorted(c.GT[0].split("/"))
        want = ["0", str(vi + 1)] if gt == 1 else [str(vi + 1), str(vi + 1)]
        wrong_gt += alleles != want
        if len(c.variants) == 1:
            gl = c.GL[0]
            assert gl != [-1, -1, -1] and max(gl) == gl[gt]          # AA, AB, BB: the true genotype is the most likely
            assert gl[0] < gl[gt] - 5
            assert c.nReads[0] > 5 and 0 <= c.GQ[0] <= 100
        fr = float(c.FR[vi])
        assert (0.3 < fr < 0.7) if gt == 1 else fr > 0.9
    assert wrong_gt <= 1
    false_pos = [k for k in called if k not in {(t[0], t[1], t[2]) for t in truth}]
    assert len(false_pos) <= 2, false_pos
    calign.closeDevice()


def test_overlapping_read_pairs():
    """pairs from fragments shorter than two reads: the forward read's overlap with its mate is set
    to quality 0 (cwindow.pyx:136-144), pairs whose fragment is shorter than one read are dropped
    (cwindow.pyx:92-98); the planted variants are still called with their genotypes"""
    from platypus_amd import calign, pipeline, simulate
    from platypus_amd.fastafile import FastaFile
    rng = np.random.default_rng(17)
    G = 16000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 40)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=40, read_len=100, error_rate=0.005, paired=True,
                                    insert=(170, 25))
    assert ((reads["insertSize"] > 0) & (reads["insertSize"] < 200)).sum() > 0.3 * len(reads["pos"])
    fa = FastaFile(sequences={b"sim": genome})
    buf = simulate.buffer_of(reads, 0, G)
    trimmed = 0
    for i in range(0, buf.readCount, 7):
        pos, end, rlen, mapq, rev, quals = buf.readInfo(i)
        trimmed += (not rev) and quals.endswith(b"\x00" * 10)
    assert trimmed > 20
    calls = pipeline.callRegion(b"sim", 0, G, [buf], fa, pipeline.Options(rlen=100))
    called = {(v.refPos, v.removed, v.added): (c, vi) for c in calls for vi, v in enumerate(c.variants)}
    missed = [t for t in truth if (t[0], t[1], t[2]) not in called]
    assert len(missed) <= 1, missed
    wrong = 0
    for pos, rem, add, gt in truth:
        if (pos, rem, add) in called:
            c, vi = called[(pos, rem, add)]
            alleles = sorted(c.GT[0].split("/"))
            wrong += alleles != (["0", str(vi + 1)] if gt == 1 else [str(vi + 1), str(vi + 1)])
    assert wrong <= 1
    assert len([k for k in called if k not in {(t[0], t[1], t[2]) for t in truth}]) <= 2
    calign.closeDevice()


def test_two_samples_and_a_crowded_window():
    """two samples with different genotypes; a window with more candidates than maxVariants goes through the
    posterior pre-filter (variantcaller.pyx:142-166)"""
    from platypus_amd import calign, pipeline, simulate
    from platypus_amd.fastafile import FastaFile
    rng = np.random.default_rng(11)
    G = 6000
    genome = simulate.random_genome(rng, G)
    snp = lambda p: (p, genome[p:p + 1], b"A" if genome[p:p + 1] != b"A" else b"C")
    shared = [snp(1000) + (2,), snp(2000) + (1,)]
    only_b = [snp(3000) + (2,)]
    crowded = [snp(4000 + 3 * i) + (2,) for i in range(4)]          # four SNPs within 10 bases
    reads_a = simulate.simulate_reads(rng, genome, sorted(shared + crowded), coverage=25, read_len=80)
    reads_b = simulate.simulate_reads(rng, genome, sorted(shared + only_b + crowded), coverage=25, read_len=80)
    fa = FastaFile(sequences={b"sim": genome})
    bufs = [simulate.buffer_of(reads_a, 0, G), simulate.buffer_of(reads_b, 0, G)]
    opts = pipeline.Options(rlen=80, maxVariants=3)
    calls = pipeline.callRegion(b"sim", 0, G, bufs, fa, opts)
    by_pos = {c.pos: c for c in calls}
    assert all(sorted(by_pos[1000].GT[i].split("/")) == ["1", "1"] for i in (0, 1))
    assert sorted(by_pos[2000].GT[0].split("/")) == ["0", "1"] and sorted(by_pos[2000].GT[1].split("/")) == ["0", "1"]
    assert sorted(by_pos[3000].GT[0].split("/")) == ["0", "0"] and sorted(by_pos[3000].GT[1].split("/")) == ["1", "1"]
    # the SNPs at 4000, 4003, 4006, 4009 are windows of their own (maxPos == refPos for SNPs, variant.pyx:48-49);
    # each window's single-SNP haplotype still mismatches the reads at the other three, the aligner explains such
    # a cluster about as cheaply either way, and the inner ones fall below minPosterior -- the model's behaviour
    assert 4000 in by_pos and 4009 in by_pos
    assert all(int(pp) < 100 for p in (4000, 4009) for pp in by_pos[p].PP)
    # a crowded window: a deletion, a SNP on its last base, a second deletion starting there and a SNP on ITS
    # last base chain into one window of four candidates (each reaches the next one's position); with
    # maxVariants = 3 the window first goes through the posterior pre-filter.  Truth: the first deletion and
    # the last SNP (both homozygous in both samples); the two in between are spurious candidates.
    from platypus_amd.variant import Variant
    dele = (4500, genome[4501:4504], b"", 2)
    last = snp(4506) + (2,)
    reads_c = simulate.simulate_reads(rng, genome, [dele, last], coverage=30, read_len=80)
    bufs_c = [simulate.buffer_of(reads_c, 0, G)]
    cands = sorted([Variant(b"sim", 4500, dele[1], b"", 10, 5), Variant(b"sim", 4503, *snp(4503)[1:], 3, 2),
                    Variant(b"sim", 4503, genome[4504:4507], b"", 3, 1), Variant(b"sim", 4506, last[1], last[2], 10, 5)])
    wins = list(pipeline.WindowGenerator().WindowsAndVariants(b"sim", 4400, 4600, cands, opts))
    assert len(wins) == 1 and len(wins[0]["variants"]) == 4
    calls2 = pipeline.callRegion(b"sim", 4400, 4600, bufs_c, fa, opts, variants=cands)
    got = {(v.refPos, v.removed, v.added) for c in calls2 for v in c.variants}
    assert got == {(4500, dele[1], b""), (4506, last[1], last[2])}
    assert all(sorted(c.GT[0].split("/")) == ["1", "1"] for c in calls2)
    calign.closeDevice()


def test_command_line_bam_to_vcf(tmp_path):
    """files in, file out: FASTA + .fai, two BAMs + .bai -> VCF through platypus_amd.runner (two chunks, two
    read buffers per chunk), checked against the planted truth"""
    from platypus_amd import bamio, calign, runner, simulate
    rng = np.random.default_rng(21)
    G = 40000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 30, spacing=500)
    fa = tmp_path / "ref.fa"
    fa.write_bytes(b">sim\n" + b"\n".join(genome[i:i + 60] for i in range(0, G, 60)) + b"\n")
    (tmp_path / "ref.fa.fai").write_bytes(b"sim\t%d\t5\t60\t61\n" % G)
    bams = []
    for k in range(2):
        reads = simulate.simulate_reads(rng, genome, truth, coverage=20, read_len=100)
        path = str(tmp_path / ("s%d.bam" % k))
        bamio.write_bam(path, [(b"sim", G)], reads, sample="S%d" % k)
        bams.append(path)
    out = str(tmp_path / "calls.vcf")
    rc = runner.main(["callVariants", "--bamFiles", ",".join(bams), "--refFile", str(fa), "--regions", "sim:1-%d" % (G - 1),
                      "--output", out, "--processRegionSize", "25000", "--bufferSize", "10000", "--maxReadLength", "100"])
    assert rc == 0
    lines = open(out).read().splitlines()
    head = [l for l in lines if l.startswith("#")]
    body = [l.split("\t") for l in lines if not l.startswith("#")]
    assert head[0] == "##fileformat=VCFv4.0" and head[-1].split("\t")[-2:] == ["s0.bam", "s1.bam"]    # cpopulation.pyx:98-99: file names
    assert sum(l.startswith("#CHROM") for l in lines) == 1                     # one header after the merge
    pos = [int(b[1]) for b in body]
    assert pos == sorted(pos) and len(body) >= len(truth) - 1
    found = {}
    for b in body:
        assert b[0] == "sim" and b[8] == "GT:GL:GQ:NR" and len(b) == 11
        found[int(b[1])] = b
    for p, rem, add, gt in truth:
        b = found.get(p + 1)
        if b is None:
            continue
        if len(rem) == len(add):
            assert (b[3], b[4]) == (rem.decode(), add.decode())
        elif add:                                                                 # insertion: REF = base at pos
            assert b[3] == genome[p:p + 1].decode() and b[4] == (genome[p:p + 1] + add).decode()
        else:                                                                     # deletion: REF = base + removed
            assert b[3] == (genome[p:p + 1] + rem).decode() and b[4] == genome[p:p + 1].decode()
        for smp in b[9:]:
            g = sorted(smp.split(":")[0].split("/"))
            assert g == (["0", "1"] if gt == 1 else ["1", "1"])
        info = dict(kv.split("=") for kv in b[7].split(";"))
        assert set(info) - {"MMLQ"} == {"FR", "PP", "TR", "NF", "NR", "TC", "TCR", "TCF", "HP", "SC", "RMP", "FPV", "RPV", "ABPV"}
        assert ("MMLQ" in info) == (len(rem) == 1 and len(add) == 1)
        assert info["SC"] == genome[p - 10:p + 11].decode() and float(info["ABPV"]) > 1e-3
        assert int(info["TC"]) == int(info["TCR"]) + int(info["TCF"]) > 10
    assert sum((p + 1) in found for p, r, a, g in truth) >= len(truth) - 1
    assert sum(found[p + 1][6] == "PASS" for p, r, a, g in truth if (p + 1) in found) >= len(truth) - 3
    calign.closeDevice()
    # the same chunks dealt to two worker processes sharing the GPU, one buffer ahead: the same records
    out2 = str(tmp_path / "calls2.vcf")
    rc = runner.main(["callVariants", "--bamFiles", ",".join(bams), "--refFile", str(fa), "--regions", "sim:1-%d" % (G - 1),
                      "--output", out2, "--processRegionSize", "25000", "--bufferSize", "10000", "--maxReadLength", "100",
                      "--workersPerGPU", "2", "--readAhead", "1"])
    assert rc == 0
    assert [l for l in open(out2).read().splitlines() if not l.startswith("##")] == [l for l in lines if not l.startswith("##")]
    # the reference's own spellings: --nCPU for the number of host processes, --labels for sample names
    out3 = str(tmp_path / "calls3.vcf")
    rc = runner.main(["callVariants", "--bamFiles", ",".join(bams), "--refFile", str(fa), "--regions", "sim:1-%d" % (G - 1),
                      "--output", out3, "--processRegionSize", "25000", "--bufferSize", "10000", "--maxReadLength", "100",
                      "--nCPU", "2", "--labels", "s[0-9]"])
    assert rc == 0
    lines3 = open(out3).read().splitlines()
    assert [l for l in lines3 if l.startswith("#CHROM")][0].split("\t")[-2:] == ["s0", "s1"]
    assert [l for l in lines3 if not l.startswith("#")] == [l for l in lines if not l.startswith("#")]
    # this process has opened the GPU, so the runs above started their workers as fresh interpreters; the same command
    # in a process of its own forks them before anything touches the GPU: the same records
    import subprocess
    import sys
    assert not runner._gpu_untouched()
    out4 = str(tmp_path / "calls4.vcf")
    cmd = [sys.executable, "-m", "platypus_amd.runner", "callVariants", "--bamFiles", ",".join(bams), "--refFile", str(fa),
           "--regions", "sim:1-%d" % (G - 1), "--output", out4, "--processRegionSize", "10000", "--bufferSize", "10000",
           "--maxReadLength", "100", "--workersPerGPU", "3"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run(cmd, cwd=root, timeout=100, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-2000:]
    assert [l for l in open(out4).read().splitlines() if not l.startswith("#")] == [l for l in lines if not l.startswith("#")]


def test_print_candidates_and_exit(tmp_path, capsys):
    """--printVarsAndExit 1 (variantcaller.pyx:110-113): the candidate list, no calls"""
    from platypus_amd import bamio, runner, simulate
    rng = np.random.default_rng(23)
    G = 12000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 10, spacing=500)
    fa = tmp_path / "ref.fa"
    fa.write_bytes(b">sim\n" + b"\n".join(genome[i:i + 60] for i in range(0, G, 60)) + b"\n")
    (tmp_path / "ref.fa.fai").write_bytes(b"sim\t%d\t5\t60\t61\n" % G)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=20, read_len=100, error_rate=0.0)
    bam = str(tmp_path / "s.bam")
    bamio.write_bam(bam, [(b"sim", G)], reads)
    out = str(tmp_path / "none.vcf")
    rc = runner.main(["callVariants", "--bamFiles", bam, "--refFile", str(fa), "--regions", "sim:1-%d" % (G - 1),
                      "--output", out, "--printVarsAndExit", "1", "--maxReadLength", "100"])
    assert rc == 0
    printed = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(truth) - 1 <= len(printed) <= len(truth)
    assert not [l for l in open(out).read().splitlines() if l and not l.startswith("#")]


def test_windows_the_device_cannot_take_cost_only_themselves(caplog):
    """A read longer than the context was made for makes the context grow; a window whose indel-bearing
    haplotypes exceed the 7-mer table's 16383 bases (calign.pyx:112) is skipped with a warning, as the
    reference skips a window that fails (variantcaller.pyx:189-191); the region's other windows are called."""
    import logging
    from platypus_amd import calign, pipeline, simulate
    from platypus_amd.fastafile import FastaFile
    rng = np.random.default_rng(23)
    G = 40000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 60, spacing=420, indel_frac=0.4)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=30, read_len=100, error_rate=0.003)
    # one 600-base read (error-free copy of the reference, one CIGAR operation) in the middle of the region
    at = int(np.searchsorted(reads["pos"], 20000))
    p600 = int(reads["pos"][at])
    seq600 = np.frombuffer(genome[p600:p600 + 600], np.uint8)
    so, co = int(reads["off"][at]), int(reads["cigarOff"][at])
    reads["seq"] = np.concatenate([reads["seq"][:so], seq600, reads["seq"][so:]])
    reads["qual"] = np.concatenate([reads["qual"][:so], np.full(600, 35, np.uint8), reads["qual"][so:]])
    reads["off"] = np.concatenate([reads["off"][:at + 1], reads["off"][at:] + 600]).astype(np.int64)
    reads["cigar"] = np.concatenate([reads["cigar"][:co], np.array([600 << 4], np.uint32), reads["cigar"][co:]])
    reads["cigarOff"] = np.concatenate([reads["cigarOff"][:at + 1], reads["cigarOff"][at:] + 1]).astype(np.int64)
    for k, val in (("pos", p600), ("mapq", 60), ("flag", 0)):
        reads[k] = np.concatenate([reads[k][:at], np.array([val], reads[k].dtype), reads[k][at:]])
    fa = FastaFile(sequences={b"sim": genome})
    calign.closeDevice()                                     # a fresh context: made for 512-base reads
    buf = simulate.buffer_of(reads, 0, G)
    calls = pipeline.callRegion(b"sim", 0, G, [buf], fa, pipeline.Options(rlen=100))
    called = {(v.refPos, v.removed, v.added) for c in calls for v in c.variants}
    assert sum((t[0], t[1], t[2]) in called for t in truth) >= len(truth) - 1
    # --maxReadLength 4200: every haplotype is padded by 8400 bases on either side, so the indel-bearing ones
    # are beyond the table's reach and their windows are skipped; windows of substitutions are still called
    with caplog.at_level(logging.WARNING, logger="Log"):
        calls = pipeline.callRegion(b"sim", 0, G, [buf], fa, pipeline.Options(rlen=4200))
    called = {(v.refPos, v.removed, v.added) for c in calls for v in c.variants}
    snps = [t for t in truth if len(t[1]) == len(t[2])]
    # (near the ends of the sequence the padding is clipped, chaplotype.pyx:70-71, and the haplotypes fit)
    indels = [t for t in truth if len(t[1]) != len(t[2]) and 8500 < t[0] < G - 8500]
    assert len(snps) > 3 and len(indels) > 3
    assert sum((t[0], t[1], t[2]) in called for t in snps) >= len(snps) - 1
    assert not any((t[0], t[1], t[2]) in called for t in indels)
    assert sum("is skipped" in r.getMessage() for r in caplog.records) >= len(indels) - 1
    calign.closeDevice()
