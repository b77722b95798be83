#!/usr/bin/env python
"""File-to-file timing: simulated BAM (+ .bai) and FASTA (+ .fai) -> VCF through platypus_amd.runner.

    python tools/time_files.py [genome_kb] [coverage] [workdir]

TF_REPEATS=k lays k repeats of the simulated stretch end to end (simulating is much slower than calling);
TF_PAIRED=1 TF_READLEN=150 simulates 150-bp pairs; TF_ARGS adds runner arguments; TF_PROFILE=path writes a cProfile dump of the second run.
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from platypus_amd import bamio, runner, simulate  # noqa: E402


def main():
    if os.environ.get("TF_DUMP_AFTER"):                         # stacks of all threads if the run is still going then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["TF_DUMP_AFTER"]), exit=True)
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    work = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp(prefix="pa_files_")
    rng = np.random.default_rng(5)
    G = kb * 1000
    t = time.perf_counter()
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, G // 1000, spacing=600)
    rlen = int(os.environ.get("TF_READLEN", "100"))
    reads = simulate.simulate_reads(rng, genome, truth, coverage=cov, read_len=rlen,
                                    paired=os.environ.get("TF_PAIRED", "0") == "1", insert=(500, 50))
    copies = int(os.environ.get("TF_REPEATS", "1"))
    if copies > 1:
        genome, truth, reads = simulate.tile(genome, truth, reads, copies)
        G *= copies
    fa = os.path.join(work, "ref.fa")
    with open(fa, "wb") as f:
        f.write(b">sim\n" + b"\n".join(genome[i:i + 60] for i in range(0, G, 60)) + b"\n")
    with open(fa + ".fai", "wb") as f:
        f.write(b"sim\t%d\t5\t60\t61\n" % G)
    bam = os.path.join(work, "s.bam")
    bamio.write_bam(bam, [(b"sim", G)], reads, sample="S")
    print("simulated and wrote %d kb, %d reads, BAM %.1f MB in %.1f s" % (G // 1000, len(reads["pos"]), os.path.getsize(bam) / 1e6,
                                                                       time.perf_counter() - t), flush=True)
    out = os.path.join(work, "calls.vcf")
    extra = os.environ.get("TF_ARGS", "").split()               # e.g. "--workersPerGPU 4 --processRegionSize 500000"
    if os.environ.get("TF_SWITCH"):                             # interpreter lock hand-over interval, seconds
        sys.setswitchinterval(float(os.environ["TF_SWITCH"]))
    if os.environ.get("TF_BAM") and os.path.exists(os.environ["TF_BAM"]):
        pass
    for rep in range(2):
        prof = None
        if rep == 1 and os.environ.get("TF_PROFILE"):
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        runner.main(["callVariants", "--bamFiles", bam, "--refFile", fa, "--regions", "sim:1-%d" % (G - 1), "--output", out,
                     "--maxReadLength", str(rlen)] + extra)
        dt = time.perf_counter() - t0
        if prof is not None:
            prof.disable()
            prof.dump_stats(os.environ["TF_PROFILE"])
        n = sum(1 for l in open(out) if not l.startswith("#"))
        print("rep %d: %.2f s -> %.1f Mbp/min, %d VCF records for %d planted variants" % (rep, dt, G / 1e6 / (dt / 60), n,
                                                                                          len(truth)), flush=True)


if __name__ == "__main__":
    main()
