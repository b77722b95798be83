#!/usr/bin/env python
"""Fixed and marginal cost of the file-to-file run: the bench line's 64-Mb data set (2 Mb simulated at 30x with 150-bp
pairs, 32 repeats end to end) called over regions of growing length, each run in a process of its own, start-up
included; a least-squares line through (length, seconds) gives the seconds a run costs before its first base and
the rate it calls at from then on.

    python tools/time_startup.py [workers ...]        (default 1 4)

TS_SIZES="32 64" limits the region lengths (Mb), TS_ARGS adds runner arguments (several sets separated by ';': each is timed).
"""
import multiprocessing
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    from platypus_amd import bamio, runner, simulate
    workers_list = [int(a) for a in sys.argv[1:]] or [1, 4]
    rng = np.random.default_rng(5)
    kb, copies = 2000, 32
    G = kb * 1000
    work = tempfile.mkdtemp(prefix="platypus_amd_startup_")
    try:
        genome = simulate.random_genome(rng, G)
        truth = simulate.plant_variants(rng, genome, G // 1000, spacing=600)
        reads = simulate.simulate_reads(rng, genome, truth, coverage=30, read_len=150, paired=True, insert=(500, 50))
        genome, truth, reads = simulate.tile(genome, truth, reads, copies)
        G *= copies
        fa = os.path.join(work, "ref.fa")
        with open(fa, "wb") as f:
            f.write(b">sim\n" + b"\n".join(genome[i:i + 60] for i in range(0, G, 60)) + b"\n")
        with open(fa + ".fai", "wb") as f:
            f.write(b"sim\t%d\t5\t60\t61\n" % G)
        bam = os.path.join(work, "s.bam")
        bamio.write_bam(bam, [(b"sim", G)], reads, sample="S")
        out = os.path.join(work, "calls.vcf")
        print("data set written", flush=True)
        sizes = [int(x) for x in os.environ.get("TS_SIZES", "1 8 16 32 64").split()]
        for workers, extra in [(w, e.split()) for w in workers_list for e in os.environ.get("TS_ARGS", "").split(";")]:
            if extra:
                print("runner arguments: %s" % " ".join(extra), flush=True)
            xs, ys = [], []
            for mb in sizes:
                L = mb * 1000000
                args = ["callVariants", "--bamFiles", bam, "--refFile", fa, "--regions", "sim:1-%d" % (L - 1), "--output", out,
                        "--maxReadLength", "150"] + extra
                if workers > 1:
                    args += ["--workersPerGPU", str(workers), "--processRegionSize", str(max(1000000, L // workers))]
                best = 1e30
                for _ in range(3):
                    t0 = time.perf_counter()
                    child = multiprocessing.get_context("spawn").Process(target=runner.main, args=(args,))
                    child.start()
                    child.join()
                    if child.exitcode != 0:
                        raise RuntimeError("run failed")
                    best = min(best, time.perf_counter() - t0)
                xs.append(mb)
                ys.append(best)
                print("workers %d  %3d Mb  %.3f s  (%.0f Mbp/min)" % (workers, mb, best, mb / best * 60), flush=True)
            if len(xs) < 3:
                continue
            slope, icpt = np.polyfit(xs[1:], ys[1:], 1)
            print("workers %d: %.3f s before the first base, then %.4f s per Mb = %.0f Mbp/min" % (workers, icpt, slope, 60.0 / slope),
                  flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
