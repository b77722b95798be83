#!/usr/bin/env python
"""End-to-end timing of a region call on simulated data (reads in memory -> calls in memory):
where the time goes between the host steps and the device batch, and the resulting Mbp/min.

    python tools/time_region.py [genome_kb] [coverage] [samples]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from platypus_amd import pipeline, simulate  # noqa: E402
from platypus_amd.fastafile import FastaFile  # noqa: E402


def main():
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    n_samples = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(5)
    G = kb * 1000
    t = time.perf_counter()
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, G // 1000, spacing=600)
    err = float(os.environ.get("TR_ERROR_RATE", "0.005"))        # substitution errors per base
    reads = [simulate.simulate_reads(rng, genome, truth, coverage=cov, read_len=100, error_rate=err,
                                     error_qual=(5, int(os.environ.get("TR_ERROR_QMAX", "25"))))
             for _ in range(n_samples)]
    copies = int(os.environ.get("TR_REPEATS", "1"))           # lay repeats of the simulated stretch end to end
    if copies > 1:
        tiled = [simulate.tile(genome, truth, r, copies) for r in reads]
        genome, truth, reads = tiled[0][0], tiled[0][1], [t[2] for t in tiled]
        G *= copies
        kb *= copies
    print("simulated %d kb, %d variants, %d x %d reads in %.1f s" % (kb, len(truth), n_samples, len(reads[0]["pos"]),
                                                                      time.perf_counter() - t), flush=True)
    fa = FastaFile(sequences={b"sim": genome})
    opts = pipeline.Options(rlen=100)
    for rep in range(2):
        t0 = time.perf_counter()
        bufs = [simulate.buffer_of(r, 0, G) for r in reads]
        t1 = time.perf_counter()
        cands = pipeline.candidatesInRegion(b"sim", 0, G, bufs, fa, opts)
        t2 = time.perf_counter()
        stats = {}
        calls = pipeline.callRegion(b"sim", 0, G, bufs, fa, opts, variants=cands, stats=stats)
        t3 = time.perf_counter()
        total = t3 - t0
        found = {(v.refPos, v.removed, v.added) for c in calls for v in c.variants}
        hit = sum((p, r, a) in found for p, r, a, g in truth)
        print("rep %d: buffers %.2f s  candidates %.2f s (%d)  call %.2f s [pack %.3f upload %.3f device %.3f unpack %.3f] "
              "-> %.2f s total = %.1f Mbp/min; windows %d, calls %d, planted variants found %d/%d, other calls %d" % (
                  rep, t1 - t0, t2 - t1, len(cands), t3 - t2, stats.get("pack", 0), stats.get("upload", 0),
                  stats.get("device", 0), stats.get("unpack", 0), total, G / 1e6 / (total / 60.0),
                  stats.get("called_windows", 0), len(calls), hit, len(truth), len(found) - hit), flush=True)
        print("       phases of the call: candidates %.3f s, windows + haplotypes + collect %.3f s, device batch (pack .. unpack) "
              "%.3f s, consume %.3f s" % (stats.get("t_candidates", 0), stats.get("t_collect", 0), stats.get("t_batch", 0),
                                          stats.get("t_consume", 0)), flush=True)
        ms = stats.get("anchor_ms", 0) + stats.get("align_ms", 0)
        print("       device: %d pairs, %.3g cells; anchor %.3f ms + alignment %.3f ms -> %.0f GCUPS on this pair stream; "
              "genotype setup %.3f ms, EM %.3f ms" % (stats.get("pairs", 0), stats.get("cells", 0), stats.get("anchor_ms", 0),
                                                     stats.get("align_ms", 0), stats.get("cells", 0) / max(ms, 1e-9) / 1e6,
                                                     stats.get("setup_ms", 0), stats.get("em_ms", 0)), flush=True)


if __name__ == "__main__":
    main()
