#!/usr/bin/env python
"""Randomised whole-region calls on simulated data for a time budget: random genome length, variant
density and mix, coverage, read length, single or paired reads, one to three samples, error rate,
buffer boundaries that cut through the data.  The truth is known by construction: every run
reports planted variants missed and calls that were not planted, and any exception stops it.
Not collected by pytest.

    python tests/fuzz_pipeline.py [seconds] [first_seed]
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
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = last = time.perf_counter()
    runs = planted = missed = extra = 0
    worst = []
    while time.perf_counter() - t0 < budget:
        rng = np.random.default_rng(seed)
        G = int(rng.integers(3000, 60000))
        genome = simulate.random_genome(rng, G, homopolymers=int(rng.integers(0, 6)))
        spacing = int(rng.choice([60, 120, 300, 1000]))
        truth = simulate.plant_variants(rng, genome, int(rng.integers(1, max(2, G // (2 * spacing)))), spacing=spacing,
                                        indel_frac=float(rng.choice([0.0, 0.3, 0.7])))
        L = int(rng.choice([50, 100, 150]))
        n_samples = int(rng.integers(1, 4))
        cov = int(rng.choice([15, 30, 60]))
        paired = bool(rng.random() < 0.5)
        err = float(rng.choice([0.0, 0.005, 0.02]))
        reads = [simulate.simulate_reads(rng, genome, truth, coverage=cov, read_len=L, error_rate=err, paired=paired,
                                         insert=(int(rng.choice([200, 350, 500])), 40)) for _ in range(n_samples)]
        fa = FastaFile(sequences={b"sim": genome})
        opts = pipeline.Options(rlen=L)
        # the region in one buffer, or cut in two at a random place (variants next to the cut may be lost to either side)
        cuts = [0, G] if rng.random() < 0.5 else [0, int(rng.integers(G // 4, 3 * G // 4)), G]
        found = set()
        for s, e in zip(cuts[:-1], cuts[1:]):
            bufs = [simulate.buffer_of(r, s, e) for r in reads]
            for c in pipeline.callRegion(b"sim", s, e, bufs, fa, opts):
                for v in c.variants:
                    found.add((v.refPos, v.removed, v.added))
        near_cut = lambda p: any(abs(p - c) < 2 * L for c in cuts[1:-1]) or p < 100 + L or p > G - 2 * L
        m = [t for t in truth if (t[0], t[1], t[2]) not in found and not near_cut(t[0])]
        x = [f for f in found if f not in {(t[0], t[1], t[2]) for t in truth}]
        runs += 1; planted += len(truth); missed += len(m); extra += len(x)
        if len(m) > 1 + len(truth) // 20 or len(x) > 2 + len(truth) // 10:
            worst.append((seed, len(truth), len(m), len(x)))
            print("seed %d: G %d, %d planted (spacing %d), %d samples x %dx, L %d, paired %s, errors %.3f: missed %d, not planted %d"
                  % (seed, G, len(truth), spacing, n_samples, cov, L, paired, err, len(m), len(x)), flush=True)
        seed += 1
        if time.perf_counter() - last > 30:
            last = time.perf_counter()
            print("... %d runs, %d planted, %d missed, %d called but not planted" % (runs, planted, missed, extra), flush=True)
    print("%d runs, %d planted variants, %d missed (%.2f %%), %d called but not planted; %d runs beyond the bounds" % (
        runs, planted, missed, 100.0 * missed / max(1, planted), extra, len(worst)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
