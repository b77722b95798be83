#!/usr/bin/env python
"""Randomised whole-region comparison (run by hand, not collected by pytest):

    python tests/fuzz_vcf.py [seconds] [first_seed]

Every region draws its own length, variant density and mix, coverage, read length, error rate, single or
paired reads and one to four samples (some with no reads over part of the region), is called by the product
(device kernels) and by the CPU restatement of the reference's procedure (oracle/vcf_oracle.py), and the VCF
body lines are compared as text.  Prints the differing lines by kind and a summary; exit status 1 if any line
differs."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from platypus_amd import calign, pipeline  # noqa: E402
from tests import vcf_compare as vc  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.perf_counter()
    n_regions = n_lines = n_skipped = 0
    kinds, why = {}, {}
    while time.perf_counter() - t0 < budget:
        rng = np.random.default_rng(seed)
        L = int(rng.choice([50, 75, 100, 150]))
        kw = dict(seed=seed, G=int(rng.integers(4000, 16000)), n_var=int(rng.integers(5, 150)),
                  coverage=int(rng.integers(8, 60)), read_len=L, n_samples=int(rng.integers(1, 5)),
                  error_rate=float(rng.choice([0.0, 0.003, 0.01, 0.02])), paired=bool(rng.random() < 0.4),
                  spacing=int(rng.choice([15, 40, 120, 300])), indel_frac=float(rng.choice([0.1, 0.3, 0.6])),
                  half_covered=bool(rng.random() < 0.3))
        genome, truth, bufs = vc.simulated_region(**kw)
        opts = pipeline.Options(rlen=L, maxVariants=int(rng.choice([8, 8, 3])), minPosterior=int(rng.choice([5, 5, 0, 20])))
        got, want, skipped = vc.call_both(b"sim", 0, len(genome), bufs, genome, opts)
        n_regions += 1; n_lines += len(want); n_skipped += len(skipped)
        for ws, we, e in skipped:
            why[type(e).__name__ + ": " + str(e)[:60]] = why.get(type(e).__name__ + ": " + str(e)[:60], 0) + 1
        for kind, items in vc.classify(got, want).items():
            kinds.setdefault(kind, []).extend((seed,) + it for it in items)
            for it in items[:2]:
                print("seed %d: %s at %s: %s" % (seed, kind, it[0], it[1]), flush=True)
        if n_regions % 20 == 0:
            print("... %d regions, %d lines, %d differing so far" % (n_regions, n_lines, sum(len(v) for v in kinds.values())), flush=True)
        seed += 1
    calign.closeDevice()
    print("%d regions, %d VCF lines in the restatement, %d windows skipped by it; differences by kind: %s" % (
        n_regions, n_lines, n_skipped, {k: len(v) for k, v in sorted(kinds.items())} or "none"))
    print("windows the restatement skipped, by exception:", why or "none")
    return 1 if kinds else 0


if __name__ == "__main__":
    sys.exit(main())
