#!/usr/bin/env python
"""Randomised file-to-file comparison (run by hand, not collected by pytest):

    python tests/fuzz_files.py [seconds] [first_seed]

Every run draws its own reference length, variant density and mix, coverage, read length, error rate, single or paired
reads (half of the runs with hardened records: duplicates, clips, bad qualities, '='/'X' operations ...), one to three
BAM files, chunk and buffer lengths, where the BGZF blocks are inflated and a few caller options,
writes BAM + FASTA files, calls them with platypus_amd.runner (device kernels) and with the chained CPU restatements
(oracle/candidates_oracle.py, oracle/vcf_oracle.py), and compares the VCF body lines as text.  Exit status 1 if any
line differs."""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tests import vcf_compare as vc  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.perf_counter()
    n_runs = n_lines = n_skipped = 0
    kinds = {}
    while time.perf_counter() - t0 < budget:
        rng = np.random.default_rng(seed)
        rlen = int(rng.choice([50, 75, 100, 150]))
        unit = int(rng.choice([5000, 15000, 40000]))
        options = []
        if rng.random() < 0.3:
            options.append(("minPosterior", int(rng.choice([0, 20]))))
        if rng.random() < 0.2:
            options.append(("maxVariants", 3))
        if rng.random() < 0.2:
            options.append(("minReads", int(rng.choice([1, 3]))))
        if rng.random() < 0.15:
            options.append(("genIndels", 0))
        kw = dict(inflate=str(rng.choice(["device", "host"])), G=int(rng.integers(30000, 120000)), unit=unit, rlen=rlen,
                  n_var=int(rng.integers(20, 400)), spacing=int(rng.choice([15, 40, 120, 300])),
                  indel_frac=float(rng.choice([0.1, 0.3, 0.6])), coverage=int(rng.integers(8, 40)),
                  paired=bool(rng.random() < 0.5), error_rate=float(rng.choice([0.0, 0.003, 0.01, 0.02])),
                  n_samples=int(rng.integers(1, 4)), chunk=int(rng.choice([unit, 3 * unit, 200000])),
                  homopolymers=int(rng.choice([0, 30])), options=tuple(options), harden=bool(rng.random() < 0.5))
        work = tempfile.mkdtemp(prefix="platypus_amd_fuzz_")
        try:
            got, want, skipped = vc.files_both(work, seed, **kw)
        finally:
            shutil.rmtree(work, ignore_errors=True)
        n_runs += 1; n_lines += len(want); n_skipped += len(skipped)
        for kind, items in vc.classify(got, want).items():
            kinds.setdefault(kind, []).extend((seed,) + it for it in items)
            for it in items[:2]:
                print("seed %d %r: %s at %s: %s" % (seed, kw, kind, it[0], it[1]), flush=True)
        if got != want and not vc.classify(got, want):
            kinds.setdefault("order", []).append((seed,))
            print("seed %d: same lines in another order" % seed, flush=True)
        if n_runs % 5 == 0:
            print("... %d runs, %d lines, %d differing so far" % (n_runs, n_lines, sum(len(v) for v in kinds.values())), flush=True)
        seed += 1
    print("%d file-to-file runs, %d VCF lines in the restatements, %d windows skipped by them; differences by kind: %s" % (
        n_runs, n_lines, n_skipped, {k: len(v) for k, v in sorted(kinds.items())} or "none"))
    return 1 if kinds else 0


if __name__ == "__main__":
    sys.exit(main())
