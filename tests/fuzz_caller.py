#!/usr/bin/env python
"""Randomised run of the genotype-likelihood / EM / posterior stage against the CPU checker: window
shapes drawn at random (2..130 haplotypes, 2..40 reads, 1..16 individuals, some without reads),
every double compared to 1e-9 relative, iteration counts and genotype calls equal
(tests/test_gpu_caller.py: run_case), for a time budget.  Not collected by pytest.

Reads per window stay at 40 or fewer on purpose.  The reference scales every individual's genotype
likelihoods by the largest value of the WINDOW (platypus/cpopulation.pyx:696,722,732-734); with
many more reads in one individual than in another the smaller individual's values sit hundreds
of orders of magnitude lower, leave the normal range of a double (denormal, then zero) and the EM
sums over them depend on the order of additions in their leading digits -- for the reference as
for any restatement of it.  FUZZ_MAX_READS lifts the limit to look at that regime.

    python tests/fuzz_caller.py [seconds] [first_seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_gpu_caller as T  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = last = time.perf_counter()
    n = bad = 0
    while time.perf_counter() - t0 < budget:
        rng = np.random.default_rng(seed)
        H = int(rng.choice([2, 2, 3, 4, 5, 8, 9, 16, 23, 24, 40, 64, 65, 100, 130]))
        R = int(rng.integers(2, int(os.environ.get("FUZZ_MAX_READS", "40")) + 1))
        n_ind = int(rng.integers(1, min(16, R) + 1))
        W = int(rng.integers(1, 5 if H > 40 else 12))
        try:
            T.run_case(W, H, R, n_ind, seed=seed, indel_frac=float(rng.choice([0.0, 0.3, 1.0])),
                       empty=int(rng.integers(0, n_ind)) if rng.random() < 0.4 and n_ind > 1 else None,
                       read_len=int(rng.choice([30, 60, 100])), hap_len=int(rng.choice([200, 300, 500])))
        except AssertionError as e:
            bad += 1
            print("seed %d (W %d, H %d, R %d, individuals %d): %s" % (seed, W, H, R, n_ind, str(e).strip().splitlines()[:3]),
                  flush=True)
        n += 1
        seed += 1
        if time.perf_counter() - last > 30:
            last = time.perf_counter()
            print("... %d cases, %d with differences so far" % (n, bad), flush=True)
    print("%d cases, %d with differences" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
