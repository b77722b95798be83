#!/usr/bin/env python
"""Time of each ABI call for one small window (4 haplotypes x 32 reads), median of many repeats:
where the 0.3 ms of a single-window call go.

    python tools/time_small_calls.py [haplotypes] [reads]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    from platypus_amd import native, synth
    b = synth.make_batch(n_windows=1, haps_per_window=H, reads_per_window=R, read_len=150, hap_len=750, indel_hap_frac=0.25, seed=5)
    ctx = native.AlignContext(0)
    spent = {}

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        r = fn(*a)
        spent.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    try:
        for rep in range(300):
            timed("haps_upload", ctx.upload_haps, b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
            timed("reads_upload", ctx.upload_reads, b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
            timed("pairs_block", ctx.pairs_block, 0, H, 0, R)
            timed("align_launch", ctx.launch)
            timed("results_fetch", ctx.fetch)
        tot = 0.0
        for k, v in spent.items():
            m = float(np.median(v[20:])) * 1e6
            tot += m
            print("%-14s %7.1f us" % (k, m))
        print("%-14s %7.1f us" % ("sum", tot))
    finally:
        ctx.close()


if __name__ == "__main__":
    main()
