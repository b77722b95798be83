#!/usr/bin/env python
"""Latency of the reference's literal call pattern through the Cython layer, one window at a time:

  prefetchWindow(haplotypes, [buffer])       one launch for the window's haplotypes x reads (Caller.setup's hook)
  Haplotype.alignReads(...) without prefetch one launch per haplotype (chaplotype.pyx:276-296 of the reference,
                                             called per genotype from cgenotype.pyx:55-56)
  alignNoTraceback(read, haplotype)          one launch per pair (the reference's innermost call)

against the same pairs as ONE batch through the ABI.  Every call uploads its haplotypes and reads, runs the
kernels and fetches the likelihoods, as a caller that keeps nothing on the device between windows would.

    python tools/time_window_calls.py [windows] [haplotypes] [reads] [read_len]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    L = int(sys.argv[4]) if len(sys.argv) > 4 else 150
    from platypus_amd import calign, chaplotype, native, synth
    from oracle import pyoracle as po
    from test_cython_layer import _buffers
    b = synth.make_batch(n_windows=W, haps_per_window=H, reads_per_window=R, read_len=L, hap_len=750,
                         indel_hap_frac=0.25, seed=3)
    cells = (L + 8) * 16
    wins = []
    for w, (order, buf) in enumerate(_buffers(b, H, R)):
        seqs = [b["hap_seq"][b["hap_off"][h]:b["hap_off"][h + 1]].tobytes() for h in range(w * H, (w + 1) * H)]
        buf.setWindow(int(b["hap_start"][w * H]), int(b["hap_start"][w * H]) + 750)
        wins.append((seqs, buf, [int(b["hap_start"][h]) for h in range(w * H, (w + 1) * H)],
                     [int(b["hap_varlen"][h]) for h in range(w * H, (w + 1) * H)]))

    def fresh(w):
        seqs, buf, starts, vl = wins[w]
        return [chaplotype.Haplotype.fromSequence(b"chr20", starts[k], starts[k] + len(seqs[k]), seqs[k], vl[k])
                for k in range(H)], buf

    def report(name, times, pairs_per_call):
        t = np.array(times) * 1e6
        print("%-44s median %8.1f us  p10 %8.1f  p90 %8.1f  per call of %5d pairs -> %7.2f GCUPS" % (
            name, np.median(t), np.percentile(t, 10), np.percentile(t, 90), pairs_per_call,
            pairs_per_call * cells / np.median(t) / 1e3), flush=True)

    haps, buf = fresh(0)
    chaplotype.prefetchWindow(haps, [buf])                       # context, first launch
    # (1) one launch per window
    times = []
    for w in range(W):
        haps, buf = fresh(w)
        t0 = time.perf_counter()
        chaplotype.prefetchWindow(haps, [buf])
        times.append(time.perf_counter() - t0)
    n_in = sum(wins[w][1].windowRange()[1] - wins[w][1].windowRange()[0] for w in range(W)) / W
    report("prefetchWindow, one window per call", times, int(H * n_in))
    # (2) one launch per haplotype
    times = []
    for w in range(min(W, 100)):
        haps, buf = fresh(w)
        for hp in haps:
            t0 = time.perf_counter()
            hp.alignReadsPy(0, buf, 1)
            times.append(time.perf_counter() - t0)
    report("alignReads, one haplotype per call", times, int(n_in))
    # (3) one launch per pair
    times = []
    for r in range(min(300, W * R)):
        h = (r // R) * H + 1
        args = (b["read_seq"][b["read_off"][r]:b["read_off"][r + 1]].tobytes(),
                b["read_qual"][b["read_off"][r]:b["read_off"][r + 1]].tobytes(), int(b["read_pos"][r]), int(b["read_mapq"][r]),
                b["hap_seq"][b["hap_off"][h]:b["hap_off"][h + 1]].tobytes(), int(b["hap_start"][h]), int(b["hap_varlen"][h]),
                po.HOMOPOLQ)
        t0 = time.perf_counter()
        calign.alignNoTracebackPy(*args)
        times.append(time.perf_counter() - t0)
    report("alignNoTraceback, one pair per call", times, 1)
    calign.closeDevice()
    # (4) the same windows as one batch through the ABI (upload + launch + fetch)
    ctx = native.AlignContext(0)
    try:
        def win(w):
            h0, r0 = w * H, w * R
            return (b["hap_seq"], b["hap_off"][h0:h0 + H + 1], b["hap_start"][h0:h0 + H], b["hap_varlen"][h0:h0 + H],
                    b["read_seq"], b["read_qual"], b["read_off"][r0:r0 + R + 1], b["read_pos"][r0:r0 + R], b["read_mapq"][r0:r0 + R])
        for which, name in ((0, "pg_align_window, one window per call"), (1, "the five upload/launch/fetch calls per window")):
            times = []
            for w in list(range(min(W, 20))) + list(range(W)):
                a = win(w)
                t0 = time.perf_counter()
                if which == 0:
                    ctx.align_window(*a)
                else:
                    ctx.upload_haps(*a[:4]); ctx.upload_reads(*a[4:]); ctx.pairs_block(0, H, 0, R); ctx.launch(); ctx.fetch()
                times.append(time.perf_counter() - t0)
            report(name, times[min(W, 20):], H * R)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.upload_batch(b)
            ctx.launch()
            ctx.fetch()
            times.append(time.perf_counter() - t0)
        report("all %d windows as one batch" % W, times, W * H * R)
    finally:
        ctx.close()


if __name__ == "__main__":
    main()
