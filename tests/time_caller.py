#!/usr/bin/env python
"""Times the genotype-likelihood / EM stage on the benchmark pair streams (GPU kernels by HIP events,
the CPU restatement on a sample of windows beside it).

    python tests/time_caller.py [wgs|panel|joint] [windows-of-wgs-shape]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from platypus_amd import native, synth  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "wgs"
    windows = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    max_iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    H, R, extra, _ = bench.WORKLOADS[wl]
    n_ind = 50 if wl == "joint" else 1
    # shape overrides for experiments: TC_H, TC_R, TC_IND
    H = int(os.environ.get("TC_H", H)); R = int(os.environ.get("TC_R", R)); n_ind = int(os.environ.get("TC_IND", n_ind))
    nw = max(1, windows * 128 // (H * R))
    b = synth.make_batch(n_windows=nw, haps_per_window=H, reads_per_window=R, read_len=150, hap_len=750,
                         seed=synth.SEED, **extra)
    ctx = native.AlignContext(0)
    ctx.upload_batch(b)
    off = np.linspace(0, R, n_ind + 1).astype(np.int32)
    ctx.upload_windows([H] * nw, [n_ind] * nw, np.arange(nw, dtype=np.int64) * H * R, np.tile(off, nw))
    for rep in range(3):
        ctx.launch()
        t0 = time.perf_counter()
        ctx.caller_launch(max_iters)
        ctx.sync()
        dt = time.perf_counter() - t0
        print("rep %d: align %.3f ms  anchor %.3f ms | setup %.3f ms  em %.3f ms  (host wall of the caller stage %.3f ms)" % (
            rep, ctx.kernel_ms(native.PG_K_ALIGN), ctx.kernel_ms(native.PG_K_ANCHOR),
            ctx.kernel_ms(native.PG_K_CALLER_SETUP), ctx.kernel_ms(native.PG_K_CALLER_EM), dt * 1e3), flush=True)
    res = ctx.caller_fetch(genotype_post=False)
    print("windows %d  haplotypes %d  individuals %d  genotypes %d  iterations: mean %.1f max %d" % (
        nw, H, n_ind, H * (H + 1) // 2, res["iters"].mean(), res["iters"].max()))
    print("iteration histogram (<=5, <=10, <=20, <=50, <100, 100):",
          [int(((res["iters"] > lo) & (res["iters"] <= hi)).sum()) for lo, hi in
           ((0, 5), (5, 10), (10, 20), (20, 50), (50, 99), (99, 100))])
    # CPU restatement on a sample of windows (single thread)
    from oracle import pyoracle as po
    _, _, like = ctx.fetch()
    ns = min(nw, 2048 if wl != "joint" else 1)
    t0 = time.perf_counter()
    for w in range(ns):
        L = like[w * H * R:(w + 1) * H * R].reshape(H, R)
        post = po.caller_setup(L, off)
        po.caller_call(post, off, H, max_iters=max_iters)
    dt = time.perf_counter() - t0
    print("CPU restatement (1 thread, one-pass EM form): %.3f ms per window on %d windows -> %.1f ms for all" % (
        dt / ns * 1e3, ns, dt / ns * nw * 1e3))
    ctx.close()


if __name__ == "__main__":
    main()
