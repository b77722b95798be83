#!/usr/bin/env python
"""Caller-stage times (genotype setup, EM) for window sets of a given shape, the alignment batch behind them included:

    python tools/em_shape.py WINDOWS HAPLOTYPES READS INDIVIDUALS [NAME=VALUE options ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    W, H, R, I = (int(x) for x in sys.argv[1:5])
    from platypus_amd import native, synth
    b = synth.make_batch(n_windows=W, haps_per_window=H, reads_per_window=R, read_len=150, hap_len=750, indel_hap_frac=0.25, seed=11)
    ctx = native.AlignContext(0)
    try:
        for kv in sys.argv[5:]:
            ctx.set_option(*kv.split("=", 1))
        ctx.upload_batch(b)
        best = None
        for _ in range(5):
            ctx.launch()
            ctx.upload_windows([H] * W, [I] * W, np.arange(W, dtype=np.int64) * H * R,
                               np.tile(np.arange(I + 1, dtype=np.int32) * (R // I), W))
            ctx.caller_launch(100)
            ctx.sync()
            res = ctx.caller_fetch(genotype_post=False)
            t = (ctx.kernel_ms(native.PG_K_CALLER_SETUP), ctx.kernel_ms(native.PG_K_CALLER_EM))
            best = t if best is None or t[1] < best[1] else best
        it = res["iters"]
        tri = float(it.sum()) * I * (H * (H + 1) // 2) * 8.0
        print("%d windows x %d haplotypes x %d individuals: setup %.3f ms, EM %.3f ms, iterations mean %.1f max %d, "
              "triangle bytes %.2f GB -> %.0f GB/s over the stage" % (W, H, I, best[0], best[1], it.mean(), it.max(), tri / 1e9,
                                                                    tri / best[1] / 1e6))
    finally:
        ctx.close()


if __name__ == "__main__":
    main()
