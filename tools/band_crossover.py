#!/usr/bin/env python
"""Device time of the two alignment kernels (one lane per pair / eight lanes per pair) over launch sizes:
where the library should switch from one to the other.

    python tools/band_crossover.py [read_len]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    from platypus_amd import native, synth
    ctx = native.AlignContext(0, 1024)
    print("%8s %12s %12s" % ("pairs", "lane us", "band us"))
    for W in (8, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
        b = synth.make_batch(n_windows=W, haps_per_window=4, reads_per_window=32, read_len=L, hap_len=5 * L,
                             indel_hap_frac=0.25, seed=7)
        row = []
        for kernel in ("lane", "band"):
            ctx.set_option("align_kernel", kernel)
            ctx.upload_batch(b)
            ts = []
            for _ in range(7):
                ctx.launch(); ctx.sync()
                ts.append(ctx.kernel_ms(native.PG_K_ALIGN) * 1e3)
            row.append(np.median(ts))
        print("%8d %12.1f %12.1f" % (W * 128, row[0], row[1]), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
