import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from platypus_amd import native, synth
ctx = native.AlignContext(0)
for (W, H, R) in [(1, 4, 32), (1, 4, 16), (1, 8, 64), (1, 2, 1), (4, 4, 32)]:
    b = synth.make_batch(n_windows=W, haps_per_window=H, reads_per_window=R, read_len=150, hap_len=750, indel_hap_frac=0.5, seed=5)
    for _ in range(3):
        ctx.upload_batch(b); ctx.launch(); ctx.fetch()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ctx.upload_batch(b); ctx.launch(); ctx.fetch(); ts.append(time.perf_counter() - t0)
    print(W, H, R, "call %.1f us" % (np.median(ts) * 1e6), "anchor %.1f us  align %.1f us" % (ctx.kernel_ms(native.PG_K_ANCHOR) * 1e3, ctx.kernel_ms(native.PG_K_ALIGN) * 1e3), flush=True)
ctx.close()
