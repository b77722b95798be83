#!/usr/bin/env python
"""Device inflaters on helper threads beside an alignment context on the main thread: does the main thread's
batch time suffer?  python tools/stress_inflate.py [threads]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from platypus_amd import native, synth  # noqa: E402


def main():
    n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    raw = np.fromfile("/tmp/time_inflate.bam", dtype=np.uint8)
    b = synth.make_batch(n_windows=512, haps_per_window=4, reads_per_window=32, read_len=150, indel_hap_frac=0.2, seed=5)
    ctx = native.AlignContext(0)

    def batch():
        t = time.perf_counter()
        ctx.upload_batch(b)
        ctx.launch()
        ctx.fetch()
        return (time.perf_counter() - t) * 1e3

    batch()
    alone = np.median([batch() for _ in range(20)])
    stop = threading.Event()
    counts = [0] * n_threads

    def helper(k):
        inf = native.Inflater(0)
        while not stop.is_set():
            inf.inflate(raw)
            counts[k] += 1
        inf.close()

    threads = [threading.Thread(target=helper, args=(k,)) for k in range(n_threads)]
    t0 = time.perf_counter()
    c0 = time.process_time()
    for t in threads:
        t.start()
    time.sleep(0.3)
    beside = [batch() for _ in range(40)]
    time.sleep(1.0)
    stop.set()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    print("batch alone %.2f ms, beside %d inflaters median %.2f ms (max %.2f); %d inflations of %.0f MB in %.2f s = %.0f MB/s in, "
          "process CPU %.2f s" % (alone, n_threads, np.median(beside), max(beside), sum(counts), raw.size / 1e6, dt,
                                 sum(counts) * raw.size / 1e6 / dt, time.process_time() - c0), flush=True)


if __name__ == "__main__":
    main()
