#!/usr/bin/env python
"""Where BamReader.fetch spends its time: file reads, inflation, record decoding, concatenation.

    python tools/time_fetch.py [genome_kb] [repeats]
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from platypus_amd import bamio, cbam, simulate  # noqa: E402


def main():
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rng = np.random.default_rng(5)
    G = kb * 1000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, G // 1000, spacing=600)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=30, read_len=100)
    genome, truth, reads = simulate.tile(genome, truth, reads, copies)
    G *= copies
    work = tempfile.mkdtemp(prefix="pa_fetch_")
    bam = os.path.join(work, "s.bam")
    bamio.write_bam(bam, [(b"sim", G)], reads)
    r = bamio.BamReader(bam)
    for rep in range(2):
        t0 = time.perf_counter()
        n = 0
        for s in range(0, G, 1000000):
            n += len(r.fetch("sim", s, s + 1000000)["pos"])
        dt = time.perf_counter() - t0
        print("fetch: %d records of %d Mb in %.2f s (%.0f MB/s of BAM)" % (n, G // 1000000, dt, os.path.getsize(bam) / 1e6 / dt),
              flush=True)
    # the parts, over the whole file
    t = {}
    t0 = time.perf_counter()
    raw = np.frombuffer(open(bam, "rb").read(), dtype=np.uint8)
    t["read file"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    data, eaten = cbam.inflateBlocks(raw, np.zeros(0, dtype=np.uint8))
    t["inflate"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    arrays, used, done = cbam.parseRecords(data[r._header_bytes:], 0, 0, G)
    t["decode records"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    bamio._concat([arrays, arrays])
    t["concatenate (two parts of this size)"] = time.perf_counter() - t0
    print("; ".join("%s %.2f s" % kv for kv in t.items()), "(%d MB of BAM, %d MB inflated, %d records)" % (
        len(raw) // 1000000, len(data) // 1000000, len(arrays["pos"])))


if __name__ == "__main__":
    main()
