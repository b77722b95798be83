#!/usr/bin/env python
"""BGZF inflation: device (pg_bgzf_inflate) against the host library on the same simulated BAM stretch.

    python tools/time_inflate.py [genome_kb] [chunk_mb]

Prints per stretch size the kernel's time, the call's wall time and CPU time, and the host library's time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from platypus_amd import bamio, cbam, native, simulate  # noqa: E402


def main():
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(5)
    G = kb * 1000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, G // 1000, spacing=600)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=30, read_len=150, paired=True, insert=(500, 50))
    path = "/tmp/time_inflate.bam"
    bamio.write_bam(path, [(b"sim", G)], reads, sample="S")
    raw = np.fromfile(path, dtype=np.uint8)
    print("BAM %.1f MB" % (raw.size / 1e6), flush=True)
    empty = np.zeros(0, np.uint8)
    t = time.perf_counter()
    want, eaten = cbam.inflateBlocks(raw, empty)
    t_host = time.perf_counter() - t
    print("host %s: %.1f MB -> %.1f MB in %.3f s (%.0f MB/s out)" % (cbam.inflateLibrary(), raw.size / 1e6, want.size / 1e6, t_host,
                                                                      want.size / 1e6 / t_host), flush=True)
    inf = native.Inflater(0)
    for lanes in (1, 8):
        inf.set_lanes(lanes)
        for mb in (1, 4, 16, 64):
            n = min(raw.size, mb << 20)
            piece = raw[:n]
            inf.inflate(piece)                                   # buffers grown, code loaded
            w0, c0 = time.perf_counter(), time.process_time()
            got, used = inf.inflate(piece)
            w1, c1 = time.perf_counter(), time.process_time()
            ok = np.array_equal(got, want[:got.size])
            print("lanes %2d, %2d MB in: %5d blocks, kernel %7.2f ms, call %7.2f ms wall / %6.2f ms cpu, %6.0f MB/s out, equal %s"
                  % (lanes, mb, native.bgzf_scan(piece)[2], inf.last_ms(), (w1 - w0) * 1e3, (c1 - c0) * 1e3,
                     got.size / 1e6 / (w1 - w0), ok), flush=True)
            if n == raw.size:
                break
    inf.close()


if __name__ == "__main__":
    main()
