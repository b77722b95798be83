#!/usr/bin/env python
"""Where a file-to-file run's time goes: the data set of one benchmark configuration (bench_e2e.make_data), one run of
platypus_amd.runner with its timing report (PLATYPUS_AMD_TIMINGS), optionally under cProfile.

    python tools/e2e_profile.py wgs|panel|joint [mb] [--cprofile] [extra runner arguments]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import bench_e2e  # noqa: E402


def main():
    args = sys.argv[1:]
    workload = args.pop(0)
    mb = float(args.pop(0)) if args and args[0].replace(".", "").isdigit() else None
    prof = "--cprofile" in args
    if prof:
        args.remove("--cprofile")
    work = tempfile.mkdtemp(prefix="e2e_profile_")
    bams, fa, G, truth, what, extra = bench_e2e.make_data(workload, work, mb)
    print(what, extra, flush=True)
    out = os.path.join(work, "calls.vcf")
    argv = ["callVariants", "--bamFiles", ",".join(bams), "--refFile", fa, "--regions", "sim:1-%d" % (G - 1), "--output", out,
            "--maxReadLength", "150", "--processRegionSize", "1000000", "--bufferSize", "1000000"] + args
    os.environ["PLATYPUS_AMD_TIMINGS"] = "1"
    from platypus_amd import runner
    t0 = time.perf_counter()
    if prof:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        runner.main(argv)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    else:
        runner.main(argv)
    dt = time.perf_counter() - t0
    print("%.2f s for %.1f Mb: %.0f Mbp/min" % (dt, G / 1e6, G / 1e6 / dt * 60), flush=True)


if __name__ == "__main__":
    main()
