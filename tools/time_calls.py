#!/usr/bin/env python
"""Where the calling thread's time goes, by callee: wraps the functions callRegion calls (compiled ones included,
which a Python profiler cannot see inside) with timers and runs tools/time_files.py's single-process case.

    TF_REPEATS=16 TF_PAIRED=1 TF_READLEN=150 python tools/time_calls.py 2000 30
"""
import functools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

spent, count = {}, {}


def timed(name, fn):
    @functools.wraps(fn)
    def wrapper(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            spent[name] = spent.get(name, 0.0) + time.perf_counter() - t0
            count[name] = count.get(name, 0) + 1
    return wrapper


def main():
    from platypus_amd import chaplotype, cpopulation, cwindow, pipeline, vcfout
    import time_files

    class Batch(cpopulation.WindowBatch):
        pass
    for m in ("add", "run", "posteriors", "siteLikelihoods", "siteStatistics", "readCounts"):
        setattr(Batch, m, timed("WindowBatch." + m, getattr(cpopulation.WindowBatch, m)))
    cpopulation.WindowBatch = Batch
    real_hap = chaplotype.Haplotype

    class Hap(real_hap):
        pass
    pipeline.chaplotype = type("M", (), dict(Haplotype=timed("Haplotype()", real_hap),
                                             variantsCompatible=timed("variantsCompatible", chaplotype.variantsCompatible)))
    pipeline.cwindow = type("M", (), dict(windowReadCount=timed("windowReadCount", cwindow.windowReadCount)))
    for name in ("siteStatistics", "windowFilters", "haplotypesInWindow", "homopolymerLength", "callRegion"):
        setattr(pipeline, name, timed(name, getattr(pipeline, name)))
    pipeline.WindowGenerator.bunches = staticmethod(timed("bunches", pipeline.WindowGenerator.bunches))
    vcfout.record = timed("vcfout.record", vcfout.record)
    time_files.main()
    for k in sorted(spent, key=spent.get, reverse=True):
        print("%-28s %9d calls  %7.3f s" % (k, count[k], spent[k]))


if __name__ == "__main__":
    main()
