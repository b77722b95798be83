#!/usr/bin/env python
"""cProfile of one region call with the region caller's Python modules loaded from their sources (the in-tree build
compiles them; a profiler sees nothing inside compiled code): which functions the calling thread spends its time in.

    python tools/profile_region.py [genome_kb]

The compiled classes the region caller uses are invisible to the profiler; they are wrapped in timing subclasses and
their seconds printed beside it (calls into the device library included).
"""
import cProfile
import importlib.util
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def source_module(name):
    path = os.path.join(ROOT, "platypus_amd", name + ".py")
    spec = importlib.util.spec_from_file_location("platypus_amd." + name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["platypus_amd." + name] = mod
    spec.loader.exec_module(mod)
    return mod


SPENT = {}


def timed_subclass(cls, names):
    """subclass of a compiled class whose named methods add their wall time to SPENT"""
    import time

    def wrap(name):
        inner = getattr(cls, name)

        def method(self, *a, **kw):
            t = time.perf_counter()
            try:
                return inner(self, *a, **kw)
            finally:
                key = cls.__name__ + "." + name
                n, sec = SPENT.get(key, (0, 0.0))
                SPENT[key] = (n + 1, sec + time.perf_counter() - t)
        return method
    return type(cls.__name__, (cls,), {n: wrap(n) for n in names})


class Shim(object):
    """a module's namespace with some names replaced"""

    def __init__(self, module, **replaced):
        self.__dict__.update({k: getattr(module, k) for k in dir(module) if not k.startswith("__")})
        self.__dict__.update(replaced)


def main():
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    import platypus_amd
    vcfout = source_module("vcfout")
    pipeline = source_module("pipeline")
    from platypus_amd import chaplotype, cpopulation, cwindow
    pipeline.cpopulation = Shim(cpopulation, WindowBatch=timed_subclass(cpopulation.WindowBatch, [
        "add", "run", "posteriors", "siteStatistics", "readCounts", "siteLikelihoods", "siteLikelihoodsResident"]))
    pipeline.chaplotype = Shim(chaplotype, Haplotype=timed_subclass(chaplotype.Haplotype, ["__init__", "variantsPy"]))
    from platypus_amd import simulate
    from platypus_amd.fastafile import FastaFile
    rng = np.random.default_rng(5)
    G = kb * 1000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, G // 1000, spacing=600)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=30, read_len=150, paired=True, insert=(500, 50))
    fa = FastaFile(sequences={b"sim": genome})
    opts = pipeline.Options(rlen=150)
    bufs = [simulate.buffer_of(reads, 0, G)]
    cands = pipeline.candidatesInRegion(b"sim", 0, G, bufs, fa, opts)
    pipeline.callRegion(b"sim", 0, G, bufs, fa, opts, variants=cands)            # warm
    SPENT.clear()
    prof = cProfile.Profile()
    prof.enable()
    calls = pipeline.callRegion(b"sim", 0, G, bufs, fa, opts, variants=cands)
    lines = [vcfout.record(c, fa) for c in calls]
    prof.disable()
    print("%d calls, %d lines" % (len(calls), len(lines)))
    pstats.Stats(prof).sort_stats("tottime").print_stats(28)
    print("compiled classes, seconds inside (profiler overhead of the wrappers included):")
    for key, (n, sec) in sorted(SPENT.items(), key=lambda kv: -kv[1][1]):
        print("  %-40s %7d calls  %.4f s" % (key, n, sec))


if __name__ == "__main__":
    main()
