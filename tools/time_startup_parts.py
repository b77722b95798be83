#!/usr/bin/env python
"""Where a worker's start-up goes: a fresh interpreter times its own steps up to the first result from the device.

    python tools/time_startup_parts.py
"""
import os
import sys
import time

T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
marks = [("interpreter up", T0)]


def mark(name):
    marks.append((name, time.perf_counter()))


import numpy as np  # noqa: E402
mark("numpy")
from platypus_amd import native  # noqa: E402
mark("libplatgpu loaded")
from platypus_amd import bamio, pipeline, cpopulation, chaplotype, calign, cbam, variant, cwindow, vcfout, simulate  # noqa: E402,F401
mark("package modules")
n = native.device_count()
mark("HIP runtime up (device count)")
ctx = native.AlignContext(0, 512)
mark("context (stream, events, tables)")
infl = [native.Inflater(0) for _ in range(3)]
mark("three inflaters")
from platypus_amd import synth  # noqa: E402
b = synth.make_batch(n_windows=4, seed=1)
mark("(a small batch made)")
ctx.upload_batch(b); ctx.launch(); ctx.fetch()
mark("first alignment launch (code objects loaded)")
ctx.upload_batch(b); ctx.launch(); ctx.fetch()
mark("second alignment launch")
prev = T0
for name, t in marks[1:]:
    print("%-48s %7.1f ms   (at %6.1f ms)" % (name, (t - prev) * 1e3, (t - T0) * 1e3))
    prev = t
