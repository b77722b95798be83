"""Build libplatgpu.so (HIP, gfx950 only) in-tree: platypus_amd/lib/libplatgpu.so.

    python -m platypus_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is rebuilt when a source is
newer than it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC_DIR = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libplatgpu.so")
SOURCES = ["platgpu.hip"]
DEPS = ["platgpu.hip", "pg_kernels.hip.h", "pg_align.hip.h", "pg_prep.hip.h", "pg_caller.hip.h", "pg_inflate.hip.h", "pg_window.hip.h", "pg_band.hip.h", os.path.join(ROOT, "include", "platgpu.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sdwa-peephole=false",
         "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for d in DEPS:
        p = d if os.path.isabs(d) else os.path.join(SRC_DIR, d)
        if os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=True, extra=()):
    if not force and not _stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + list(extra) + ["-o", LIB] + [os.path.join(SRC_DIR, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
