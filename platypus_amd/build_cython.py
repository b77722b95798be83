"""Build the Cython 3 layer (calign, cwindow, chaplotype, cgenotype, cpopulation) in-tree.

    python -m platypus_amd.build_cython [--force]

The extension modules link against lib/libplatgpu.so (rpath $ORIGIN/lib).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MODULES = ["cbam", "csimulate", "variant", "calign", "cwindow", "chaplotype", "cgenotype", "cpopulation"]
# plain Python modules of the region caller, compiled as they are (same source, same semantics, less interpreter
# time per window); the .py files stay the source, and are what runs where the build has not been done
PY_MODULES = ["pipeline", "vcfout"]
PXD = ["samtoolsWrapper.pxd", "platgpu.pxd", "calign.pxd", "cwindow.pxd", "chaplotype.pxd", "cgenotype.pxd", "cpopulation.pxd", "variant.pxd"]


def _ext_path(mod):
    return os.path.join(HERE, mod + sysconfig.get_config_var("EXT_SUFFIX"))


def _source(src_dir, mod):
    pyx = os.path.join(src_dir, mod + ".pyx")
    return pyx if os.path.exists(pyx) else os.path.join(src_dir, mod + ".py")


def _stale(mod):
    out = _ext_path(mod)
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [_source(HERE, mod)] + [os.path.join(HERE, p) for p in PXD]
    deps.append(os.path.join(ROOT, "include", "platgpu.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# test-only module that cimports the layer the way the reference's callers do (tests/cy_dropin/refcalls.pyx)
TEST_MODULES = [(os.path.join(ROOT, "tests", "cy_dropin"), "refcalls")]


def build(force=False, verbose=True):
    from Cython.Build import cythonize  # noqa: F401  (fails early if Cython is missing)
    import numpy as np
    inc = [sysconfig.get_paths()["include"], np.get_include(), os.path.join(ROOT, "include")]
    build_dir = os.path.join(HERE, "_cybuild")
    os.makedirs(build_dir, exist_ok=True)
    for src_dir, mod in [(HERE, m) for m in MODULES + PY_MODULES] + TEST_MODULES:
        out = os.path.join(src_dir, mod + sysconfig.get_config_var("EXT_SUFFIX"))
        if src_dir == HERE:
            stale = _stale(mod)
        else:       # after the layer's modules: rebuilt whenever one of them or its own source is newer
            deps = [os.path.join(src_dir, mod + ".pyx")] + [os.path.join(HERE, p) for p in PXD]
            stale = not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)
        if not force and not stale:
            continue
        c_file = os.path.join(build_dir, mod + ".c")
        cmd = [sys.executable, "-m", "cython", "-3", "-I", ROOT, "-o", c_file, _source(src_dir, mod)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=ROOT)
        cc = ["gcc", "-O3", "-fPIC", "-shared", "-fno-strict-aliasing", "-Wno-unused-function",
              "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION"] + ["-I" + i for i in inc] + \
             ["-o", out, c_file, "-L" + os.path.join(HERE, "lib"), "-lplatgpu",
              "-Wl,-rpath," + ("$ORIGIN/lib" if src_dir == HERE else "$ORIGIN/../../platypus_amd/lib"), "-lm"] + (["-lz", "-ldl"] if mod == "cbam" else [])
        if verbose:
            print(" ".join(cc), flush=True)
        subprocess.check_call(cc)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
