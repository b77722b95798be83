"""A run that loses a stretch must not report success: an error inside a worker (other than the per-window errors
callRegion itself skips, platypus/variantcaller.pyx:189-191) ends the run with a non-zero status, as the reference's
region wrapper re-raises (platypus/variantcaller.pyx:238-248).  CPU only: the device and the region call are stand-ins."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
from platypus_amd import runner                      # (sets the BLAS thread default before numpy is loaded)
from platypus_amd import calign, pipeline
calign.openDevice = lambda *a, **k: None             # no GPU here
def failing(chrom, s, e, *a, **k):
    if s >= %(fail_from)d:
        raise RuntimeError("injected device failure")
    return []
pipeline.callRegion = failing
print("fork mode:", runner._gpu_untouched() and runner._single_threaded(), flush=True)
sys.exit(runner.main(%(argv)r))
"""


def _files(tmp_path):
    from platypus_amd import bamio, simulate
    rng = np.random.default_rng(3)
    G = 12000
    genome = simulate.random_genome(rng, G)
    truth = simulate.plant_variants(rng, genome, 8, spacing=500)
    fa = tmp_path / "ref.fa"
    fa.write_bytes(b">sim\n" + b"\n".join(genome[i:i + 60] for i in range(0, G, 60)) + b"\n")
    (tmp_path / "ref.fa.fai").write_bytes(b"sim\t%d\t5\t60\t61\n" % G)
    reads = simulate.simulate_reads(rng, genome, truth, coverage=8, read_len=100)
    bam = str(tmp_path / "s.bam")
    bamio.write_bam(bam, [(b"sim", G)], reads)
    return G, str(fa), bam


def _run(tmp_path, fail_from, extra):
    G, fa, bam = _files(tmp_path)
    out = str(tmp_path / "calls.vcf")
    argv = ["callVariants", "--bamFiles", bam, "--refFile", fa, "--regions", "sim:1-%d" % (G - 1), "--output", out,
            "--processRegionSize", "3000", "--bufferSize", "3000", "--maxReadLength", "100", "--inflate", "host"] + extra
    script = tmp_path / "run.py"
    script.write_text(_SCRIPT % {"root": ROOT, "fail_from": fail_from, "argv": argv})
    done = subprocess.run([sys.executable, str(script)], cwd=ROOT, timeout=120, capture_output=True, text=True)
    return done, out


def test_a_failing_stretch_fails_a_single_process_run(tmp_path):
    done, out = _run(tmp_path, 6000, [])
    assert done.returncode != 0 and "injected device failure" in done.stderr
    assert not os.path.exists(out)                       # no merged file pretends to be the run's result


def test_a_failing_worker_fails_the_run_in_fork_mode(tmp_path):
    done, out = _run(tmp_path, 6000, ["--workersPerGPU", "2"])
    assert "fork mode: True" in done.stdout                # workers forked: the stand-ins reach them
    assert done.returncode != 0 and "injected device failure" in done.stderr
    assert not os.path.exists(out)


def test_stand_ins_alone_give_a_clean_run(tmp_path):
    """the same command with nothing failing: exit status 0 and a header-only file (the stand-in calls nothing)"""
    done, out = _run(tmp_path, 10 ** 9, ["--workersPerGPU", "2"])
    assert done.returncode == 0, done.stderr[-2000:]
    lines = open(out).read().splitlines()
    assert lines and all(l.startswith("#") for l in lines) and sum(l.startswith("#CHROM") for l in lines) == 1
