"""A few seconds of each randomised check (tests/fuzz_*.py) inside the GPU suite: random batches against the
reference's DP, random window sets of the genotype / EM stage against the checker, random simulated regions
through the whole pipeline.  The long runs are made by hand; their logs are under profiles/."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(name, seconds, seed, monkeypatch):
    sys.path.insert(0, HERE)
    try:
        mod = __import__(name)
        monkeypatch.setattr(sys, "argv", [name + ".py", str(seconds), str(seed)])
        return mod.main()
    finally:
        sys.path.remove(HERE)


def test_random_batches_against_the_reference_dp(monkeypatch):
    assert _run("fuzz_parity", 4, 77001, monkeypatch) == 0


def test_random_windows_against_the_caller_checker(monkeypatch):
    assert _run("fuzz_caller", 4, 77001, monkeypatch) == 0


def test_random_regions_run_through(monkeypatch):
    from platypus_amd import calign
    assert _run("fuzz_pipeline", 4, 77001, monkeypatch) == 0
    calign.closeDevice()
