"""The CPU checker against the golden vectors made from the reference's own
compiled routine (tests/golden/gen_golden.py), and, where that build is
present, against the routine itself on fresh random cases."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from platypus_amd import synth


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "align_golden.npz"))
    len2 = g["len2"]
    o2 = np.concatenate([[0], np.cumsum(len2)])
    o1 = np.concatenate([[0], np.cumsum(len2 + 15)])
    for i in range(len(len2)):
        yield (g["seq1"][o1[i]:o1[i + 1]].tobytes(), g["seq2"][o2[i]:o2[i + 1]].tobytes(),
               g["qual"][o2[i]:o2[i + 1]].tobytes(), int(g["score"][i]))


def test_known_answers_and_golden(golden_dir):
    n = bad = 0
    for s1, s2, q, exp in _cases(golden_dir):
        n += 1
        bad += po.fast_align(s1, s2, q) != exp
    assert n > 3000 and bad == 0


def test_homopolq_literal():
    # chaplotype.pyx:37-38 computes the table; chaplotype.pyx:40 holds the same string
    import math
    errs = [2.9e-5] * 4 + [4.3e-5, 1.1e-4, 2.4e-4, 5.7e-4, 1.0e-3, 1.4e-3] + \
           [1.4e-3 + 4.3e-4 * (n - 10) for n in range(11, 50)]
    s = bytes(int(33.5 + 10 * math.log((i + 1) * q) / math.log(0.1)) for i, q in enumerate(errs))
    assert s == po.HOMOPOLQ and len(s) == 49


def test_window_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "window_golden.npz"))
    for name in ("wgs", "panel"):
        b = {k.split("__", 1)[1]: g[k] for k in g.files if k.startswith(name + "__")}
        score, anchor = po.run_pairs(synth.oracle_view(b))
        assert np.array_equal(score, b["score"]) and np.array_equal(anchor, b["anchor"])
        # hash table built once per haplotype gives the same answers as once per pair
        s2, a2 = po.run_pairs(synth.oracle_view(b), reuse_hash=True, threads=2)
        assert np.array_equal(s2, score) and np.array_equal(a2, anchor)


@pytest.mark.skipif(not po.have_ref(), reason="reference build absent")
def test_restatement_equals_reference_routine_random():
    rng = np.random.default_rng(7)
    for _ in range(1500):
        L = int(rng.integers(8, 300))
        s1 = rng.choice(np.frombuffer(b"ACGTN", np.uint8), p=[.24, .24, .24, .24, .04], size=L + 15)
        if rng.random() < 0.3:
            p = int(rng.integers(0, L)); s1[p:p + int(rng.integers(2, 70))] = s1[p]
        off = int(rng.integers(0, 16))
        s2 = np.resize(s1[off:], L).copy()
        m = rng.random(L) < 0.05
        s2[m] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(m.sum()))
        q = rng.integers(0, 94, size=L).astype(np.uint8)
        assert po.fast_align(s1.tobytes(), s2.tobytes(), q.tobytes()) == \
            po.ref_fast_align(s1.tobytes(), s2.tobytes(), q.tobytes())


def test_vote_matches_bruteforce():
    """map_read against an independent dictionary implementation of
    calign.pyx:126-158 (first position to reach the final maximum wins)."""
    code = {65: 1, 67: 3, 71: 2, 84: 0, 78: 2}
    rng = np.random.default_rng(3)
    for _ in range(200):
        hl = int(rng.integers(60, 900))
        hap = rng.choice(np.frombuffer(b"ACGTN", np.uint8), p=[.245] * 4 + [.02], size=hl)
        rl = int(rng.integers(10, 160))
        st = int(rng.integers(0, max(1, hl - rl)))
        read = np.resize(hap[st:], rl).copy()
        m = rng.random(rl) < 0.03
        read[m] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(m.sum()))
        h = lambda s, i: sum(code[int(s[i + j])] << (2 * (6 - j)) for j in range(7))
        first, cnt = {}, {}
        for i in range(hl - 7):
            k = h(hap, i); cnt[k] = cnt.get(k, 0) + 1; first.setdefault(k, i)
        votes, best, bestpos = {}, 0, 0
        for i in range(rl - 7):
            k = h(read, i)
            if cnt.get(k, 0) == 1:
                p = first[k] - i
                votes[p] = votes.get(p, 0) + 1
                if votes[p] > best:
                    best, bestpos = votes[p], p
        tab = po.hash_sequence(hap.tobytes())
        assert po.map_read(read.tobytes(), hl, tab) == bestpos


def test_likelihood_formula():
    import math
    for sc, mq in [(0, 60), (30, 20), (2048, 0), (5, 255)]:
        pw = math.exp(-0.23025850929940459 * mq)
        assert po.likelihood(sc, mq) == min(1.0, math.exp(-0.23025850929940459 * sc) * (1 - pw) + pw)


def test_restatements_under_address_and_ub_sanitizers(golden_dir, tmp_path):
    """make -C oracle asan: both C restatements built with ASan + UBSan; the driver replays every golden
    fastAlignmentRoutine call from exact-size heap blocks (a read past either end aborts), then random
    windows through the hash / vote / likelihood wrapper and the genotype / EM / posterior stage."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    g = np.load(os.path.join(golden_dir, "align_golden.npz"))
    len2 = g["len2"].astype(np.int64)
    o1 = np.concatenate([[0], np.cumsum(len2 + 15)])
    o2 = np.concatenate([[0], np.cumsum(len2)])
    path = tmp_path / "golden.bin"
    with open(path, "wb") as f:
        f.write(np.int32(len(len2)).tobytes())
        for i in range(len(len2)):
            f.write(np.array([len2[i], g["score"][i]], np.int32).tobytes())
            f.write(g["seq1"][o1[i]:o1[i + 1]].tobytes())
            f.write(g["seq2"][o2[i]:o2[i + 1]].tobytes())
            f.write(g["qual"][o2[i]:o2[i + 1]].tobytes())
    r = subprocess.run([os.path.join(root, "oracle", "_build", "sanitize_driver"), str(path)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert "%d golden calls, 0 mismatches" % len(len2) in r.stdout
