#!/usr/bin/env python
"""Randomised parity run: batches of random shape and content through the C ABI against the CPU
checker (DP by the reference's compiled routine when oracle/_ref is built), for a time budget.

    python tests/fuzz_parity.py [seconds] [first_seed]

Every batch draws its own window count, haplotypes and reads per window, read length (8..300,
single or ragged), haplotype length (60..2000), share of indel haplotypes, homopolymer runs, 'N'
share, substitution and read-indel rates; then reads are moved (some far enough to hang off the
haplotype or to trip the assertion of calign.pyx:61), qualities get extremes (0, 93, 127), read
bases get 'N's, mapping qualities cover 0..255, the pair list is a shuffled subset with repeats,
and one batch in four uses other gap parameters.  The alignment kernel (one lane per pair, eight
lanes per pair) is the library's choice for the batch's size in half of the batches and forced to one
or the other in the rest; one batch in three also goes window by window through pg_align_window
(dense windows, before the pair list is shuffled).  Scores and anchors must be equal, likelihoods
the same doubles.  Prints one line per mismatching batch and a summary; exit status 1 on any."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as po  # noqa: E402
from platypus_amd import native, synth  # noqa: E402


def make(seed):
    rng = np.random.default_rng(seed)
    L = int(rng.choice([8, 12, 25, 36, 60, 100, 150, 151, 250, 300]))
    hap_len = int(rng.integers(max(60, L + 80), 2001))
    b = synth.make_batch(n_windows=int(rng.integers(1, 40)), haps_per_window=int(rng.integers(1, 8)),
                         reads_per_window=int(rng.integers(1, 70)), read_len=L, hap_len=hap_len,
                         indel_hap_frac=float(rng.choice([0.0, 0.3, 1.0])), homopolymers=bool(rng.random() < 0.4),
                         n_frac=float(rng.choice([0.0, 0.0, 0.005, 0.05])), sub_rate=float(rng.choice([0.0, 0.01, 0.1])),
                         read_indel_frac=float(rng.choice([0.0, 0.1, 0.5])), seed=seed, ragged=bool(rng.random() < 0.5))
    nR = len(b["read_pos"])
    # positions: mostly as simulated, some shifted, a few far off either end
    shift = np.where(rng.random(nR) < 0.15, rng.integers(-40, 41, nR), 0)
    far = rng.random(nR) < 0.03
    shift = np.where(far, rng.integers(-hap_len, hap_len, nR), shift)
    b["read_pos"] = (b["read_pos"].astype(np.int64) + shift).astype(np.int32)
    q = b["read_qual"]
    r = rng.random(len(q))
    q[r < 0.01] = 0
    q[(r >= 0.01) & (r < 0.02)] = 93
    q[(r >= 0.02) & (r < 0.025)] = 127
    s = b["read_seq"]
    s[rng.random(len(s)) < float(rng.choice([0.0, 0.002, 0.05]))] = ord("N")
    b["read_mapq"] = np.where(rng.random(nR) < 0.1, rng.integers(0, 256, nR), b["read_mapq"]).astype(np.int32)
    n = len(b["pair_hap"])
    b["dense"] = (b["pair_hap"].copy(), b["pair_read"].copy()) if rng.random() < 1.0 / 3 else None
    b["align_kernel"] = str(rng.choice(["default", "default", "lane", "band"]))
    # how the batch goes up: haplotypes whole or as stretches and variant lists (pg_haps_build), read bases as characters
    # or two per byte (pg_reads_upload_nibbles)
    b["build_haps"] = bool(rng.random() < 0.4)
    b["nibbles"] = bool(rng.random() < 0.4)
    pick = rng.integers(0, n, size=int(rng.integers(1, 2 * n + 1)))
    b["pair_hap"], b["pair_read"] = b["pair_hap"][pick].copy(), b["pair_read"][pick].copy()
    params = (3, 2, po.HOMOPOLQ)
    if rng.random() < 0.25:
        tab = bytes(int(v) for v in np.sort(rng.integers(34, 100, int(rng.integers(5, 60))))[::-1])
        params = (int(rng.integers(1, 9)), int(rng.integers(0, 7)), tab)
    return b, params


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    use_ref = po.have_ref()
    ctx = native.AlignContext(0)
    t0 = time.perf_counter()
    n_batches = n_pairs = n_bad = n_assert = n_refused = n_window_pairs = 0
    mLTOT = -0.23025850929940459
    last_report = t0
    while time.perf_counter() - t0 < budget:
        if time.perf_counter() - last_report > 30:             # a line every half minute: the run is alive
            last_report = time.perf_counter()
            print("... %d batches, %d pairs, %d with differences so far" % (n_batches, n_pairs, n_bad), flush=True)
        b, (ge, npr, tab) = make(seed)
        try:
            ctx.set_params(ge, npr, tab)
            ctx.set_option("align_kernel", b["align_kernel"])
            windows = None
            if b["dense"] is not None:                           # window by window, every haplotype x every read
                m = b["meta"]
                H, R = m["haps_per_window"], m["reads_per_window"]
                windows = [ctx.align_window(b["hap_seq"], b["hap_off"][w * H:(w + 1) * H + 1], b["hap_start"][w * H:(w + 1) * H],
                                            b["hap_varlen"][w * H:(w + 1) * H], b["read_seq"], b["read_qual"],
                                            b["read_off"][w * R:(w + 1) * R + 1], b["read_pos"][w * R:(w + 1) * R],
                                            b["read_mapq"][w * R:(w + 1) * R]) for w in range(m["windows"])]
            if b["build_haps"]:
                ctx.build_haps(**b["built"])
            else:
                ctx.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"])
            if b["nibbles"]:
                ctx.upload_reads_nibbles(native.pack_nibbles(b["read_seq"]), b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
            else:
                ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"])
            ctx.upload_pairs(b["pair_hap"], b["pair_read"])
            ctx.launch()
            score, anchor, like = ctx.fetch()
        except native.PlatGpuError as e:
            if "unsupported" in str(e).lower() or "shorter" in str(e) or "15" in str(e):
                n_refused += 1
                seed += 1
                continue
            raise
        cs, ca = po.run_pairs(synth.oracle_view(b), use_ref=use_ref, gapextend=ge, nucprior=npr, table=tab)
        ok = np.array_equal(cs, score) and np.array_equal(ca, anchor)
        mq = b["read_mapq"][b["pair_read"]]
        want = np.array([0.0 if s == native.PG_SCORE_ASSERT_FAILED else
                         min(1.0, math.exp(mLTOT * s) * (1.0 - math.exp(mLTOT * m)) + math.exp(mLTOT * m))
                         for s, m in zip(cs.tolist(), mq.tolist())])
        ok = ok and np.array_equal(want, like)
        if windows is not None:
            d = dict(synth.oracle_view(b))
            d["pair_hap"], d["pair_read"] = b["dense"]
            ds, da = po.run_pairs(d, use_ref=use_ref, gapextend=ge, nucprior=npr, table=tab)
            dq = b["read_mapq"][d["pair_read"]]
            dl = np.array([0.0 if s == native.PG_SCORE_ASSERT_FAILED else
                           min(1.0, math.exp(mLTOT * s) * (1.0 - math.exp(mLTOT * m)) + math.exp(mLTOT * m))
                           for s, m in zip(ds.tolist(), dq.tolist())])
            ws = np.concatenate([w[0] for w in windows]); wa = np.concatenate([w[1] for w in windows])
            wl = np.concatenate([w[2] for w in windows])
            if not (np.array_equal(ws, ds) and np.array_equal(wa, da) and np.array_equal(wl, dl)):
                ok = False
                print("seed %d: the window calls differ from the checker" % seed, flush=True)
            n_window_pairs += len(ds)
        n_assert += int((cs == native.PG_SCORE_ASSERT_FAILED).sum())
        if not ok:
            n_bad += 1
            bad = np.flatnonzero((cs != score) | (ca != anchor) | (want != like))
            print("seed %d: %d of %d pairs differ, first %d: gpu (%d, %d, %r) cpu (%d, %d, %r)" % (
                seed, len(bad), len(cs), bad[0], score[bad[0]], anchor[bad[0]], like[bad[0]], cs[bad[0]], ca[bad[0]],
                want[bad[0]]), flush=True)
        n_batches += 1
        n_pairs += len(cs)
        seed += 1
    ctx.close()
    print("%d batches, %d pairs (%d with the swallowed assertion) and %d more through window calls, %d batches refused as "
          "outside the domain, %d batches with differences; DP checker: %s" % (n_batches, n_pairs, n_assert, n_window_pairs, n_refused, n_bad,
                                                            "reference align.c" if use_ref else "restatement"))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
