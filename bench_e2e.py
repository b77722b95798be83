"""The file-to-file leg of bench.py: simulated BAM + .bai and FASTA + .fai -> VCF through platypus_amd.runner,
the second figure BASELINE.json names (Mbp/min).  One data set per benchmark configuration:

  wgs    configs[1] / configs[2]: ONE sample, 30x, 150-bp read pairs (fragments 500 +- 50) over a chr20-sized
         (64.4 Mb) chromosome-like sequence -- GC 44 %, homopolymer runs and short tandem repeats, variants at
         wgsim's rates (0.001 per base, 15 % indels, extension 0.3), qualities that mean what they say with about
         0.02 errors per base where the qualities are low, indel errors in homopolymer runs.  The whole sequence is
         simulated once; nothing is tiled.
  panel  configs[3]: 100x over the targets of a panel (600-bp targets every 2 kb, five overlapping amplicon-like
         fragments of 250-300 bp each) on homopolymer-rich sequence (a run of 8-20 every 400 bases).
  joint  configs[4]: 50 samples at 10x, their genotypes drawn from one founder set of variants.

Every run is `python -m platypus_amd.runner callVariants ...` in a process of its own, start-up included, with
--processRegionSize 1000000 --bufferSize 1000000 whatever the number of processes, so that every run writes the same
VCF; the bench compares them byte for byte (header aside: it quotes the command line)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

NO_CPU_WHY = ("no CPU figure stands beside this one: the product has no CPU path (by design: the library fails without a "
              "gfx950 device), and the reference is Python-2 / Cython-0.25 code that cannot run in this image, so the same "
              "pipeline cannot be timed on the host cores; the alignment path alone has its CPU baseline in cpu_baseline")


def _write_fasta(path, name, genome):
    G = len(genome)
    a = np.frombuffer(genome, dtype=np.uint8)
    full = G // 60 * 60
    body = np.empty((G // 60, 61), dtype=np.uint8)
    body[:, :60] = a[:full].reshape(-1, 60)
    body[:, 60] = 10
    with open(path, "wb") as f:
        f.write(b">" + name + b"\n")
        f.write(body.tobytes())
        if full < G:
            f.write(genome[full:] + b"\n")
    with open(path + ".fai", "wb") as f:
        f.write(b"%s\t%d\t%d\t60\t61\n" % (name, G, len(name) + 2))


def make_data(workload, work, mb=None, seed=5):
    """files of one configuration under `work`; returns (bam paths, fasta path, genome length, truth, text, extra)"""
    from platypus_amd import bamio, simulate
    rng = np.random.default_rng(seed)
    extra = {}
    if workload == "wgs":
        G = int((mb or 64.4) * 1e6)
        genome = simulate.chr_like_genome(rng, G)
        truth = simulate.plant_variants_rate(rng, genome)
        reads = simulate.simulate_pairs(seed, genome, truth, coverage=30, read_len=150, insert=(500, 50))
        bam = os.path.join(work, "s.bam")
        bamio.write_bam(bam, [(b"sim", G)], reads, sample="S")
        bams, n_reads = [bam], len(reads["pos"])
        what = ("one sample, 30x, 150-bp pairs over a %.1f-Mb chromosome-like sequence simulated whole (GC 44 %%, "
                "homopolymer runs and short repeats; variants 0.001 per base, 15 %% indels; quality-dependent base errors, "
                "about 0.02 per base; indel errors in homopolymer runs)" % (G / 1e6))
    elif workload == "panel":
        G = int((mb or 6.0) * 1e6)
        genome = simulate.chr_like_genome(rng, G, homopolymer_every=400, repeat_every=3000)
        truth = simulate.plant_variants_rate(rng, genome, rate=0.002, indel_frac=0.3)
        t_start = np.arange(1000, G - 1600, 2000, dtype=np.int64)
        fs, fl = [], []
        for k in range(5):                                    # five overlapping amplicons per target, 40 fragments each
            n = 40
            fs.append((t_start[:, None] + 90 * k + rng.integers(-8, 9, (len(t_start), n))).ravel())
            fl.append(rng.integers(250, 301, len(t_start) * n))
        reads = simulate.simulate_pairs(seed, genome, truth, read_len=150, fragments=(np.concatenate(fs), np.concatenate(fl)))
        bam = os.path.join(work, "s.bam")
        bamio.write_bam(bam, [(b"sim", G)], reads, sample="S")
        bams, n_reads = [bam], len(reads["pos"])
        extra["targets"] = int(len(t_start))
        extra["target_mb"] = len(t_start) * 660 / 1e6
        truth = [t for t in truth if ((t[0] - 1000) % 2000) < 660]            # the variants a target covers
        what = ("one sample, about 100x over %d targets of ~660 bp (five overlapping 250-300-bp fragments each, 150-bp pairs) "
                "every 2 kb of a %.1f-Mb homopolymer-rich sequence (a run of 8-20 every 400 bases); the region called is "
                "the whole sequence" % (len(t_start), G / 1e6))
    else:
        G = int((mb or 2.0) * 1e6)
        genome = simulate.chr_like_genome(rng, G)
        founders = simulate.plant_variants_rate(rng, genome)
        af = rng.beta(0.5, 1.5, len(founders))
        bams, n_reads = [], 0
        carried = np.zeros(len(founders), dtype=bool)
        for s in range(50):
            gt = rng.binomial(2, af)
            carried |= gt > 0
            mine = [(p, r, a, int(g)) for (p, r, a, _), g in zip(founders, gt) if g > 0]
            reads = simulate.simulate_pairs(seed + 1 + s, genome, mine, coverage=10, read_len=150, insert=(500, 50))
            bam = os.path.join(work, "s%02d.bam" % s)
            bamio.write_bam(bam, [(b"sim", G)], reads, sample="S%02d" % s)
            bams.append(bam)
            n_reads += len(reads["pos"])
        truth = [f for f, c in zip(founders, carried) if c]
        what = ("50 samples at 10x (150-bp pairs) over a %.1f-Mb chromosome-like sequence, genotypes drawn from one founder "
                "set of %d variants (allele frequencies Beta(0.5, 1.5)); called jointly with --maxVariants 8 "
                "--maxHaplotypes 256, the reference's defaults" % (G / 1e6, len(founders)))
    fa = os.path.join(work, "ref.fa")
    _write_fasta(fa, b"sim", genome)
    extra.update(reads=int(n_reads), bam_mb=sum(os.path.getsize(b) for b in bams) / 1e6)
    return bams, fa, G, truth, what, extra


def run_once(bams, fa, G, out, extra_args, timings=None):
    """one run of the command line in a process of its own: (seconds, host core-seconds of it and its children)"""
    cmd = [sys.executable, "-m", "platypus_amd.runner", "callVariants", "--bamFiles", ",".join(bams), "--refFile", fa,
           "--regions", "sim:1-%d" % (G - 1), "--output", out, "--maxReadLength", "150",
           "--processRegionSize", "1000000", "--bufferSize", "1000000"] + list(extra_args)
    env = dict(os.environ)
    if timings is not None:
        env["PLATYPUS_AMD_TIMINGS"] = "1"
    t0, c0 = time.perf_counter(), os.times()
    done = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    dt, c1 = time.perf_counter() - t0, os.times()
    if done.returncode != 0:
        raise RuntimeError("the file-to-file run failed (exit code %s): %s" % (done.returncode, done.stderr[-800:]))
    cpu = (c1.children_user - c0.children_user) + (c1.children_system - c0.children_system)
    if timings is not None:
        timings.append(done.stderr)
    return dt, cpu


def _body(path):
    with open(path, "rb") as f:
        return b"".join(l for l in f if not l.startswith(b"##"))


def _timing_report(stderr_texts):
    """from the runner's timing report (one line per worker): seconds its alignment / EM batches spent on the device,
    candidates, windows, windows that went to the device"""
    import re
    dev, seen = 0.0, False
    counts = {"candidates": 0, "windows": 0, "windows_called": 0}
    for text in stderr_texts:
        for m in re.finditer(r"\bdevice (\d+\.\d+)", text):
            dev += float(m.group(1))
            seen = True
        for m in re.finditer(r"from (\d+) candidates in (\d+) windows \((\d+) with reads", text):
            counts["candidates"] += int(m.group(1)); counts["windows"] += int(m.group(2)); counts["windows_called"] += int(m.group(3))
    k = {"pairs": 0, "cells": 0, "vote_pairs": 0, "vote_named": 0, "vote_full": 0, "anchor_ms": 0.0, "align_ms": 0.0, "setup_ms": 0.0, "em_ms": 0.0}
    for text in stderr_texts:
        for m in re.finditer(r"device batches: (\d+) pairs, (\d+) cells, (\d+) pairs anchored by votes \((\d+) \+ (\d+) of them counted by a wavefront\), anchoring ([\d.]+) ms, alignment ([\d.]+) ms, "
                             r"genotype set-up ([\d.]+) ms, EM ([\d.]+) ms", text):
            k["pairs"] += int(m.group(1)); k["cells"] += int(m.group(2))
            k["vote_pairs"] += int(m.group(3)); k["vote_named"] += int(m.group(4)); k["vote_full"] += int(m.group(5))
            for name, g in zip(("anchor_ms", "align_ms", "setup_ms", "em_ms"), m.groups()[5:]):
                k[name] += float(g)
    if k["pairs"]:
        # the DP kernel on the run's own pair stream (cells as the reference's loop executes them, 16 (L + 8) per pair)
        k["align_gcups"] = k["cells"] / (k["align_ms"] * 1e-3) / 1e9 if k["align_ms"] > 0 else None
        k["what"] = ("every device batch of the one-process run: pairs aligned, DP cells, kernel times from HIP events; one batch per "
                     "1-Mb stretch, so the launches are a few thousand to a few hundred thousand pairs each")
        counts["pair_stream"] = k
    return (dev if seen else None), counts


def leg(workload="wgs", n_gpus=1, workers=4, mb=None, reps=2):
    """the `end_to_end` object of the bench line"""
    work = tempfile.mkdtemp(prefix="platypus_amd_bench_")
    try:
        t0 = time.perf_counter()
        bams, fa, G, truth, what, extra = make_data(workload, work, mb)
        t_sim = time.perf_counter() - t0
        mbp = G / 1e6
        out_d = {"workload": workload, "what": what, "genome_mb": mbp, "simulate_seconds": t_sim,
                 "command": "python -m platypus_amd.runner callVariants --processRegionSize 1000000 --bufferSize 1000000 "
                            "--maxReadLength 150 [--nGPU N --workersPerGPU M --inflate device|host]",
                 "cpu_baseline": None, "why": NO_CPU_WHY}
        out_d.update(extra)

        def timed(tag, args, reps=reps):
            best, cpu = 1e30, None
            out = os.path.join(work, "calls_%s.vcf" % tag)
            for _ in range(reps):
                dt, c = run_once(bams, fa, G, out, args)
                if dt < best:
                    best, cpu = dt, c
            return out, {"mbp_per_min": mbp / (best / 60.0), "seconds": best, "host_core_seconds_per_mb": cpu / mbp}

        runs = {}
        ref_vcf, runs["1_process"] = timed("p1", ["--inflate", "device"])
        vcfs = [ref_vcf]
        if n_gpus == 1:
            if workload == "wgs":
                v, runs["1_process_host_inflate"] = timed("p1h", ["--inflate", "host"])
                vcfs.append(v)
                for m in (4, 8):
                    for mode in ("device", "host"):
                        v, runs["%d_processes_%s_inflate" % (m, mode)] = timed("p%d%s" % (m, mode[0]), ["--workersPerGPU", str(m), "--inflate", mode])
                        vcfs.append(v)
            else:
                v, runs["%d_processes" % workers] = timed("pm", ["--workersPerGPU", str(workers)])
                vcfs.append(v)
            top = runs["1_process"]
        else:
            tag = "%d_gpus_x_%d_processes" % (n_gpus, workers)
            v, runs[tag] = timed("gN", ["--nGPU", str(n_gpus), "--workersPerGPU", str(workers)])
            vcfs.append(v)
            top = runs[tag]
        # the device's share of a one-process run (its own timing report)
        texts = []
        run_once(bams, fa, G, os.path.join(work, "calls_t.vcf"), ["--inflate", "device"], timings=texts)
        dev, counts = _timing_report(texts)
        ref_body = _body(ref_vcf)
        identical = all(_body(v) == ref_body for v in vcfs[1:])
        called = set()
        for line in ref_body.split(b"\n"):
            if line and not line.startswith(b"#"):
                called.add(int(line.split(b"\t")[1]) - 1)
        # VCF POS of an indel is its anchor base = the package's `pos` of the left-aligned variant; of a SNP its base
        planted = {t[0] for t in truth}
        found = sum(p in called for p in planted)
        out_d.update(top)
        out_d.update({
            "n_gpus": n_gpus, "processes": 1 if n_gpus == 1 else n_gpus * workers, "runs": runs,
            "all_runs_same_vcf": bool(identical),
            "device_seconds_1_process": dev,
            "device_share_1_process": (dev / runs["1_process"]["seconds"]) if dev is not None else None,
            "vcf_records": len(called), "planted_variants": len(planted), "planted_found": found,
            "calls_not_planted": len(called - planted),
        })
        out_d.update(counts)                                   # candidates and windows of the one-process run
        return out_d
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="wgs")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--mb", type=float, default=None)
    a = ap.parse_args()
    print(json.dumps(leg(a.workload, a.gpus, a.workers, a.mb)))
