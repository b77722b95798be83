#!/usr/bin/env python
"""Benchmark of the read x haplotype alignment hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--windows NW]

A *step* is one pass of the whole hot path (7-mer anchoring of pairs on
indel-bearing haplotypes, banded DP, likelihood) over one synthetic batch that
is already resident in HBM.  The workload is the pair stream of BASELINE.json's
configs[1] (single-sample 30x WGS, 150-bp reads): windows of 4 candidate
haplotypes (~750 bp) x 32 overlapping reads, 2^22 pairs per GPU by default.
With N > 1 every rank owns its own shard of windows (different seed), nothing
is exchanged on the data path, and the value is all ranks' cells over the
slowest rank's time (weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

OPS_PER_CELL = 14          # int16 ALU operations per DP cell of the reference recurrence
NOMINAL_LANE_RATE = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes per clock x 2.4 GHz: VALU lane-instructions per second
LDS_LOOKUP_PEAK = 75e12 / 4                # 4-byte LDS reads per second, all CUs streaming
# VALU instructions one wavefront of k_align2 executes for reads of L bases: VALU_A + VALU_B * (L + 8)
VALU_A, VALU_B = 653.8, 164.367
VALU_SOURCE = ("profiles/r02_valu_insts.csv: SQ_INSTS_VALU / SQ_WAVES of k_align2 at L = 100, 150, 250 "
               "(18405.4, 26625.4, 43060.4 per wavefront; the line fits them within 2; tools/valu_count.sh)")


# Pair streams of the benchmark configurations: (haplotypes/window, reads/window, extra synth arguments, text).
# --windows counts windows of the wgs shape; the other shapes are scaled to the same number of pairs.
WORKLOADS = {
    "wgs": (4, 32, dict(indel_hap_frac=0.25),
            "wgs30x-like windows: %d windows x 4 haplotypes (~750 bp, 25%% of the variant haplotypes carry an "
            "indel) x 32 reads of %d bp = %d pairs per GPU"),
    "panel": (6, 128, dict(indel_hap_frac=1.0, homopolymers=True, n_frac=0.005),
              "100x amplicon-panel-like windows: %d windows x 6 haplotypes (~750 bp with homopolymer runs of 5-40 "
              "and 0.5%% N, every variant haplotype carries an indel) x 128 reads of %d bp = %d pairs per GPU"),
    "joint": (257, 550, dict(indel_hap_frac=0.25),
              "50-sample joint-calling-like windows: %d windows x 257 haplotypes (~750 bp) x 550 reads of %d bp "
              "= %d pairs per GPU"),
}


def pmc_traffic(n_pairs, read_len):
    """HBM bytes per k_align2 launch from the committed counter passes (separate rocprofv3 --pmc runs of
    this same command; counters cannot be read from inside the process).  FETCH_SIZE and WRITE_SIZE are
    in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so it is doubled (with the
    doubling it equals the bytes of the resident read and haplotype arrays, which the kernel reads once).
    Only valid for the workload the passes were collected on."""
    import csv
    import glob
    if n_pairs != (1 << 22) or read_len != 150:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")))
    if not files:
        return None, None
    val = {}
    with open(files[-1], newline="") as f:
        for row in csv.DictReader(f):
            if row["kernel"] == "pg::k_align2":
                val[row["counter"]] = float(row["mean_per_dispatch"])
    if "FETCH_SIZE" not in val or "WRITE_SIZE" not in val:
        return None, None
    return (2.0 * val["FETCH_SIZE"] + val["WRITE_SIZE"]) * 1024.0, os.path.relpath(files[-1], ROOT)


def end_to_end_leg(kb=2000, coverage=30, copies=32):
    """Simulate kb kilobases at the given coverage (one planted variant per kb, 150-bp read pairs from 500 +- 50 bp
    fragments with true CIGARs),
    lay `copies` repeats of them end to end as one reference (simulating is far slower than calling),
    write BAM + .bai and FASTA + .fai, call them to a VCF with platypus_amd.runner, time the call."""
    import multiprocessing
    import shutil
    import tempfile
    from platypus_amd import bamio, runner, simulate
    rng = np.random.default_rng(5)
    G = kb * 1000
    work = tempfile.mkdtemp(prefix="platypus_amd_bench_")
    try:
        genome = simulate.random_genome(rng, G)
        truth = simulate.plant_variants(rng, genome, G // 1000, spacing=600)
        reads = simulate.simulate_reads(rng, genome, truth, coverage=coverage, read_len=150, paired=True, insert=(500, 50))
        genome, truth, reads = simulate.tile(genome, truth, reads, copies)
        G *= copies
        fa = os.path.join(work, "ref.fa")
        with open(fa, "wb") as f:
            f.write(b">sim\n" + b"\n".join(genome[i:i + 60] for i in range(0, G, 60)) + b"\n")
        with open(fa + ".fai", "wb") as f:
            f.write(b"sim\t%d\t5\t60\t61\n" % G)
        bam = os.path.join(work, "s.bam")
        bamio.write_bam(bam, [(b"sim", G)], reads, sample="S")
        out = os.path.join(work, "calls.vcf")
        base = ["callVariants", "--bamFiles", bam, "--refFile", fa, "--regions", "sim:1-%d" % (G - 1), "--output", out,
                "--maxReadLength", "150"]
        best, cpu = {}, {}
        for mode in ("device", "host"):                      # where the BGZF blocks are inflated (runner --inflate)
            for workers in (1, 4, 8):                        # host processes sharing the one GPU
                extra = ["--inflate", mode]
                if workers > 1:
                    extra += ["--workersPerGPU", str(workers), "--processRegionSize", str(G // workers)]
                for _ in range(2):
                    # in a process of its own, start-up included, as a user would run it: inside this one the caller
                    # shares the interpreter with torch and a second HIP runtime, which costs it a sixth of its rate
                    t0, c0 = time.perf_counter(), os.times()
                    child = multiprocessing.get_context("spawn").Process(target=runner.main, args=(base + extra,))
                    child.start()
                    child.join()
                    if child.exitcode != 0:
                        raise RuntimeError("the file-to-file run failed (exit code %s)" % child.exitcode)
                    dt, c1 = time.perf_counter() - t0, os.times()
                    if dt < best.get((mode, workers), 1e30):     # host core-seconds of the run: this process, its threads, its children
                        best[mode, workers] = dt
                        cpu[mode, workers] = (c1.user - c0.user) + (c1.system - c0.system) + \
                                             (c1.children_user - c0.children_user) + (c1.children_system - c0.children_system)
        by_mode = {}
        for mode in ("device", "host"):
            d = {"mbp_per_min": G / 1e6 / (best[mode, 1] / 60.0), "seconds": best[mode, 1],
                 "host_core_seconds_per_mb": cpu[mode, 1] / (G / 1e6)}
            for workers in (4, 8):
                d["mbp_per_min_%d_host_processes" % workers] = G / 1e6 / (best[mode, workers] / 60.0)
                d["seconds_%d_host_processes" % workers] = best[mode, workers]
                d["host_core_seconds_per_mb_%d_host_processes" % workers] = cpu[mode, workers] / (G / 1e6)
            by_mode[mode] = d
        called = set()
        for line in open(out):
            if not line.startswith("#"):
                called.add(int(line.split("\t")[1]) - 1)
        # top level: what the runner does by itself (--inflate auto): one process inflates the BGZF blocks on the GPU,
        # several processes that share a GPU leave it to the host library
        out_d = {k: by_mode["host" if k.endswith("8_host_processes") else "device"][k] for k in by_mode["device"]}
        out_d["by_inflate_mode"] = by_mode
        out_d.update({
                "genome_kb": kb * copies, "simulated_kb": kb, "repeats": copies, "coverage": coverage,
                "read_length": 150, "paired": True,
                "reads": int(len(reads["pos"])), "bam_mb": os.path.getsize(bam) / 1e6, "vcf_records": len(called),
                "planted_variants": len(truth), "planted_found": sum(p in called for p, r, a, g in truth),
                "processes": 1, "what": "simulated BAM + .bai and FASTA + .fai -> VCF, platypus_amd.runner in a process of its own, "
                                        "start-up included; BGZF blocks inflated on the GPU with one or four host processes, by the host "
                                        "library with eight (runner --inflate auto); by_inflate_mode: every combination"})
        return out_d
    finally:
        shutil.rmtree(work, ignore_errors=True)


def max_over_ranks(value, dist, device):
    """MAX all-reduce of one float (the elapsed time): the only cross-rank operation of the run"""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=32768, help="windows per GPU (128 pairs each)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-to-file end-to-end leg")
    ap.add_argument("--no-stream", action="store_true", help="skip the PCIe-inclusive (streamed batches) leg")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="kernel-choice override passed to pg_set_option (measurement sweeps)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="wgs",
                    help="pair stream to synthesise; wgs is the configuration the metric is quoted on")
    args = ap.parse_args()

    from platypus_amd import shard
    rank, world, local_rank = shard.rank_info()
    import torch                                           # (before the caller's library is loaded: torch brings a HIP runtime of its own)
    # ---- file to file (simulated BAM + FASTA -> VCF through the region caller), not part of the metric:
    # the second figure BASELINE.json names, Mbp/min, for ONE process / one GPU.  It runs first, while this process
    # has not opened the GPU: the caller runs in processes of its own, and a parent that holds streams and memory on
    # the same device takes a tenth off their rate ------------------------------------------------------
    e2e = None
    if world == 1 and not args.no_e2e and args.workload == "wgs":
        try:
            e2e = end_to_end_leg()
        except Exception as e:                             # reported, never hidden
            e2e = {"error": "%s: %s" % (type(e).__name__, e)}

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU; on a box with fewer GPUs than ranks (rehearsals) ranks share devices
        device = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(device)
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")   # the process group only carries a barrier and one MAX
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", device))
                sync_dev = torch.device("cuda", device)
                dist.barrier()
            except Exception:                                 # control plane only: gloo will do
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
        if backend != "nccl":
            dist.init_process_group("gloo", rank=rank, world_size=world)
            sync_dev = torch.device("cpu")
    else:
        device = 0
        torch.cuda.set_device(0)

    from platypus_amd import native, synth
    L = args.read_len
    H, R, extra, text = WORKLOADS[args.workload]
    n_windows = max(1, args.windows * 128 // (H * R))
    batch = synth.make_batch(n_windows=n_windows, haps_per_window=H, reads_per_window=R,
                             read_len=L, hap_len=750, seed=synth.SEED + rank, **extra)
    n_pairs = len(batch["pair_hap"])
    ctx = native.AlignContext(device)
    for opt in args.option:
        ctx.set_option(*opt.split("=", 1))
    ctx.upload_batch(batch)                                    # inputs resident in HBM from here on

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        ctx.launch()
    barrier()
    t0 = time.perf_counter()
    k_ms, v_ms = [], []
    for _ in range(args.steps):
        ctx.launch()
        # event times of this step's kernels, read after the step has drained;
        # the read is part of the timed region (it is what a caller would wait for)
        ctx.sync()
        k_ms.append(ctx.kernel_ms(native.PG_K_ALIGN))
        v_ms.append(ctx.kernel_ms(native.PG_K_ANCHOR))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = max_over_ranks(dt, dist, sync_dev)

    cells_per_pair = 16 * (L + 8)
    total_cells = float(n_pairs) * cells_per_pair * world * args.steps
    gcups = total_cells / dt / 1e9
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (k_align2): share of the VALU issue rate ------------------
    # The kernel is bound by VALU instruction issue (integer DP; no MFMA, 10 % of the HBM rate).
    #   achieved = VALU instructions the kernel EXECUTES per launch x 64 lanes / its average duration
    #   peak     = 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz (a wavefront's VALU instruction issues
    #              over 2 cycles), and beside it the rate a probe of the kernel's own instruction forms
    #              (v_add_u16 / v_min_u16) reaches in this process
    # Executed instructions per wavefront are a property of the shipped binary: VALU_A + VALU_B x (L + 8),
    # from SQ_INSTS_VALU / SQ_WAVES of k_align2 at L = 100, 150, 250 (profiles/, see VALU_SOURCE).
    k_avg = float(np.mean(k_ms)) * 1e-3
    score, anchor, like = ctx.fetch()
    n_waves = (n_pairs + 63) // 64
    insts_per_wave = VALU_A + VALU_B * (L + 8)
    lane_instr = float(n_waves) * insts_per_wave * 64.0
    achieved = lane_instr / k_avg
    probe = ctx.valu_vop2_rate()
    pk_peak = ctx.valu_int16_peak()
    algo_ops = float(n_pairs) * cells_per_pair * OPS_PER_CELL       # int16 operations of the reference recurrence
    traffic, traffic_src = pmc_traffic(n_pairs, L) if args.workload == "wgs" else (None, None)
    roofline = {
        "bound": "valu_issue",
        "kernel": "k_align2",
        "achieved": achieved / 1e12, "peak": NOMINAL_LANE_RATE / 1e12, "unit": "T lane-instructions/s",
        "frac": achieved / NOMINAL_LANE_RATE,
        "peak_source": "256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz",
        "probe_peak": probe / 1e12, "frac_of_probe": achieved / probe,
        "probe_source": "v_add_u16/v_min_u16 issue-rate probe run in this process",
        "valu_insts_per_wave": insts_per_wave, "valu_insts_source": VALU_SOURCE,
        "kernel_ms": k_avg * 1e3, "anchor_kernel_ms": float(np.mean(v_ms)),
        "kernel_gcups": float(n_pairs) * cells_per_pair / k_avg / 1e9,
        # the survey's convention, kept for comparison: 14 int16 operations per cell of the reference
        # recurrence over the packed-int16 probe rate (the kernel needs fewer instructions per cell than
        # that, so this is not a fraction of anything the hardware limits)
        "algorithmic_ops_per_s": algo_ops / k_avg / 1e12, "packed_int16_probe": pk_peak / 1e12,
        "algorithmic_frac": algo_ops / k_avg / pk_peak,
        "hbm_algorithmic_gbps": float(n_pairs) * (2 * L + 24) / k_avg / 1e9,
        "hbm_peak_gbps": 8000.0,
        "hbm_algorithmic_bytes": float(n_pairs) * (2 * L + 24),
        # HBM bytes per launch from the counter passes of a committed profile (not measured in this run)
        "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
    }
    # the anchoring kernel in front of it (k_vote4) is bound by LDS look-up latency / issue: one table
    # look-up per 7-mer of every read of a pair on an indel-bearing haplotype
    n_vote_pairs = int((batch["hap_varlen"][batch["pair_hap"]] != 0).sum())
    v_avg = float(np.mean(v_ms)) * 1e-3
    lookups = float(n_vote_pairs) * max(0, L - 7)
    roofline_anchor = {
        "bound": "lds_lookup", "kernel": "k_vote4", "kernel_ms": v_avg * 1e3, "pairs": n_vote_pairs,
        "achieved": lookups / v_avg / 1e12 if v_avg > 0 else None, "peak": LDS_LOOKUP_PEAK / 1e12,
        "unit": "T look-ups/s", "frac": lookups / v_avg / LDS_LOOKUP_PEAK if v_avg > 0 else None,
        "peak_source": "ds_read_b32 aggregate of the chip, 75 TB/s / 4 B",
    }

    # ---- the consumer of the likelihood matrix (genotype likelihoods + EM per window), not part of
    # the metric: kernel times of one pass over the same batch; one individual per window, except the
    # joint-calling stream whose 550 reads per window are 50 individuals' 11 reads each ----------------
    caller = None
    n_inds = 50 if args.workload == "joint" else 1
    try:
        ctx.launch()
        ctx.upload_windows([H] * n_windows, [n_inds] * n_windows, np.arange(n_windows, dtype=np.int64) * H * R,
                           np.tile(np.arange(n_inds + 1, dtype=np.int32) * (R // n_inds), n_windows))
        ctx.caller_launch(100)
        ctx.sync()
        res = ctx.caller_fetch(genotype_post=False)
        caller = {"windows": n_windows, "haplotypes": H, "individuals": n_inds,
                  "setup_ms": ctx.kernel_ms(native.PG_K_CALLER_SETUP), "em_ms": ctx.kernel_ms(native.PG_K_CALLER_EM),
                  "em_iterations_mean": float(res["iters"].mean()), "em_iterations_max": int(res["iters"].max())}
        if args.workload == "joint":
            # The EM stage streams every individual's genotype-likelihood matrix once per iteration of its window:
            # algorithmic bytes = iterations x individuals x H (H + 1) / 2 doubles (the matrix is symmetric), over
            # the whole stage's time (steps, updates and the final calls).
            em_bytes = float(res["iters"].sum()) * n_inds * (H * (H + 1) // 2) * 8.0
            em_s = caller["em_ms"] * 1e-3
            caller["roofline"] = {"bound": "hbm", "kernel": "k_em_step_sym", "achieved": em_bytes / em_s / 1e9, "peak": 8000.0,
                                  "unit": "GB/s", "frac": em_bytes / em_s / 8e12, "algorithmic_bytes": em_bytes,
                                  "traffic": None}
    except native.PlatGpuError as e:                       # reported, never hidden
        caller = {"error": str(e)}

    # ---- PCIe-inclusive rate, never `value`: the same batch streamed from page-locked host memory through the
    # ABI (everything uploaded and fetched per batch, four contexts fed by a host thread each), beside the link's own rate ----
    streamed = None
    if world == 1 and not args.no_stream and args.workload == "wgs":
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import time_stream
            streamed = time_stream.measure(W=n_windows, N=48, pinned=True, device=device, batch=batch, threads=4)
            streamed["link_gbps"] = time_stream.link_rate()
            streamed["what"] = ("haplotypes, reads and window blocks uploaded, kernels run, score / anchor / likelihood "
                                "fetched for every batch; bytes_up / link_gbps is the least a batch can take")
        except Exception as e:                             # reported, never hidden
            streamed = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- CPU baseline: the reference's own compiled routine on the host cores ---------
    cpu = None
    if world == 1 and not args.no_cpu:                       # rank 0 at N = 1 only: the other ranks would wait
        from oracle import pyoracle as po
        cores = min(16, os.cpu_count() or 1)
        kind = "reference" if po.have_ref() else "port"
        sample = min(n_pairs, 1 << 22)
        ov = synth.oracle_view(batch)
        reps = 3                                              # about 25 s of CPU work in total
        t1 = time.perf_counter()
        for _ in range(reps):
            cs, ca = po.run_pairs(ov, use_ref=(kind == "reference"), threads=cores, hi=sample)
        cdt = (time.perf_counter() - t1) / reps
        same = bool(np.array_equal(cs, score[:sample]) and np.array_equal(ca, anchor[:sample]))
        cpu = {"value": sample * cells_per_pair / cdt / 1e9, "unit": "GCUPS", "cores": cores,
               "kind": kind, "pairs_per_s": sample / cdt,
               "sample": "first %d pairs of the GPU batch (same arrays), %d passes of %.2f s wall each, "
                         "per-pair 7-mer table rebuild as in the reference; DP by %s" %
                         (sample, reps, cdt, "the reference's compiled align.c" if kind == "reference"
                          else "the repo's C restatement"),
               "gpu_equals_cpu_on_sample": same}

    out = {
        "metric": "read_x_haplotype_align_gcups", "value": gcups, "unit": "GCUPS",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int16", "data": "synthetic",
        "config": {"workload": (text % (n_windows, L, n_pairs)) +
                               "; whole hot path (anchoring + banded DP + likelihood)",
                   "workload_name": args.workload,
                   "pairs_per_gpu": n_pairs, "read_len": L, "cells_per_pair": cells_per_pair,
                   "parallelism": "region shards, %d process(es), no collective" % world},
        "pairs_per_s": float(n_pairs) * world * args.steps / dt,
        "roofline": roofline,
        "roofline_anchor": roofline_anchor,
        "cpu_baseline": cpu,
        "caller_stage": caller,
        "pcie_inclusive": streamed,
        "pcie_inclusive_gcups": streamed.get("gcups") if streamed else None,
        "end_to_end": e2e,
    }
    print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
