#!/usr/bin/env python
"""Benchmark of the read x haplotype alignment hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--windows NW] [--workload wgs|panel|joint]

A *step* is one pass of the whole hot path (7-mer anchoring of pairs on
indel-bearing haplotypes, banded DP, likelihood) over one synthetic batch that
is already resident in HBM.  The workload is the pair stream of BASELINE.json's
configs[1] (single-sample 30x WGS, 150-bp reads): windows of 4 candidate
haplotypes (~750 bp) x 32 overlapping reads, 2^22 pairs per GPU by default.

Ranks.  The path shards by independent genomic regions, so N GPUs are N
processes that exchange nothing on the data path (weak scaling): every rank
owns its own shard of windows (seed + rank) on its own device, the value is all
ranks' cells over the slowest rank's time.  `--gpus N` started plainly launches
the N ranks itself: the parent touches no GPU, starts N fresh interpreters (one
per device, rank k on device k mod the number of devices), lines them up on a
barrier before and after the K timed steps and prints ONE JSON line.  Started
under `python -m torch.distributed.run` (RANK / WORLD_SIZE in the environment)
the given ranks are used instead and the barrier and the MAX over ranks go
through a gloo process group; nothing of torch touches the GPU in either mode,
and without a launcher torch is not imported at all.

The second figure BASELINE.json names, Mbp/min from files to a VCF, is the
`end_to_end` object: a simulated 30x BAM + FASTA called by platypus_amd.runner,
at N > 1 with `--nGPU N`, the merged VCF compared byte for byte with the
one-process run's.
"""
import argparse
import json
import os
import sys
import time

# nothing here calls a BLAS routine: without its thread pool the runner's workers can be forked (see runner.py)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

OPS_PER_CELL = 14          # int16 ALU operations per DP cell of the reference recurrence
NOMINAL_LANE_RATE = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes per clock x 2.4 GHz: VALU lane-instructions per second
# VALU instructions one wavefront of 64 pairs of k_vote5 executes (SQ_INSTS_VALU over the wavefronts that hold pairs)
VOTE_VALU_PER_WAVE = 1740.0
VOTE_VALU_SOURCE = ("profiles/r03_pmc_summary.csv (wgs stream: 21.79 M per launch of 782 336 pairs = 1782 per wavefront) and "
                    "profiles/r03_pmc_vote_panel.csv (92.46 M per 3 495 040 pairs = 1693)")
# VALU instructions one wavefront of k_align2 executes for reads of L bases: VALU_A + VALU_B * (L + 8)
VALU_A, VALU_B = 653.8, 164.367
VALU_SOURCE = ("profiles/r03_valu_insts.csv (the same counts as r02: the kernel did not change): SQ_INSTS_VALU / SQ_WAVES of k_align2 at L = 100, 150, 250 "
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


# ------------------------------------------------------------------------------------------------------
# the timed loop of one rank
def make_batch(args, rank):
    from platypus_amd import synth
    H, R, extra, text = WORKLOADS[args.workload]
    n_windows = max(1, args.windows * 128 // (H * R))
    batch = synth.make_batch(n_windows=n_windows, haps_per_window=H, reads_per_window=R,
                             read_len=args.read_len, hap_len=750, seed=synth.SEED + rank, **extra)
    return batch, n_windows


def rank_steps(ctx, args, barrier):
    """W untimed steps, then exactly K timed steps between two barriers, each with the device drained (pg_sync is a
    hipStreamSynchronize of the context's stream, the only stream the path uses).  Returns this rank's seconds and
    the per-step event times of the two kernels."""
    from platypus_amd import native
    for _ in range(args.warmup):
        ctx.launch()
    ctx.sync()
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
    ctx.sync()
    barrier()
    return time.perf_counter() - t0, k_ms, v_ms


def _child(rank, world, args, barrier, conn):
    """one rank of a self-launched run: a fresh interpreter that opens its own device and nothing else"""
    try:
        from platypus_amd import native
        ndev = native.device_count()
        if ndev < 1:
            raise RuntimeError("no gfx950 device")
        device = rank % ndev
        batch, n_windows = make_batch(args, rank)
        ctx = native.AlignContext(device)
        for opt in args.option:
            ctx.set_option(*opt.split("=", 1))
        ctx.upload_batch(batch)                                # inputs resident in HBM from here on
        dt, k_ms, v_ms = rank_steps(ctx, args, lambda: barrier.wait(600))
        n_vote = int((batch["hap_varlen"][batch["pair_hap"]] != 0).sum())
        conn.send(dict(rank=rank, device=device, devices=ndev, dt=dt, k_ms=k_ms, v_ms=v_ms,
                       n_pairs=len(batch["pair_hap"]), n_windows=n_windows, n_vote_pairs=n_vote))
        ctx.close()
    except BaseException as e:                                 # reported to the parent, never hidden
        try:
            barrier.abort()
        except Exception:
            pass
        conn.send(dict(rank=rank, error="%s: %s" % (type(e).__name__, e)))
    finally:
        conn.close()


def launch_ranks(args, world):
    """--gpus N without a launcher: N fresh interpreters, one per device; returns their reports in rank order.
    The parent makes no GPU call."""
    import multiprocessing
    mp = multiprocessing.get_context("spawn")
    barrier = mp.Barrier(world)
    procs, pipes = [], []
    for r in range(world):
        here, there = mp.Pipe(duplex=False)
        p = mp.Process(target=_child, args=(r, world, args, barrier, there))
        p.start()
        there.close()
        procs.append(p)
        pipes.append(here)
    reports = []
    for r, (p, c) in enumerate(zip(procs, pipes)):
        try:
            reports.append(c.recv())
        except EOFError:
            reports.append(dict(rank=r, error="rank %d ended without a report" % r))
    for p in procs:
        p.join()
    bad = [rep for rep in reports if "error" in rep] + \
          [dict(rank=r, error="exit code %s" % p.exitcode) for r, p in enumerate(procs) if p.exitcode != 0]
    if bad:
        raise RuntimeError("; ".join("rank %d: %s" % (b["rank"], b["error"]) for b in bad))
    return reports


def slowest(value, dist):
    """MAX over ranks of one float (the elapsed time): with the barriers, the only cross-rank operation of a run
    under a launcher.  Host objects through gloo; no device tensor."""
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, float(value))
    return max(gathered)


# ------------------------------------------------------------------------------------------------------
def roofline_of(n_pairs, L, k_ms, v_ms, n_vote_pairs, workload, probe=None, pk_peak=None):
    """Roofline of the dominant kernel (k_align2): share of the VALU issue rate.
    The kernel is bound by VALU instruction issue (integer DP; no MFMA, 10 % of the HBM rate).
      achieved = VALU instructions the kernel EXECUTES per launch x 64 lanes / its average duration
      peak     = 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz (a wavefront's VALU instruction issues
                 over 2 cycles), and beside it the rate a probe of the kernel's own instruction forms
                 (v_add_u16 / v_min_u16) reaches in this process
    Executed instructions per wavefront are a property of the shipped binary: VALU_A + VALU_B x (L + 8),
    from SQ_INSTS_VALU / SQ_WAVES of k_align2 at L = 100, 150, 250 (profiles/, see VALU_SOURCE)."""
    cells_per_pair = 16 * (L + 8)
    k_avg = float(np.mean(k_ms)) * 1e-3
    n_waves = (n_pairs + 63) // 64
    insts_per_wave = VALU_A + VALU_B * (L + 8)
    achieved = float(n_waves) * insts_per_wave * 64.0 / k_avg
    algo_ops = float(n_pairs) * cells_per_pair * OPS_PER_CELL       # int16 operations of the reference recurrence
    traffic, traffic_src = pmc_traffic(n_pairs, L) if workload == "wgs" else (None, None)
    roofline = {
        "bound": "valu_issue",
        "kernel": "k_align2",
        "achieved": achieved / 1e12, "peak": NOMINAL_LANE_RATE / 1e12, "unit": "T lane-instructions/s",
        "frac": achieved / NOMINAL_LANE_RATE,
        "peak_source": "256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz",
        "valu_insts_per_wave": insts_per_wave, "valu_insts_source": VALU_SOURCE,
        "kernel_ms": k_avg * 1e3, "anchor_kernel_ms": float(np.mean(v_ms)),
        "kernel_gcups": float(n_pairs) * cells_per_pair / k_avg / 1e9,
        "hbm_algorithmic_gbps": float(n_pairs) * (2 * L + 24) / k_avg / 1e9,
        "hbm_peak_gbps": 8000.0,
        "hbm_algorithmic_bytes": float(n_pairs) * (2 * L + 24),
        # HBM bytes per launch from the counter passes of a committed profile (not measured in this run)
        "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
    }
    if probe:
        roofline.update({"probe_peak": probe / 1e12, "frac_of_probe": achieved / probe,
                         "probe_source": "v_add_u16/v_min_u16 issue-rate probe run in this process"})
    if pk_peak:
        # the survey's convention, kept for comparison: 14 int16 operations per cell of the reference
        # recurrence over the packed-int16 probe rate (the kernel needs fewer instructions per cell than
        # that, so this is not a fraction of anything the hardware limits)
        roofline.update({"algorithmic_ops_per_s": algo_ops / k_avg / 1e12, "packed_int16_probe": pk_peak / 1e12,
                         "algorithmic_frac": algo_ops / k_avg / pk_peak})
    # the anchoring kernel in front of it (k_vote5: exact bit-parallel vote counts, one lane per pair) executes about
    # VOTE_VALU_PER_WAVE VALU instructions per wavefront of 64 pairs; its share of the issue rate is low because one
    # launch of many short workgroups spends as long setting up (rows of four haplotypes into LDS, a barrier) as counting
    v_avg = float(np.mean(v_ms)) * 1e-3
    vote_rate = float((n_vote_pairs + 63) // 64) * VOTE_VALU_PER_WAVE * 64.0 / v_avg if v_avg > 0 else None
    roofline_anchor = {
        "bound": "valu_issue", "kernel": "k_vote5", "kernel_ms": v_avg * 1e3, "pairs": n_vote_pairs,
        "achieved": vote_rate / 1e12 if vote_rate else None, "peak": NOMINAL_LANE_RATE / 1e12,
        "unit": "T lane-instructions/s", "frac": vote_rate / NOMINAL_LANE_RATE if vote_rate else None,
        "valu_insts_per_wave": VOTE_VALU_PER_WAVE, "valu_insts_source": VOTE_VALU_SOURCE,
    }
    return roofline, roofline_anchor


def caller_stage(ctx, args, n_windows):
    """The consumer of the likelihood matrix (genotype likelihoods + EM per window), not part of the metric: kernel
    times of one pass over the same batch; one individual per window, except the joint-calling stream whose 550
    reads per window are 50 individuals' 11 reads each."""
    from platypus_amd import native
    H, R, _, _ = WORKLOADS[args.workload]
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
        return caller
    except native.PlatGpuError as e:                       # reported, never hidden
        return {"error": str(e)}


def cpu_baseline(batch, n_pairs, L, score, anchor):
    """The reference's own compiled routine (oracle/_ref) on the host cores, over the same arrays; its results are
    compared with the GPU's."""
    from oracle import pyoracle as po
    from platypus_amd import synth
    cells_per_pair = 16 * (L + 8)
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
    return {"value": sample * cells_per_pair / cdt / 1e9, "unit": "GCUPS", "cores": cores,
            "kind": kind, "pairs_per_s": sample / cdt,
            "sample": "first %d pairs of the GPU batch (same arrays), %d passes of %.2f s wall each, "
                      "per-pair 7-mer table rebuild as in the reference; DP by %s" %
                      (sample, reps, cdt, "the reference's compiled align.c" if kind == "reference"
                       else "the repo's C restatement"),
            "gpu_equals_cpu_on_sample": same}


# ------------------------------------------------------------------------------------------------------
def join_group(rank, world):
    """The control plane of a run started by torch.distributed.run: a gloo group on the host (a barrier and one MAX);
    nothing of torch touches the GPU.  gloo announces its connections on the C++ standard output; this program's
    standard output carries one JSON line and nothing else, so descriptor 1 points at standard error while the group
    is being set up."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
        dist.barrier()
    finally:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    return dist


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
    ap.add_argument("--stream-threads", type=int, default=6, help="contexts (a host thread each) of the PCIe-inclusive leg")
    ap.add_argument("--e2e-mb", type=float, default=None, help="length of the simulated reference of the end-to-end leg, Mb")
    ap.add_argument("--e2e-workers", type=int, default=4, help="host processes per GPU of the multi-process end-to-end runs")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch the ranks, line them up on the barriers and report, without opening a GPU (tests)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="kernel-choice override passed to pg_set_option (measurement sweeps)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="wgs",
                    help="pair stream to synthesise; wgs is the configuration the metric is quoted on")
    args = ap.parse_args()

    from platypus_amd import shard
    rank, env_world, local_rank = shard.rank_info()
    launched = env_world > 1                     # started by torch.distributed.run: the ranks are given
    world = env_world if launched else max(1, args.gpus)
    if launched and args.gpus not in (1, env_world):
        sys.stderr.write("bench.py: --gpus %d under a launcher of %d ranks: the launcher's count is used\n" % (args.gpus, env_world))
    if args.dry_run:
        return dry_run(args, world, launched, rank)

    dist = None
    if launched:
        dist = join_group(rank, world)

    # ---- file to file (simulated BAM + FASTA -> VCF through the region caller), not part of the metric:
    # the second figure BASELINE.json names, Mbp/min.  It runs first, while no process of this run has
    # opened a GPU: the caller runs in processes of its own, and a parent that holds streams and memory on
    # the same device takes a tenth off their rate ---------------------------------------------------------
    e2e = None
    if rank == 0 and not args.no_e2e:
        try:
            import bench_e2e
            e2e = bench_e2e.leg(args.workload, n_gpus=world, workers=args.e2e_workers, mb=args.e2e_mb)
        except Exception as e:                             # reported, never hidden
            e2e = {"error": "%s: %s" % (type(e).__name__, e)}
    if dist is not None:
        dist.barrier()

    L = args.read_len
    H, R, extra, text = WORKLOADS[args.workload]
    cells_per_pair = 16 * (L + 8)
    extras = {}
    if launched or world == 1:
        from platypus_amd import native
        ndev = native.device_count()
        device = local_rank % max(1, ndev)
        batch, n_windows = make_batch(args, rank)
        n_pairs = len(batch["pair_hap"])
        ctx = native.AlignContext(device)
        for opt in args.option:
            ctx.set_option(*opt.split("=", 1))
        ctx.upload_batch(batch)                                    # inputs resident in HBM from here on
        dt_mine, k_ms, v_ms = rank_steps(ctx, args, (lambda: None) if dist is None else dist.barrier)
        dts = [dt_mine]
        if dist is not None:
            gathered = [None] * world
            dist.all_gather_object(gathered, dict(rank=rank, device=device, dt=dt_mine, k_ms=float(np.mean(k_ms))))
            dts = [g["dt"] for g in gathered]
            if rank != 0:
                ctx.close()
                dist.barrier()
                dist.destroy_process_group()
                return
            ranks = [dict(rank=g["rank"], device=g["device"], seconds=g["dt"], kernel_ms=g["k_ms"]) for g in gathered]
        else:
            ranks = [dict(rank=0, device=device, seconds=dt_mine, kernel_ms=float(np.mean(k_ms)))]
        n_vote_pairs = int((batch["hap_varlen"][batch["pair_hap"]] != 0).sum())
        score, anchor, like = ctx.fetch()
        n_anchored, n_named, n_full = ctx.anchor_stats()
        probe = ctx.valu_vop2_rate()
        pk_peak = ctx.valu_int16_peak()
        roofline, roofline_anchor = roofline_of(n_pairs, L, k_ms, v_ms, n_vote_pairs, args.workload, probe, pk_peak)
        roofline_anchor["pairs_counted_over_named_positions"] = n_named
        roofline_anchor["pairs_counted_over_every_position"] = n_full
        extras["caller_stage"] = caller_stage(ctx, args, n_windows)
        # ---- PCIe-inclusive rate, never `value`: the same batch streamed from page-locked host memory through the
        # ABI (everything uploaded and fetched per batch, six contexts fed by a host thread each), beside the link's own rate
        streamed = None
        if world == 1 and not args.no_stream and args.workload == "wgs":
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import time_stream
                streamed = time_stream.measure(W=n_windows, N=48, pinned=True, device=device, batch=batch, threads=args.stream_threads,
                                               built=True, nibbles=True)
                streamed["link_gbps"] = time_stream.link_rate()
                streamed["what"] = ("haplotypes (as reference stretches and variant lists, put together on the device), reads (bases two "
                                    "per byte, as a BAM record holds them) and window blocks uploaded, kernels run, score / anchor / "
                                    "likelihood fetched for every batch; bytes_up / link_gbps is the least a batch can take")
                # the same stream with the larger forms of the inputs: every way leaves the same results on the host
                same = True
                for key, kw in (("with_read_bases_as_characters", dict(built=True)), ("with_whole_haplotype_sequences_too", dict())):
                    other = time_stream.measure(W=n_windows, N=48, pinned=True, device=device, batch=batch, threads=args.stream_threads, **kw)
                    same = same and other["checksum"] == streamed["checksum"]
                    streamed[key] = {k: other[k] for k in ("ms_per_batch", "gcups", "bytes_up", "upload_gbps")}
                streamed.pop("checksum")
                streamed["results_equal_every_way"] = bool(same)
            except Exception as e:                             # reported, never hidden
                streamed = {"error": "%s: %s" % (type(e).__name__, e)}
        extras["pcie_inclusive"] = streamed
        extras["pcie_inclusive_gcups"] = streamed.get("gcups") if streamed else None
        # ---- CPU baseline: rank 0 at N = 1 only (the other ranks would wait) ---------------------------
        cpu = None
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline(batch, n_pairs, L, score, anchor)
        ctx.close()
    else:
        reports = launch_ranks(args, world)
        dts = [r["dt"] for r in reports]
        r0 = reports[0]
        n_pairs, n_windows = r0["n_pairs"], r0["n_windows"]
        roofline, roofline_anchor = roofline_of(n_pairs, L, r0["k_ms"], r0["v_ms"], r0["n_vote_pairs"], args.workload)
        ranks = [dict(rank=r["rank"], device=r["device"], seconds=r["dt"], kernel_ms=float(np.mean(r["k_ms"]))) for r in reports]
        extras["devices_visible"] = r0["devices"]
        cpu = None

    dt = max(dts)                                                  # the slowest rank's time
    total_cells = float(n_pairs) * cells_per_pair * world * args.steps
    out = {
        "metric": "read_x_haplotype_align_gcups", "value": total_cells / dt / 1e9, "unit": "GCUPS",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int16", "data": "synthetic",
        "config": {"workload": (text % (n_windows, L, n_pairs)) +
                               "; whole hot path (anchoring + banded DP + likelihood)",
                   "workload_name": args.workload,
                   "pairs_per_gpu": n_pairs, "read_len": L, "cells_per_pair": cells_per_pair,
                   "parallelism": "region shards, %d process(es) (%s), one per GPU, no collective" % (
                       world, "given by the launcher" if launched else "started by this command")},
        "pairs_per_s": float(n_pairs) * world * args.steps / dt,
        "ranks": ranks,
        "roofline": roofline,
        "roofline_anchor": roofline_anchor,
        "cpu_baseline": cpu,
    }
    if cpu is None:
        out["cpu_baseline_why"] = ("measured on rank 0 at N = 1 only" if world > 1 else "skipped with --no-cpu")
    out.update(extras)
    out["end_to_end"] = e2e
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------
def _dry_child(rank, world, barrier, conn):
    barrier.wait(60)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))                             # a stand-in for the K steps: rank r takes 50 (r + 1) ms
    barrier.wait(60)
    conn.send(dict(rank=rank, dt=time.perf_counter() - t0, pid=os.getpid()))
    conn.close()


def dry_run(args, world, launched, rank):
    """The launcher without a GPU: starts the ranks the way the real run does, lines them up on the same barriers
    and prints one line with n_gpus = the ranks that ran (what the CPU tests call)."""
    if launched:                                 # under torch.distributed.run: the group, a barrier, one line from rank 0
        dist = join_group(rank, world)
        t0 = time.perf_counter()
        dist.barrier()
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(rank=rank, dt=time.perf_counter() - t0, pid=os.getpid()))
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "read_x_haplotype_align_gcups", "dry_run": True, "n_gpus": world,
                              "seconds": max(g["dt"] for g in gathered),
                              "ranks": [dict(rank=g["rank"], seconds=g["dt"], pid=g["pid"]) for g in gathered]}), flush=True)
        return
    import multiprocessing
    mp = multiprocessing.get_context("spawn")
    barrier = mp.Barrier(world)
    procs, pipes = [], []
    for r in range(world):
        here, there = mp.Pipe(duplex=False)
        p = mp.Process(target=_dry_child, args=(r, world, barrier, there))
        p.start()
        there.close()
        procs.append(p)
        pipes.append(here)
    reports = [c.recv() for c in pipes]
    for p in procs:
        p.join()
    assert all(p.exitcode == 0 for p in procs)
    dt = max(r["dt"] for r in reports)
    print(json.dumps({"metric": "read_x_haplotype_align_gcups", "dry_run": True, "n_gpus": len(reports),
                      "seconds": dt, "ranks": [dict(rank=r["rank"], seconds=r["dt"], pid=r["pid"]) for r in reports],
                      "parent_pid": os.getpid()}), flush=True)


if __name__ == "__main__":
    main()
