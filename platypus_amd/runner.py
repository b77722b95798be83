"""Command line of the region caller: BAM files + indexed FASTA -> VCF, one worker process per GPU.

    python -m platypus_amd.runner callVariants --bamFiles a.bam,b.bam --refFile ref.fa \\
        [--regions chr:start-end,...] [--output AllVariants.vcf] [--nGPU 1] [options of Options]

The scheduling is the reference's (platypus/runner.py:98-165): regions are cut into chunks of
--processRegionSize, every chunk is called on its own into a temporary file and the files are
concatenated in (chromosome, start) order.  Where the reference deals chunks to --nCPU worker
processes, here they are dealt to --nGPU x --workersPerGPU processes, each bound to one device;
nothing is exchanged between them.  Inside a chunk reads are buffered --bufferSize bases at a time
(platypus/variantcaller.pyx:229-233)."""
import argparse
import collections
import multiprocessing
import os
import sys
import tempfile
import threading
import time

# Nothing here calls a BLAS routine, and the thread pool numpy's BLAS library starts when it is loaded would make this
# a process with native threads, whose workers have to be started as fresh interpreters instead of forked
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

from platypus_amd import shard  # noqa: E402


def parse_regions(text, fasta):
    """platypus/bamfileutils.pyx:162-246 for 'chr', 'chr:start-end' lists; default: every sequence of the FASTA index"""
    out = []
    if not text:
        for name, t in sorted(fasta.refs.items()):
            out.append((name.decode(), None, t.SeqLength))
        return out
    for item in text.split(","):
        if ":" in item:
            chrom, span = item.split(":")
            s, e = span.split("-")
            out.append((chrom, int(s), int(e)))
        else:
            out.append((item, None, fasta.refs[item.encode()].SeqLength))
    return out


def sample_names(args):
    """cpopulation.pyx:98-101: the file name of every BAM, or what --labels matches in its path"""
    import re
    files = args.bamFiles.split(",")
    if args.labels is None:
        return [x.split("/")[-1] for x in files]
    return [re.search(args.labels, x).group(0) for x in files]


def call_chunks(chunks, args, tmpdir):
    """Calls every chunk of the list into its own temporary file.  The read buffers of the next
    stretches are prepared on helper threads (file reads, BGZF inflation, record decoding, filtering and
    trimming, the scan for variant candidates: all of it runs outside the interpreter lock) while the current
    stretch is being called."""
    from concurrent.futures import ThreadPoolExecutor
    from platypus_amd import bamio, pipeline, simulate, vcfout
    import gc
    # A stretch makes a few hundred thousand small objects (variants, haplotypes, calls) that live until it has been
    # written, and the interpreter's cycle collector walks the young ones after every 700 allocations by default: a sixth
    # of the calling thread's time for nothing -- the stretch's objects hold no cycles.  It stays on, at a longer interval.
    gc.set_threshold(100000, 50, 100)
    # A helper thread needs the interpreter lock for a moment between its long steps outside it; at the default interval a
    # helper waits up to 5 ms each time while the calling thread computes, and the stretches arrive late (measured: 4 %
    # of a one-process run).  The calling thread gives way after 50 us instead.
    sys.setswitchinterval(5e-5)
    from platypus_amd.fastafile import FastaFile
    fa = FastaFile(args.refFile, args.refFile + ".fai", parseNCBI=bool(args.parseNCBI))
    readers = [bamio.BamReader(b) for b in args.bamFiles.split(",")]
    samples = sample_names(args)
    opts = pipeline.Options(**{k: getattr(args, k) for k in vars(pipeline.Options()) if hasattr(args, k)})
    units = [(ch, s, min(s + args.bufferSize, ch[2])) for ch in chunks for s in range(ch[1], ch[2], args.bufferSize)]

    spent = {"fetch": 0.0, "buffers": 0.0, "candidates": 0.0, "wait": 0.0, "call": 0.0, "write": 0.0}
    counts = {"windows": 0, "called_windows": 0, "candidates": 0}
    phases = {}                                           # seconds per phase of callRegion, summed (timing report only)
    kernels = {"pairs": 0, "cells": 0, "vote_pairs": 0, "vote_named": 0, "vote_full": 0, "anchor_ms": 0.0, "align_ms": 0.0, "setup_ms": 0.0, "em_ms": 0.0}   # the device batches' own counts
    timed = bool(os.environ.get("PLATYPUS_AMD_TIMINGS"))

    local = threading.local()
    opened = []
    inflaters = None                                      # --inflate device: BGZF blocks are inflated on the GPU
    if args.inflate == "device":
        inflaters = bamio.InflaterPool.shared(int(os.environ.get("PLATGPU_DEVICE", "0")), args.inflaters)

    def load_piece(unit, k, parts):
        """records of piece k of the unit's stretch cut into `parts`, one set of arrays per BAM file.  A piece after the
        first keeps the records that start inside it, so every record of the stretch is in exactly one piece.  Pieces exist
        at the start of a run only, when nothing can be called before the first stretches are there; they are inflated
        by the host library whatever --inflate says (a device call takes tens of milliseconds however small)."""
        (chrom, _, _), s, e = unit
        if not hasattr(local, "piece_readers"):
            local.piece_readers = [bamio.BamReader(b) for b in args.bamFiles.split(",")]
            opened.extend(local.piece_readers)
        lo, hi = s + (e - s) * k // parts, s + (e - s) * (k + 1) // parts
        out = []
        for r in local.piece_readers:
            try:
                a = r.fetch(chrom, lo - (1 if k else 0), hi)      # (a record without extent that starts on the edge)
            except KeyError:
                a = bamio._concat([])
            out.append(a if k == 0 else bamio.drop_before(a, lo))
        return out

    bam_names = args.bamFiles.split(",")

    def load_bam(unit, bi, pieces=None):
        """the read buffer of one BAM file for the unit's stretch: fetch, inflate, decode, filter and trim"""
        (chrom, _, _), s, e = unit
        if not hasattr(local, "readers"):                 # a file handle per helper thread and file, opened when first used
            local.readers = {}
        if bi not in local.readers:
            local.readers[bi] = bamio.BamReader(bam_names[bi], inflater=inflaters)
            opened.append(local.readers[bi])
        t0 = time.perf_counter()
        if pieces is not None:                            # the stretch was read in pieces side by side
            got = [p.result()[bi] for p in pieces]
            reads = bamio._concat([a for a in got if len(a["pos"])])
        else:
            try:
                reads = local.readers[bi].fetch(chrom, s, e)
            except KeyError:                              # this BAM has no such sequence: no reads from it
                reads = bamio._concat([])
        reads["chrom"] = chrom.encode()
        t1 = time.perf_counter()
        buf = simulate.buffer_of(reads, s, e, minMapQual=opts.minMapQual, maxReads=int(opts.maxReads),
                                 minGoodQualBases=opts.minGoodQualBases, maxBadQualBases=opts.maxBadQualBases)
        spent["fetch"] += t1 - t0
        spent["buffers"] += time.perf_counter() - t1
        return buf

    def load(unit, pieces=None, per_bam=None):            # fetch, inflate, decode, filter and trim: all outside the lock
        (chrom, _, _), s, e = unit
        if per_bam is not None:                           # every file's buffer was made by a task of its own
            bufs = [f.result() for f in per_bam]
        else:
            bufs = [load_bam(unit, bi, pieces) for bi in range(len(bam_names))]
        if sum(b.readCount for b in bufs) > opts.maxReads:                # variantcaller.pyx:84-88: the stretch is skipped
            return bufs, None
        t2 = time.perf_counter()
        if not hasattr(local, "fa"):                      # a reference file handle per helper thread
            local.fa = FastaFile(args.refFile, args.refFile + ".fai", parseNCBI=bool(args.parseNCBI))
        local.fa.setCacheSequence(chrom.encode(), s - 25000, e + 25000)
        cands = pipeline.candidatesInRegion(chrom.encode(), s, e, bufs, local.fa, opts)
        spent["candidates"] += time.perf_counter() - t2
        return bufs, cands

    header = "\n".join(vcfout.header(samples, " ".join(sys.argv[1:]))) + "\n"
    for ch in chunks:
        if ch[1] >= ch[2]:                                                # an empty chunk still has its (header only) file
            with open(os.path.join(tmpdir, "output_temp_" + shard.region_name(ch)), "w") as f:
                f.write(header)
    n_calls, out, current = 0, None, None
    ahead = max(1, args.readAhead)
    with ThreadPoolExecutor(ahead) as pool:
        # Nothing can be called before the first stretch is there, and the calling thread takes a stretch in a tenth of the
        # time one helper thread needs to prepare it: the first stretches are read in pieces (12, 6, 4, 3, 2, 2 of them
        # with twelve helper threads), queued in the order of the stretches, so that the helpers finish stretch 0 together
        # first, then stretch 1, ... instead of every helper starting a whole stretch of its own.
        # With many BAM files a stretch's files are read side by side instead (one task per file, queued before the
        # task that waits for them: the pool starts tasks in the order they were queued, so a waiting task never holds
        # up what it waits for).
        def submit(k, unit):
            if len(bam_names) >= 4:
                return pool.submit(load, unit, None, [pool.submit(load_bam, unit, bi) for bi in range(len(bam_names))])
            parts = max(1, ahead // (k + 1)) if k < ahead else 1
            if parts <= 1:
                return pool.submit(load, unit)
            return pool.submit(load, unit, [pool.submit(load_piece, unit, j, parts) for j in range(parts)])

        pending = collections.deque(submit(k, u) for k, u in enumerate(units[:ahead]))
        if units and not args.printVarsAndExit:
            from platypus_amd import calign
            calign.openDevice(int(opts.rlen))             # while the helpers read the first stretch
        for i, (ch, s, e) in enumerate(units):
            t0 = time.perf_counter()
            bufs, cands = pending.popleft().result()
            spent["wait"] += time.perf_counter() - t0
            if i + ahead < len(units):
                pending.append(submit(i + ahead, units[i + ahead]))
            if ch is not current:
                if out is not None:
                    out.close()
                out = open(os.path.join(tmpdir, "output_temp_" + shard.region_name(ch)), "w")
                out.write(header)
                current = ch
            chrom = ch[0]
            if cands is None:                                 # more reads than maxReads
                continue
            if args.printVarsAndExit:                         # variantcaller.pyx:110-113
                for v in cands:
                    print(v)
                continue
            t0 = time.perf_counter()
            # A window that cannot be called costs only itself: callRegion skips it with a log line, as the reference does
            # (variantcaller.pyx:189-191).  Anything else -- a device error, a fault of this code -- ends the worker with a
            # non-zero status, as the reference's region wrapper re-raises (variantcaller.pyx:238-248): no stretch is ever
            # dropped from a run that reports success.
            stats = {} if timed else None
            calls = pipeline.callRegion(chrom.encode(), s, e, bufs, fa, opts, variants=cands, stats=stats)
            if timed:
                for k, v in stats.items():
                    if k.startswith("t_") or k in ("pack", "upload", "device", "unpack"):
                        phases[k] = phases.get(k, 0.0) + v
                    elif k in kernels:
                        kernels[k] += v
                    elif k in counts:
                        counts[k] += v
            t1 = time.perf_counter()
            for c in calls:
                out.write(vcfout.record(c, fa) + "\n")
                n_calls += 1
            spent["call"] += t1 - t0
            spent["write"] += time.perf_counter() - t1
    if out is not None:
        out.close()
    for r in readers + opened:
        r.close()
    if timed:                                             # where a worker's time went (helper thread: fetch, buffers)
        sys.stderr.write("worker %d: %d calls from %d candidates in %d windows (%d with reads and more than one haplotype); helper threads: fetch %.2f s, buffers %.2f s, candidates %.2f s; main thread: "
                         "waiting %.2f s, calling %.2f s, writing %.2f s; phases of calling: %s\n" % (
                             os.getpid(), n_calls, counts["candidates"], counts["windows"], counts["called_windows"], spent["fetch"], spent["buffers"], spent["candidates"], spent["wait"],
                             spent["call"], spent["write"],
                             ", ".join("%s %.2f" % (k, v) for k, v in sorted(phases.items()))))
        sys.stderr.write("worker %d: device batches: %d pairs, %d cells, %d pairs anchored by votes (%d + %d of them counted by a wavefront), anchoring %.3f ms, alignment %.3f ms, genotype set-up %.3f ms, EM %.3f ms\n" % (
            os.getpid(), kernels["pairs"], kernels["cells"], kernels["vote_pairs"], kernels["vote_named"], kernels["vote_full"], kernels["anchor_ms"], kernels["align_ms"], kernels["setup_ms"], kernels["em_ms"]))
    return n_calls


def _gpu_untouched():
    """True when this process holds no handle of the GPU driver's device node (the HIP runtime opens it with its first
    call, not when its library is loaded)"""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink("/proc/self/fd/" + fd).startswith(("/dev/kfd", "/dev/dri/")):
                    return False
            except OSError:
                pass
        return True
    except OSError:
        return False


def _single_threaded():
    """True when the process has one thread in all, native threads of loaded libraries included (threading's own count
    sees interpreter threads only); a process that cannot tell says no"""
    try:
        return len(os.listdir("/proc/self/task")) == 1
    except OSError:
        return False


def _worker(rank, chunks, args, tmpdir, queue, forked=False):
    # workers of one GPU share its device; on a machine with fewer devices than --nGPU names (a rehearsal of a larger
    # run) the GPUs asked for are dealt onto the ones there are
    from platypus_amd import native
    os.environ["PLATGPU_DEVICE"] = str((rank % max(1, args.nGPU)) % max(1, native.device_count()))
    queue.put((rank, call_chunks(chunks, args, tmpdir)))
    if forked:
        # multiprocessing ends a forked child without the C library's exit handlers, so the HIP runtime would never shut
        # down, and the driver takes 0.2 s to clean up after a process that dies holding queues and pinned memory (0.01 s
        # after an orderly shutdown).  Everything this worker owes is on disk or in the queue: end through exit().
        import ctypes
        queue.close()
        queue.join_thread()
        sys.stdout.flush()
        sys.stderr.flush()
        ctypes.CDLL(None).exit(0)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="platypus_amd.runner")
    ap.add_argument("command", choices=["callVariants"])
    ap.add_argument("--bamFiles", required=True)
    ap.add_argument("--refFile", required=True)
    ap.add_argument("--regions", default=None)
    ap.add_argument("--output", "-o", default="AllVariants.vcf")
    ap.add_argument("--nGPU", type=int, default=1)
    ap.add_argument("--workersPerGPU", type=int, default=1,
                    help="host processes per device: the host side (BAM decoding, candidates, windows) dominates the "
                         "run time, so several processes can share one GPU")
    ap.add_argument("--readAhead", type=int, default=None,
                    help="buffers prepared ahead of the one being called, each on a helper thread of its own "
                         "(default: 12 for a single host process, 6 per process otherwise); a shared pool of piece readers "
                         "under a few buffer threads was measured and is slower")
    ap.add_argument("--inflate", choices=["auto", "device", "host"], default="auto",
                    help="where the BGZF blocks of the BAM files are inflated: on the GPU (one lane per block; a third fewer "
                         "host core-seconds per Mb) or by the host library on the helper threads.  auto: on the GPU while at most "
                         "four host processes share a GPU, by the host library beyond (measured on 16 cores per GPU: with eight "
                         "processes their inflater streams take turns on the device and the cores are there to be used)")
    ap.add_argument("--inflaters", type=int, default=None,
                    help="device inflaters of a host process, shared by its helper threads (default: 4 with up to three "
                         "processes per GPU, 3 with four, 2 beyond)")
    ap.add_argument("--nCPU", type=int, default=None,
                    help="the reference's option: taken as the total number of host processes when --workersPerGPU is not given")
    ap.add_argument("--labels", default=None, help="regular expression that extracts a sample name from a BAM path")
    ap.add_argument("--printVarsAndExit", type=int, default=0, help="print the variant candidates and stop")
    ap.add_argument("--parseNCBI", type=int, default=0)
    ap.add_argument("--nIndividuals", "-n", type=int, default=1, help="accepted; the number of BAM files decides")
    ap.add_argument("--dataType", default="population")
    ap.add_argument("--verbosity", type=int, default=2)
    ap.add_argument("--logFileName", default=None)
    ap.add_argument("--getVariantsFromBAMs", type=int, default=1)
    for unsupported in ("--source", "--useCortex", "--freqAsPrior"):
        ap.add_argument(unsupported, default=None, help="not supported by this build")
    ap.add_argument("--processRegionSize", type=int, default=30000000)       # the reference's default (platypus/runner.py:181)
    ap.add_argument("--bufferSize", type=int, default=1000000)
    from platypus_amd import pipeline
    for k, v in vars(pipeline.Options()).items():
        ap.add_argument("--" + ("maxReadLength" if k == "rlen" else k), dest=k, type=type(v), default=v)
    args = ap.parse_args(argv)
    for name in ("source", "useCortex", "freqAsPrior"):
        if getattr(args, name) not in (None, "0", 0):
            ap.error("--%s is not supported: candidates come from the BAM files only" % name)
    if args.ploidy != 2 or args.dataType != "population" or not args.getVariantsFromBAMs:
        ap.error("only --ploidy 2, --dataType population and --getVariantsFromBAMs 1 are supported")
    if args.nCPU is not None and "--workersPerGPU" not in (argv if argv is not None else sys.argv):
        args.workersPerGPU = max(1, -(-args.nCPU // max(1, args.nGPU)))
    from platypus_amd.fastafile import FastaFile
    fa = FastaFile(args.refFile, args.refFile + ".fai", parseNCBI=bool(args.parseNCBI))
    chunks = shard.split_regions(parse_regions(args.regions, fa), args.processRegionSize)
    tmpdir = tempfile.mkdtemp(prefix="platypus_amd_")
    n_workers = max(1, args.nGPU) * max(1, args.workersPerGPU)
    if args.readAhead is None:
        args.readAhead = 12 if n_workers == 1 else 6
    if args.inflaters is None:
        args.inflaters = max(2, min(4, 12 // max(1, args.workersPerGPU)))     # twelve streams per GPU between its processes
    if args.inflate == "auto":
        args.inflate = "device" if max(1, args.workersPerGPU) <= 4 else "host"
    weights = None
    if n_workers > 1:                                       # longest first by the reads a chunk holds, as far as the
        from platypus_amd import bamio                      # BAM indexes tell (compressed bytes), else by length
        sizes = []
        for b in args.bamFiles.split(","):
            r = bamio.BamReader(b)
            sizes.append([r.regionBytes(ch[0], ch[1], ch[2]) for ch in chunks])
            r.close()
        if all(v is not None for row in sizes for v in row):
            weights = [sum(row[i] for row in sizes) + 1 for i in range(len(chunks))]
    per_rank = shard.assign(chunks, n_workers, weights)
    if n_workers <= 1:
        call_chunks(chunks, args, tmpdir)
    else:
        # Workers are forked while this process has not opened the GPU (no second interpreter start-up and import); a
        # process that has, or that runs other threads, starts them fresh: nothing of its GPU state may reach them.
        forked = _gpu_untouched() and _single_threaded()
        ctx = multiprocessing.get_context("fork" if forked else "spawn")
        queue = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, per_rank[r], args, tmpdir, queue, forked)) for r in range(n_workers)]
        for p in procs:
            p.start()
        for p in procs:
            p.join()
            if p.exitcode != 0:
                raise SystemExit("a worker failed (exit code %s)" % p.exitcode)
    parts = {}
    for ch in chunks:
        name = shard.region_name(ch)
        with open(os.path.join(tmpdir, "output_temp_" + name)) as f:
            parts[name] = f.read().splitlines()
    with open(args.output, "w") as out:
        out.write("\n".join(shard.merge_outputs(parts)) + "\n")
    for ch in chunks:
        os.remove(os.path.join(tmpdir, "output_temp_" + shard.region_name(ch)))
    os.rmdir(tmpdir)
    return 0


if __name__ == "__main__":
    sys.exit(main())
