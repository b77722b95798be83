"""One process per GPU over genomic-region shards; no collective on the data path.

The reference parallelises by regions only: ``getRegions`` cuts every
chromosome into chunks of ``processRegionSize`` (platypus/bamfileutils.pyx:248-261),
``multiprocessing.Pool.map`` gives each chunk to a worker that writes its own
temporary VCF (platypus/runner.py:119-135), and the parent concatenates the
files sorted by chromosome name, then start (platypus/runner.py:19-30,137-165).
Workers exchange nothing.  Here the workers are GPU ranks: the same chunks,
dealt to ranks by estimated work, the same merge order on the host.  Ranks
exchange nothing here either (a benchmark that wants the slowest rank's time
reduces it itself).
"""
import functools
import os


def split_regions(regions, process_region_size):
    """The chunking of platypus/bamfileutils.pyx:248-261.

    ``regions`` holds (chrom, start, end); start may be None for "whole
    sequence of length end", in which case chunks start at 1 as the reference's do.
    """
    out = []
    for chrom, start, end in regions:
        if start is None:
            for i in range(1, end, process_region_size):
                out.append((chrom, i, min(i + process_region_size, end)))
        elif end - start > process_region_size:
            for i in range(start, end, process_region_size):
                out.append((chrom, i, min(i + process_region_size, end)))
        else:
            out.append((chrom, start, end))
    return out


def assign(chunks, world_size, weights=None):
    """Deal chunks to ranks: longest-processing-time-first on ``weights`` (e.g. read
    counts from the BAM index; chunk length if absent).  Deterministic, every
    chunk goes to exactly one rank.  Returns a list of chunk lists, one per rank."""
    if weights is None:
        weights = [c[2] - c[1] for c in chunks]
    order = sorted(range(len(chunks)), key=lambda i: (-weights[i], i))
    load = [0] * world_size
    mine = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        mine[r].append(i)
        load[r] += weights[i]
    return [[chunks[i] for i in sorted(idx)] for idx in mine]


def region_name(chunk):
    return "%s:%s-%s" % chunk           # the temp-file suffix of platypus/runner.py:116,128


def _chrom_and_pos_cmp(x, y):
    """platypus/runner.py:19-30 (chromosome as a STRING, then start as a number)."""
    xc, yc = x.split(":")[0], y.split(":")[0]
    xs, ys = int(x.split(":")[1].split("-")[0]), int(y.split(":")[1].split("-")[0])
    return (xc > yc) - (xc < yc) or (xs > ys) - (xs < ys)


def merge_order(names):
    """Order in which per-chunk outputs are concatenated (platypus/runner.py:146)."""
    return sorted(names, key=functools.cmp_to_key(_chrom_and_pos_cmp))


def merge_outputs(parts):
    """``parts`` maps region name -> list of text lines of that chunk's output, header
    lines starting with '#'.  The header is taken from the first file only and body
    lines follow in merge order (platypus/runner.py:149-160)."""
    out = []
    for k, name in enumerate(merge_order(list(parts))):
        for line in parts[name]:
            if line.startswith("#"):
                if k == 0:
                    out.append(line)
            else:
                out.append(line)
    return out


def rank_info():
    """(rank, world_size, local_rank) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))
