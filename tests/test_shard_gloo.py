"""Region sharding across ranks (CPU, gloo, world size 2): every chunk is owned by
exactly one rank, the merged output is independent of the number of ranks, and
the only cross-rank operation is a MAX over timings."""
import os
import socket
import subprocess
import sys

from platypus_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_matches_reference_chunking():
    # whole chromosome: chunks start at 1 (bamfileutils.pyx:248-252)
    c = shard.split_regions([("chr20", None, 250)], 100)
    assert c == [("chr20", 1, 101), ("chr20", 101, 201), ("chr20", 201, 250)]
    c = shard.split_regions([("chr2", 1000, 1250), ("chr10", 5, 50)], 100)
    assert c == [("chr2", 1000, 1100), ("chr2", 1100, 1200), ("chr2", 1200, 1250), ("chr10", 5, 50)]


def test_merge_order_is_string_then_number():
    names = ["chr2:100-200", "chr10:5-50", "chr2:20-100", "chr10:500-600"]
    assert shard.merge_order(names) == ["chr10:5-50", "chr10:500-600", "chr2:20-100", "chr2:100-200"]
    parts = {"chr2:100-200": ["#h", "b2"], "chr10:5-50": ["#h", "a1", "a2"]}
    assert shard.merge_outputs(parts) == ["#h", "a1", "a2", "b2"]


def test_assign_covers_once_and_balances():
    chunks = shard.split_regions([("chr20", None, 64_400_000)], 1_000_000)
    assert len(chunks) == 65
    w = [(i * 7919) % 1000 + 1 for i in range(len(chunks))]
    for world in (1, 2, 4, 8):
        parts = shard.assign(chunks, world, w)
        flat = sorted(c for p in parts for c in p)
        assert flat == sorted(chunks)
        loads = [sum(w[chunks.index(c)] for c in p) for p in parts]
        assert max(loads) - min(loads) <= max(w)


_WORKER = r"""
import os, sys, json, time
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from platypus_amd import shard
import bench                                  # the benchmark's own MAX over ranks
rank, world, _ = shard.rank_info()
dist.init_process_group("gloo", rank=rank, world_size=world)
chunks = shard.split_regions([("chr20", None, 3_000_000), ("chr3", None, 900_000)], 500_000)
mine = shard.assign(chunks, world)[rank]
# a stand-in for "call variants in my chunks": one body line per chunk
parts = {shard.region_name(c): ["#header", "%%s\t%%d" %% (c[0], c[1])] for c in mine}
dist.barrier()
t = bench.slowest(0.25 * (rank + 1), dist)
gathered = [None] * world
dist.all_gather_object(gathered, parts)      # host-side merge of per-shard outputs, not a data-path collective
if rank == 0:
    allparts = {}
    for g in gathered:
        assert not (set(g) & set(allparts))
        allparts.update(g)
    print(json.dumps({"t": t, "merged": shard.merge_outputs(allparts), "n": len(allparts)}))
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["t"] == 0.5                                  # MAX over ranks
    chunks = shard.split_regions([("chr20", None, 3_000_000), ("chr3", None, 900_000)], 500_000)
    single = shard.merge_outputs({shard.region_name(c): ["#header", "%s\t%d" % (c[0], c[1])] for c in chunks})
    assert res["n"] == len(chunks) and res["merged"] == single   # same file as one process would write


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` without a launcher starts two fresh interpreters itself (the parent opens no GPU), lines
    them up on the run's barriers and prints ONE line with n_gpus = 2 and the slowest rank's time; --dry-run replaces
    the device work by a sleep of 50 ms x (rank + 1)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [l for l in done.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and [r["rank"] for r in res["ranks"]] == [0, 1]
    pids = {r["pid"] for r in res["ranks"]}
    assert len(pids) == 2 and res["parent_pid"] not in pids            # ranks are processes of their own
    assert res["seconds"] == max(r["seconds"] for r in res["ranks"]) and 0.09 < res["seconds"] < 5.0
    assert res["ranks"][1]["seconds"] > res["ranks"][0]["seconds"] - 0.02  # both left the first barrier together


def test_bench_under_a_launcher_prints_one_line_and_nothing_else():
    """started by torch.distributed.run the ranks join a gloo group; gloo announces its connections on the C++ standard
    output, which must not reach this program's: one JSON line from rank 0, byte for byte"""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    done = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                           "--gpus", "2", "--dry-run"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    assert done.stdout.count("\n") == 1 and done.stdout.startswith("{"), done.stdout[:400]
    res = json.loads(done.stdout)
    assert res["n_gpus"] == 2 and [r["rank"] for r in res["ranks"]] == [0, 1]
    assert len({r["pid"] for r in res["ranks"]}) == 2
