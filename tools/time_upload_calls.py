"""Wall time of each call of one streamed batch (one context, nothing else running): where a batch's host time goes.

    python tools/time_upload_calls.py [windows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from platypus_amd import native, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
H, R, L = 4, 32, 150
b = synth.make_batch(n_windows=W, haps_per_window=H, reads_per_window=R, read_len=L)


def host(a):
    p = native.pinned_array(a.shape, a.dtype)
    p[...] = a
    return p


src = {k: host(b[k]) for k in ("hap_seq", "hap_off", "hap_start", "hap_varlen", "read_seq", "read_qual", "read_off", "read_pos", "read_mapq")}
bsrc = {k: host(v) for k, v in b["built"].items()}
blk = dict(hap_lo=np.arange(W, dtype=np.int32) * H, n_haps=np.full(W, H, np.int32), read_lo=np.arange(W, dtype=np.int32) * R,
           n_reads=np.full(W, R, np.int32))
n_pairs = W * H * R
out = (host(np.empty(n_pairs, np.int32)), host(np.empty(n_pairs, np.int32)), host(np.empty(n_pairs, np.float64)))
c = native.AlignContext(0)
steps = [("pg_haps_upload", lambda: c.upload_haps(src["hap_seq"], src["hap_off"], src["hap_start"], src["hap_varlen"])),
         ("pg_haps_build", lambda: c.build_haps(**bsrc)),
         ("pg_reads_upload", lambda: c.upload_reads(src["read_seq"], src["read_qual"], src["read_off"], src["read_pos"], src["read_mapq"])),
         ("pg_pairs_blocks", lambda: c.pairs_blocks(blk["hap_lo"], blk["n_haps"], blk["read_lo"], blk["n_reads"])),
         ("pg_align_launch + pg_sync", lambda: (c.launch(), c.sync())),
         ("pg_results_fetch_async + pg_sync", lambda: (c._check(native.LIB.pg_results_fetch_async(c._h, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data)), c.sync()))]
times = {k: [] for k, _ in steps}
for rep in range(6):
    for k, f in steps:
        t0 = time.perf_counter()
        f()
        times[k].append(time.perf_counter() - t0)
for k, _ in steps:
    print("%-36s %7.3f ms" % (k, 1e3 * min(times[k][1:])))
sizes = dict(haps=sum(src[k].nbytes for k in ("hap_seq", "hap_off", "hap_start", "hap_varlen")), built=sum(v.nbytes for v in bsrc.values()),
             reads=sum(src[k].nbytes for k in ("read_seq", "read_qual", "read_off", "read_pos", "read_mapq")), blocks=sum(v.nbytes for v in blk.values()),
             results=sum(o.nbytes for o in out))
print("bytes:", sizes)
c.close()
