"""Wall time of each stage of one batch through the C ABI (host buffers -> results on the host)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from platypus_amd import native, synth

b = synth.make_batch(n_windows=int(sys.argv[1]) if len(sys.argv) > 1 else 32768)
if len(sys.argv) > 2 and sys.argv[2] == "pinned":          # the caller's buffers in page-locked memory (pg_host_alloc)
    for k in list(b):
        if isinstance(b[k], np.ndarray):
            p = native.pinned_array(b[k].shape, b[k].dtype)
            p[...] = b[k]
            b[k] = p
ctx = native.AlignContext(0)
for rep in range(3):
    t = [time.perf_counter()]
    ctx.upload_haps(b["hap_seq"], b["hap_off"], b["hap_start"], b["hap_varlen"]); t.append(time.perf_counter())
    ctx.upload_reads(b["read_seq"], b["read_qual"], b["read_off"], b["read_pos"], b["read_mapq"]); t.append(time.perf_counter())
    ctx.upload_pairs(b["pair_hap"], b["pair_read"]); t.append(time.perf_counter())
    ctx.launch(); ctx.sync(); t.append(time.perf_counter())
    s, a, l = ctx.fetch(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    n = len(s)
    print("rep %d: haps %.1f ms  reads %.1f ms  pairs %.1f ms  kernels %.2f ms  fetch %.1f ms  total %.1f ms -> %.1f Mpairs/s, %.0f GCUPS incl. PCIe and host packing"
          % (rep, d[0], d[1], d[2], d[3], d[4], d.sum(), n / d.sum() / 1e3, n * 2528 / d.sum() / 1e6))
bytes_in = b["hap_seq"].nbytes + 2 * b["read_seq"].nbytes + 2 * b["pair_hap"].nbytes
print("H2D payload %.0f MB, D2H %.0f MB" % (bytes_in / 1e6, n * 16 / 1e6))
