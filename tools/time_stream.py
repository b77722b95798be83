"""PCIe-inclusive rate of the hot path: batches streamed through the C ABI from host memory to results on the
host, two contexts (one stream each) taking turns so that one batch's transfers overlap the other's kernels.

    python tools/time_stream.py [windows per batch] [batches] [pinned 0/1]

Every batch is uploaded in full (haplotypes, reads, block descriptors), launched and fetched (score, anchor,
likelihood): nothing stays resident between batches.  Prints wall time per batch and GCUPS."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from platypus_amd import native, synth  # noqa: E402

H, R, L = 4, 32, 150


def link_rate():
    """GB/s of one 256-MiB copy from page-locked host memory to the device: the ceiling of any upload"""
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    probe = native.pinned_array((256 << 20,), np.uint8)
    probe[...] = 1
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), probe.nbytes) == 0
    rates = []
    for rep in range(4):
        t0 = time.perf_counter()
        assert hip.hipMemcpy(dev, probe.ctypes.data, probe.nbytes, 1) == 0
        rates.append(probe.nbytes / (time.perf_counter() - t0) / 1e9)
    hip.hipFree(dev)
    return max(rates[1:])


def measure(W=32768, N=12, pinned=True, device=0, batch=None, threads=2, built=False, nibbles=False):
    """built: the haplotypes go up as reference stretches and variant lists and are put together on the device
    (pg_haps_build) instead of as whole sequences (pg_haps_upload); nibbles: the reads' bases go up two per byte in
    BAM's 4-bit code (pg_reads_upload_nibbles) instead of as characters"""
    b = batch if batch is not None else synth.make_batch(n_windows=W, haps_per_window=H, reads_per_window=R, read_len=L)
    n_pairs = W * H * R

    def host(a):
        if not pinned:
            return np.ascontiguousarray(a)
        p = native.pinned_array(a.shape, a.dtype)
        p[...] = a
        return p

    src = {k: host(b[k]) for k in ("hap_seq", "hap_off", "hap_start", "hap_varlen", "read_seq", "read_qual", "read_off",
                                   "read_pos", "read_mapq")}
    bsrc = {k: host(v) for k, v in b["built"].items()} if built else None
    seq4 = host(native.pack_nibbles(b["read_seq"])) if nibbles else None
    blk = dict(hap_lo=np.arange(W, dtype=np.int32) * H, n_haps=np.full(W, H, np.int32),
               read_lo=np.arange(W, dtype=np.int32) * R, n_reads=np.full(W, R, np.int32))
    n_ctx = max(2, int(threads))
    ctxs = [native.AlignContext(device) for _ in range(n_ctx)]
    outs = [(host(np.empty(n_pairs, np.int32)), host(np.empty(n_pairs, np.int32)), host(np.empty(n_pairs, np.float64)))
            for _ in ctxs]

    def upload_and_launch(c):
        if built:
            c.build_haps(**bsrc)
        else:
            c.upload_haps(src["hap_seq"], src["hap_off"], src["hap_start"], src["hap_varlen"])
        if nibbles:
            c.upload_reads_nibbles(seq4, src["read_qual"], src["read_off"], src["read_pos"], src["read_mapq"])
        else:
            c.upload_reads(src["read_seq"], src["read_qual"], src["read_off"], src["read_pos"], src["read_mapq"])
        c.pairs_blocks(blk["hap_lo"], blk["n_haps"], blk["read_lo"], blk["n_reads"])
        c.launch()                               # asynchronous: the kernels run while the other context uploads

    def fetch_async(c, o):                       # enqueued behind the context's kernels; complete after c.sync()
        c._check(native.LIB.pg_results_fetch_async(c._h, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data))

    try:
        for warm in range(2):
            for c, o in zip(ctxs, outs):
                upload_and_launch(c)
                fetch_async(c, o)
                c.sync()
        ref = [x.copy() for x in outs[0]]
        for o in outs:
            for x in o:
                x[...] = 0
        t0 = time.perf_counter()
        if threads:
            # one host thread per context: an upload call returns when its copies and packing kernels have ended, so a
            # single thread that feeds both contexts leaves the link idle while it waits (the library calls run
            # without the interpreter lock)
            import threading

            def feed(k0):
                for k in range(k0, N, n_ctx):
                    upload_and_launch(ctxs[k0])
                    fetch_async(ctxs[k0], outs[k0])
                    ctxs[k0].sync()

            workers = [threading.Thread(target=feed, args=(k0,)) for k0 in range(n_ctx)]
            for w in workers:
                w.start()
            for w in workers:
                w.join()
        else:
            upload_and_launch(ctxs[0])
            fetch_async(ctxs[0], outs[0])
            for k in range(1, N):
                upload_and_launch(ctxs[k & 1])       # batch k goes up while batch k - 1 is computed and comes down
                fetch_async(ctxs[k & 1], outs[k & 1])
                ctxs[(k - 1) & 1].sync()             # batch k - 1 is on the host now
            ctxs[(N - 1) & 1].sync()
        dt = (time.perf_counter() - t0) / N
        same = all(np.array_equal(x, y) for o in outs for x, y in zip(o, ref))
        import zlib
        checksum = [zlib.crc32(x.tobytes()) for x in ref]
    finally:
        for c in ctxs:
            c.close()
    hap_keys = ("hap_seq", "hap_off", "hap_start", "hap_varlen")
    up = sum(src[k].nbytes for k in src if not (built and k in hap_keys)) + sum(v.nbytes for v in blk.values())
    if built:
        up += sum(v.nbytes for v in bsrc.values())
    if nibbles:
        up += seq4.nbytes - src["read_seq"].nbytes
    return dict(ms_per_batch=dt * 1e3, gcups=n_pairs * 16 * (L + 8) / dt / 1e9, pairs_per_s=n_pairs / dt, batches=N,
                pairs_per_batch=n_pairs, bytes_up=up, bytes_down=n_pairs * 16, upload_gbps=up / dt / 1e9,
                host_memory="page-locked (pg_host_alloc)" if pinned else "pageable", results_equal_across_batches=same,
                host_threads=n_ctx if threads else 1, checksum=checksum,
                haplotypes="reference stretches and variant lists, put together on the device" if built else "whole sequences",
                read_bases="two per byte (BAM's 4-bit code)" if nibbles else "characters")


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    pinned = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
    print("host-to-device copy of 256 MiB from page-locked memory: %.1f GB/s" % link_rate())
    for threads, built, nibbles in ((0, False, False), (2, False, False), (4, False, False), (4, True, False), (6, True, False), (4, True, True), (6, True, True), (8, True, True)):
      m = measure(W, N, pinned, threads=threads, built=built, nibbles=nibbles)
      assert m["results_equal_across_batches"]
      print("host threads: %d; haplotypes: %s; read bases: %s" % (m["host_threads"], m["haplotypes"], m["read_bases"]))
      print("%s host memory, %d batches of %d pairs: %.2f ms per batch = %.0f GCUPS (%.1f M pairs/s); up %.0f MB (%.1f GB/s if "
            "transfers alone filled the time), down %.0f MB" % (m["host_memory"], m["batches"], m["pairs_per_batch"],
                                                                m["ms_per_batch"], m["gcups"], m["pairs_per_s"] / 1e6,
                                                                m["bytes_up"] / 1e6, m["upload_gbps"], m["bytes_down"] / 1e6))
