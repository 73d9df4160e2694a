#!/usr/bin/env python3
"""Which decode kernel serves a host-pointer batch of n streams fastest: 1 = one stream per lane,
65 = one wavefront per stream with the tables in LDS, 64/16/8 = the cooperative kernel.  Times the C call
(best of three) for each n and kernel, 64 KiB text streams; the thresholds in yeast_host.cpp follow it."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import brotli_amd as ba
L = ba.lib()
libc = ctypes.CDLL(None)
libc.free.argtypes = [ctypes.c_void_p]
meta, distinct = H.packed_streams("cfg2_streams")
res = {}
for n in (16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768):
    streams = [distinct[(k * 37) % len(distinct)] for k in range(n)]
    src = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(s), ctypes.c_void_p) for s in streams])
    row = {}
    for lanes in (1, 65, 64, 16, 8):
        os.environ["YEAST_FORCE_LANES"] = str(lanes)
        best = None
        for _ in range(3):
            lens = (ctypes.c_size_t * n)(*[len(s) for s in streams])
            dest = (ctypes.c_void_p * n)()
            got = (ctypes.c_size_t * n)()
            rc = (ctypes.c_int * n)()
            chk = (ctypes.c_uint64 * n)()
            t0 = time.perf_counter()
            e = L.yeast_batch(n, src, lens, dest, got, rc, ba.CHECK_XXH64, chk)
            dt = time.perf_counter() - t0
            assert e == 0 and not any(rc)
            for p in dest:
                libc.free(p)
            best = dt if best is None or dt < best else best
        row[lanes] = round(best * 1e3, 2)
    res[n] = row
    print(n, row, flush=True)
