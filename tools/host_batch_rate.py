#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer batch call (yeast_batch: host buffers in, one malloc'd
buffer per stream out, XXH64 check): never bench.py's `value`, reported in DESIGN.md beside it."""
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
for n in (2048, 16384, 65536):
    streams = [distinct[(k * 37) % len(distinct)] for k in range(n)]
    src = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(s), ctypes.c_void_p) for s in streams])
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
        total = sum(got)
        for p in dest:
            libc.free(p)
        best = dt if best is None or dt < best else best
    res[n] = {"ms": best * 1e3, "GBps_out": total / best / 1e9}
print(json.dumps(res))
