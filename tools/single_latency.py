#!/usr/bin/env python3
"""BASELINE config 1 beside the GPU: one yeast() call on testdata/alice29.txt.compressed through the
drop-in (host pointers in, malloc'd output back), timed; the reference on one host core where
oracle/_ref exists, else this repo's CPU restatement."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import brotli_amd as ba
res = {}
for name in ("alice29.txt", "plrabn12.txt", "quickfox"):
    comp = open(os.path.join(H.GOLD, "testdata", name + ".compressed"), "rb").read()
    orig = open(os.path.join(H.GOLD, "testdata", name), "rb").read()
    ba.yeast(comp)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); rc, out, got, used = ba.yeast(comp); t.append(time.perf_counter() - t0)
    assert rc == 0 and out == orig
    t0 = time.perf_counter(); rc, out, got, used = ba.yeast(comp, orig); tc = time.perf_counter() - t0
    assert rc == 0
    ref = H.reference()
    fn = H.run_yeast
    t0 = time.perf_counter()
    r = fn(ref.yeast if ref is not None else H.oracle().oracle_yeast, comp)
    tr = time.perf_counter() - t0
    res[name] = {"bytes_out": len(orig), "gpu_yeast_ms": min(t) * 1e3, "gpu_compare_mode_ms": tc * 1e3,
                 "cpu_%s_ms" % ("reference" if ref is not None else "restatement"): tr * 1e3}
print(json.dumps(res))
