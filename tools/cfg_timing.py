#!/usr/bin/env python3
"""BASELINE configs 4 and 5 timed through the host-pointer calls (PCIe and host work included) beside
the CPU: cfg4 = the synthesised dictionary/transform/context-mode streams as one batch, cfg5 = each
large-window stream as a single call (a latency case: one stream is one serial chain)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import brotli_amd as ba
ref = H.reference()
cpu = (lambda s: H.run_yeast(ref.yeast, s)) if ref is not None else H.oracle_yeast
res = {"cpu": "reference" if ref is not None else "restatement"}
meta, streams = H.packed_streams("cfg4_streams")
ba.yeast_batch(streams)
t0 = time.perf_counter(); r = ba.yeast_batch(streams, ba.CHECK_XXH64); tg = time.perf_counter() - t0
t0 = time.perf_counter(); outs = [cpu(s) for s in streams]; tc = time.perf_counter() - t0
out_bytes = sum(x[2] for x in r)
res["cfg4"] = {"streams": len(streams), "out_bytes": out_bytes, "gpu_ms": tg * 1e3, "cpu_1core_ms": tc * 1e3}
meta, streams = H.packed_streams("cfg5_streams")
res["cfg5"] = []
for m, s in zip(meta["streams"], streams):
    ba.yeast_batch([s])  # the CPU beside it computes no check either
    t0 = time.perf_counter(); (rc, out, got, used, chk), = ba.yeast_batch([s]); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); c = cpu(s); tc = time.perf_counter() - t0
    assert rc == m["rc"] and got == m["got"]
    res["cfg5"].append({"name": m["name"], "out_bytes": got, "gpu_ms": tg * 1e3, "cpu_1core_ms": tc * 1e3})
print(json.dumps(res))
