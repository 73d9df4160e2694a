#!/usr/bin/env python3
"""BASELINE config 5 taken apart: the two large-window streams decoded device-resident (decode kernel alone,
then with the CRC-32C check), timed with events on the stream, beside the host-pointer call of
tools/cfg_timing.py, so that the share of the kernel, the check and the host work (copies, malloc) shows."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers as H
import brotli_amd as ba
from brotli_amd.device import DeviceBatch

meta, streams = H.packed_streams("cfg5_streams")
ref = H.reference()
res = []
for m, s in zip(meta["streams"], streams):
    row = {"name": m["name"], "out_bytes": m["got"]}
    for lanes in (65, 64):
        for check in (ba.CHECK_NONE, ba.CHECK_CRC32C):
            b = DeviceBatch([s], [m["got"]])
            b.decode(check, lanes)
            torch.cuda.synchronize()
            t = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); b.decode(check, lanes); e1.record(); torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1))
            r = b.records()[0]
            assert r["rc"] == m["rc"] and r["got"] == m["got"], (r, m)
            row["lanes%d_%s_ms" % (lanes, "crc32c" if check else "decode")] = min(t)
    def best(fn, k=5):
        fn()
        t = []
        for _ in range(k):
            t0 = time.perf_counter(); fn(); t.append((time.perf_counter() - t0) * 1e3)
        return {"min": min(t), "median": sorted(t)[len(t) // 2]}
    row["host_call_ms"] = best(lambda: ba.yeast_batch([s]))
    row["host_call_crc32c_ms"] = best(lambda: ba.yeast_batch([s], ba.CHECK_CRC32C))
    row["host_call_xxh64_ms"] = best(lambda: ba.yeast_batch([s], ba.CHECK_XXH64), 2)
    row["yeast_call_ms"] = best(lambda: ba.yeast(s))
    if ref is not None:
        row["cpu_reference_ms"] = best(lambda: H.run_yeast(ref.yeast, s))
    res.append(row)
print(json.dumps(res))
