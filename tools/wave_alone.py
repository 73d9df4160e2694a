#!/usr/bin/env python3
"""The wavefront-per-stream kernel with a wavefront alone on its CU (1 and 256 streams of BASELINE config 2, events on
the stream): the product build's cycles per command, without the marks of the profile build."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
import brotli_amd as ba
from brotli_amd.device import DeviceBatch
meta, distinct = H.packed_streams("cfg2_streams")
res = {}
for n in (1, 256, 1024, 2048):
    order = np.random.RandomState(3).randint(0, len(distinct), size=n)
    b = DeviceBatch([distinct[j] for j in order], [65536] * n)
    b.decode(0, 65); torch.cuda.synchronize()
    t = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.decode(0, 65); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    assert (b.records()["rc"] == 0).all()
    res[str(n)] = {"ms": min(t), "in_bytes": int(sum(len(distinct[j]) for j in order))}
print(json.dumps(res))
