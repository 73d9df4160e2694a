#!/usr/bin/env python3
"""The two side figures SURVEY section 8(d) asks for beside the bench line, measured on the GPU box:
the device-copy bandwidth of this GPU (a 4 GiB HBM-to-HBM copy, bytes read + bytes written) and the
CPU decoder on ONE host thread over a sample of the cfg2 batch (reference build if present, else the
restatement), beside the 16-thread figure bench.py prints."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench

n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(1)
for _ in range(2):
    b.copy_(a)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for s, e in ev:
    s.record(); b.copy_(a); e.record()
torch.cuda.synchronize()
ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
out = {"device_copy": {"bytes_each_way": n, "ms": ms, "GBps_read_plus_write": 2 * n / ms / 1e6}}
del a, b

meta, distinct = bench.load_distinct()
order = bench.build_order(65536, len(distinct), 0x5EED, "shuffled")
out["cpu_1_thread"] = bench.cpu_baseline(distinct, order, 1, 2048)
out["cpu_16_threads"] = bench.cpu_baseline(distinct, order, 16, 16384)
print(json.dumps(out))
