#!/usr/bin/env python3
"""Phase timing of the lean batch kernel (library built with -DYS_PROFILE, loaded through YEAST_LIB):
lane 0 of every wavefront leaves its cycle counts in the last 128 bytes of its output region."""
import os, sys
os.environ["YEAST_SORT"] = "0"  # lane 0 of wavefront k must hold stream k * S: the marks are read from its output region
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H
import brotli_amd as ba
from brotli_amd.device import DeviceBatch
meta, distinct = H.packed_streams("cfg2_streams")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
order = np.random.RandomState(3).randint(0, len(distinct), size=n)
b = DeviceBatch([distinct[j] for j in order], [65536 + 128] * n)
b.decode(check=0, lanes=1); torch.cuda.synchronize()
b.decode(check=0, lanes=1); torch.cuda.synchronize()
r = b.records()
print("rc ok:", (r["rc"] == 0).all())
host = b.dst.cpu().numpy()
offs = np.asarray(b.host_dst_off[:n + 1], dtype=np.int64)
rows = []
for i in range(0, n, S):
    e = int(offs[i + 1])
    rows.append(host[e - 128:e - 128 + 120].view(np.uint64).astype(np.float64))
m = np.stack(rows).mean(0)
names = ["block switch", "A issue (piece loads, lookup)", "A wait", "A store piece + command", "B issue", "B wait", "B literal", "C issue",
         "C wait", "C distance", "D copy set-up", "loop head"]
order_ = [11, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
tot = m[:12].sum()
print("passes per wavefront %.0f; cycles per pass %.0f" % (m[12], tot / m[12]))
print("header and tables %.2f ms, loop %.2f ms per wavefront (mean); slowest loop %.2f ms; clock in the loop %.2f GHz" % (m[13] / 1e5, m[14] / 1e5, np.stack(rows)[:, 14].max() / 1e5, tot / (m[14] * 10)))
for k in order_:
    print("  %-34s %8.0f cycles/pass  %5.1f%%" % (names[k], m[k] / m[12], 100 * m[k] / tot))
