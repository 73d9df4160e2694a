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
b = DeviceBatch([distinct[j] for j in order], [65536 + 256] * n)
b.decode(check=0, lanes=1); torch.cuda.synchronize()
b.decode(check=0, lanes=1); torch.cuda.synchronize()
r = b.records()
print("rc ok:", (r["rc"] == 0).all())
host = b.dst.cpu().numpy()
offs = np.asarray(b.host_dst_off[:n + 1], dtype=np.int64)
rows = []
for i in range(0, n, S):
    e = int(offs[i + 1])
    rows.append(host[e - 128:e].view(np.uint64).astype(np.float64))
A = np.stack(rows)
H = np.stack([host[int(offs[i + 1]) - 256:int(offs[i + 1]) - 176].view(np.uint64).astype(np.float64) for i in range(0, n, S)])
hm = np.median(H, axis=0)
print("header of lane 0's stream, median (ms): block headers + context maps %.3f, literal codes %.3f, insert-and-copy codes %.3f, distance codes %.3f, map resolution %.3f" % tuple(hm[:5] / 1e5))
print("complex prefix codes of a stream, median (ms): code length code %.3f, code lengths %.3f, counting sort %.3f, table %.3f" % tuple(hm[6:10] / 1e5))
cnt = H[:, 5].astype(np.uint64)
print("mean number of literal codes %.1f, insert-and-copy %.1f, distance %.1f, literal block types %.1f" % ((cnt & 0xffff).mean(), ((cnt >> 16) & 0xffff).mean(), ((cnt >> 32) & 0xffff).mean(), (cnt >> 48).mean()))
m = A.mean(0)
# prof[12] = trips of the wavefront, [13] = header and tables, [14] = the loop (both in 10 ns units)
print("trips per wavefront: mean %.0f max %.0f" % (m[12], A[:, 12].max()))
print("header and tables %.2f ms, loop %.2f ms per wavefront (mean); slowest loop %.2f ms" % (m[13] / 1e5, m[14] / 1e5, A[:, 14].max() / 1e5))
print("time per trip: %.0f ns" % (10 * m[14] / m[12]))
print("cycles per trip waiting: for the entry (head) %.0f, for the input words %.0f, for the piece %.0f" % (m[2] / m[12], m[3] / m[12], m[4] / m[12]))
slow = A[A[:, 14].argmax()]
print("slowest wavefront: %.0f trips, %.0f ns per trip, waits %.0f + %.0f + %.0f cycles per trip" % (slow[12], 10 * slow[14] / slow[12], slow[2] / slow[12], slow[3] / slow[12], slow[4] / slow[12]))
names = ["loop back, wait for the entry, entry read", "1 consume, block switch", "wait for the input words, merge, peek, flags",
         "wait for the piece, stores, context bytes", "3 context -> table, enter", "request (entry, top-up)",
         "4 distance, dictionary, copy set-up", "piece request"]
for k, name in enumerate(names):
    v = m[5 + k] if k < 7 else m[15]
    print("  %-48s %6.0f" % (name, v / m[12]))
