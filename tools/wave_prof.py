#!/usr/bin/env python3
"""Phase timing of the wavefront-per-stream kernel (library built with -DYS_PROFILE, loaded through
YEAST_LIB): per-stream cycle counts written over the first bytes of each output region."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H
import brotli_amd as ba
from brotli_amd.device import DeviceBatch
meta, distinct = H.packed_streams("cfg2_streams")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
order = np.random.RandomState(3).randint(0, len(distinct), size=n)
b = DeviceBatch([distinct[j] for j in order], [65536] * n)
b.decode(check=0, lanes=65); torch.cuda.synchronize()
b.decode(check=0, lanes=65); torch.cuda.synchronize()
r = b.records()
print("rc ok:", (r["rc"] == 0).all(), "rc counts", np.bincount(r["rc"].astype(np.int64)))
host = b.dst.cpu().numpy()
offs = np.asarray(b.host_dst_off[:n], dtype=np.int64)
step = max(1, n // 4096)
sel = offs[::step]
prof = np.stack([host[o:o + 96].view(np.uint64) for o in sel]).astype(np.float64)
names = ["header", "command symbol + lengths", "literals", "distance", "complete (wait + store)", "issue copy/dict", "stream total", "loop setup"]
m = prof.mean(0)
for k in range(8):
    print("  %-28s %12.0f cycles/stream  %5.1f%%" % (names[k], m[k], 100 * m[k] / m[6]))
cnt = m[8:12]
print("per stream: commands %.0f literals %.0f copies %.0f dictionary words %.0f" % tuple(cnt))
print("cycles per command symbol %.0f, per literal %.0f, per distance %.0f, per complete %.0f, per issue %.0f" %
      (m[1] / cnt[0], m[2] / max(cnt[1], 1), m[3] / cnt[0], m[4] / max(cnt[2] + cnt[3], 1), m[5] / max(cnt[2] + cnt[3], 1)))
