#!/usr/bin/env python3
"""Phase timing of the batch kernel (library built with -DYS_PROFILE): per-stream cycle counts of
header parsing, command loop and queue drains, packed into record.check (check kind 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H
import brotli_amd as ba
from brotli_amd.device import DeviceBatch
meta, distinct = H.packed_streams("cfg2_streams")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
order = np.random.RandomState(3).randint(0, len(distinct), size=n)
b = DeviceBatch([distinct[j] for j in order], [65536] * n)
b.decode(check=0, lanes=lanes); torch.cuda.synchronize()
b.decode(check=0, lanes=lanes); torch.cuda.synchronize()
r = b.records()
ck = r["check"]
hdr = (ck & 0xfffff).astype(np.float64) * 1024
loop = ((ck >> 20) & 0xfffff).astype(np.float64) * 1024
drain = ((ck >> 40) & 0xfffff).astype(np.float64) * 1024
print("rc ok:", (r["rc"] == 0).all())
look = (r["used"] & 0xffffffff).astype(np.float64) * 1024
trips = (r["used"] >> 32).astype(np.float64)
drains = r["flags"].astype(np.float64)
print("per stream: trips mean %.0f max %.0f; drains mean %.0f; lookup wait %.0f cycles (%.1f%% of loop), per trip %.0f; drain per drain %.0f cycles; loop per trip %.0f"
      % (trips.mean(), trips.max(), drains.mean(), look.mean(), 100 * look.mean() / loop.mean(), look.mean() / trips.mean(), drain.mean() / max(drains.mean(), 1), loop.mean() / trips.mean()))
print("cycles/stream (constant 100 MHz counter?) mean: header %.0f  loop %.0f  of which drain %.0f" % (hdr.mean(), loop.mean(), drain.mean()))
print("header share %.1f%%, drain share of loop %.1f%%" % (100 * hdr.mean() / (hdr.mean() + loop.mean()), 100 * drain.mean() / loop.mean()))
g = r["got"]
print("per stream at drain triggers: need %.0f  queue full %.0f;  uppercase words %.0f; drain load phases %.0f cycles (text streams: need %.0f; markov: need %.0f)"
      % ((g & 0xfff).mean(), ((g >> 12) & 0xfff).mean(), ((g >> 24) & 0xfff).mean(), (g >> 36).astype(np.float64).mean() * 1024,
         (g & 0xfff)[order < 192].mean(), (g & 0xfff)[order >= 192].mean()))
sec = np.zeros((n, 6))
host = b.dst.cpu().numpy()
for i in range(0, n, 64):  # lane 0 of every wavefront
    o = int(b.host_dst_off[i])
    sec[i] = host[o:o + 48].view(np.uint64)
m = sec[::64].mean(0)
names = ["top (refill, address)", "lookup wait", "decode + extra bits", "command / literal", "distance + copy (odd trips)", "bottom (trigger, next literal table)"]
tr = trips[::64].mean()
print("main-loop sections, cycles per trip (lane 0 of each wavefront; drain excluded):")
for k in range(6):
    print("  %-40s %7.0f" % (names[k], m[k] / tr))
