#!/usr/bin/env python3
"""One stream of BASELINE config 2 through the wavefront-per-stream kernel, twice (for counter runs: rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
import brotli_amd as ba
from brotli_amd.device import DeviceBatch
meta, distinct = H.packed_streams("cfg2_streams")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
order = np.random.RandomState(3).randint(0, len(distinct), size=n)
b = DeviceBatch([distinct[j] for j in order], [65536] * n)
for _ in range(2):
    b.decode(0, 65); torch.cuda.synchronize()
assert (b.records()["rc"] == 0).all()
