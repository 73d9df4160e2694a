#!/usr/bin/env python3
"""Summarise rocprofv3 csv output directories produced by tools/prof.sh."""
import csv, glob, os, sys, collections
base = sys.argv[1]
for f in sorted(glob.glob(os.path.join(base, "**", "*kernel_stats.csv"), recursive=True)):
    print("==", f)
    print(open(f).read())
for f in sorted(glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    print("==", f)
    for k, v in agg.items():
        print(k, {a: b for a, b in v.items()})

# HBM traffic of the dominant kernel per launch, for bench.py's roofline.traffic, as MI355X_MICROARCH.md
# (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE from separate --pmc passes, KiB x 1024, and on
# gfx950 FETCH_SIZE doubled (it tallies 128-byte requests at 64 bytes).  The guide calibrated the factor
# on wide streaming reads; this kernel's reads are scattered, so the doubled figure is an upper bound
# and the raw one is kept beside it.
import json
tot = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r["Counter_Name"] == name and "k_decode_lean" in r["Kernel_Name"]]
        if vals:
            tot[name] = sum(vals) / len(vals) * 1024.0
if len(tot) == 2 and len(sys.argv) > 2:
    out = {"streams": int(sys.argv[2]), "kernel": "k_decode_lean<32, 2>", "fetch_bytes": tot["FETCH_SIZE"],
           "fetch_bytes_gfx950_corrected": 2 * tot["FETCH_SIZE"],
           "write_bytes": tot["WRITE_SIZE"], "bytes_per_launch": 2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"],
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB x 1024, FETCH_SIZE x 2 "
                     "(gfx950), mean over launches"}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "hbm_traffic.json"), "w"), indent=1)
    print("hbm_traffic.json:", out)
