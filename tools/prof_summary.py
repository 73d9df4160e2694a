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

# HBM traffic of the dominant kernel per launch, for bench.py's roofline.traffic, as MI355X_MICROARCH.md (HBM section)
# prescribes: FETCH_SIZE and WRITE_SIZE from separate --pmc passes, KiB x 1024.  The guide's gfx950 correction (FETCH_SIZE
# reports half the bytes) is for wide streaming reads; it says to calibrate other patterns.  tools/ubench/fetch_calib.hip
# (profiles/r03/fetch_calibration.txt) did: scattered 2-byte and 16-byte loads, one per 64-byte line -- this kernel's table
# lookups and copy pieces -- are counted exactly (x 1.000, one 64-byte fetch per L2 miss), the streaming read x 0.5.  So
# the factor here is 1, and the x 2 figure is kept beside it as the upper bound round 2 quoted.
import json, re
tot, kname = {}, None
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and "k_decode_lean" in r["Kernel_Name"]]
        if rows:
            tot[name] = sum(float(r["Counter_Value"]) for r in rows) / len(rows) * 1024.0
            kname = re.sub(r"^void |\(.*$", "", rows[0]["Kernel_Name"])
if len(tot) == 2 and len(sys.argv) > 2:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    out = {"streams": int(sys.argv[2]), "kernel": kname, "csrc": bench.kernel_sources_hash(), "fetch_bytes": tot["FETCH_SIZE"], "fetch_factor": 1.0,
           "fetch_bytes_if_streaming_x2": 2 * tot["FETCH_SIZE"],
           "write_bytes": tot["WRITE_SIZE"], "bytes_per_launch": tot["FETCH_SIZE"] + tot["WRITE_SIZE"],
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB x 1024, mean over launches; FETCH_SIZE x 1: "
                     "calibrated on scattered 2- and 16-byte loads (profiles/r03/fetch_calibration.txt)"}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "hbm_traffic.json"), "w"), indent=1)
    # the ceilings file follows the same counters (lines = bytes / 64); its trip and instruction counts are edited by hand
    cf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r03", "ceilings.json")
    if os.path.exists(cf):
        c = json.load(open(cf))
        c["kernel"], c["streams"], c["csrc"] = kname, int(sys.argv[2]), out["csrc"]
        c["hbm_lines"]["lines_per_launch"] = out["bytes_per_launch"] / 64.0
        json.dump(c, open(cf, "w"), indent=1)
    print("hbm_traffic.json:", out)
