#!/bin/bash
# usage: tools/calib.sh   FETCH_SIZE / TCC_MISS of tools/ubench/fetch_calib per access pattern (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03/calib
for mode in 0 1 2; do
  for ctr in FETCH_SIZE TCC_MISS_sum; do
    timeout -s KILL 60 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/r03/calib/m${mode}_$ctr -- $R/tools/ubench/fetch_calib $mode > $R/gpurun_out/r03/calib/m${mode}_$ctr.log 2>&1 || exit 1
  done
done
python3 - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03/calib"
for mode in (0, 1, 2):
    row = {}
    for ctr in ("FETCH_SIZE", "TCC_MISS_sum"):
        for f in glob.glob("%s/m%d_%s/**/*counter_collection.csv" % (root, mode, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_lines" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    row[ctr] = float(r["Counter_Value"])
    want = (1 << 24) * 64
    print("mode %d: bytes touched %d; FETCH_SIZE %.0f KiB = %.0f bytes = %.3f x; TCC_MISS %.0f = %.3f per line" %
          (mode, want, row.get("FETCH_SIZE", 0), row.get("FETCH_SIZE", 0) * 1024, row.get("FETCH_SIZE", 0) * 1024 / want,
           row.get("TCC_MISS_sum", 0), row.get("TCC_MISS_sum", 0) / (1 << 24)))
PY
