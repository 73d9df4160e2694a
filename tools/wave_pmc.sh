#!/bin/bash
# usage: tools/wave_pmc.sh <tag> [streams]   instruction and stall counters of k_decode_wave (runs on the GPU box)
TAG=$1; N=${2:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/$TAG/w1 -- python3 $R/tools/wave_one.py $N > $R/gpurun_out/$TAG/w1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH --output-format csv -d $R/gpurun_out/$TAG/w2 -- python3 $R/tools/wave_one.py $N > $R/gpurun_out/$TAG/w2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
for d in ("w1", "w2"):
    for f in glob.glob("$R/gpurun_out/$TAG/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "k_decode_wave" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
        for k, v in acc.items():
            print("%-24s %16.0f  (two launches)" % (k, v))
PY
