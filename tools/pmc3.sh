#!/bin/bash
# usage: tools/pmc3.sh <tag> [bench flags]  three counter passes of bench.py (runs on the GPU box)
TAG=$1; shift
tools/pmc_once.sh ${TAG}a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "$@" || exit 1
tools/pmc_once.sh ${TAG}b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" "$@" || exit 1
tools/pmc_once.sh ${TAG}c "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "$@" || exit 1
python3 - $TAG <<'PY'
import csv,sys,collections,glob,os
tag=sys.argv[1]
root=os.environ.get("GRAFT_REPO_ROOT",".")
for sfx in "abc":
    for f in glob.glob("%s/gpurun_out/%s%s/pmc/**/*counter_collection.csv"%(root,tag,sfx), recursive=True):
        agg=collections.defaultdict(float); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'k_decode' in k and ('lean' in k or 'flat' in k or 'wave' in k):
                agg[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
        for k,v in sorted(agg.items()): print("%-34s %18.0f  per dispatch %16.0f (%d)"%(k,v,v/cnt[k],cnt[k]))
PY
