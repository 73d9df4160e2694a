#!/bin/bash
# usage: tools/hitrate.sh "<env settings>" ...   L2 hit rate, read requests and mean read latency of the batch kernel
# for each setting (one counter pass each; runs on the GPU box)
k=0
for cfg in "$@"; do
  k=$((k+1))
  export $cfg >/dev/null 2>&1
  tools/pmc_once.sh hr$k "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" >/dev/null 2>&1
  for v in $cfg; do unset ${v%%=*}; done
  python3 - "$cfg" $k <<'PY'
import csv,sys,glob,os,collections
cfg,k=sys.argv[1],sys.argv[2]
root=os.environ.get("GRAFT_REPO_ROOT",".")
agg=collections.defaultdict(float); n=0
for f in glob.glob("%s/gpurun_out/hr%s/pmc/**/*counter_collection.csv"%(root,k), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_decode_lean' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
h,m=agg['TCC_HIT_sum'],agg['TCC_MISS_sum']
rq,lat=agg['TCP_TCC_READ_REQ_sum'],agg['TCP_TCC_READ_REQ_LATENCY_sum']
print("%-52s L2 hit rate %.3f  read requests %.3g  mean read latency %.0f cycles" % ("["+cfg+"]", h/(h+m) if h+m else 0, rq, lat/rq if rq else 0))
PY
done
