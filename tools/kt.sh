#!/bin/bash
# usage: tools/kt.sh <tag> <lanes> <streams>   kernel trace only (runs on the GPU box); env passes through
TAG=$1; L=$2; N=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/kt -- python3 $R/bench.py --streams $N --steps 2 --warmup 1 --cpu-sample 0 --lanes $L --check 0 > $R/gpurun_out/$TAG/bench_kt.log 2>&1
echo exit $?
find $R/gpurun_out/$TAG/kt -name "*kernel_stats.csv" | xargs cat
