#!/bin/bash
# usage: tools/pmc_ws.sh <tag> "<counters>" <streams> [bench flags]
TAG=$1; CTRS=$2; N=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
timeout -s KILL 150 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/$TAG/pmc -- python3 $R/bench.py --streams $N --steps 1 --warmup 1 --cpu-sample 0 --check 0 "$@" > $R/gpurun_out/$TAG/bench.log 2>&1
echo pmc exit $?
