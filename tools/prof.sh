#!/bin/bash
# usage: tools/prof.sh <tag> <lanes> <streams>   (runs on the GPU box)
set -e
TAG=$1; L=$2; N=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
# the kernel trace is of the default bench command (what the driver runs)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/kt -- python3 $R/bench.py > $R/gpurun_out/$TAG/bench_kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/$TAG/pmc1 -- python3 $R/bench.py --streams $N --steps 1 --warmup 1 --cpu-sample 0 --lanes $L --check 0 > $R/gpurun_out/$TAG/bench_pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/$TAG/pmc2 -- python3 $R/bench.py --streams $N --steps 1 --warmup 1 --cpu-sample 0 --lanes $L --check 0 > $R/gpurun_out/$TAG/bench_pmc2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $R/gpurun_out/$TAG/pmc3 -- python3 $R/bench.py --streams $N --steps 1 --warmup 1 --cpu-sample 0 --lanes $L --check 0 > $R/gpurun_out/$TAG/bench_pmc3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc4 -- python3 $R/bench.py --streams $N --steps 1 --warmup 1 --cpu-sample 0 --lanes $L --check 0 > $R/gpurun_out/$TAG/bench_pmc4.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc5 -- python3 $R/bench.py --streams $N --steps 1 --warmup 1 --cpu-sample 0 --lanes $L --check 0 > $R/gpurun_out/$TAG/bench_pmc5.log 2>&1
echo done
