#!/bin/bash
# usage: tools/prof_all.sh <tag>   everything profiles/<tag>/ holds: kernel trace of the default bench, five PMC passes,
# the stream-count series and the section profile of the instrumented build (runs on the GPU box, ~5 minutes)
TAG=$1
tools/prof.sh $TAG 0 65536 || exit 1
tools/series.sh 8192 16384 32768 49152 65536 98304 131072 262144 > gpurun_out/$TAG/series.txt 2>&1 || exit 1
if [ -f brotli_amd/libyeast_prof.so ]; then
  (YEAST_LIB=$PWD/brotli_amd/libyeast_prof.so timeout -k 10 200 python3 tools/lean_prof.py 2048 32; YEAST_LIB=$PWD/brotli_amd/libyeast_prof.so timeout -k 10 200 python3 tools/lean_prof.py 65536 32) > gpurun_out/$TAG/lean_profile.txt 2>&1
fi
echo profiles done
