#!/bin/sh
# Build libyeast_mi355x.so (gfx950 only) in-tree.  Usage: brotli_amd/csrc/build.sh
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/.."
DATA="$OUT/data"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -enable-ipra=false: with inter-procedural register allocation (the AMDGPU default) the caller of an out-of-line device
# function keeps values in registers the callee is recorded not to touch; in the bounds-checked build (tools/build_bounds.sh)
# the cooperative kernel came back from wave_copy_units() with two of its scalars holding other variables' values (the
# stream "overlaps" of BASELINE config 4 decoded wrongly; with the flag it decodes).  The product build showed no such
# failure, but nothing says it could not: every build uses the standard calling convention.  No measurable cost
# (45.8 ms / 39.9 ms / 17.1 ms at 65,536 / 8,192 / 2,048 streams either way).
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -enable-ipra=false"
cd "$HERE"
$HIPCC $FLAGS -c yeast_decode.hip -o yeast_decode.o ${YEAST_EXTRA_FLAGS} &
P1=$!
$HIPCC $FLAGS -c yeast_check.hip -o yeast_check.o ${YEAST_EXTRA_FLAGS} &
P2=$!
$HIPCC $FLAGS -c yeast_wave.hip -o yeast_wave.o ${YEAST_EXTRA_FLAGS} &
P4=$!
# the batch kernel's loop is bound by the latency of dependent instructions of one wavefront: schedule it for
# instruction-level parallelism (43.4 -> 42.1 ms at 8,192 streams; the full batch within noise)
$HIPCC $FLAGS -mllvm -amdgpu-sched-strategy=max-ilp -c yeast_lane.hip -o yeast_lane.o ${YEAST_EXTRA_FLAGS} &
P5=$!
$HIPCC $FLAGS -c yeast_host.cpp -o yeast_host.o ${YEAST_EXTRA_FLAGS} &
P3=$!
$HIPCC $FLAGS -c yeast_br.cpp -o yeast_br.o &
P6=$!
gcc -O2 -fPIC -DYS_DATA_DIR="\"$DATA\"" -c yeast_data.c -o yeast_data.o
wait $P1
wait $P2
wait $P3
wait $P4
wait $P5
wait $P6
$HIPCC -shared -fPIC -pthread --offload-arch=gfx950 -o "$OUT/libyeast_mi355x.so" yeast_decode.o yeast_wave.o yeast_lane.o yeast_check.o yeast_host.o yeast_br.o yeast_data.o
echo "built $OUT/libyeast_mi355x.so"
