#!/bin/bash
# usage: tools/sweep.sh "<lanes list>" "<blocks/cu list>" streams [extra bench.py flags]
for L in $1; do for B in $2; do
  printf "lanes %s blocks/cu %s %s: " $L $B "$4"
  YEAST_BLOCKS_PER_CU=$B timeout -k 10 120 python3 bench.py --streams $3 --steps 2 --cpu-sample 0 --lanes $L --check 0 $4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms %.2f  GB/s %.2f' % (d['roofline']['kernel_ms'], d['value']))" || exit 1
done; done
