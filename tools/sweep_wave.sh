#!/bin/bash
# usage: tools/sweep_wave.sh "<waves/CU list>" "<root lit list>" "<root iac list>" streams
for W in $1; do for RL in $2; do for RI in $3; do
  printf "waves/cu %s root lit %s iac %s: " $W $RL $RI
  YEAST_WAVES_PER_CU=$W YEAST_ROOT_LIT=$RL YEAST_ROOT_IAC=$RI YEAST_RETRY_REPORT=1 timeout -k 10 120 python3 bench.py --streams $4 --steps 2 --cpu-sample 0 --lanes 65 --check 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.2f  GB/s %.2f' % (d['roofline']['kernel_ms'], d['value']))" || exit 1
done; done; done
