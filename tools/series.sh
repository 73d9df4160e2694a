#!/bin/bash
# usage: tools/series.sh <stream counts...>   kernel time of the batch kernel over batch sizes (GPU box)
for n in "$@"; do
  printf "%-10s " "$n"
  timeout -k 10 300 python3 bench.py --steps 3 --cpu-sample 0 --streams "$n" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel %s %.2f ms  step_ms %.2f  GB/s %.2f' % (d['roofline']['kernel'], d['roofline']['kernel_ms'], d['ms_per_step'], d['value']))" || exit 1
done
