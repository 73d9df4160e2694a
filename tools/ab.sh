#!/bin/bash
# usage: tools/ab.sh "<env settings> [-- <bench.py flags>]" ...   one short bench.py run per argument
# (runs on the GPU box); prints the decode kernel time and the end-to-end rate of each.  "" = defaults.
for cfg in "$@"; do
  printf "%-56s " "[$cfg]"
  envs="${cfg%%--*}"; args=""
  case "$cfg" in *--*) args="--${cfg#*--}";; esac
  env $envs timeout -k 10 200 python3 bench.py --steps 3 --cpu-sample 0 $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.2f  step_ms %.2f  GB/s %.2f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['value']))" || exit 1
done
