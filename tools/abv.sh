#!/bin/bash
# usage: tools/abv.sh "<variant or ''> [bench flags]" ...   like tools/ab.sh, the first word names a library variant
# (brotli_amd/libyeast_<variant>.so, built by tools/build_variant.sh; "-" = the product library)
for cfg in "$@"; do
  v="${cfg%% *}"; rest="${cfg#* }"; [ "$rest" = "$cfg" ] && rest=""
  lib="$PWD/brotli_amd/libyeast_mi355x.so"; [ "$v" != "-" ] && lib="$PWD/brotli_amd/libyeast_$v.so"
  printf "%-40s " "[$cfg]"
  YEAST_LIB=$lib timeout -k 10 200 python3 bench.py --steps 3 --cpu-sample 0 $rest 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.2f  step_ms %.2f  GB/s %.2f  %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['value'], '' if d['verified'] else 'NOT VERIFIED'))" || exit 1
done
