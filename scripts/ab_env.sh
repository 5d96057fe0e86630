#!/bin/bash
# A/B of one environment switch in ONE box: scripts/ab_env.sh VAR valueA valueB [bench args]  (an empty value = unset)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
var=$1; a=$2; b=$3; shift 3
for r in 1 2; do for val in "$a" "$b"; do
  env $var="$val" python bench.py --steps 8 --warmup 2 --no-config4 --no-comm-model --no-cpu-baseline --no-pmc "$@" 2>/dev/null \
   | python -c "import json,sys; o=json.loads(sys.stdin.readline()); k=o['kernel_ms_per_step']; print('$var=$val', 'ms_per_step', round(o['ms_per_step'],2), 'kernel_sum', round(o['kernel_sum_ms'],2), {a:k[a] for a in k if 'bt_col' in a or 'btcalc' in a})"
done; done 2>&1 | tee gpurun_out/ab_env.log
