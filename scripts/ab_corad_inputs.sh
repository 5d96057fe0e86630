#!/bin/bash
# A/B of CorAdCalc's default kernel (k_corad_lds) against k_corad_fused<LEAN> (MOM6X_CORAD_INPUTS=global) in ONE box
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for r in 1 2; do for m in lds global; do
  MOM6X_CORAD_INPUTS=$m python bench.py --steps 8 --warmup 2 --no-config4 --no-comm-model --no-cpu-baseline --no-pmc ${AB_ARGS:-} 2>/dev/null \
   | python -c "import json,sys; o=json.loads(sys.stdin.readline()); k=o['kernel_ms_per_step']; print('inputs=$m', 'ms_per_step', round(o['ms_per_step'],2), {a:k[a] for a in k if 'corad' in a})"
done; done 2>&1 | tee gpurun_out/ab_corad_inputs.log
