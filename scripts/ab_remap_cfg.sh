#!/bin/bash
# A/B of the remap merge's compile-time OM4 instantiation (MOM6X_REMAP_CFG=0: the generic kernel) in ONE box.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for r in 1 2; do for cfg in 1 0; do
  MOM6X_REMAP_CFG=$cfg python bench.py --steps 4 --warmup 1 --no-config4 --no-comm-model --no-cpu-baseline --no-pmc 2>/dev/null \
   | python -c "import json,sys; o=json.loads(sys.stdin.readline()); a=o['ale_remap_leg']; print('cfg=$cfg', 'ms_per_step', o['ms_per_step'], {k: a[k] for k in a if 'ms' in k or 'frac' in k})"
done; done 2>&1 | tee gpurun_out/ab_remap_cfg.log
