#!/bin/bash
# A/B of a run-time switch of the mass-flux kernel inside one call: bash scripts/r04_ab_env.sh VAR val1 val2 ...
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
VAR=$1; shift
for val in "$@"; do
  echo "=== $VAR=$val"
  export $VAR=$val
  PROF_MODES=adjust,bt_cont timeout 100 python scripts/prof_continuity.py 2>&1 | grep '^lds'
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc --tracers -1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench dyn-only ms/step', round(j['ms_per_step'],2), {k:v for k,v in j['kernel_ms_per_step'].items() if 'mass_flux' in k}, 'x avg launch', j['roofline']['avg_launch_ms'])"
done
