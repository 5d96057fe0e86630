#!/bin/bash
# k_bt_substep2 (two barotropic sub-steps per launch) against the three kernels per sub-step at 1440 x 1080 x 75 (round 5)
export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
run() { MOM6X_BT_SUBSTEP=$1 python bench.py --steps 8 --warmup 2 --no-config4 --no-comm-model --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$1 $2', round(d['ms_per_step'],2), {a:b for a,b in k.items() if 'bt_' in a})"; }
for bsy in "$@"; do
  touch mom6_amd/csrc/barotropic.hip
  MOM6X_CFLAGS="-DMOM6X_BT_PAIR_BSY=$bsy" python -m mom6_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  run kernels "(BSY=$bsy build)"; run pair "BSY=$bsy"; run pair "BSY=$bsy"
done
touch mom6_amd/csrc/barotropic.hip; python -m mom6_amd.build > /dev/null 2>&1
