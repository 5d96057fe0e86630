#!/bin/bash
export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
run() { MOM6X_BT_SUBSTEP=$1 python bench.py --steps 8 --warmup 2 --no-config4 --no-comm-model --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],2), {a:b for a,b in k.items() if 'bt_' in a})"; }
MOM6X_BT_SUBSTEP=pair python -m pytest tests/test_barotropic_gpu.py tests/test_rk2_gpu.py -x -q -m gpu -k "barotropic or double_gyre_bitexact or 75_layers_on_chip" --tb=short 2>&1 | tail -4
run kernels; run pair; run kernels; run pair
