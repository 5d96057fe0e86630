#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_barotropic_gpu.py tests/test_rk2_gpu.py tests/test_layout_gpu.py tests/test_bench_layout_gpu.py tests/test_restart_gpu.py tests/test_step_mom_gpu.py -x -q > $OUT/pytest_gpu_c6.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_c6.log
cd /tmp
for m in local_wrap rccl_self; do
  for w in kernels fused; do
    MOM6X_BT_SUBSTEP=$w timeout 200 python $ROOT/scripts/prof_tile.py $m 20 2>&1 | grep ms_per_step | sed "s/^/substep=$w /"
  done
done
cd $ROOT
for w in kernels fused; do
  MOM6X_BT_SUBSTEP=$w timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc --breakdown > $OUT/bench_sub_$w.json 2> $OUT/bench_sub_$w.err
  python -c "
import json,sys
j=json.loads(open('$OUT/bench_sub_$w.json').read().strip().splitlines()[-1]); print('$w', j['ms_per_step'], {k:v for k,v in j['kernel_ms_per_step'].items() if 'bt_' in k})"
  grep -i "k_bt_" $OUT/bench_sub_$w.err | head -8
done
