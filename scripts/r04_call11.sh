ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
timeout 900 python -m pytest tests/test_barotropic_gpu.py tests/test_rk2_gpu.py tests/test_layout_gpu.py tests/test_switches_gpu.py -q -x -k "not MFW and not REMAP and not TRIDIAG and not HV_KC" 2>&1 | grep -E "passed|failed|Error" | tail -3
cd /tmp
for m in local_wrap rccl_self; do
  for w in kernels fused; do
    MOM6X_BT_SUBSTEP=$w timeout 200 python $ROOT/scripts/prof_tile.py $m 30 2>&1 | grep ms_per_step | sed "s/^/substep=$w /"
  done
done
cd $ROOT
for w in kernels fused; do
  MOM6X_BT_SUBSTEP=$w timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc --tracers -1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w full size', round(j['ms_per_step'],2), {k:v for k,v in j['kernel_ms_per_step'].items() if 'bt_' in k})"
done
