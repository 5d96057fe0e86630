ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
timeout 900 python -m pytest tests/test_tracer_gpu.py tests/test_layout_gpu.py tests/test_configs_gpu.py tests/test_switches_gpu.py -q -x -k "not MFW and not BT_" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc --breakdown 2>&1 >/dev/null | grep "thermo: k_ta_\|thermo: k_tridiag" | head -4
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step', round(j['ms_per_step'],2), j['thermo']['advect_tracer_ms'], j['thermo']['advect_frac_of_hbm_peak'])"
