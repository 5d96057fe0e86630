ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
timeout 600 python -m pytest tests/test_continuity_gpu.py tests/test_rk2_gpu.py tests/test_layout_gpu.py -q -x 2>&1 | tail -1
PROF_MODES=plain,adjust,bt_cont timeout 100 python scripts/prof_continuity.py 2>&1 | grep '^lds' | sed 's/k_convergence<DIR>=[0-9.]* //; s/lds //; s/k_mass_flux_wave//g' | tr '\n' '|'; echo
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc --tracers -1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench dyn-only ms/step', round(j['ms_per_step'],2), {k:v for k,v in j['kernel_ms_per_step'].items() if 'mass_flux' in k}, 'x avg launch', j['roofline']['avg_launch_ms'])"
