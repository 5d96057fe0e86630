#!/bin/bash
# A/B of a compile-time variant of continuity_wave.hip: bash scripts/r04_ab_branchy.sh "<flag A>" "<flag B>" ...
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
for fl in "$@"; do
  echo "=== variant [$fl]"
  touch mom6_amd/csrc/continuity_wave.hip
  MOM6X_CFLAGS="$fl" python -m mom6_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python - <<'PY'
import json
r=json.load(open('mom6_amd/lib/kernel_resources.json'))['continuity_wave.hip']
for k,v in r.items():
    if 'Li5ELb0' in k: print(k[28:48], "vgprs", v['vgprs'], "scratch", v['scratch'], "sgpr_spill", v.get('sgpr_spill'))
PY
  timeout 600 python -m pytest tests/test_continuity_gpu.py -q -x 2>&1 | tail -1
  PROF_MODES=adjust,bt_cont timeout 100 python scripts/prof_continuity.py 2>&1 | grep '^lds'
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc --tracers -1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench dyn-only ms/step', round(j['ms_per_step'],2), {k:v for k,v in j['kernel_ms_per_step'].items() if 'mass_flux' in k}, 'x avg launch', j['roofline']['avg_launch_ms'])"
done
touch mom6_amd/csrc/continuity_wave.hip
