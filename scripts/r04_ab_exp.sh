#!/bin/bash
# timing experiments of continuity_wave.hip with parts compiled out (results wrong on purpose): bash scripts/r04_ab_exp.sh "<flags A>" "<flags B>" ...
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
  PROF_MODES=plain,adjust,bt_cont timeout 100 python scripts/prof_continuity.py 2>&1 | grep '^lds'
done
touch mom6_amd/csrc/continuity_wave.hip
