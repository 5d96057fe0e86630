#!/bin/bash
# A/B of dyn_kernels.hip build variants (the one-kernel vertical viscosity) on the GPU box in the benchmark's state: bash scripts/r05_ab_vv.sh "<cflags>|<env>" ...  (round 5)
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT
for spec in "$@"; do
  IFS='|' read -r fl ev <<< "$spec"
  echo "=== variant cflags=[$fl] env=[$ev]"
  touch mom6_amd/csrc/dyn_kernels.hip
  MOM6X_CFLAGS="$fl" python -m mom6_amd.build > /dev/null 2>$OUT/ab_build.err || { echo BUILD FAILED; tail -5 $OUT/ab_build.err; }
  python - <<'PY'
import json
r=json.load(open('mom6_amd/lib/kernel_resources.json'))
print('   spills:', sorted({(v['vgpr_spill'], v['scratch']) for f,ks in r.items() for k,v in ks.items() if 'coef_cols' in k}))
PY
  for rep in 1 2; do
    env $ev timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc --no-config4 --no-comm-model --breakdown 2>$OUT/ab_breakdown.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('ms_per_step', j['ms_per_step'])"
    grep -E 'k_vertvisc_' $OUT/ab_breakdown.txt | grep -v thermo
  done
done
touch mom6_amd/csrc/dyn_kernels.hip
python -m mom6_amd.build > /dev/null 2>&1
