#!/bin/bash
# A/B of continuity_wave.hip build variants on the GPU box: bash scripts/r05_ab_mfw.sh "<cflags>|<env>" ...  (round 5)
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
MODES=${PROF_MODES:-plain,bt_cont,full,adjust}
for spec in "$@"; do
  fl="${spec%%|*}"; ev="${spec#*|}"; [ "$ev" = "$spec" ] && ev=""
  echo "=== variant cflags=[$fl] env=[$ev]"
  touch mom6_amd/csrc/continuity_wave.hip
  MOM6X_CFLAGS="$fl" python -m mom6_amd.build > /dev/null 2>$OUT/ab_build.err || { echo BUILD FAILED; tail -5 $OUT/ab_build.err; }
  for rep in 1 2; do
    env $ev PROF_MODES=$MODES timeout 200 python scripts/prof_continuity.py 2>&1 | grep -E '^lds|phases|re-evaluations|one-way|Error|error'
  done
done
touch mom6_amd/csrc/continuity_wave.hip
python -m mom6_amd.build > /dev/null 2>&1
