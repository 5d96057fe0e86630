#!/bin/bash
# A/B of continuity_wave.hip build variants on the GPU box (round 6): bash scripts/r06_ab_mfw.sh "<cflags>|<env>|<ni nj nk>" ...
# one build per variant, two repetitions of scripts/prof_continuity.py at the given size (default 1440 1080 75)
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
MODES=${PROF_MODES:-bt_cont,full,adjust}
for spec in "$@"; do
  IFS='|' read -r fl ev sz <<< "$spec"
  echo "=== variant cflags=[$fl] env=[$ev] size=[${sz:-1440 1080 75}]"
  touch mom6_amd/csrc/continuity_wave.hip
  MOM6X_CFLAGS="$fl" python -m mom6_amd.build > /dev/null 2>$OUT/ab_build.err || { echo BUILD FAILED; tail -5 $OUT/ab_build.err; }
  for rep in 1 2; do
    env $ev PROF_MODES=$MODES timeout 200 python scripts/prof_continuity.py ${sz:-1440 1080 75} 2>&1 | grep -E '^lds|phases|re-evaluations|one-way|Error|error'
  done
done
