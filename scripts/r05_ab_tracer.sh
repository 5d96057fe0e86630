#!/bin/bash
# A/B of tracer.hip build variants on the GPU box in the BENCHMARK's state: bash scripts/r05_ab_tracer.sh "<cflags>|<env>|<test -k>" ...  (round 5)
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT
for spec in "$@"; do
  IFS='|' read -r fl ev tst <<< "$spec"
  echo "=== variant cflags=[$fl] env=[$ev]"
  touch mom6_amd/csrc/tracer.hip
  MOM6X_CFLAGS="$fl" python -m mom6_amd.build > /dev/null 2>$OUT/ab_build.err || { echo BUILD FAILED; tail -5 $OUT/ab_build.err; }
  if [ -n "$tst" ]; then
    env $ev timeout 900 python -m pytest tests/test_tracer_gpu.py -q -x -m gpu -k "$tst" -p no:cacheprovider > $OUT/ab_tracer_test.log 2>&1
    grep -E "passed|failed|error" $OUT/ab_tracer_test.log | tail -3
  fi
  for rep in 1 2; do
    env $ev timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc --no-config4 --no-comm-model --breakdown 2>$OUT/ab_breakdown.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('ms_per_step', j['ms_per_step'], 'thermo', j.get('thermo_ms_per_call', j.get('config', {}).get('thermo_ms_per_call')))"
    grep -E 'k_ta_[xy]_|k_ta_save|k_ta_init' $OUT/ab_breakdown.txt
  done
done
touch mom6_amd/csrc/tracer.hip
python -m mom6_amd.build > /dev/null 2>&1
