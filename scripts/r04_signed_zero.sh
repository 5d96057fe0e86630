#!/bin/bash
# A/B of continuity_wave.hip's per-file flags: the sign of zeros against the oracle, and the kernel time
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT
for flags in "-ffinite-math-only -fno-signed-zeros" "-ffinite-math-only" ""; do
  echo "=== flags: [$flags]"
  touch mom6_amd/csrc/continuity_wave.hip
  MOM6X_WAVE_FLAGS="$flags" python -m mom6_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python - <<'PY'
import json
r=json.load(open('mom6_amd/lib/kernel_resources.json'))['continuity_wave.hip']
for k,v in r.items():
    if 'Li5ELb0' in k: print(k[-40:], v['vgprs'], v['scratch'])
PY
  MOM6X_TEST_SIGNED_ZERO=strict timeout 600 python -m pytest tests/test_continuity_gpu.py tests/test_restart_gpu.py -q 2>&1 | grep -E "zeros of opposite|passed|failed" | sort | uniq -c | sort -rn | head -8
  PROF_MODES=adjust,bt_cont timeout 100 python scripts/prof_continuity.py 2>&1 | grep '^lds'
done
touch mom6_amd/csrc/continuity_wave.hip
