#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_restart_gpu.py tests/test_fortran_gpu.py tests/test_configs_gpu.py -x -q > $OUT/pytest_gpu_c3.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_c3.log
for r in 4 6 8 10 12 14 16 20 25 27 32 45; do
  echo "rows=$r $(MOM6X_MFW_ROWS=$r PROF_MODES=adjust,bt_cont timeout 100 python scripts/prof_continuity.py 360 540 75 2>&1 | grep '^lds' | tr '\n' ' ')"
done
