#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_restart_gpu.py tests/test_fortran_gpu.py tests/test_layout_gpu.py tests/test_bench_layout_gpu.py tests/test_halo_gpu.py tests/test_rk2_gpu.py -x -q > $OUT/pytest_gpu_c4.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_c4.log
for r in 8 12 16 24; do
  echo "full rows=$r $(MOM6X_MFW_ROWS=$r PROF_MODES=adjust,bt_cont timeout 100 python scripts/prof_continuity.py 2>&1 | grep '^lds' | tr '\n' ' ')"
done
cd /tmp
for m in local_wrap rccl_self; do
  for w in full ref; do
    MOM6X_PASS_WIDTHS=$w timeout 200 python $ROOT/scripts/prof_tile.py $m 20 2>&1 | grep ms_per_step | sed "s/^/widths=$w /"
  done
done
