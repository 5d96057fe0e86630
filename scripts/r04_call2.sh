#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_c2.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu_c2.log
cd /tmp
for m in local_wrap rccl_self; do
  for r in 0 16; do
    MOM6X_MFW_ROWS=$r timeout 200 python $ROOT/scripts/prof_tile.py $m 20 2>&1 | grep ms_per_step | sed "s/^/rows=$r /"
  done
done
rm -rf $OUT/tile2_local_wrap
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tile2_local_wrap -o t -- python $ROOT/scripts/prof_tile.py local_wrap 10 > $OUT/tile2_local_wrap.log 2>&1
find $OUT/tile2_local_wrap -name "*kernel_trace*" -delete
