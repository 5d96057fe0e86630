#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
for t in 0 1 2 3; do
  MOM6X_BT_TILE=$t timeout 200 python $ROOT/scripts/prof_tile.py local_wrap 20 2>&1 | grep ms_per_step | sed "s/^/tile=$t /"
done
MOM6X_BT_SUBSTEP=kernels timeout 200 python $ROOT/scripts/prof_tile.py local_wrap 20 2>&1 | grep ms_per_step | sed "s/^/kernels /"
