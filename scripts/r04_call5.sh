#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
echo skip-tests
cd /tmp
for m in rccl_self local_wrap; do
rm -rf $OUT/trace_$m
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$m -o t -- python $ROOT/scripts/prof_tile.py $m 10 > $OUT/trace_$m.log 2>&1
f=$(find $OUT/trace_$m -name "*kernel_trace.csv" | head -1)
head -2 $f | cut -c1-600
python $ROOT/scripts/trace_gaps.py $f 10 > $OUT/trace_gaps_$m.txt
rm -f $f
done
