#!/bin/bash
# rocprofv3 kernel trace + stats of bench.py on the GPU box (no counters).  Output: gpurun_out/prof_stats/
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
export MOM6X_BENCH_NO_PMC=1   # (bench.py would otherwise start its own rocprofv3 passes for roofline.traffic)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
rm -rf $OUT/prof_stats
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
cd $ROOT
find $OUT/prof_stats -name "*kernel_trace*" -delete     # the raw trace is large; the stats are what is kept
find $OUT/prof_stats -type f | head
tail -1 $OUT/prof_stats.log | cut -c1-400
