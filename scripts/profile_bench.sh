#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box: (1) kernel trace + stats, (2) FETCH_SIZE pass, (3) WRITE_SIZE pass.
# Counters are collected in their own runs (no sys/hip/hsa tracing next to --pmc).  Outputs: gpurun_out/prof_*/
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1
cd $ROOT
find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -type f | head -20
tail -2 $OUT/prof_stats.log | cut -c1-300
