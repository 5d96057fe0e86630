#!/bin/bash
# round-5 evidence in one call: rocprofv3 stats + FETCH / WRITE / SQ passes of the headline (scripts/profile_bench.sh), condensed under
# gpurun_out/r05_${TAG:-v2}_*; then the default bench line (the driver's command)
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq
PMC_TIMEOUT=200 bash scripts/profile_bench.sh all 2>&1 | tail -6
find $OUT/prof_stats -name "*kernel_trace*" -delete
python scripts/rocprof_summary.py $OUT $OUT/r05_${TAG:-v2} 2>&1 | tail -3
grep "^{\"metric" $OUT/prof_stats.log > $OUT/r05_${TAG:-v2}_bench_under_rocprof.json
python scripts/pmc_summary.py prof_sq > $OUT/r05_${TAG:-v2}_sq_summary.txt 2>&1
( time python bench.py --steps 20 --warmup 5 > $OUT/r05_${TAG:-v2}_bench.json 2> $OUT/r05_${TAG:-v2}_bench.err ) 2> $OUT/r05_${TAG:-v2}_bench.time
ls -la $OUT | grep r05_${TAG:-v2} | head -20
