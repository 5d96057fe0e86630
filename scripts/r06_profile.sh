#!/bin/bash
# round-6 evidence in one call: rocprofv3 stats + FETCH / WRITE / SQ passes of the headline (scripts/profile_bench.sh), condensed under
# gpurun_out/r06_${TAG:-final}_*; then the default bench line (the driver's command)
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT; T=${TAG:-final}
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq
PMC_TIMEOUT=200 bash scripts/profile_bench.sh all 2>&1 | tail -6
find $OUT/prof_stats -name "*kernel_trace*" -delete
python scripts/rocprof_summary.py $OUT $OUT/r06_${T} 2>&1 | tail -3
grep "^{\"metric" $OUT/prof_stats.log > $OUT/r06_${T}_bench_under_rocprof.json
python scripts/pmc_summary.py prof_sq > $OUT/r06_${T}_sq_summary.txt 2>&1
( time python bench.py > $OUT/r06_${T}_bench.json 2> $OUT/r06_${T}_bench.err ) 2> $OUT/r06_${T}_bench.time
ls -la $OUT | grep r06_${T} | head -20
