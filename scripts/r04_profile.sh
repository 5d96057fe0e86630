#!/bin/bash
# round-4 evidence in one call: rocprofv3 stats + FETCH / WRITE / SQ passes of the headline (scripts/profile_bench.sh), condensed under
# gpurun_out/r04_${TAG:-v2}_*; the default bench line; the tile's per-kernel table and compute-stream gaps in both modes
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq
PMC_TIMEOUT=200 bash scripts/profile_bench.sh all 2>&1 | tail -6
find $OUT/prof_stats -name "*kernel_trace*" -delete
python scripts/rocprof_summary.py $OUT $OUT/r04_${TAG:-v2} 2>&1 | tail -3
grep "^{\"metric" $OUT/prof_stats.log > $OUT/r04_${TAG:-v2}_bench_under_rocprof.json
python scripts/pmc_summary.py prof_sq > $OUT/r04_${TAG:-v2}_sq_summary.txt 2>&1
( time python bench.py > $OUT/r04_${TAG:-v2}_bench.json 2> $OUT/r04_${TAG:-v2}_bench.err ) 2> $OUT/r04_${TAG:-v2}_bench.time
export MOM6X_BENCH_NO_PMC=1
cd /tmp
for m in local_wrap rccl_self; do
  rm -rf $OUT/tile_$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tile_$m -o t -- python $ROOT/scripts/prof_tile.py $m 10 > $OUT/tile_$m.log 2>&1
  f=$(find $OUT/tile_$m -name "*kernel_trace.csv" | head -1)
  python $ROOT/scripts/trace_gaps.py $f 10 > $OUT/r04_tile_${m}_gaps.txt 2>&1
  rm -f $f
  cp $OUT/tile_$m/t_kernel_stats.csv $OUT/r04_tile_${m}_kernel_stats.csv
  grep ms_per_step $OUT/tile_$m.log
done
cd $ROOT
ls -la $OUT | grep r04_ | head -20
du -sh $OUT
