ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; cd /tmp
for m in local_wrap rccl_self; do
  rm -rf $OUT/tile_$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tile_$m -o t -- python $ROOT/scripts/prof_tile.py $m 10 > $OUT/tile_$m.log 2>&1
  f=$(find $OUT/tile_$m -name "*kernel_trace.csv" | head -1)
  python $ROOT/scripts/trace_gaps.py $f 10 > $OUT/r04_tile_${m}_gaps.txt 2>&1
  rm -f $f
  cp $OUT/tile_$m/t_kernel_stats.csv $OUT/r04_tile_${m}_kernel_stats.csv
  grep ms_per_step $OUT/tile_$m.log
done
