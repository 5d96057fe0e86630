#!/bin/bash
# round-6 evidence in one call: the default bench line (the driver's command), then the per-kernel tables of the 8-GPU tile
# (360 x 540 x 75, local wrap copies / RCCL self-sends) -> gpurun_out/r06_${TAG:-v0}_*
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT; T=${TAG:-v0}
( time python bench.py --steps 20 --warmup 5 > $OUT/r06_${T}_bench.json 2> $OUT/r06_${T}_bench.err ) 2> $OUT/r06_${T}_bench.time
tail -c 600 $OUT/r06_${T}_bench.json
export MOM6X_BENCH_NO_PMC=1; cd /tmp
for m in local_wrap rccl_self; do
  rm -rf $OUT/tile_$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tile_$m -o t -- python $ROOT/scripts/prof_tile.py $m 10 > $OUT/tile_$m.log 2>&1
  f=$(find $OUT/tile_$m -name "*kernel_trace.csv" | head -1)
  python $ROOT/scripts/trace_gaps.py $f 10 > $OUT/r06_${T}_tile_${m}_gaps.txt 2>&1
  rm -f $f
  cp $OUT/tile_$m/t_kernel_stats.csv $OUT/r06_${T}_tile_${m}_kernel_stats.csv
  grep ms_per_step $OUT/tile_$m.log
done
