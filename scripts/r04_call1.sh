#!/bin/bash
# round 4, GPU call 1: data for the work list (no code change yet)
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp
# 1. per-kernel table of the 360 x 540 x 75 tile
for m in local_wrap rccl_self; do
  rm -rf $OUT/tile_$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tile_$m -o t -- python $ROOT/scripts/prof_tile.py $m 10 > $OUT/tile_$m.log 2>&1
  find $OUT/tile_$m -name "*kernel_trace*" -delete
done
# 2. per-mode times of the mass-flux kernels
( cd $ROOT && timeout 300 python scripts/prof_continuity.py > $OUT/cont_modes.log 2>&1 )
# 3. tracer kernels: stats + counters
( cd $ROOT && timeout 300 python scripts/prof_tracer.py 1440 1080 75 4 > $OUT/tracer_prof.log 2>&1 )
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf $OUT/prof_ta$i
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "k_ta_" --output-format csv -d $OUT/prof_ta$i -o ta -- python $ROOT/scripts/prof_tracer.py 1440 1080 75 4 > $OUT/prof_ta$i.log 2>&1; echo "ta set $i rc=$?"
done
cd $ROOT
python scripts/pmc_summary.py prof_ta > $OUT/ta_pmc_summary.txt 2>&1
tail -3 $OUT/tile_local_wrap.log $OUT/tile_rccl_self.log
cat $OUT/cont_modes.log | tail -12
cat $OUT/tracer_prof.log | tail -14
cat $OUT/ta_pmc_summary.txt
