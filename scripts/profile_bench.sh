#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box: (1) kernel trace + stats of the headline model (the legs on other tile
# sizes are switched off so that the averages are those of the 1440 x 1080 x 75 tile), (2) a FETCH_SIZE pass, (3) a
# WRITE_SIZE pass of the dynamics alone.  Counters are collected in their own runs (no sys/hip/hsa tracing next to --pmc),
# each on a short leash (rocprofv3 counter collection hangs now and then).  Outputs: gpurun_out/prof_*/
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
export MOM6X_BENCH_NO_PMC=1   # (bench.py would otherwise start its own rocprofv3 passes for roofline.traffic)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
if [ "${1:-all}" != "pmc" ]; then
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python $ROOT/bench.py --no-config4 --no-comm-model --no-cpu-baseline > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
fi
if [ "${1:-all}" != "stats" ]; then
timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-comm-model --tracers -1 > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-comm-model --tracers -1 > $OUT/prof_write.log 2>&1; echo "write rc=$?"
timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/prof_sq -o sq -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-comm-model --tracers -1 > $OUT/prof_sq.log 2>&1; echo "sq rc=$?"
fi
cd $ROOT
find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -type f 2>/dev/null | head -20
tail -1 $OUT/prof_stats.log 2>/dev/null | cut -c1-400
