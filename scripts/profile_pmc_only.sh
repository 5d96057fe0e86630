#!/bin/bash
# the two PMC passes of profile_bench.sh alone, with a short leash (rocprofv3 counter collection hangs now and then)
set -u
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp
timeout ${PMC_TIMEOUT:-90} rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --tracers 0 > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout ${PMC_TIMEOUT:-90} rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --tracers 0 > $OUT/prof_write.log 2>&1; echo "write rc=$?"
