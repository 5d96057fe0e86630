#!/bin/bash
# SQ counters of the kernels whose name matches $1 in a short bench run (dynamics only): scripts/pmc_kernel.sh <regex> "<counters>" ...
export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
R=$(pwd); re=$1; shift; n=0
for set in "$@"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$re" --output-format csv -d $R/gpurun_out/prof_k$n -o sq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-comm-model ${PMC_BENCH_ARGS---tracers -1} > $R/gpurun_out/prof_k$n.log 2>&1; echo "set $n rc=$?"
done
python scripts/pmc_summary.py prof_k
