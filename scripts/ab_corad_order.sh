#!/bin/bash
# FETCH+WRITE of k_corad_fused for the two orders
export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
ROOT=$(pwd); cd /tmp
for m in xcd plain; do
  for c in FETCH_SIZE WRITE_SIZE; do
    MOM6X_CORAD_ORDER=$m timeout 150 rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/ab_${m}_$c -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-comm-model --tracers -1 > /dev/null 2>&1
    python3 - <<P
import csv,collections
s=collections.Counter(); n=collections.Counter()
for r in csv.DictReader(open("$ROOT/gpurun_out/ab_${m}_$c/p_counter_collection.csv")):
    k=r["Kernel_Name"]
    if "corad_fused" in k or "hv_fused" in k: s[k[:30]]+=float(r["Counter_Value"]); n[k[:30]]+=1
for k in s: print("$m $c", k, n[k], s[k]/n[k]/1e6)
P
  done
done
