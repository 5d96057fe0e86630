#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the continuity kernels in one mode (dev tool; separate passes)
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp
export PROF_MODES=${PROF_MODES:-full}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- env PYTHONPATH=$ROOT python $ROOT/scripts/prof_continuity.py > $OUT/pmc_$c.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, collections, re
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = sorted(csv.DictReader(open(f"gpurun_out/pmc_{c}/p_counter_collection.csv")), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        n = r["Kernel_Name"]
        m = re.search(r"(k_mass_flux_lds<[^>]*>|k_convergence<[^>]*>|k_h_av)", n)
        if m: print(c, m.group(1).replace(" ", ""), "%.3f GB (raw, KB units x1024)" % (float(r["Counter_Value"]) * 1024 / 1e9))
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
