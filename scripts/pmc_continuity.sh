#!/bin/bash
# SQ counters of the continuity kernels (dev tool): is k_mass_flux_lds issue-bound or waiting?
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq -o sq -- env PYTHONPATH=$ROOT python $ROOT/scripts/prof_continuity.py > $OUT/pmc_sq.log 2>&1
cd $ROOT
python - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open("gpurun_out/pmc_sq/sq_counter_collection.csv")):
    n = r["Kernel_Name"]
    if "k_mass_flux" not in n and "k_edge" not in n: continue
    n = n.split("(")[1 if n.startswith("void (") else 0] if False else n
    key = ("lds<0" if "lds<0" in n else "lds<1" if "lds<1" in n else "legacy" if "k_mass_flux<" in n else "other")
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[key] += 1
for k, v in agg.items():
    wc = v["SQ_WAVE_CYCLES"]
    print(k, "launches", cnt[k], " ".join(f"{c}={x/wc:.3f}" for c, x in sorted(v.items()) if c != "SQ_WAVE_CYCLES"), "wave_cycles/launch=%.3e" % (wc / cnt[k]))
PY
