#!/bin/bash
# Instruction mix of k_mass_flux_lds in one mode (dev tool): PROF_MODES=full bash scripts/pmc_massflux.sh
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp
export PROF_MODES=${PROF_MODES:-full}
i=0
for set in ${PMC_SETS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"}; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc ${set//,/ } --output-format csv -d $OUT/pmc_mf$i -o mf -- env PYTHONPATH=$ROOT python $ROOT/scripts/prof_continuity.py > $OUT/pmc_mf$i.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, collections, glob, re
tab = collections.defaultdict(dict)   # (short name, ordinal) -> {counter: value}
for f in sorted(glob.glob("gpurun_out/pmc_mf*/mf_counter_collection.csv")):
    seen = collections.Counter(); disp = {}
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        n = r["Kernel_Name"]
        if "k_mass_flux_lds" not in n: continue
        m = re.search(r"k_mass_flux_lds<([^>]*)>", n)
        short = m.group(1).replace(" ", "") if m else n[:60]
        key = (short, r["Dispatch_Id"])
        if key not in disp:
            disp[key] = seen[short]; seen[short] += 1
        tab[(short, disp[key])][r["Counter_Name"]] = tab[(short, disp[key])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(tab):
    print("k_mass_flux_lds<%s> launch %d" % k)
    for c, x in sorted(tab[k].items()):
        print("   %-28s %.4e" % (c, x))
PY
