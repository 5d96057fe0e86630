#!/bin/bash
# SQ counters of k_mass_flux_wave in one mode (dev tool): PROF_MODES=full bash scripts/pmc_wave.sh
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp
export PROF_MODES=${PROF_MODES:-full}
i=0
SETS=${PMC_SETS:-SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAVES,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY SQ_INSTS_SALU,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM,SQ_INST_LEVEL_VMEM,SQ_ACTIVE_INST_VMEM,SQ_WAVE_CYCLES SQC_ICACHE_REQ,SQC_ICACHE_HITS,SQC_ICACHE_MISSES,GRBM_GUI_ACTIVE}
for set in $SETS; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc ${set//,/ } --output-format csv -d $OUT/pmc_mw$i -o mw -- env PYTHONPATH=$ROOT python $ROOT/scripts/prof_continuity.py > $OUT/pmc_mw$i.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, collections, glob, re
tab = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmc_mw*/mw_counter_collection.csv")):
    seen = collections.Counter(); disp = {}
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        n = r["Kernel_Name"]
        if "k_mass_flux_wave" not in n: continue
        m = re.search(r"k_mass_flux_wave<([^>]*)>", n)
        short = m.group(1).replace(" ", "") if m else n[:60]
        key = (short, r["Dispatch_Id"])
        if key not in disp:
            disp[key] = seen[short]; seen[short] += 1
        tab[(short, disp[key])][r["Counter_Name"]] = tab[(short, disp[key])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(tab):
    if k[1] != 1: continue   # the profiled (second) launch of the mode
    print("k_mass_flux_wave<%s> launch %d" % k)
    for c, x in sorted(tab[k].items()):
        print("   %-28s %.4e" % (c, x))
PY
