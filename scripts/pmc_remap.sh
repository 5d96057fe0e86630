#!/bin/bash
# SQ counters of the ALE remapping kernels (scripts/prof_remap.py), per launch.  Counter collection is limited to the
# remapping kernels (rocprofv3 has crashed inside torch's own kernels under --pmc on this image).
#   scripts/pmc_remap.sh "SQ_WAVES SQ_INSTS_VALU ..." [more sets ...]
export TMPDIR=/tmp
R=$(pwd)
n=0
for set in "${@:-SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU}"; do
  n=$((n+1))
  timeout 200 rocprofv3 --pmc $set --kernel-include-regex "k_remap.*" --output-format csv -d $R/gpurun_out/prof_remap_sq$n -o sq -- python $R/scripts/prof_remap.py ${PROF_REMAP_ARGS:-} > $R/gpurun_out/prof_remap_sq$n.log 2>&1; echo "set $n rc=$?"
done
python - <<EOF
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/prof_remap_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in acc:
    print(k, {a: round(b/n[k][a]/1e6,3) for a,b in sorted(acc[k].items())})
EOF
