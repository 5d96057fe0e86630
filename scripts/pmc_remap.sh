#!/bin/bash
# SQ counters of the ALE remapping kernels (scripts/prof_remap.py), per launch.
export TMPDIR=/tmp
R=$(pwd)
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-include-regex "k_remap.*" --output-format csv -d $R/gpurun_out/prof_remap_sq -o sq -- python $R/scripts/prof_remap.py > $R/gpurun_out/prof_remap_sq.log 2>&1; echo rc=$?
python - <<EOF
import csv,glob,collections
f=glob.glob("gpurun_out/prof_remap_sq/**/*counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "remap" in k:
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]);
        if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
for k in acc:
    print(k, n[k], {a: round(b/n[k]/1e6,2) for a,b in acc[k].items()}, "valu/wave", round(acc[k]["SQ_INSTS_VALU"]/acc[k]["SQ_WAVES"]))
EOF
