"""Per-launch averages of the counters rocprofv3 --pmc wrote under gpurun_out/<prefix>*/ (dev tool): python scripts/pmc_summary.py prof_remap_sq [kernel-substring]"""
import csv, glob, collections, re, sys
pre = sys.argv[1] if len(sys.argv) > 1 else "prof_remap_sq"
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/%s*/**/*counter_collection.csv" % pre, recursive=True):
    for r in csv.DictReader(open(f)):
        mk = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
        if not mk: continue
        k = mk.group(0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if sub in k:
        print(k, max(n[k].values()), {a: round(b / n[k][a] / 1e6, 2) for a, b in sorted(acc[k].items())})
