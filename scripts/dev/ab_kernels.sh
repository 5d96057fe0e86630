#!/bin/bash
# A/B of build variants by kernel time (dev tool): bash scripts/dev/ab_kernels.sh "<kernel name regex>" "<cflags>" "<cflags>" ...
# one build per variant (every .hip file is rebuilt, or the files AB_TOUCH names), one short bench.py run under rocprofv3 --kernel-trace --stats each.
ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; OUT=$ROOT/gpurun_out; mkdir -p $OUT
PAT="$1"; shift
for fl in "$@"; do
  echo "=== variant cflags=[$fl]"
  touch ${AB_TOUCH:-mom6_amd/csrc/*.hip}
  a=""; case "$fl" in ARCH=*) a="${fl#ARCH=}"; fl="";; esac
  MOM6X_ARCH="${a:-gfx950}" MOM6X_CFLAGS="$fl" python -m mom6_amd.build > /dev/null 2>$OUT/ab_build.err || { echo BUILD FAILED; tail -5 $OUT/ab_build.err; continue; }
  for rep in 1 2; do
    D=/tmp/abk_$$_$rep; rm -rf $D
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ab -- python $ROOT/bench.py --steps ${AB_STEPS:-6} --warmup 2 --no-config4 --no-cpu-baseline --no-comm-model ${AB_BENCH_ARGS} > $D.log 2>&1)
    python - "$D" "$PAT" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
    if re.search(sys.argv[2], n): print("   %-52s calls %4s  avg %9.1f us" % (n[:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    grep -o '"ms_per_step": [0-9.]*' $D.log | head -1
  done
done
