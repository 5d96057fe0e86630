ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
for pf in 1 3; do
  touch mom6_amd/csrc/tracer.hip
  MOM6X_CFLAGS="-DMOM6X_TA_PF=$pf" python -m mom6_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python - <<'PY'
import json
r=json.load(open('mom6_amd/lib/kernel_resources.json'))['tracer.hip']
for k,v in r.items():
    if 'y_marchILi4' in k: print(k[:42], {a:v[a] for a in ('vgprs','scratch','occupancy')})
PY
  echo "PF=$pf"
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc --breakdown 2>&1 >/dev/null | grep "thermo: k_ta_y" | head -3
done
touch mom6_amd/csrc/tracer.hip
