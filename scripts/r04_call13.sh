ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
for xr in 2 4 8; do
  touch mom6_amd/csrc/tracer.hip
  MOM6X_CFLAGS="-DMOM6X_TA_XR=$xr" python -m mom6_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  echo "XR=$xr"
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-comm-model --no-pmc --breakdown 2>&1 >/dev/null | grep "thermo: k_ta_\|thermo: k_tridiag" | head -6
done
touch mom6_amd/csrc/tracer.hip
