ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1; cd /tmp
for r in 0 2 3 4 5 6 8; do
  echo "rows=$r $(MOM6X_MFW_ROWS=$r timeout 200 python $ROOT/scripts/prof_tile.py local_wrap 30 2>&1 | grep ms_per_step)"
done
