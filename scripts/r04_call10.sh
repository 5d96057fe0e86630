ROOT=$(pwd); export TMPDIR=/tmp MOM6X_BENCH_NO_PMC=1
timeout 900 python -m pytest tests/test_continuity_gpu.py tests/test_layout_gpu.py tests/test_bench_layout_gpu.py tests/test_halo_gpu.py tests/test_rk2_gpu.py -q -x 2>&1 | tail -2
cd /tmp
for m in local_wrap rccl_self; do timeout 200 python $ROOT/scripts/prof_tile.py $m 20 2>&1 | grep ms_per_step; done
