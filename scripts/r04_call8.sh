ROOT=$(pwd); OUT=$ROOT/gpurun_out
MOM6X_TEST_SIGNED_ZERO=strict timeout 900 python -m pytest tests/test_continuity_gpu.py tests/test_restart_gpu.py tests/test_rk2_gpu.py tests/test_layout_gpu.py -x -q 2>&1 | grep -E "^E .*Assert|passed|failed" | head
PROF_MODES=adjust,bt_cont timeout 100 python scripts/prof_continuity.py 2>&1 | grep '^lds'
