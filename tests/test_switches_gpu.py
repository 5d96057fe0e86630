"""Every run-time switch that selects another kernel (README.md "Run-time switches") is held to the same bit-for-bit parity as
the default path: the GPU tests of the routine it touches, once more in a process that has the switch set.  (The switches are read
once per process, hence the subprocess; tests/test_rk2_gpu.py::test_rk2_75_layers_with_btcalc_written_out does the same for
MOM6X_BTCALC, test_continuity_gpu.py / test_dyn_gpu.py / test_horvisc_gpu.py / test_tracer_gpu.py for theirs.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # switch, value, test file, -k selection
    ("MOM6X_BT_SUBSTEP", "kernels", "test_barotropic_gpu.py", ""),                      # three launches per barotropic sub-step
    ("MOM6X_BT_SUBSTEP", "kernels", "test_rk2_gpu.py", "double_gyre_bitexact or 75_layers_on_chip"),
    ("MOM6X_BT_SUBSTEP", "fused", "test_layout_gpu.py", "wide_halos"),                  # (the default at these sizes; named)
    ("MOM6X_MFW_SPEC", "0", "test_continuity_gpu.py", "many_layers"),                   # the general mass-flux kernel where the one compiled for the launch's switches would run
    ("MOM6X_MFW_SPEC", "0", "test_rk2_gpu.py", "75_layers_on_chip"),
    ("MOM6X_VERTVISC", "walk", "test_rk2_gpu.py", "75_layers_on_chip or one_kernel"),                 # the column solve through HBM at nk = 75
    ("MOM6X_VERTVISC", "pair", "test_rk2_gpu.py", "75_layers_on_chip or one_kernel"),                 # vertvisc_coef and the solve as two kernels (on chip each)
    ("MOM6X_PASS_WIDTHS", "full", "test_layout_gpu.py", "tile_layout_gives"),           # NIHALO rows in every group pass of the step
    ("MOM6X_POISON_HALO", "1", "test_layout_gpu.py", "tile_layout_gives"),              # NaNs in the halo rows beyond the width of each narrow pass
    ("MOM6X_BT_MASS_SOURCE", "own", "test_rk2_gpu.py", "double_gyre_bitexact or 75_layers_on_chip"),
    ("MOM6X_REMAP_MERGE", "apply", "test_remap_gpu.py", "ALE_remap"),                   # the one-field streamed merge everywhere
    ("MOM6X_TRIDIAG", "walk", "test_tracer_gpu.py", "tridiagonal"),                     # the tracer solves through HBM
]


@pytest.mark.parametrize("switch,value,file,select", CASES, ids=[f"{c[0]}={c[1]}:{c[2][5:-7]}" for c in CASES])
def test_switch_keeps_parity(switch, value, file, select):
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", file), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"]
    if select:
        cmd += ["-k", select]
    r = subprocess.run(cmd, env=dict(os.environ, **{switch: value}), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
