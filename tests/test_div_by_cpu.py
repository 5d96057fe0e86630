"""The three-operation division by 3 and 6 of the tracer kernels (mom6_amd/csrc/tracer.hip `div_by<C>`: Markstein's correction of
x * RN(1/c) with one exact remainder) equals the division bit for bit wherever the kernel uses it (|x| >= 2**-1000, zeros included) --
checked here on the host with the same operations (C `fma`), tens of millions of values of every exponent plus the patterns that
break naive x * (1/3): tests/native/div_by_check.c.  The device side is held to the oracle, which divides, by tests/test_tracer_gpu.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_three_operation_division_is_the_division(tmp_path):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    exe = tmp_path / "div_by_check"
    subprocess.run([cc, "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "native", "div_by_check.c"), "-lm"], check=True)
    r = subprocess.run([str(exe), "30"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " mismatches 0 " in r.stdout, r.stdout[-500:]
    assert int(r.stdout.split("checked ")[1].split()[0]) > 3e7
