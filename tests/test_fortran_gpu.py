"""A step driven from FORTRAN: fortran/drive_double_gyre (amdflang; fortran/mom6x_c_api.F90 + fortran/mom6x_host.F90 +
libmom6x.so, nothing else) reads a case written here, packs the metric block from arrays with MOM6's symmetric-memory
extents, creates the context, initialises the modules, uploads the state once, runs three steps of
step_MOM_dyn_split_RK2 on the resident state and compares all eight prognostic arrays with the committed fixture
tests/golden/rk2_double_gyre_strong_drag_3steps -- bit for bit, in Fortran.  This is the ISO_C_BINDING boundary of
SURVEY.md 8(b) exercised end to end (the Python mirror the other tests use never touches these code paths:
mom6x_upload / mom6x_download with Fortran extents, the bind(C) struct layouts, mom6x_dev_alloc)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from mom6_amd import abi
from tests import cases
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "fortran", "drive_double_gyre")
STATE = ["u", "v", "h", "uh", "vh", "uhtr", "vhtr", "eta_av"]


def _stagger(name):
    """Staggering code (mom6x_upload) of a metric plane from its MOM6 name."""
    if name.endswith("Bu"):
        return 3
    if name.endswith("Cu") or name == "dy_Cu":
        return 1
    if name.endswith("Cv") or name == "dx_Cv":
        return 2
    return 0


def _f_extent(d, a, stagger):
    """The part of a pitched array that a Fortran array with MOM6's symmetric-memory extents holds (C order [k][j][i])."""
    xB, yB = int(stagger in (1, 3)), int(stagger in (2, 3))
    return np.ascontiguousarray(a[..., d.joff - d.halo - yB: d.joff + d.nj + d.halo, d.ioff - d.halo - xB: d.ioff + d.ni + d.halo])


def _fields(s):
    """A bind(C) struct as Fortran unformatted stream I/O transfers it: component by component, no padding."""
    out = b""
    for name, ctype in s._fields_:
        v = getattr(s, name)
        out += struct.pack("<i", v) if ctype is C.c_int else struct.pack("<d", v)
    return out


def test_three_steps_driven_from_fortran(tmp_path):
    if not os.path.exists(DRIVER):
        pytest.fail("fortran/drive_double_gyre is missing: __graft_entry__.build() compiles it with amdflang")
    cfg = H.double_gyre()
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg)
    cont, bt, cor, pgf, rk2 = cases.rk2_params(d, inp["GV"], dict(strong_drag=1))
    gold = H.load_golden("rk2_double_gyre_strong_drag_3steps")
    nsteps, first_direction = 3, 0
    a_u, a_v, h_u, h_v, Ray_u, Ray_v = inp["coefs"][0]
    path = tmp_path / "case.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<10i", 1297042742, d.ni, d.nj, d.nk, d.halo, nsteps, first_direction, abi.G_COUNT, d.reentrant_x, d.reentrant_y))
        f.write(struct.pack("<d", inp["dt"]))
        for s in (inp["GV"], cont, bt, cor, pgf, rk2):
            f.write(_fields(s))
        f.write(np.ascontiguousarray(inp["Rlay"], dtype="<f8").tobytes()); f.write(np.ascontiguousarray(inp["gp"], dtype="<f8").tobytes())
        for m, name in enumerate(abi.METRICS):
            st = _stagger(name)
            f.write(struct.pack("<i", st)); f.write(_f_extent(d, M[m], st).astype("<f8").tobytes())
        for a, st in ((inp["u"], 1), (inp["v"], 2), (inp["h"], 0), (a_u, 1), (a_v, 2), (h_u, 1), (h_v, 2), (Ray_u, 1), (Ray_v, 2), (inp["taux"], 1), (inp["tauy"], 2)):
            f.write(_f_extent(d, a, st).astype("<f8").tobytes())
        for n in STATE:
            f.write(np.ascontiguousarray(gold[n], dtype="<f8").tobytes())
    r = subprocess.run([DRIVER, str(path)], capture_output=True, text=True, timeout=300)
    print(r.stdout); print(r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count(": bit-identical") == len(STATE)
