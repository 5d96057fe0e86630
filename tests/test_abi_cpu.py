"""CPU-side checks of the drop-in boundary: the HIP C-ABI library loads without a GPU, exports every symbol
that include/mom6x.h declares, its structs match the ctypes mirrors, and the product path fails LOUDLY
(no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mom6_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return abi.load_library()


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mom6x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mom6x_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/mom6x.h but not exported: {missing}"


def test_struct_sizes_match_ctypes_mirrors(lib):
    mirrors = [abi.Dims, abi.VGrid, abi.ContinuityParams, abi.BTCont, abi.BarotropicParams, abi.CoriolisParams,
               abi.PGFParams, abi.RK2Params, abi.RK2Hooks, abi.EOSParams, abi.VertviscParams, abi.HorViscParams,
               abi.RemappingParams, abi.RegridZstarParams, abi.ChksumResult,
               abi.SumOutputParams, abi.EnergySums, abi.RegridRhoParams]
    for which, cls in enumerate(mirrors):
        assert lib.mom6x_struct_size(which) == C.sizeof(cls), cls.__name__
    assert lib.mom6x_abi_version() == abi.ABI_VERSION == 6


@pytest.mark.parametrize("ni,nj,nk,halo", [(44, 40, 2, 4), (1440, 1080, 75, 4), (7, 5, 1, 3), (720, 540, 75, 4)])
def test_dims_init_matches_python_twin(lib, ni, nj, nk, halo):
    d = abi.Dims()
    assert lib.mom6x_dims_init(C.byref(d), ni, nj, nk, halo) == 0
    p = abi.dims_init(ni, nj, nk, halo)
    for f, _ in abi.Dims._fields_:
        assert getattr(d, f) == getattr(p, f), f
    assert d.pitch % 16 == 0 and d.ioff % 16 == 0 and d.pitch >= d.ioff + ni + halo
    assert lib.mom6x_dims_init(C.byref(d), 0, nj, nk, halo) != 0     # bad sizes are rejected with a message
    assert b"bad sizes" in lib.mom6x_last_error()


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tests import helpers as H
    gg, d, M = H.double_gyre()
    ctx = C.c_void_p()
    GV = abi.vgrid_default()
    rc = lib.mom6x_ctx_create(C.byref(ctx), C.byref(d), 0, M.ctypes.data_as(C.c_void_p), C.byref(GV), 0)
    assert rc != 0 and not ctx.value
    assert lib.mom6x_last_error()
    from mom6_amd.dycore import Dycore
    with pytest.raises(RuntimeError):
        Dycore(d, M, GV)


def test_metric_enum_matches_header():
    txt = open(os.path.join(ROOT, "include", "mom6x.h")).read()
    body = txt[txt.index("enum mom6x_metric"):]
    body = body[:body.index("MOM6X_G_COUNT")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"MOM6X_G_([A-Za-z0-9_]+)", body)
    assert names == abi.METRICS


def test_fortran_binding(lib):
    """The ISO_C_BINDING mirror (fortran/mom6x_c_api.F90) compiles with amdflang, links to the library and
    agrees with it on every struct size (north_star: "Fortran host via ISO_C_BINDING")."""
    import subprocess
    import __graft_entry__ as g
    if not g._build_fortran_binding():
        pytest.skip("amdflang not available")
    out = subprocess.run([os.path.join(ROOT, "fortran", "check_abi")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "fortran ABI check OK" in out.stdout, out.stdout + out.stderr
