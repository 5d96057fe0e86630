"""The restart file layer (mom6_amd/restart.py; MOM_restart.F90:1567 save_restart, MOM_io.F90:254-587 create_MOM_file):
axes, staggering, attributes and the `checksum` attribute, on numpy data (no GPU)."""
import numpy as np
import pytest

from mom6_amd import restart as R
from tests import helpers as H


def test_restart_file_layout_and_checksum_attribute(orc, tmp_path):
    gg, d, M = H.benchmark_small(nk=3)
    rng = np.random.default_rng(3)
    full = {n: rng.standard_normal((3,) + tuple(d.shape2())) for n in ("u", "v", "h")}
    full["sfc"] = rng.standard_normal(d.shape2())
    axes = R.axes_of(gg, d)
    assert len(axes["lonq"]) == d.ni + 1 and len(axes["latq"]) == d.nj + 1 and axes["lonh"][0] == 0.5 and axes["lath"][0] == -39.5
    variables = []
    for n, hg, zg in (("u", "u", "L"), ("v", "v", "L"), ("h", "h", "L"), ("sfc", "h", "1")):
        sl = d.sl(-1 if hg == "u" else 0, d.ni - 1, -1 if hg == "v" else 0, d.nj - 1)
        a = full[n][(Ellipsis,) + tuple(sl)]
        # the checksum range is the h-point computational domain whatever the staggering (MOM_restart.F90:2416-2431)
        chk = orc.field_chksum(d, full[n], 0, d.ni - 1, 0, d.nj - 1)
        dims = ("Time",) + (("Layer",) if zg == "L" else ()) + (R._YAX[hg], R._XAX[hg])
        variables.append((n, a[None], dims, dict(long_name=n, units="m", checksum="%016X" % (chk % 2 ** 64))))
    variables.append(("DTBT", np.array([37.5]), ("Time",), dict(long_name="Barotropic timestep", units="seconds",
                                                                  checksum="%016X" % int(np.array([37.5]).view(np.int64)[0]))))
    p = tmp_path / "MOM.res.nc"
    R.write_restart_file(p, axes, 1.25, variables)
    assert open(p, "rb").read(4) == b"CDF\x02"                       # netCDF-3, 64-bit offset
    from scipy.io import netcdf_file
    with netcdf_file(str(p), "r", mmap=False) as nc:
        assert nc.dimensions["lonh"] == d.ni and nc.dimensions["lonq"] == d.ni + 1 and nc.dimensions["latq"] == d.nj + 1
        assert nc.dimensions["Layer"] == 3 and nc.dimensions["Time"] is None and "Interface" not in nc.dimensions
        assert nc.variables["u"].dimensions == ("Time", "Layer", "lath", "lonq")
        assert nc.variables["v"].dimensions == ("Time", "Layer", "latq", "lonh")
        assert nc.variables["sfc"].dimensions == ("Time", "lath", "lonh") and nc.variables["DTBT"].dimensions == ("Time",)
        assert nc.variables["Time"].units == b"days" and nc.variables["lonq"].cartesian_axis == b"X"
        assert len(nc.variables["h"].checksum) == 16
    t, data, atts = R.read_restart_file(p)
    assert t == 1.25 and set(data) == {"u", "v", "h", "sfc", "DTBT"}
    for n, a, _, at in variables:
        assert np.array_equal(data[n], a) and atts[n]["checksum"] == at["checksum"]
    # the u variable carries the western boundary face, which the checksum does not cover
    assert data["u"].shape == (1, 3, d.nj, d.ni + 1)
    assert int(atts["h"]["checksum"], 16) == int(np.sum(data["h"].view(np.uint64).astype(object))) % 2 ** 64
    assert int(atts["u"]["checksum"], 16) == int(np.sum(np.ascontiguousarray(data["u"][..., 1:]).view(np.uint64).astype(object))) % 2 ** 64
