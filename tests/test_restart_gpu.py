"""The reference's restart regression (.testing `test.restart`: a run interrupted by save_restart / restore_state gives the
same ocean.stats as the uninterrupted one) on the device: 4 baroclinic steps against 2 + restart file + 2 in a new
context.  Every prognostic field must be bit-identical; the checksums in the file are verified on the device."""
import numpy as np
import pytest

from mom6_amd import abi, restart as R, sum_output as SO
from tests import helpers as H

pytestmark = pytest.mark.gpu

STATE = ["u", "v", "h", "uh", "vh", "uhtr", "vhtr", "eta_av"]


def new_model(cfg, inp, bt_mod, hv=None):
    from mom6_amd.dycore import Dycore
    from tests import cases
    gg, d, M = cfg
    GV, Rlay, gp, dt = inp["GV"], inp["Rlay"], inp["gp"], inp["dt"]
    cont2, bt2, cor2, pgf2, rk22 = cases.rk2_params(d, GV, bt_mod, None, None)
    dyc = Dycore(d, M, GV, 0)
    dyc.continuity_init(cont2); dyc.barotropic_init(bt2); dyc.CoriolisAdv_init(cor2); dyc.PressureForce_init(pgf2, Rlay, gp)
    dyc.initialize_dyn_split_RK2(rk22)
    dyc.vertvisc_set_coef(*[dyc.to_dev(a) if a is not None else None for a in inp["coefs"][0]])
    if hv is not None:
        dyc.hor_visc_init(hv)
    dyc.sum_output_init(abi.sum_output_params_default(dt), gp)
    forcing = (dyc.to_dev(inp["taux"]), dyc.to_dev(inp["tauy"]))
    return dyc, forcing


def step(dyc, sg, forcing, dt, calc_dtbt):
    dyc.step_MOM_dyn_split_RK2(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], sg["uhtr"], sg["vhtr"], sg["eta_av"], forcing[0],
                               forcing[1], dt, calc_dtbt=calc_dtbt)


def registry(dyc, sg, gg, d):
    CS = R.MOM_restart_CS(dyc, R.axes_of(gg, d))
    CS.register_restart_field(sg["u"], "u", "u", "L", "Zonal velocity", "m s-1")          # MOM.F90:3863-3875
    CS.register_restart_field(sg["v"], "v", "v", "L", "Meridional velocity", "m s-1")
    CS.register_restart_field(sg["h"], "h", "h", "L", "Layer Thickness", "m")
    CS.register_restarts_dyn_split_RK2()
    return CS


@pytest.mark.parametrize("cfg_name,bt_mod", [("double_gyre", dict(strong_drag=1)), ("channel", dict()), ("benchmark_small", dict())])
def test_restarted_run_is_bit_identical(cfg_name, bt_mod, tmp_path):
    from tests import cases
    cfg = getattr(H, cfg_name)()
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg, False, False)
    dt = inp["dt"]

    def fresh_state(dyc):
        return dict(u=dyc.to_dev(inp["u"]), v=dyc.to_dev(inp["v"]), h=dyc.to_dev(inp["h"]), uh=dyc.zeros3(), vh=dyc.zeros3(),
                    uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())

    # ---- the uninterrupted run
    dyc, forcing = new_model(cfg, inp, bt_mod)
    sg = fresh_state(dyc)
    dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)
    stats_a = SO.SumOutput()
    for n in range(4):
        step(dyc, sg, forcing, dt, n == 0)
        stats_a.record(dyc.write_energy(sg["u"], sg["v"], sg["h"]), dt * (n + 1), n + 1)
    ref = {k: sg[k].cpu().numpy() for k in ("u", "v", "h", "eta_av")}
    ref_cs = {k: dyc.rk2_field(k).cpu().numpy() for k in ("eta", "u_av", "v_av", "CAu_pred", "diffu")}
    dyc.close()

    # ---- two steps, save_restart
    dyc, forcing = new_model(cfg, inp, bt_mod)
    sg = fresh_state(dyc)
    dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)
    stats_b = SO.SumOutput()
    for n in range(2):
        step(dyc, sg, forcing, dt, n == 0)
        stats_b.record(dyc.write_energy(sg["u"], sg["v"], sg["h"]), dt * (n + 1), n + 1)
    path = tmp_path / "MOM.res.nc"
    sums = registry(dyc, sg, gg, d).save_restart(path, 2 * dt / 86400.0)
    dtbt = dyc.barotropic_dtbt()
    assert set(sums) == {"u", "v", "h", "sfc", "u2", "v2", "CAu", "CAv", "diffu", "diffv", "ubtav", "vbtav", "DTBT"} and dtbt > 0
    dyc.close()

    # ---- a new context: restore_state instead of the new-run initialisation, two more steps
    dyc, forcing = new_model(cfg, inp, bt_mod)
    sg = dict(u=dyc.zeros3(), v=dyc.zeros3(), h=dyc.zeros3(), uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=dyc.zeros3(), vhtr=dyc.zeros3(),
              eta_av=dyc.zeros2())
    CS = registry(dyc, sg, gg, d)
    assert CS.restore_state(path) == 2 * dt / 86400.0
    assert dyc.barotropic_dtbt() == dtbt
    dyc.rk2_set_CAu_pred_stored(True)                    # query_initialized(CS%CAu_pred, "CAu") :1616
    for n in range(2, 4):
        step(dyc, sg, forcing, dt, False)
        stats_b.record(dyc.write_energy(sg["u"], sg["v"], sg["h"]), dt * (n + 1), n + 1)
    for k in ref:
        H.assert_bitwise(sg[k].cpu().numpy(), ref[k], "restart:" + k, H.interior(d, {"u": "u", "v": "v"}.get(k, "h")))
    for k in ref_cs:
        H.assert_bitwise(dyc.rk2_field(k).cpu().numpy(), ref_cs[k], "restart:" + k, H.interior(d, "u" if k in ("u_av", "CAu_pred", "diffu") else "h"))
    assert stats_b.lines == stats_a.lines                # what .testing's test.restart compares
    # a corrupted file is refused (RESTART_CHECKSUMS_REQUIRED)
    t, data, atts = R.read_restart_file(path)
    data["h"][0, 0, 3, 3] += 1.0e-9
    bad = tmp_path / "bad.res.nc"
    variables = []
    with __import__("scipy.io", fromlist=["netcdf_file"]).netcdf_file(str(path), "r", mmap=False) as nc:
        for n in data:
            variables.append((n, data[n], nc.variables[n].dimensions, atts[n]))
    R.write_restart_file(bad, CS.axes, t, variables)
    with pytest.raises(RuntimeError, match="Checksum of input field h"):
        CS.restore_state(bad)
    dyc.close()
