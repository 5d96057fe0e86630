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

    def outside_as_allocated(a, stag, fill):
        """MOM.F90 allocates u, v with zeros and h with Angstrom_H and the initialisation fills the computational domain (+ the
        connected halos): what lies beyond a closed edge keeps the allocation value -- in a new run and in a restarted one alike.
        The seeded state has values there; masked faces take the SIGN of their zero from them."""
        b = np.full_like(a, fill)
        sl = (Ellipsis,) + tuple(H.interior(d, stag))
        b[sl] = a[sl]
        return b

    def fresh_state(dyc):
        from mom6_amd import parallel
        st = dict(u=dyc.to_dev(outside_as_allocated(inp["u"], "u", 0.0)), v=dyc.to_dev(outside_as_allocated(inp["v"], "v", 0.0)),
                  h=dyc.to_dev(outside_as_allocated(inp["h"], "h", inp["GV"].Angstrom_H)), uh=dyc.zeros3(), vh=dyc.zeros3(),
                  uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
        import torch
        torch.cuda.synchronize()
        parallel.pass_fields(dyc, [st["u"], st["v"], st["h"]], [1, 2, 0])     # the connected (re-entrant) halos
        dyc.sync()
        return st

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
    dyc.sync()
    sg["h"].fill_(inp["GV"].Angstrom_H)                  # (MOM.F90 allocates h with Angstrom_H)
    import torch
    torch.cuda.synchronize()
    CS = registry(dyc, sg, gg, d)
    assert CS.restore_state(path) == 2 * dt / 86400.0
    assert dyc.barotropic_dtbt() == dtbt
    dyc.rk2_set_CAu_pred_stored(True)                    # query_initialized(CS%CAu_pred, "CAu") :1616
    for n in range(2, 4):
        step(dyc, sg, forcing, dt, False)
        stats_b.record(dyc.write_energy(sg["u"], sg["v"], sg["h"]), dt * (n + 1), n + 1)
    for k in ref:      # every bit, zeros of opposite sign included (what the reference's chksum bit counts would see)
        H.assert_bitwise(sg[k].cpu().numpy(), ref[k], "restart:" + k, H.interior(d, {"u": "u", "v": "v"}.get(k, "h")), signed_zero_ok=False)
    for k in ref_cs:
        H.assert_bitwise(dyc.rk2_field(k).cpu().numpy(), ref_cs[k], "restart:" + k, H.interior(d, "u" if k in ("u_av", "CAu_pred", "diffu") else "h"),
                         signed_zero_ok=False)
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


def test_restart_file_without_coriolis_accelerations(orc):
    """A restart file that holds sfc, u2, v2, diffu, diffv but not CAu, CAv (an older format, or a run that turned
    STORE_CORIOLIS_ACCEL on at the restart): initialize_dyn_split_RK2 :1620-1640 then forms h_av with one continuity call on the
    auxiliary velocities and CAu_pred, CAv_pred with CorAdCalc.  mom6x_dyn_split_RK2_restart_fills against the oracle's restatement
    of the same branch, bit for bit, after two steps + the restart + two steps; nothing may be left at its zero fill (round 3 left
    h_av = 0 on this path: q ~ abs_vort / vol_neglect)."""
    from tests import cases
    cfg = H.double_gyre()
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg, False, False)
    dt = inp["dt"]
    bt_mod = dict(strong_drag=1)
    have = abi.RK2_HAVE_ETA | abi.RK2_HAVE_DIFFU | abi.RK2_HAVE_U2
    # ---- oracle: two steps, "file" = the registered variables but CAu, CAv; a fresh model; restart fills; two steps
    so, m = cases.oracle_rk2(orc, cfg, inp, 2, bt_mod)
    cont, bt, cor, pgf, rk2 = cases.rk2_params(d, inp["GV"], bt_mod, None, None)
    m2 = orc.OrcModel(d, M, inp["GV"], cont, bt, cor, pgf, rk2, inp["Rlay"], inp["gp"], 0)
    for n in ("eta", "u_av", "v_av", "diffu", "diffv"):
        m2[n][...] = m[n]
    for n in ("ubtav", "vbtav"):
        m2.btcs[n][...] = m.btcs[n]
    bt.dtbt = m.bt.dtbt     # the restart scalar DTBT
    so2 = {k: v.copy() for k, v in so.items()}
    so2["uh"][...] = 0.0; so2["vh"][...] = 0.0
    m2.restart_fills(so2["u"], so2["v"], so2["h"], so2["uh"], so2["vh"], dt, have)
    m2f = {n: m2[n].copy() for n in ("h_av", "CAu_pred")}
    for n in range(2):
        m2.step(so2["u"], so2["v"], so2["h"], so2["uh"], so2["vh"], so2["uhtr"], so2["vhtr"], so2["eta_av"], inp["taux"], inp["tauy"], dt,
                inp["coefs"], calc_dtbt=False)
    assert np.isfinite(so2["u"]).all() and np.abs(so2["u"]).max() < 10.0
    # ---- device: the same
    dyc, forcing = new_model(cfg, inp, bt_mod)
    sg = dict(u=dyc.to_dev(inp["u"]), v=dyc.to_dev(inp["v"]), h=dyc.to_dev(inp["h"]), uh=dyc.zeros3(), vh=dyc.zeros3(),
              uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
    dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)
    for n in range(2):
        step(dyc, sg, forcing, dt, n == 0)
    dyc.sync()    # (the clones below run on torch's stream, the steps on the context's)
    keep = {k: sg[k].clone() for k in ("u", "v", "h", "uhtr", "vhtr")}
    keep_cs = {k: dyc.rk2_field(k).clone() for k in ("eta", "u_av", "v_av", "diffu", "diffv")}
    keep_bt = {k: dyc.barotropic_field(k).clone() for k in ("ubtav", "vbtav")}
    dtbt = dyc.barotropic_dtbt()
    import torch
    torch.cuda.synchronize(); dyc.close()
    dyc, forcing = new_model(cfg, inp, bt_mod)
    dyc.sync()    # (the context zeroes its arrays on its own stream; the copies below run on torch's)
    sg = dict(u=keep["u"], v=keep["v"], h=keep["h"], uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=keep["uhtr"], vhtr=keep["vhtr"], eta_av=dyc.zeros2())
    for k, a in keep_cs.items():
        dyc.rk2_field(k).copy_(a)
    for k, a in keep_bt.items():
        dyc.barotropic_field(k).copy_(a)
    dyc.barotropic_dtbt(dtbt)
    import torch
    torch.cuda.synchronize()
    dyc.dyn_split_RK2_restart_fills(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt, have)
    dyc.sync()
    h_av = dyc.rk2_field("h_av").cpu().numpy()
    H.assert_bitwise(h_av, m2f["h_av"], "restart fills: h_av", H.interior(d, "h"))
    H.assert_bitwise(dyc.rk2_field("CAu_pred").cpu().numpy(), m2f["CAu_pred"], "restart fills: CAu_pred", H.interior(d, "u"))
    assert h_av[(Ellipsis,) + tuple(H.interior(d, "h"))].min() > 0.0
    for n in range(2):
        step(dyc, sg, forcing, dt, False)
    dyc.sync()
    for k in ("u", "v", "h", "uh", "vh", "eta_av"):
        H.assert_bitwise(sg[k].cpu().numpy(), so2[k], "restart without CAu:" + k, H.interior(d, {"u": "u", "uh": "u", "v": "v", "vh": "v"}.get(k, "h")))
    dyc.close()
