"""step_MOM in miniature (MOM.F90:step_MOM -> step_MOM_dynamics :1312, step_MOM_tracer_dyn :1395, step_MOM_thermo :1751):
two baroclinic steps, tracer advection with the accumulated transports, then an ALE step -- z* regridding, remapping of
T, S, u, v and of the auxiliary restart variables (remap_dyn_split_RK2_aux_vars) -- and a line of ocean.stats, repeated.
The switches are those of .testing/tc2 that touch this path (Z* + REMAPPING_SCHEME = PPM_IH4, BOUND_CORIOLIS,
MASS_WEIGHT_IN_PRESSURE_GRADIENT with an equation of state, Smagorinsky KH / AH, BT_PROJECT_VELOCITY, BEBT = 0.2,
DTBT = -0.95, ETA_TOLERANCE = 1e-6, VELOCITY_TOLERANCE = 1e-3, DT_THERM = 2 DT) plus REMAP_AUXILIARY_VARS.
Every array the next stage reads must be bit-identical to the oracle's, cycle after cycle."""
import numpy as np
import pytest

from mom6_amd import abi, synth, sum_output as SO
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = abi.G


@pytest.mark.parametrize("cfg_name,remap_aux", [("benchmark_small", 1), ("island_basin", 0)])
def test_dynamics_tracers_ALE_cycle(orc, cfg_name, remap_aux, sums):
    import torch
    from mom6_amd.dycore import Dycore
    from tests import cases
    from tests.test_dyn_gpu import visc_inputs
    cfg = getattr(H, cfg_name)(nk=8)
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg, False, False)
    GV, Rlay, gp, dt = inp["GV"], inp["Rlay"], inp["gp"], inp["dt"]
    bt_mod = dict(strong_drag=1, BT_project_velocity=1, bebt=0.2, dtbt_fraction=0.95)
    cont_mod = dict(tol_eta=1.0e-6, tol_vel=1.0e-3)
    T0, S0 = cases.thermo_state(d, M)
    eos = abi.eos_params_default(abi.WRIGHT); eos.MassWghtInterp = 1
    hvP = abi.hor_visc_params_default(dt)
    for k_, v_ in dict(Laplacian=1, Kh_vel_scale=0.05, Smagorinsky_Kh=1, Smag_Lap_const=0.06, Ah_vel_scale=0.05, Smagorinsky_Ah=1,
                       Smag_bi_const=0.06).items():
        setattr(hvP, k_, v_)
    hvP.dt = dt
    vvP = abi.vertvisc_params_default()
    vis = visc_inputs(d, M)
    depth = float(M[G["bathyT"]].max())
    cr = np.linspace(1.0, 4.0, d.nk); cr *= depth / cr.sum()
    RP = abi.regrid_zstar_params_default(min_thickness=1.0e-3)
    RS = abi.remapping_params_default(abi.REMAP_PPM_IH4, GV.H_subroundoff, boundary_extrapolation=0)
    SP = abi.sum_output_params_default(dt, use_temperature=1, C_p=3925.0)

    def params():
        cont, bt, cor, pgf, rk2 = cases.rk2_params(d, GV, bt_mod, dict(remap_aux=remap_aux), dict(bound_Coriolis=1))
        for k_, v_ in cont_mod.items():
            setattr(cont, k_, v_)
        return cont, bt, cor, pgf, rk2

    # ---------------- oracle model
    cont, bt, cor, pgf, rk2 = params()
    m = orc.OrcModel(d, M, GV, cont, bt, cor, pgf, rk2, Rlay, gp, 0)
    so = dict(u=inp["u"].copy(), v=inp["v"].copy(), h=inp["h"].copy(), uh=np.zeros_like(inp["h"]), vh=np.zeros_like(inp["h"]),
              uhtr=np.zeros_like(inp["h"]), vhtr=np.zeros_like(inp["h"]), eta_av=np.zeros(d.shape2()), T=T0.copy(), S=S0.copy())
    m.set_tv(so["T"], so["S"], eos)
    m.set_vertvisc(vvP, *vis, inp["coefs"][0][4], inp["coefs"][0][5])
    m.set_hor_visc(hvP)
    m.initialize(so["u"], so["v"], so["h"], so["uh"], so["vh"], dt)
    st = orc.SumOutputState(d, M, GV, gp, SP)
    # ---------------- device model
    cont2, bt2, cor2, pgf2, rk22 = params()
    dyc = Dycore(d, M, GV, 0)
    dyc.continuity_init(cont2); dyc.barotropic_init(bt2); dyc.CoriolisAdv_init(cor2); dyc.PressureForce_init(pgf2, Rlay, gp)
    dyc.initialize_dyn_split_RK2(rk22)
    sg = {n: dyc.to_dev(a) for n, a in (("u", inp["u"]), ("v", inp["v"]), ("h", inp["h"]), ("T", T0), ("S", S0))}
    sg.update(uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
    dyc.PressureForce_set_tv(sg["T"], sg["S"], eos)
    dyc.vertvisc_init(vvP)
    dyc.vertvisc_set_visc(*[dyc.to_dev(a) if a is not None else None for a in list(vis) + [inp["coefs"][0][4], inp["coefs"][0][5]]])
    dyc.hor_visc_init(hvP)
    dyc.tracer_advect_init(dt, 2)            # TRACER_ADVECTION_SCHEME = "PPM"
    dyc.sum_output_init(SP, gp)
    txd, tyd = dyc.to_dev(inp["taux"]), dyc.to_dev(inp["tauy"])
    torch.cuda.synchronize()
    dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)
    h_new = torch.zeros_like(sg["h"]); dzI = torch.zeros((d.nk + 1,) + d.shape2(), dtype=torch.float64, device=dyc.device)
    hv = [torch.full_like(sg["h"], 1.0e-3) for _ in range(4)]
    torch.cuda.synchronize()              # (torch's fills run on its own stream)

    stats_o, stats_g = SO.SumOutput(use_temperature=True, C_p=3925.0), SO.SumOutput(use_temperature=True, C_p=3925.0)
    stag = dict(u="u", v="v", h="h", T="h", S="h", uh="u", vh="v")
    nstep = 0
    for cycle in range(2):
        for n in range(2):                                                            # DT_THERM = 2 DT
            m.step(so["u"], so["v"], so["h"], so["uh"], so["vh"], so["uhtr"], so["vhtr"], so["eta_av"], inp["taux"], inp["tauy"], dt,
                   inp["coefs"], calc_dtbt=(nstep == 0))
            dyc.step_MOM_dyn_split_RK2(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], sg["uhtr"], sg["vhtr"], sg["eta_av"], txd, tyd,
                                       dt, calc_dtbt=(nstep == 0))
            nstep += 1
            dyc.sync()
            for n_ in ("h", "u", "v", "uh", "vh"):
                H.assert_bitwise(sg[n_].cpu().numpy(), so[n_], f"step {nstep}: {n_}", H.interior(d, stag[n_]))
        # step_MOM_tracer_dyn: advect_tracer with the transports accumulated over the two steps, which are then cleared
        orc.advect_tracer(d, M, GV, 0, dt, 2, so["h"], so["uhtr"], so["vhtr"], 2 * dt, [so["T"], so["S"]])
        dyc.advect_tracer(sg["h"], sg["uhtr"], sg["vhtr"], 2 * dt, [sg["T"], sg["S"]])
        dyc.sync()                                   # torch works on its own stream: wait for the context's before touching its arrays
        so["uhtr"][:] = 0.0; so["vhtr"][:] = 0.0; sg["uhtr"].zero_(); sg["vhtr"].zero_()
        torch.cuda.synchronize()
        for n_ in ("T", "S"):
            H.assert_bitwise(sg[n_].cpu().numpy(), so[n_], f"cycle {cycle}: advected {n_}", H.interior(d, "h"))
        # step_MOM_thermo -> ALE_regridding_and_remapping (MOM.F90:1751)
        hn_o = np.zeros_like(so["h"]); dz_o = np.zeros((d.nk + 1,) + d.shape2())
        orc.ALE_regrid_zstar(d, M, GV, RP, cr, so["h"], hn_o, dz_o)
        orc.ALE_remap_tracers(d, M, RS, so["h"], hn_o, [so["T"], so["S"]])
        ho = [np.full_like(so["h"], 1.0e-3) for _ in range(4)]
        orc.ALE_remap_set_h_vel(d, M, so["h"], ho[0], ho[1]); orc.ALE_remap_set_h_vel(d, M, hn_o, ho[2], ho[3])
        orc.ALE_remap_velocities(d, M, RS, ho[0], ho[1], ho[2], ho[3], so["u"], so["v"])
        m.remap_aux_vars(RS, ho[0], ho[1], ho[2], ho[3])
        so["h"][:] = hn_o
        dyc.ALE_regrid_zstar(RP, cr, sg["h"], h_new, dzI)
        dyc.ALE_remap_tracers(RS, sg["h"], h_new, [sg["T"], sg["S"]])
        dyc.ALE_remap_set_h_vel(sg["h"], hv[0], hv[1]); dyc.ALE_remap_set_h_vel(h_new, hv[2], hv[3])
        dyc.ALE_remap_velocities(RS, hv[0], hv[1], hv[2], hv[3], sg["u"], sg["v"])
        dyc.remap_dyn_split_RK2_aux_vars(RS, hv[0], hv[1], hv[2], hv[3])
        dyc.sync()
        sg["h"].copy_(h_new)
        torch.cuda.synchronize()
        for n_ in ("u", "v", "h", "T", "S", "uh", "vh"):
            H.assert_bitwise(sg[n_].cpu().numpy(), so[n_], f"cycle {cycle}: {n_}", H.interior(d, stag[n_]))
        for n_ in ("u_av", "CAu_pred", "diffu", "eta"):
            H.assert_bitwise(dyc.rk2_field(n_).cpu().numpy(), m[n_], f"cycle {cycle}: CS%{n_}", H.interior(d, "h" if n_ == "eta" else "u"))
        stats_o.record(orc.write_energy(st, so["u"], so["v"], so["h"], so["T"], so["S"]), nstep * dt, nstep)
        stats_g.record(dyc.write_energy(sg["u"], sg["v"], sg["h"], sg["T"], sg["S"]), nstep * dt, nstep)
    assert stats_g.lines == stats_o.lines and len(stats_g.lines) == 2
    assert np.abs(dz_o).max() > 0.0 and np.isfinite(so["u"]).all()
    # remapping and advection conserve salt and heat: the anomaly columns of the second line are round-off
    se, te = (float(stats_g.lines[1].split(tag)[1].split(",")[0]) for tag in ("Se ", "Te "))
    assert abs(se) < 1e-12 and abs(te) < 1e-12
    if remap_aux:
        assert np.abs(m["diffu"]).max() > 0.0
    dyc.close()
