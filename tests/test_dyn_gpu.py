"""GPU parity, bit for bit: CorAdCalc, PressureForce_FV_Bouss (layered path), vertvisc, vertvisc_remnant."""
import numpy as np
import pytest

from mom6_amd import abi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = abi.G


def visc_coefs(d, M, seed=3):
    """Plausible vertvisc_coef outputs: a_u [nk+1] (H T-1), h_u [nk] (H)."""
    nk = d.nk
    h, _, _ = synth.make_state(d, M)
    a = np.zeros((nk + 1,) + d.shape2())
    for K in range(1, nk + 1):
        a[K] = 1e-4 * (1.0 + 0.5 * synth.smooth_field(d, seed + K, ox=1.0, oy=0.5)) / 10.0
    a[nk] *= 20.0   # bottom drag
    a_u = a * M[G["mask2dCu"]][None]; a_v = a * M[G["mask2dCv"]][None]
    h_u = np.zeros_like(h); h_v = np.zeros_like(h)
    h_u[:, :, :-1] = 0.5 * (h[:, :, :-1] + h[:, :, 1:]); h_v[:, :-1, :] = 0.5 * (h[:, :-1, :] + h[:, 1:, :])
    h_u = np.maximum(h_u, 1e-9); h_v = np.maximum(h_v, 1e-9)
    Ray = 1e-5 * (1 + synth.smooth_field(d, seed + 50, nk=nk, ox=0.5, oy=0.5))
    return [np.ascontiguousarray(x) for x in (a_u, a_v, h_u, h_v, Ray, Ray.copy())]


@pytest.mark.parametrize("cfg", ["double_gyre", "channel", "benchmark_small"])
@pytest.mark.parametrize("mods", [dict(), dict(KE_Scheme=abi.KE_GUDONOV), dict(KE_Scheme=abi.KE_SIMPLE_GUDONOV, no_slip=1),   # (k_corad_lds)
                                  dict(bound_Coriolis=1), dict(Coriolis_Scheme=abi.ARAKAWA_HSU90, KE_Scheme=abi.KE_GUDONOV),
                                  dict(Coriolis_Scheme=abi.SADOURNY75_ENSTRO, KE_Scheme=abi.KE_SIMPLE_GUDONOV, no_slip=1, bound_Coriolis=1),
                                  dict(Coriolis_En_Dis=1), dict(Coriolis_En_Dis=1, bound_Coriolis=1, KE_Scheme=abi.KE_GUDONOV),
                                  dict(Coriolis_En_Dis=1, Coriolis_Scheme=abi.ARAKAWA_HSU90),
                                  dict(Coriolis_Scheme=abi.ARAKAWA_LAMB81),
                                  dict(Coriolis_Scheme=abi.ARAKAWA_LAMB81, bound_Coriolis=1, KE_Scheme=abi.KE_SIMPLE_GUDONOV, no_slip=1),
                                  dict(Coriolis_Scheme=abi.AL_BLEND, rough=1),
                                  dict(Coriolis_Scheme=abi.AL_BLEND, F_eff_max_blend=3.0, wt_lin_blend=0.5, bound_Coriolis=1, rough=1),
                                  dict(Coriolis_Scheme=abi.AL_BLEND, F_eff_max_blend=2.0, wt_lin_blend=0.0),
                                  dict(Coriolis_Scheme=abi.ROBUST_ENSTRO, rough=1),
                                  dict(Coriolis_Scheme=abi.ROBUST_ENSTRO, PV_Adv_Scheme=abi.PV_ADV_UPWIND1, bound_Coriolis=1,
                                       Coriolis_En_Dis=1, rough=1)],
                         ids=lambda m: "-".join(f"{k}={v}" for k, v in m.items()) or "default")
def test_CorAdCalc(orc, cfg, mods):
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = getattr(H, cfg)()
    GV = abi.vgrid_default()
    CS = abi.coriolis_params_default()
    mods = dict(mods); rough = mods.pop("rough", 0)
    for k, v in mods.items():
        setattr(CS, k, v)
    h, u, v = synth.make_state(d, M, thin_frac=0.05)
    if rough:    # sharp thickness contrasts: every regime of the blend's weights, ROBUST_ENSTRO's Heff clamps binding
        h = H.roughen(h)
    uh = u * 1.0e5 * (1 + 0.1 * synth.smooth_field(d, 7, nk=d.nk)); vh = v * 1.0e5
    if mods.get("Coriolis_En_Dis"):
        # CORIOLIS_EN_DIS (tc4) compares the continuity solver's transport with the centred one: give it ratios from
        # 0.05 to 3, both signs and exact equality so that every branch of the bracketing is taken
        rng = np.random.default_rng(9)
        G = abi.G
        uc = 0.5 * M[G["dy_Cu"]][None] * u * (h + np.roll(h, -1, axis=2))
        vc = 0.5 * M[G["dx_Cv"]][None] * v * (h + np.roll(h, -1, axis=1))
        fu = rng.choice([0.05, 0.2, 0.3, 0.6, 1.0, 1.5, 2.0, 3.0, -0.5], size=u.shape)
        fv = rng.choice([0.05, 0.2, 0.3, 0.6, 1.0, 1.5, 2.0, 3.0, -0.5], size=v.shape)
        uh, vh = uc * fu, vc * fv
    uh = np.ascontiguousarray(uh); vh = np.ascontiguousarray(vh)
    CAu = np.zeros_like(h); CAv = np.zeros_like(h)
    orc.CorAdCalc(d, M, GV, CS, u, v, h, uh, vh, CAu, CAv)
    dyc = Dycore(d, M, GV)
    dyc.CoriolisAdv_init(CS)
    gu, gv = dyc.zeros3(), dyc.zeros3()
    T = [dyc.to_dev(x) for x in (u, v, h, uh, vh)]
    torch.cuda.synchronize()
    dyc.CorAdCalc(*T, gu, gv)
    dyc.sync()
    H.assert_bitwise(gu.cpu().numpy(), CAu, "CAu", H.interior(d, "u"))
    H.assert_bitwise(gv.cpu().numpy(), CAv, "CAv", H.interior(d, "v"))
    assert np.abs(CAu).max() > 0
    dyc.close()


def test_CorAdCalc_other_kernels_are_bit_identical_too():
    """The default configuration runs k_corad_lds (inputs, q and KE through LDS), every other one k_corad_fused (the cases above
    with BOUND_CORIOLIS, CORIOLIS_EN_DIS or another scheme).  The two-kernel form through HBM (MOM6X_CORAD=legacy) is held to the
    same oracle: the cases above again in a process with the switch set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env in (dict(MOM6X_CORAD="legacy"),):
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_dyn_gpu.py"), "-m", "gpu", "-q", "-x",
                            "-k", "test_CorAdCalc and not other_kernels"], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=900, cwd=root)
        assert r.returncode == 0 and " passed" in r.stdout, str(env) + r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("cfg", ["double_gyre", "channel", "benchmark_small"])
@pytest.mark.parametrize("bug", [1, 0])
def test_PressureForce(orc, cfg, bug):
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = getattr(H, cfg)()
    GV = abi.vgrid_default()
    CS = abi.pgf_params_default(GV.Rho0)
    CS.rho_ref_bug = bug
    if not bug:
        CS.rho_ref = 1030.0
    Rlay, gp = abi.layer_densities(d.nk, GV.Rho0, GV.g_Earth)
    h, _, _ = synth.make_state(d, M, thin_frac=0.05)
    o = dict(PFu=np.zeros_like(h), PFv=np.zeros_like(h), pbce=np.zeros_like(h), eta=np.zeros(d.shape2()))
    orc.PressureForce(d, M, GV, CS, Rlay, gp, h, o["PFu"], o["PFv"], o["pbce"], o["eta"])
    dyc = Dycore(d, M, GV)
    dyc.PressureForce_init(CS, Rlay, gp)
    g = dict(PFu=dyc.zeros3(), PFv=dyc.zeros3(), pbce=dyc.zeros3(), eta=dyc.zeros2())
    hd = dyc.to_dev(h)
    torch.cuda.synchronize()
    dyc.PressureForce(hd, g["PFu"], g["PFv"], g["pbce"], g["eta"])
    dyc.sync()
    H.assert_bitwise(g["PFu"].cpu().numpy(), o["PFu"], "PFu", H.interior(d, "u"))
    H.assert_bitwise(g["PFv"].cpu().numpy(), o["PFv"], "PFv", H.interior(d, "v"))
    sl = d.sl(-1, d.ni, -1, d.nj)
    H.assert_bitwise(g["pbce"].cpu().numpy(), o["pbce"], "pbce", sl)
    H.assert_bitwise(g["eta"].cpu().numpy(), o["eta"], "eta", sl)
    assert np.abs(o["PFu"]).max() > 0
    dyc.close()


@pytest.mark.parametrize("cfg", ["double_gyre", "benchmark_small"])
@pytest.mark.parametrize("form", ["LINEAR", "WRIGHT", "WRIGHT_FULL", "WRIGHT_REDUCED", "UNESCO", "ROQUET_RHO", "JACKETT06", "ROQUET_SPV"])
@pytest.mark.parametrize("mods", [dict(), dict(MassWghtInterp=1), dict(MassWghtInterp=3, use_SSH_in_Z0p=1, bug=0, dRho_dp=4.5e-7),
                                  # the ALE path: TS_PLM_edge_values + int_density_dz_generic_plm (PRESSURE_RECONSTRUCTION_SCHEME = 1)
                                  dict(Recon_Scheme=1), dict(Recon_Scheme=1, boundary_extrap=0, MassWghtInterp=1),
                                  dict(Recon_Scheme=1, MassWghtInterp=3, MassWghtInterpVanOnly=1, h_nonvanished=5.0, use_SSH_in_Z0p=1, bug=0),
                                  # PRESSURE_RECONSTRUCTION_SCHEME = 2: TS_PPM_edge_values + int_density_dz_generic_ppm
                                  dict(Recon_Scheme=2), dict(Recon_Scheme=2, boundary_extrap=0, MassWghtInterp=1),
                                  dict(Recon_Scheme=2, MassWghtInterp=3, MassWghtInterpVanOnly=1, h_nonvanished=5.0, use_SSH_in_Z0p=1, bug=0),
                                  # EOS_QUADRATURE: int_density_dz_generic_pcm
                                  dict(EOS_quadrature=1), dict(EOS_quadrature=1, MassWghtInterp=1, dRho_dp=4.5e-7),
                                  dict(EOS_quadrature=1, MassWghtInterp=3, MassWghtInterpVanOnly=1, h_nonvanished=5.0, use_SSH_in_Z0p=1, bug=0)],
                         ids=lambda m: "-".join(f"{k}={v}" for k, v in m.items()) or "default")
def test_PressureForce_with_equation_of_state(orc, cfg, form, mods):
    """The use_EOS branch (int_density_dz -> analytic linear / Wright integrals, or with Recon_Scheme = 1 the PLM edge values
    and the generic 5-point quadratures; Set_pbce_Bouss with T and S)."""
    import torch
    from mom6_amd.dycore import Dycore
    from tests import cases
    gg, d, M = getattr(H, cfg)()
    GV = abi.vgrid_default()
    CS = abi.pgf_params_default(GV.Rho0)
    eos = abi.eos_params_default(getattr(abi, form))
    for k, v in mods.items():
        if k == "bug":
            CS.rho_ref_bug = v; CS.rho_ref = 1030.0
        else:
            setattr(eos, k, v)
    Rlay, gp = abi.layer_densities(d.nk, GV.Rho0, GV.g_Earth)
    h, _, _ = synth.make_state(d, M, thin_frac=0.05)
    T, S = cases.thermo_state(d, M)
    o = dict(PFu=np.zeros_like(h), PFv=np.zeros_like(h), pbce=np.zeros_like(h), eta=np.zeros(d.shape2()))
    if eos.form in (abi.UNESCO, abi.ROQUET_RHO, abi.JACKETT06, abi.ROQUET_SPV) and not (eos.Recon_Scheme or eos.EOS_quadrature):   # MOM_EOS.F90:1495
        dyc = Dycore(d, M, GV)
        dyc.PressureForce_init(CS, Rlay, gp)
        with pytest.raises(abi.Mom6xError, match="No analytic integration option is available with this EOS!"):
            dyc.PressureForce_set_tv(dyc.to_dev(T), dyc.to_dev(S), eos)
        dyc.close()
        return
    if eos.Recon_Scheme == 2 and d.nk < 4:   # edge_values_implicit_h4 needs four layers: refused, not approximated
        dyc = Dycore(d, M, GV)
        dyc.PressureForce_init(CS, Rlay, gp)
        with pytest.raises(abi.Mom6xError, match="NK >= 4"):
            dyc.PressureForce_set_tv(dyc.to_dev(T), dyc.to_dev(S), eos)
        dyc.close()
        return
    orc.PressureForce(d, M, GV, CS, Rlay, gp, h, o["PFu"], o["PFv"], o["pbce"], o["eta"], T=T, S=S, eos=eos)
    dyc = Dycore(d, M, GV)
    dyc.PressureForce_init(CS, Rlay, gp)
    Td, Sd = dyc.to_dev(T), dyc.to_dev(S)
    dyc.PressureForce_set_tv(Td, Sd, eos)
    g = dict(PFu=dyc.zeros3(), PFv=dyc.zeros3(), pbce=dyc.zeros3(), eta=dyc.zeros2())
    hd = dyc.to_dev(h)
    torch.cuda.synchronize()
    dyc.PressureForce(hd, g["PFu"], g["PFv"], g["pbce"], g["eta"])
    dyc.sync()
    H.assert_bitwise(g["PFu"].cpu().numpy(), o["PFu"], "PFu", H.interior(d, "u"))
    H.assert_bitwise(g["PFv"].cpu().numpy(), o["PFv"], "PFv", H.interior(d, "v"))
    sl = d.sl(-1, d.ni, -1, d.nj)
    H.assert_bitwise(g["pbce"].cpu().numpy(), o["pbce"], "pbce", sl)
    H.assert_bitwise(g["eta"].cpu().numpy(), o["eta"], "eta", sl)
    assert np.abs(o["PFu"]).max() > 0 and np.isfinite(o["PFu"]).all()
    if eos.Recon_Scheme == 1:   # ALE_PLM_edge_values itself (public in MOM_ALE), through its own entry point
        Qt_o, Qb_o = np.zeros_like(h), np.zeros_like(h)
        orc.ALE_PLM_edge_values(d, GV, h, T, eos.boundary_extrap, Qt_o, Qb_o)
        Qt, Qb = dyc.zeros3(), dyc.zeros3()
        dyc.ALE_PLM_edge_values(hd, Td, eos.boundary_extrap, Qt, Qb); dyc.sync()
        H.assert_bitwise(Qt.cpu().numpy(), Qt_o, "T_t", sl); H.assert_bitwise(Qb.cpu().numpy(), Qb_o, "T_b", sl)
    if eos.Recon_Scheme == 2:   # one field of TS_PPM_edge_values through its own entry point
        Qt_o, Qb_o = np.zeros_like(h), np.zeros_like(h)
        orc.ALE_PPM_edge_values(d, GV, h, S, eos.boundary_extrap, Qt_o, Qb_o)
        Qt, Qb = dyc.zeros3(), dyc.zeros3()
        dyc.ALE_PPM_edge_values(hd, Sd, eos.boundary_extrap, Qt, Qb); dyc.sync()
        H.assert_bitwise(Qt.cpu().numpy(), Qt_o, "S_t", sl); H.assert_bitwise(Qb.cpu().numpy(), Qb_o, "S_b", sl)
    # back to the layered path
    dyc.PressureForce_set_tv(None, None, None)
    o2 = dict(PFu=np.zeros_like(h), PFv=np.zeros_like(h))
    orc.PressureForce(d, M, GV, CS, Rlay, gp, h, o2["PFu"], o2["PFv"])
    dyc.PressureForce(hd, g["PFu"], g["PFv"]); dyc.sync()
    H.assert_bitwise(g["PFu"].cpu().numpy(), o2["PFu"], "PFu layered", H.interior(d, "u"))
    dyc.close()


@pytest.mark.parametrize("cfg", ["double_gyre", "benchmark_small"])
@pytest.mark.parametrize("use_ray", [False, True])
@pytest.mark.parametrize("Hmix_stress", [0.0, 20.0, 900.0])
def test_vertvisc_and_remnant(orc, cfg, use_ray, Hmix_stress):
    """Hmix_stress > 0: DIRECT_STRESS (.testing/tc3, tc4) with HMIX_STRESS inside the top layer (20 m) or spanning
    several layers (900 m)."""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = getattr(H, cfg)()
    GV = abi.vgrid_default()
    a_u, a_v, h_u, h_v, Ray_u, Ray_v = visc_coefs(d, M)
    if not use_ray:
        Ray_u = Ray_v = None
    h, u, v = synth.make_state(d, M)
    taux = np.ascontiguousarray(0.1 * synth.smooth_field(d, 41, ox=1.0, oy=0.5) * M[G["mask2dCu"]])
    tauy = np.ascontiguousarray(0.05 * synth.smooth_field(d, 42, ox=0.5, oy=1.0) * M[G["mask2dCv"]])
    dt = 600.0
    uo, vo = u.copy(), v.copy()
    tbu, tbv = np.zeros(d.shape2()), np.zeros(d.shape2())
    orc.vertvisc(d, M, GV, uo, vo, a_u, a_v, h_u, h_v, Ray_u, Ray_v, taux, tauy, dt, tbu, tbv, Hmix_stress=Hmix_stress, h=h)
    vru, vrv = np.zeros_like(u), np.zeros_like(u)
    orc.vertvisc_remnant(d, M, vru, vrv, a_u, a_v, h_u, h_v, Ray_u, Ray_v, dt)
    dyc = Dycore(d, M, GV)
    Ts = [dyc.to_dev(x) if x is not None else None for x in (a_u, a_v, h_u, h_v, Ray_u, Ray_v)]
    dyc.vertvisc_set_coef(*Ts)
    ug, vg = dyc.to_dev(u), dyc.to_dev(v)
    tbug, tbvg, vrug, vrvg = dyc.zeros2(), dyc.zeros2(), dyc.zeros3(), dyc.zeros3()
    tx, ty = dyc.to_dev(taux), dyc.to_dev(tauy)
    hd = dyc.to_dev(h)
    if Hmix_stress > 0.0:
        dyc.vertvisc_set_direct_stress(Hmix_stress, hd)
    torch.cuda.synchronize()
    dyc.vertvisc(ug, vg, tx, ty, dt, tbug, tbvg)
    dyc.vertvisc_remnant(vrug, vrvg, dt)
    dyc.sync()
    for name, a, b, st in (("u", ug, uo, "u"), ("v", vg, vo, "v"), ("taux_bot", tbug, tbu, "u"), ("tauy_bot", tbvg, tbv, "v"),
                           ("visc_rem_u", vrug, vru, "u"), ("visc_rem_v", vrvg, vrv, "v")):
        H.assert_bitwise(a.cpu().numpy(), b, name, H.interior(d, st))
    assert 0 < vru[(Ellipsis,) + H.interior(d, "u")].max() <= 1.0 + 1e-12
    dyc.close()


def visc_inputs(d, M, with_shear=True):
    """visc%Kv_bbl_u/v, visc%bbl_thick_u/v (set_viscous_BBL outputs) and visc%Kv_shear: plausible fields."""
    Kv_bbl_u = np.ascontiguousarray((2e-3 * (1 + 0.5 * synth.smooth_field(d, 91, ox=1, oy=.5))) * M[G["mask2dCu"]])
    Kv_bbl_v = np.ascontiguousarray((2e-3 * (1 + 0.5 * synth.smooth_field(d, 92, ox=.5, oy=1))) * M[G["mask2dCv"]])
    bt_u = np.ascontiguousarray(8.0 * (1 + 0.6 * synth.smooth_field(d, 93, ox=1, oy=.5)))
    bt_v = np.ascontiguousarray(8.0 * (1 + 0.6 * synth.smooth_field(d, 94, ox=.5, oy=1)))
    Kv_shear = None
    if with_shear:
        Kv_shear = np.ascontiguousarray(1e-3 * np.abs(synth.smooth_field(d, 95, nk=d.nk + 1, ox=.5, oy=.5)))
    return Kv_bbl_u, Kv_bbl_v, bt_u, bt_v, Kv_shear


@pytest.mark.parametrize("cfg", ["double_gyre", "benchmark_small"])
@pytest.mark.parametrize("mods", [dict(), dict(harmonic_visc=1), dict(bottomdraglaw=0, Kv_extra_bbl=5e-4, harm_BL_val=0.5),
                                  dict(bottomdraglaw=0), dict(Kvml_invZ2=1e-3, answer_date=20181231, harm_BL_val=1.0)])
def test_vertvisc_coef(orc, cfg, mods):
    """vertvisc_coef + find_coupling_coef on the device against the oracle, bit for bit."""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = getattr(H, cfg)()
    GV = abi.vgrid_default()
    P = abi.vertvisc_params_default()
    for k, v in mods.items():
        setattr(P, k, v)
    h, u, v = synth.make_state(d, M, thin_frac=0.1)
    Kbu, Kbv, btu, btv, Ksh = visc_inputs(d, M, with_shear=("harmonic_visc" not in mods))
    o = dict(a_u=np.zeros((d.nk + 1,) + d.shape2()), a_v=np.zeros((d.nk + 1,) + d.shape2()), h_u=np.zeros_like(h), h_v=np.zeros_like(h))
    orc.vertvisc_coef(d, M, GV, P, u, v, h, 1200.0, o["a_u"], o["a_v"], o["h_u"], o["h_v"], Kbu, Kbv, btu, btv, Ksh)
    dyc = Dycore(d, M, GV)
    dyc.vertvisc_init(P)
    dev = [dyc.to_dev(a) if a is not None else None for a in (Kbu, Kbv, btu, btv, Ksh)]
    dyc.vertvisc_set_visc(*dev)
    ud, vd, hd = dyc.to_dev(u), dyc.to_dev(v), dyc.to_dev(h)
    torch.cuda.synchronize()
    dyc.vertvisc_coef(ud, vd, hd, 1200.0)
    dyc.sync()
    for n, stg in (("a_u", "u"), ("a_v", "v"), ("h_u", "u"), ("h_v", "v")):
        H.assert_bitwise(dyc.vertvisc_field(n).cpu().numpy(), o[n], n, H.interior(d, stg))
    assert o["a_u"][1:].max() > 0 and np.isfinite(o["a_u"]).all() and o["h_u"].max() > 0
    dyc.close()
