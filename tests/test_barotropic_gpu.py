"""GPU parity: HIP barotropic solver (btcalc, bt_mass_source, set_dtbt, btstep) vs the oracle.

Everything except the single `av_rem**(1/nstep)` (libm pow vs device pow) is bit-exact by
construction, so:  BT_STRONG_DRAG=True (no pow) must be BIT-IDENTICAL; the default path must agree
to 1e-12 of each field's range (FP64, ~100 sub-steps of ulp-level drift)."""
import numpy as np
import pytest

from mom6_amd import abi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = abi.G


def make_inputs(orc, cfg, first_direction=0, u_max=0.1):
    gg, d, M = cfg
    GV = abi.vgrid_default()
    CS = abi.continuity_params_default(d.nk, GV.Angstrom_H)
    h, u, v = synth.make_state(d, M, u_max=u_max)
    dt = 1200.0
    vr_u = np.clip(0.8 + 0.3 * synth.smooth_field(d, 11, nk=d.nk, ox=1.0, oy=0.5), 0, 1) * (M[G["mask2dCu"]] > 0)
    vr_v = np.clip(0.8 + 0.3 * synth.smooth_field(d, 12, nk=d.nk, ox=0.5, oy=1.0), 0, 1) * (M[G["mask2dCv"]] > 0)
    bt = orc.new_bt_cont(d)
    hp = np.zeros_like(h); uh = np.zeros_like(h); vh = np.zeros_like(h)
    orc.continuity_PPM(d, M, GV, CS, first_direction, u, v, h, hp, uh, vh, dt, visc_rem_u=vr_u, visc_rem_v=vr_v, BT_cont=bt)
    eta = (h.sum(0) - M[G["bathyT"]]) * M[G["mask2dT"]]
    pbce = np.zeros_like(h)
    pbce[0] = 9.8
    for k in range(1, d.nk):
        pbce[k] = pbce[k - 1] + 0.02 * (1.0 + 0.1 * synth.smooth_field(d, 30 + k, ox=0.5, oy=0.5))
    taux = 0.1 * synth.smooth_field(d, 41, ox=1.0, oy=0.5) * M[G["mask2dCu"]]
    tauy = 0.05 * synth.smooth_field(d, 42, ox=0.5, oy=1.0) * M[G["mask2dCv"]]
    bcu = 1e-6 * synth.smooth_field(d, 21, nk=d.nk, ox=1, oy=.5) * M[G["mask2dCu"]]
    bcv = 1e-6 * synth.smooth_field(d, 22, nk=d.nk, ox=.5, oy=1) * M[G["mask2dCv"]]
    eta_PF = eta * (1.0 + 0.01 * synth.smooth_field(d, 43, ox=0.5, oy=0.5))
    ucor = u * 0.9; vcor = v * 0.9
    return dict(d=d, M=M, GV=GV, h=h, u=u, v=v, dt=dt, vr_u=np.ascontiguousarray(vr_u), vr_v=np.ascontiguousarray(vr_v),
                bt=bt, uh=uh, vh=vh, eta=np.ascontiguousarray(eta), pbce=pbce, taux=np.ascontiguousarray(taux),
                tauy=np.ascontiguousarray(tauy), bcu=np.ascontiguousarray(bcu), bcv=np.ascontiguousarray(bcv),
                eta_PF=np.ascontiguousarray(eta_PF), ucor=ucor, vcor=vcor, first_direction=first_direction)


def run_both(orc, I, pmod=None, use_uh0=True, use_etaav=True, bottom=False, default_thick=False, no_bt_cont=False, eta_ms=None):
    import torch
    from mom6_amd.dycore import Dycore, BTContDev
    d, M, GV = I["d"], I["M"], I["GV"]
    P = abi.barotropic_params_default(20.0)
    for k, v in (pmod or {}).items():
        setattr(P, k, v)
    # ---------------- oracle
    cs = orc.BtState(d)
    orc.barotropic_init(d, M, GV, P, cs)
    if default_thick:
        orc.btcalc(d, M, GV, I["h"], None, None, cs, scheme=P.bt_thick_scheme)
    else:
        orc.btcalc(d, M, GV, I["h"], I["bt"]["h_u"], I["bt"]["h_v"], cs)
    dtbt, _ = orc.set_dtbt(d, M, GV, P, cs, gtot_est=9.8 + 0.02 * d.nk, SSH_add=10.0)
    P.dtbt = dtbt
    eta_ms = I["eta"] if eta_ms is None else eta_ms   # (the eta bt_mass_source compares sum(h) with: eta_cor is their difference)
    orc.bt_mass_source(d, M, GV, I["h"], eta_ms, True, cs)
    z3 = lambda: np.zeros(d.shape3()); z2 = lambda: np.zeros(d.shape2())
    o = dict(alu=z3(), alv=z3(), eta_out=z2(), uhbtav=z2(), vhbtav=z2(), etaav=z2() if use_etaav else None)
    kw = {}
    if use_uh0:
        kw.update(uh0=I["uh"], vh0=I["vh"], u_uh0=I["u"], v_vh0=I["v"])
    if bottom:
        kw.update(taux_bot=0.3 * I["taux"], tauy_bot=0.3 * I["tauy"])
    nstep = orc.btstep(d, M, GV, P, cs, I["first_direction"], I["u"], I["v"], I["eta"], I["dt"], I["bcu"], I["bcv"],
                       I["taux"], I["tauy"], I["pbce"], I["eta_PF"], I["ucor"], I["vcor"], o["alu"], o["alv"],
                       o["eta_out"], o["uhbtav"], o["vhbtav"], I["vr_u"], I["vr_v"], None if no_bt_cont else I["bt"], etaav=o["etaav"], **kw)
    assert nstep >= 2
    # ---------------- device
    dyc = Dycore(d, M, GV, I["first_direction"])
    P2 = abi.barotropic_params_default(20.0)
    for k, v in (pmod or {}).items():
        setattr(P2, k, v)
    dyc.barotropic_init(P2)
    T = {k: dyc.to_dev(I[k]) for k in ("h", "u", "v", "eta", "bcu", "bcv", "taux", "tauy", "pbce", "eta_PF", "ucor", "vcor",
                                       "vr_u", "vr_v", "uh", "vh")}
    btd = BTContDev(dyc)
    for n in abi.BTCont._names:
        btd[n].copy_(torch.from_numpy(I["bt"][n]))
    torch.cuda.synchronize()
    if default_thick:
        dyc.btcalc(T["h"])
    else:
        dyc.btcalc(T["h"], btd["h_u"], btd["h_v"])
    dtbt_g = dyc.set_dtbt(gtot_est=9.8 + 0.02 * d.nk, SSH_add=10.0)
    dyc.bt_mass_source(T["h"], dyc.to_dev(eta_ms), True)
    g = dict(alu=dyc.zeros3(), alv=dyc.zeros3(), eta_out=dyc.zeros2(), uhbtav=dyc.zeros2(), vhbtav=dyc.zeros2(),
             etaav=dyc.zeros2() if use_etaav else None)
    kwg = {}
    if use_uh0:
        kwg.update(uh0=T["uh"], vh0=T["vh"], u_uh0=T["u"], v_vh0=T["v"])
    if bottom:
        kwg.update(taux_bot=dyc.to_dev(0.3 * I["taux"]), tauy_bot=dyc.to_dev(0.3 * I["tauy"]))
    torch.cuda.synchronize()
    dyc.btstep(T["u"], T["v"], T["eta"], I["dt"], T["bcu"], T["bcv"], T["taux"], T["tauy"], T["pbce"], T["eta_PF"],
               T["ucor"], T["vcor"], g["alu"], g["alv"], g["eta_out"], g["uhbtav"], g["vhbtav"], T["vr_u"], T["vr_v"],
               None if no_bt_cont else btd, etaav=g["etaav"], **kwg)
    dyc.sync()
    res = dict(dtbt=(dtbt_g, dtbt))
    res["frhatu"] = (dyc.barotropic_field("frhatu").cpu().numpy(), cs["frhatu"], "u")
    res["frhatv"] = (dyc.barotropic_field("frhatv").cpu().numpy(), cs["frhatv"], "v")
    res["IDatu"] = (dyc.barotropic_field("IDatu").cpu().numpy(), cs["IDatu"], "u")
    res["q_D"] = (dyc.barotropic_field("q_D").cpu().numpy(), cs["q_D"], "q")
    res["eta_cor"] = (dyc.barotropic_field("eta_cor").cpu().numpy(), cs["eta_cor"], "h")
    res["ubtav"] = (dyc.barotropic_field("ubtav").cpu().numpy(), cs["ubtav"], "u")
    res["vbtav"] = (dyc.barotropic_field("vbtav").cpu().numpy(), cs["vbtav"], "v")
    stag = dict(alu="u", alv="v", eta_out="h", uhbtav="u", vhbtav="v", etaav="h")
    for k in o:
        if o[k] is not None:
            res[k] = (g[k].cpu().numpy(), o[k], stag[k])
    dyc.close()
    return d, res


PRE = ("frhatu", "frhatv", "IDatu", "q_D", "eta_cor")


def check(d, res, exact, rtol=1e-12):
    assert res["dtbt"][0] == res["dtbt"][1], f"set_dtbt: {res['dtbt']}"
    for k, val in res.items():
        if k == "dtbt":
            continue
        a, b, st = val
        sl = H.interior(d, st)
        if exact or k in PRE:
            H.assert_bitwise(a, b, k, sl)
        else:
            H.assert_close(a, b, k, rtol, sl)


@pytest.mark.parametrize("first_direction", [0, 1])
@pytest.mark.parametrize("project", [0, 1])
def test_btstep_bitexact_strong_drag(orc, first_direction, project):
    I = make_inputs(orc, H.double_gyre(), first_direction)
    d, res = run_both(orc, I, dict(strong_drag=1, BT_project_velocity=project))
    check(d, res, exact=True)


@pytest.mark.parametrize("project", [0, 1])
def test_btstep_default_path(orc, project):
    I = make_inputs(orc, H.double_gyre())
    d, res = run_both(orc, I, dict(BT_project_velocity=project))
    check(d, res, exact=False)


def test_btstep_tc1_settings_channel(orc):
    # tc1/p0: BT_PROJECT_VELOCITY=T, BEBT=0.2, DTBT=-0.95 on a re-entrant channel (halo wrap inside the loop)
    I = make_inputs(orc, H.channel())
    d, res = run_both(orc, I, dict(strong_drag=1, BT_project_velocity=1, bebt=0.2, dtbt_fraction=0.95), bottom=True)
    check(d, res, exact=True)


def test_btstep_option_flags(orc):
    I = make_inputs(orc, H.benchmark_small())
    d, res = run_both(orc, I, dict(strong_drag=1, wt_uv_bug=0, Sadourny=0, visc_rem_u_uh0=1, clip_velocity=1,
                                   bound_BT_corr=1, vel_underflow=1e-15, G_extra=0.1, use_old_coriolis_bracket_bug=1),
                      use_etaav=False)
    check(d, res, exact=True)


def test_btstep_no_uh0_default_thickness(orc):
    I = make_inputs(orc, H.benchmark_small())
    d, res = run_both(orc, I, dict(strong_drag=1), use_uh0=False, default_thick=True)
    check(d, res, exact=True)


@pytest.mark.parametrize("cfg", ["double_gyre", "channel", "benchmark_small"])
@pytest.mark.parametrize("project", [0, 1])
@pytest.mark.parametrize("nonlinear", [0, 1, 2, -1])   # NONLINEAR_BT_CONTINUITY off / updated every sub-step / every second / only before the loop
def test_btstep_without_BT_cont(orc, cfg, project, nonlinear):
    """USE_BT_CONT_TYPE = False (BT_cont not associated; NONLINEAR_BT_CONTINUITY = False): the barotropic continuity equation linear in
    the velocities with the face areas of find_face_areas (MOM_barotropic.F90:5146-5237, :1131-1136, :1221, :2639, :3053), the default
    BT_THICK_SCHEME without a BT_cont_type (HYBRID: btcalc without h_u / h_v) -- bit for bit with BT_STRONG_DRAG, 1e-12 on the default
    drag path.  NONLINEAR_BT_CONTINUITY: the face areas from bathymetry + eta (:5171-5186), recomputed every
    NONLIN_BT_CONT_UPDATE_PERIOD sub-steps with a stencil of 2 (:767-768, :2539-2543)."""
    nl = {} if nonlinear == 0 else dict(nonlinear_continuity=1, nonlin_cont_update_period=max(nonlinear, 0))
    I = make_inputs(orc, dict(double_gyre=H.double_gyre, channel=H.channel, benchmark_small=H.benchmark_small)[cfg](), project)
    d, res = run_both(orc, I, dict(strong_drag=1, BT_project_velocity=project, **nl), default_thick=True, no_bt_cont=True)
    check(d, res, exact=True)
    assert np.abs(res["uhbtav"][1]).max() > 0
    d, res = run_both(orc, I, dict(BT_project_velocity=project, **nl), default_thick=True, no_bt_cont=True, use_uh0=False)
    check(d, res, exact=False)


@pytest.mark.parametrize("cfg", ["channel", "benchmark_small"])
@pytest.mark.parametrize("scheme", [abi.BT_THICK_HYBRID, abi.BT_THICK_HARMONIC, abi.BT_THICK_ARITHMETIC])
@pytest.mark.parametrize("bt_cont", [0, 1])
def test_btstep_thick_schemes_and_eta_cor_bound(orc, cfg, scheme, bt_cont):
    """BT_THICK_SCHEME = HYBRID / HARMONIC / ARITHMETIC for btcalc without h_u, h_v (MOM_barotropic.F90:4448-4483) and
    BOUND_BT_CORRECTION through eta_cor_bound (:6164-6173, :1582-1585) -- without a BT_cont_type, and behind one with
    BT_CONT_CORR_BOUNDS = False.  MAXVEL is small enough that the bound acts on most columns.  Bit for bit."""
    I = make_inputs(orc, dict(channel=H.channel, benchmark_small=H.benchmark_small)[cfg]())
    mod = dict(strong_drag=1, bt_thick_scheme=scheme, bound_BT_corr=1, BT_cont_bounds=0, maxvel=2.0e-5)
    eta_ms = I["eta"] + 0.05 * synth.smooth_field(I["d"], 41) * (I["M"][G["mask2dT"]] > 0)   # eta_cor of a few centimetres, both signs
    d, res = run_both(orc, I, mod, default_thick=True, no_bt_cont=not bt_cont, eta_ms=eta_ms)
    check(d, res, exact=True)
    # the bound did something: the same run without it differs
    mod0 = dict(mod, bound_BT_corr=0)
    d, res0 = run_both(orc, I, mod0, default_thick=True, no_bt_cont=not bt_cont, eta_ms=eta_ms)
    assert not np.array_equal(res["eta_out"][1], res0["eta_out"][1])


def test_set_dtbt_face_areas_on_a_sloping_bottom(orc):
    """set_dtbt(pbce, eta=eta) without BT_cont and without NONLINEAR_BT_CONTINUITY takes find_face_areas(add_max=0)
    (MOM_barotropic.F90:3576-3582, :5208-5219): dy_Cu Z_to_H max(max(D_i, D_i+1) + Z_ref, 0), the DEEPER neighbour, not the
    harmonic mean of :5221-5236 -- checked against the formula written out in numpy on a bowl-shaped bottom, and equal to the
    oracle's bits.  With NONLINEAR_BT_CONTINUITY and eta: the harmonic means of bathyT Z_to_H + eta (:5171-5186)."""
    import torch
    from mom6_amd.dycore import Dycore
    I = make_inputs(orc, H.benchmark_small())
    d, M, GV = I["d"], I["M"], I["GV"]
    bathy = M[G["bathyT"]]
    assert np.ptp(bathy[d.joff:d.joff + d.nj, d.ioff:d.ioff + d.ni]) > 100.0   # a sloping bottom
    for nonlin in (0, 1):
        P = abi.barotropic_params_default(20.0)
        P.nonlinear_continuity = nonlin
        cs = orc.BtState(d)
        orc.barotropic_init(d, M, GV, P, cs)
        orc.btcalc(d, M, GV, I["h"], I["bt"]["h_u"], I["bt"]["h_v"], cs)
        dt_o = orc.set_dtbt_pbce_eta(d, M, GV, P, cs, I["pbce"], I["eta"])
        dyc = Dycore(d, M, GV, I["first_direction"])
        P2 = abi.barotropic_params_default(20.0)
        P2.nonlinear_continuity = nonlin
        dyc.barotropic_init(P2)
        dyc.btcalc(dyc.to_dev(I["h"]), dyc.to_dev(I["bt"]["h_u"]), dyc.to_dev(I["bt"]["h_v"]))
        dt_g = dyc.set_dtbt_pbce(dyc.to_dev(I["pbce"]), eta=dyc.to_dev(I["eta"]))
        assert dt_g == dt_o, (nonlin, dt_g, dt_o)
        # the formula, independently
        sl = (slice(d.joff, d.joff + d.nj), slice(d.ioff, d.ioff + d.ni))
        def sh(a, di, dj):
            return a[d.joff + dj:d.joff + dj + d.nj, d.ioff + di:d.ioff + di + d.ni]
        g = lambda n: M[G[n]]
        Z = GV.Z_to_H
        if nonlin:
            Hc = bathy * Z + I["eta"]
            hm = lambda a, b: np.where((a > 0) & (b > 0), 2.0 * a * b / np.where(a + b != 0, a + b, 1.0), 0.0)
            DuE, DuW = sh(g("dy_Cu"), 0, 0) * hm(sh(Hc, 0, 0), sh(Hc, 1, 0)), sh(g("dy_Cu"), -1, 0) * hm(sh(Hc, -1, 0), sh(Hc, 0, 0))
            DvN, DvS = sh(g("dx_Cv"), 0, 0) * hm(sh(Hc, 0, 0), sh(Hc, 0, 1)), sh(g("dx_Cv"), 0, -1) * hm(sh(Hc, 0, -1), sh(Hc, 0, 0))
        else:
            mx = lambda a, b: np.maximum(np.maximum(a, b) + P.Z_ref, 0.0)
            DuE, DuW = sh(g("dy_Cu"), 0, 0) * Z * mx(sh(bathy, 1, 0), sh(bathy, 0, 0)), sh(g("dy_Cu"), -1, 0) * Z * mx(sh(bathy, 0, 0), sh(bathy, -1, 0))
            DvN, DvS = sh(g("dx_Cv"), 0, 0) * Z * mx(sh(bathy, 0, 1), sh(bathy, 0, 0)), sh(g("dx_Cv"), 0, -1) * Z * mx(sh(bathy, 0, 0), sh(bathy, 0, -1))
        fr_u, fr_v = cs["frhatu"], cs["frhatv"]
        k3 = lambda a, di, dj: a[:, d.joff + dj:d.joff + dj + d.nj, d.ioff + di:d.ioff + di + d.ni]
        pb = k3(I["pbce"], 0, 0)
        gE, gW = (pb * k3(fr_u, 0, 0)).sum(0), (pb * k3(fr_u, -1, 0)).sum(0)
        gN, gS = (pb * k3(fr_v, 0, 0)).sum(0), (pb * k3(fr_v, 0, -1)).sum(0)
        f2 = g("Coriolis2Bu")
        Idt2 = 0.5 * (1.0 + 2.0 * P.bebt) * (sh(g("IareaT"), 0, 0) * (gE * DuE * sh(g("IdxCu"), 0, 0) + gW * DuW * sh(g("IdxCu"), -1, 0) +
                                              gN * DvN * sh(g("IdyCv"), 0, 0) + gS * DvS * sh(g("IdyCv"), 0, -1)) +
                                             (sh(f2, 0, 0) + sh(f2, -1, -1) + sh(f2, -1, 0) + sh(f2, 0, -1)))
        want = P.dtbt_fraction * np.sqrt(1.0 / Idt2.max())
        assert abs(dt_g - want) <= 1e-12 * want, (nonlin, dt_g, want)
