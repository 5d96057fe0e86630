"""CPU tests of the oracle (test infrastructure).  PARITY UNPINNED: the reference holds no golden vectors for
this path and cannot be built here, so the oracle is checked through the reference's own differential
invariants (.testing/README.rst: dim.* rescaling must be bit-identical; conservation; known answers)."""
import numpy as np
import pytest

from mom6_amd import abi, grid, synth
from tests import helpers as H

G = abi.G


def _cont(orc, d, M, GV, CS, u, v, h, dt, **kw):
    hn = np.zeros_like(h); uh = np.zeros_like(h); vh = np.zeros_like(h)
    orc.continuity_PPM(d, M, GV, CS, 0, u, v, h, hn, uh, vh, dt, **kw)
    return hn, uh, vh


def test_continuity_conserves_volume_exactly_and_matches_uhbt(orc):
    gg, d, M = H.double_gyre()
    GV = abi.vgrid_default(); CS = abi.continuity_params_default(d.nk)
    h, u, v = synth.make_state(d, M, thin_frac=0.1)
    hn, uh, vh = _cont(orc, d, M, GV, CS, u, v, h, 1200.0)
    sl = H.interior(d, "h"); A = M[G["areaT"]][sl]
    v0 = (h[(Ellipsis,) + sl] * A).sum(1).sum(1); v1 = (hn[(Ellipsis,) + sl] * A).sum(1).sum(1)
    assert np.all(np.abs(v1 / v0 - 1) < 1e-14)            # each layer's volume is conserved in a closed basin
    assert (hn[(Ellipsis,) + sl] >= GV.Angstrom_H).all()   # positive-definite
    uhbt = np.ascontiguousarray(uh.sum(0) * 1.03); vhbt = np.ascontiguousarray(vh.sum(0) * 0.97)
    vr = np.ones_like(h)
    uc = np.zeros_like(h); vc = np.zeros_like(h)
    hn2, uh2, vh2 = _cont(orc, d, M, GV, CS, u, v, h, 1200.0, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vr, visc_rem_v=vr.copy(), u_cor=uc, v_cor=vc)
    su = H.interior(d, "u")
    err = np.abs((uh2.sum(0) - uhbt)[su]) * 1200.0 * M[G["IareaT"]][su]
    assert err.max() <= CS.tol_eta * 1.0001 + 1e-12        # the Newton solve meets ETA_TOLERANCE


@pytest.mark.parametrize("flags", [dict(vol_CFL=1), dict(aggress_adjust=1, vol_CFL=1), dict(aggress_adjust=1)])
def test_continuity_aggress_adjust_and_volume_based_cfl(orc, flags):
    """CONT_PPM_AGGRESS_ADJUST / CONT_PPM_VOLUME_BASED_CFL (MOM_continuity_PPM.F90:612, :651-716, :938-945): volume is still
    conserved layer by layer and the switches do change the answer -- the volume-based CFL number wherever dy_Cu / areaT is not
    1 / dxT (faces narrower than their cells), the aggressive limits where a limit on du binds (a strong adjustment)."""
    GV = abi.vgrid_default()
    gg, d, M = H.double_gyre()
    M = H.narrowed_faces(d, M)
    CS = abi.continuity_params_default(d.nk)
    CS.sum_order = abi.SUM_REFERENCE   # (what these switches run in whatever sum_order says: the base run is compared with them)
    h, u, v = synth.make_state(d, M, thin_frac=0.1)
    base = _cont(orc, d, M, GV, CS, u, v, h, 1200.0)
    for k, val in flags.items():
        setattr(CS, k, val)
    hn, uh, vh = _cont(orc, d, M, GV, CS, u, v, h, 1200.0)
    sl = H.interior(d, "h"); A = M[G["areaT"]][sl]
    v0 = (h[(Ellipsis,) + sl] * A).sum(1).sum(1); v1 = (hn[(Ellipsis,) + sl] * A).sum(1).sum(1)
    assert np.all(np.abs(v1 / v0 - 1) < 1e-14)
    assert np.array_equal(uh, base[1]) == (not flags.get("vol_CFL"))   # without uhbt / BT_cont aggress_adjust alone changes nothing
    # (as tests/test_continuity_gpu.py::test_continuity_newton_reaches_cfl_limits: fast flow, a strong adjustment)
    gg, d, M = H.benchmark_small()
    M = H.narrowed_faces(d, M)
    h, u, v = synth.make_state(d, M, thin_frac=0.1)
    u = np.ascontiguousarray(20.0 * u); v = np.ascontiguousarray(20.0 * v)
    CS = abi.continuity_params_default(d.nk)
    hn, uh, vh = _cont(orc, d, M, GV, CS, u, v, h, 1200.0)
    uhbt = np.ascontiguousarray(uh.sum(0) * (1.0 + 3.0 * synth.smooth_field(d, 13, ox=1.0, oy=0.5)))
    vhbt = np.ascontiguousarray(vh.sum(0) * (1.0 - 3.0 * synth.smooth_field(d, 14, ox=0.5, oy=1.0)))
    vr = np.ascontiguousarray(np.clip(0.5 + 0.6 * synth.smooth_field(d, 11, nk=d.nk, ox=1.0, oy=0.5), 0.0, 1.0))
    outs = []
    for on in (False, True):
        for k, val in flags.items():
            setattr(CS, k, val if on else 0)
        uc = np.zeros_like(h); vc = np.zeros_like(h)
        hn2, uh2, vh2 = _cont(orc, d, M, GV, CS, u, v, h, 1200.0, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vr, visc_rem_v=vr.copy(), u_cor=uc, v_cor=vc)
        outs.append(uh2)
        assert np.isfinite(uh2).all() and np.isfinite(uc).all()
    if flags.get("aggress_adjust"):
        assert not np.array_equal(outs[0], outs[1])        # the adjustment runs into the limits, and they are different limits


@pytest.mark.parametrize("case,nk", [("double_gyre", 2), ("benchmark_small", 8), ("benchmark_small", 75)])
def test_tree16_sum_order_agrees_with_the_reference_order(orc, case, nk):
    """mom6x_continuity_params.sum_order = MOM6X_SUM_TREE16 (the order of the wave-owned device kernel) against the
    reference's sequential k sums: every output of the corrector-shaped and of the BT_cont-shaped call agrees within
    1e-13 of the field's range (north_star: "to a stated floating-point tolerance"), the adjusted transports still sum
    to uhbt within ETA_TOLERANCE, and each layer's volume is still conserved to round-off."""
    gg, d, M = getattr(H, case)(nk=nk)
    GV = abi.vgrid_default()
    h, u, v = synth.make_state(d, M, thin_frac=0.1)
    vr_u = np.ascontiguousarray(np.clip(0.5 + 0.6 * synth.smooth_field(d, 11, nk=d.nk, ox=1.0, oy=0.5), 0.0, 1.0))
    vr_v = np.ascontiguousarray(np.clip(0.5 + 0.6 * synth.smooth_field(d, 12, nk=d.nk, ox=0.5, oy=1.0), 0.0, 1.0))
    dt = 1200.0
    res = {}
    for order in (abi.SUM_REFERENCE, abi.SUM_TREE16):
        CS = abi.continuity_params_default(d.nk, GV.Angstrom_H); CS.sum_order = order
        hn, uh, vh = _cont(orc, d, M, GV, CS, u, v, h, dt)
        uhbt = np.ascontiguousarray(uh.sum(0) * (1.0 + 0.05 * synth.smooth_field(d, 13, ox=1.0, oy=0.5)))
        vhbt = np.ascontiguousarray(vh.sum(0) * (1.0 - 0.05 * synth.smooth_field(d, 14, ox=0.5, oy=1.0)))
        if order == abi.SUM_REFERENCE:
            bt_in = (uhbt, vhbt)
        uhbt, vhbt = bt_in                                   # the same target transports for both orders
        bt = orc.new_bt_cont(d)
        o = dict(u_cor=np.zeros_like(h), v_cor=np.zeros_like(h), du_cor=np.zeros(d.shape2()), dv_cor=np.zeros(d.shape2()))
        o["h"], o["uh"], o["vh"] = _cont(orc, d, M, GV, CS, u, v, h, dt, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vr_u, visc_rem_v=vr_v,
                                        BT_cont=bt, **o)
        o.update({"BT_" + n: bt[n] for n in abi.BTCont._names})
        res[order] = o
        su, sv = H.interior(d, "u"), H.interior(d, "v")
        assert (np.abs((o["uh"].sum(0) - uhbt)[su]) * dt * M[G["IareaT"]][su]).max() <= CS.tol_eta * 1.0001 + 1e-12
        assert (np.abs((o["vh"].sum(0) - vhbt)[sv]) * dt * M[G["IareaT"]][sv]).max() <= CS.tol_eta * 1.0001 + 1e-12
    ndiff = 0
    for n, a in res[abi.SUM_REFERENCE].items():
        b = res[abi.SUM_TREE16][n]
        if n.startswith(("BT_uBT", "BT_vBT")):
            continue     # break points of the fit: see _bt_cont_transport below
        st = "u" if (n in ("uh", "u_cor", "du_cor") or "_u" in n or n.startswith("BT_uBT")) else ("h" if n == "h" else "v")
        sl = (Ellipsis,) + tuple(H.interior(d, st))
        scale = np.abs(a[sl]).max()
        assert np.abs(a[sl] - b[sl]).max() <= 1e-13 * scale, (n, np.abs(a[sl] - b[sl]).max() / scale)
        ndiff += np.count_nonzero(a[sl] != b[sl])
    if nk > 2:
        assert ndiff > 0     # the two orders are different computations (with two layers the tree IS the sequence)
    # uBT_WW .. vBT_NN are ratios of differences of nearly equal face areas (:1376-1404), i.e. ill-conditioned by
    # construction (and zeroed by a threshold test where the areas agree to 1e-12); what btstep uses is the transport
    # they parametrise, and that is what must agree: find_uhbt at velocities on both sides of both break points.
    for dirn, names in (("u", ("FA_u_EE", "FA_u_E0", "FA_u_W0", "FA_u_WW", "uBT_WW", "uBT_EE")),
                        ("v", ("FA_v_NN", "FA_v_N0", "FA_v_S0", "FA_v_SS", "vBT_SS", "vBT_NN"))):
        sl = H.interior(d, dirn)
        A = [res[abi.SUM_REFERENCE]["BT_" + n][sl] for n in names]
        B = [res[abi.SUM_TREE16]["BT_" + n][sl] for n in names]
        for t in (-3.0, -1.0, -0.3, -0.01, 0.01, 0.3, 1.0, 3.0):
            uu = t * np.maximum(np.abs(A[4]), np.abs(A[5])) + 1e-3 * t
            ta, tb = _bt_cont_transport(uu, *A), _bt_cont_transport(uu, *B)
            assert np.abs(ta - tb).max() <= 1e-12 * np.abs(ta).max(), (dirn, t, np.abs(ta - tb).max() / np.abs(ta).max())


def _bt_cont_transport(u, FA_EE, FA_E0, FA_W0, FA_WW, uBT_WW, uBT_EE):
    """find_uhbt (MOM_barotropic.F90:4610-4631) with the local fit of set_local_BT_cont_types (:4963-4972), vectorised."""
    C1_3 = 1.0 / 3.0
    uh_EE = uBT_EE * (C1_3 * (2.0 * FA_E0 + FA_EE)); uh_WW = uBT_WW * (C1_3 * (2.0 * FA_W0 + FA_WW))
    with np.errstate(divide="ignore", invalid="ignore"):
        crvW = np.where(np.abs(uBT_WW) > 0.0, (C1_3 * (FA_WW - FA_W0)) / uBT_WW**2, 0.0)
        crvE = np.where(np.abs(uBT_EE) > 0.0, (C1_3 * (FA_EE - FA_E0)) / uBT_EE**2, 0.0)
    return np.where(u == 0.0, 0.0,
           np.where(u < uBT_EE, (u - uBT_EE) * FA_EE + uh_EE,
           np.where(u < 0.0, u * (FA_E0 + crvE * u**2),
           np.where(u <= uBT_WW, u * (FA_W0 + crvW * u**2), (u - uBT_WW) * FA_WW + uh_WW))))


def test_continuity_known_answer_uniform_flow(orc):
    # uniform h and a uniform zonal velocity on a re-entrant channel: uh = dy_Cu*u*h exactly, h unchanged
    gg, d, M = H.channel()
    GV = abi.vgrid_default(); CS = abi.continuity_params_default(d.nk)
    h = np.full(d.shape3(), 100.0); u = 0.25 * np.ones(d.shape3()) * M[G["mask2dCu"]][None]; v = np.zeros(d.shape3())
    hn, uh, vh = _cont(orc, d, M, GV, CS, u, v, h, 600.0)
    su = H.interior(d, "u")
    np.testing.assert_array_equal(uh[(Ellipsis,) + su], (M[G["dy_Cu"]] * 0.25 * 100.0)[su][None].repeat(d.nk, 0))
    sl = d.sl(0, d.ni - 1, 2, d.nj - 3)                     # away from the walls
    np.testing.assert_allclose(hn[(Ellipsis,) + sl], 100.0, rtol=0, atol=1e-11)


@pytest.mark.parametrize("p", [11, -7])
def test_dim_l_rescaling_is_bit_identical(orc, p):
    """.testing dim.l: rescale horizontal lengths by 2**p; answers must be bit-identical after unscaling."""
    gg, d, M = H.benchmark_small()
    GV = abi.vgrid_default(); CS = abi.continuity_params_default(d.nk)
    h, u, v = synth.make_state(d, M)
    s = 2.0 ** p
    M2 = M.copy()
    for n in abi.METRICS:
        if n.startswith(("dx", "dy")): M2[G[n]] = M[G[n]] * s
        elif n.startswith(("Idx", "Idy")): M2[G[n]] = M[G[n]] / s
        elif n.startswith("area"): M2[G[n]] = M[G[n]] * s * s
        elif n.startswith("Iarea"): M2[G[n]] = M[G[n]] / (s * s)
    CS2 = abi.continuity_params_default(d.nk); CS2.tol_vel = CS.tol_vel * s
    hn, uh, vh = _cont(orc, d, M, GV, CS, u, v, h, 900.0)
    hn2, uh2, vh2 = _cont(orc, d, M2, GV, CS2, u * s, v * s, h, 900.0)
    np.testing.assert_array_equal(hn2, hn)
    np.testing.assert_array_equal(uh2 / (s * s), uh)
    np.testing.assert_array_equal(vh2 / (s * s), vh)
    # CorAdCalc: accelerations scale as L T-2
    cor = abi.coriolis_params_default()
    CAu = np.zeros_like(h); CAv = np.zeros_like(h); CAu2 = np.zeros_like(h); CAv2 = np.zeros_like(h)
    orc.CorAdCalc(d, M, GV, cor, u, v, h, uh, vh, CAu, CAv)
    GV2 = abi.vgrid_default(); GV2.H_subroundoff = GV.H_subroundoff   # vol_neglect carries m_to_L**2
    orc.CorAdCalc(d, M2, GV2, cor, u * s, v * s, h, uh2, vh2, CAu2, CAv2)
    sl = H.interior(d, "u")
    a, b = CAu2[(Ellipsis,) + sl] / s, CAu[(Ellipsis,) + sl]
    assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max()   # vol_neglect (1e-4 m)**2 is not rescaled here: round-off level only


def test_resting_ocean_stays_at_rest(orc):
    """Known answer: level interfaces + no wind => PFu = PFv = 0 exactly and the state does not change."""
    ni, nj, nk = 24, 20, 4
    gg = grid.GlobalGrid(ni, nj, kind="spherical", lon0=0, lat0=20, dlon=1.0, dlat=1.0, depth_fn=grid.flat_depth(ni, nj, 400.0))
    d, M = gg.tile(nk)
    GV = abi.vgrid_default(); Rlay, gp = abi.layer_densities(nk)
    bt = abi.barotropic_params_default(30.0)
    m = orc.OrcModel(d, M, GV, abi.continuity_params_default(nk), bt, abi.coriolis_params_default(), abi.pgf_params_default(),
                     abi.rk2_params_default(), Rlay, gp)
    h = np.where(M[G["mask2dT"]][None] > 0, 100.0, 1e-10) * np.ones(d.shape3())
    u = np.zeros_like(h); v = np.zeros_like(h)
    a = np.zeros((nk + 1,) + d.shape2()); a[1:] = 1e-5
    coefs = (a * M[G["mask2dCu"]][None], a * M[G["mask2dCv"]][None], np.maximum(h, 1e-9), np.maximum(h, 1e-9), None, None)
    coefs = tuple(np.ascontiguousarray(x) if x is not None else None for x in coefs)
    z = lambda: np.zeros_like(h)
    uh, vh, uhtr, vhtr, eta_av = z(), z(), z(), z(), np.zeros(d.shape2())
    tau = np.zeros(d.shape2())
    h0 = h.copy()
    m.initialize(u, v, h, uh, vh, 600.0)
    for n in range(3):
        m.step(u, v, h, uh, vh, uhtr, vhtr, eta_av, tau, tau, 600.0, coefs, calc_dtbt=(n == 0))
    # (PFu is not masked in the reference either; the coastal faces next to Angstrom-thick land columns are closed)
    assert np.abs(m["PFu"] * M[G["mask2dCu"]][None]).max() == 0.0 and np.abs(m["PFv"] * M[G["mask2dCv"]][None]).max() == 0.0
    assert np.abs(u).max() == 0.0 and np.abs(v).max() == 0.0
    np.testing.assert_array_equal(h[(Ellipsis,) + H.interior(d, "h")], h0[(Ellipsis,) + H.interior(d, "h")])


def test_rk2_conserves_volume_and_btstep_is_consistent(orc):
    gg, d, M = H.benchmark_small()
    GV = abi.vgrid_default(); Rlay, gp = abi.layer_densities(d.nk)
    bt = abi.barotropic_params_default(30.0)
    m = orc.OrcModel(d, M, GV, abi.continuity_params_default(d.nk), bt, abi.coriolis_params_default(), abi.pgf_params_default(),
                     abi.rk2_params_default(), Rlay, gp)
    h, u, v = synth.make_state(d, M, u_max=0.05, h_pert=0.001)
    a = np.zeros((d.nk + 1,) + d.shape2()); a[1:] = 1e-5; a[d.nk] = 3e-4
    hu = np.maximum(h, 1e-9)
    coefs = tuple(np.ascontiguousarray(x) if x is not None else None for x in
                  (a * M[G["mask2dCu"]][None], a * M[G["mask2dCv"]][None], hu, hu.copy(), None, None))
    z = lambda: np.zeros_like(h)
    uh, vh, uhtr, vhtr, eta_av = z(), z(), z(), z(), np.zeros(d.shape2())
    taux = np.ascontiguousarray(0.1 * synth.smooth_field(d, 41, ox=1, oy=.5) * M[G["mask2dCu"]]); tauy = np.zeros(d.shape2())
    sl = H.interior(d, "h"); A = M[G["areaT"]][sl]
    vol0 = (h[(Ellipsis,) + sl] * A).sum()
    m.initialize(u, v, h, uh, vh, 900.0)
    for n in range(3):
        m.step(u, v, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy, 900.0, coefs, calc_dtbt=(n == 0))
        assert abs((h[(Ellipsis,) + sl] * A).sum() / vol0 - 1) < 1e-14
        # the layer transports sum to the barotropic solver's time-mean transport to within ETA_TOLERANCE
        su = H.interior(d, "u")
        err = np.abs((uh.sum(0) - m["uhbt"])[su]) * 900.0 * M[G["IareaT"]][su]
        assert err.max() < 1e-6
        # eta (barotropic) tracks the layer-thickness sum: eta_cor stays tiny
        eta_h = (h.sum(0) - M[G["bathyT"]])[sl]
        assert np.abs(eta_h - m["eta"][sl]).max() < 1e-3
    assert np.isfinite(u).all() and np.abs(u).max() < 1.0


def test_pressure_force_eos_consistency(orc):
    """The use_EOS branch of PressureForce_FV_Bouss against the layered branch: with a linear equation of state,
    dRho_dp = 0 and T, S chosen so that every layer has exactly its target density Rlay(k), the analytic density
    integrals reduce to the layered formulas, so PFu/PFv agree to round-off; and a level, horizontally uniform
    ocean feels no force under either equation of state."""
    gg, d, M = H.benchmark_small()
    GV = abi.vgrid_default(); CS = abi.pgf_params_default(GV.Rho0)
    Rlay, gp = abi.layer_densities(d.nk, GV.Rho0, GV.g_Earth)
    h, _, _ = synth.make_state(d, M, thin_frac=0.05)
    z = lambda: np.zeros_like(h)
    P0u, P0v = z(), z()
    orc.PressureForce(d, M, GV, CS, Rlay, gp, h, P0u, P0v)
    eos = abi.eos_params_default(abi.LINEAR)
    T = np.full_like(h, 10.0); S = z()
    for k in range(d.nk):
        S[k] = (Rlay[k] - eos.Rho_T0_S0 - eos.dRho_dT * 10.0) / eos.dRho_dS
    P1u, P1v = z(), z()
    orc.PressureForce(d, M, GV, CS, Rlay, gp, h, P1u, P1v, T=T, S=S, eos=eos)
    su, sv = H.interior(d, "u"), H.interior(d, "v")
    scale = np.abs(P0u[(Ellipsis,) + su]).max()
    assert scale > 0
    assert np.abs(P1u - P0u)[(Ellipsis,) + su].max() < 1e-9 * scale
    assert np.abs(P1v - P0v)[(Ellipsis,) + sv].max() < 1e-9 * scale
    # resting, level ocean (flat bottom): no pressure force with either EOS
    gg2, d2, M2 = H.channel()
    h2 = np.full(d2.shape3(), 1000.0 / d2.nk)
    T2 = np.zeros_like(h2); S2 = np.zeros_like(h2)
    for k in range(d2.nk):
        T2[k] = 18.0 - 4.0 * k; S2[k] = 34.0 + 0.3 * k
    Rl2, gp2 = abi.layer_densities(d2.nk, GV.Rho0, GV.g_Earth)
    for form in (abi.LINEAR, abi.WRIGHT):
        e2 = abi.eos_params_default(form)
        Pu, Pv = np.zeros_like(h2), np.zeros_like(h2)
        orc.PressureForce(d2, M2, GV, CS, Rl2, gp2, h2, Pu, Pv, T=T2, S=S2, eos=e2)
        su2, sv2 = H.interior(d2, "u"), H.interior(d2, "v")
        assert np.abs(Pu[(Ellipsis,) + su2] * M2[G["mask2dCu"]][su2]).max() < 1e-12
        assert np.abs(Pv[(Ellipsis,) + sv2] * M2[G["mask2dCv"]][sv2]).max() < 1e-12


def test_vertvisc_coef_known_answers(orc):
    """vertvisc_coef on uniform layers at rest: hvel is the (arithmetic = harmonic) thickness, the interior coupling
    coefficient is Kv / (h + dz_neglect) (find_coupling_coef :2536-2539) and the bottom one Kv / (h/2 + dz_neglect);
    with BOTTOMDRAGLAW the bottom coefficient is kv_bbl / (min(h/2, bbl_thick) + dz_neglect) (:2543-2547)."""
    gg, d, M = H.channel()
    GV = abi.vgrid_default()
    hk = 1000.0 / d.nk
    h = np.full(d.shape3(), hk); u = np.zeros_like(h); v = np.zeros_like(h)
    P = abi.vertvisc_params_default(Kv=1e-3)
    P.bottomdraglaw = 0
    out = dict(a_u=np.zeros((d.nk + 1,) + d.shape2()), a_v=np.zeros((d.nk + 1,) + d.shape2()), h_u=np.zeros_like(h), h_v=np.zeros_like(h))
    orc.vertvisc_coef(d, M, GV, P, u, v, h, 600.0, out["a_u"], out["a_v"], out["h_u"], out["h_v"])
    su = H.interior(d, "u"); wet = M[G["mask2dCu"]][su] > 0
    assert np.all(out["h_u"][(Ellipsis,) + su][:, wet] == hk + GV.H_subroundoff)
    assert np.all(out["a_u"][(0,) + su] == 0.0)
    for K in range(1, d.nk):
        np.testing.assert_allclose(out["a_u"][(K,) + su][wet], 1e-3 / (hk + GV.dZ_subroundoff), rtol=1e-15)
    np.testing.assert_allclose(out["a_u"][(d.nk,) + su][wet], 1e-3 / (0.5 * hk + GV.dZ_subroundoff), rtol=1e-15)
    P.bottomdraglaw = 1
    kvb = np.full(d.shape2(), 5e-3); bth = np.full(d.shape2(), 20.0)
    orc.vertvisc_coef(d, M, GV, P, u, v, h, 600.0, out["a_u"], out["a_v"], out["h_u"], out["h_v"], kvb, kvb, bth, bth)
    np.testing.assert_allclose(out["a_u"][(d.nk,) + su][wet], 5e-3 / (min(0.5 * hk, 20.0 + GV.dZ_subroundoff) + GV.dZ_subroundoff), rtol=1e-15)
    # far above the bottom boundary layer the drag-law correction vanishes: botfn = 1/(1+0.09 z^6) with z >> 1
    np.testing.assert_allclose(out["a_u"][(1,) + su][wet], 1e-3 / (hk + GV.dZ_subroundoff), rtol=1e-6)


def test_equation_of_state_against_reference_known_answers(orc):
    """PINNED: the reference's own unit test EOS_unit_tests holds check values for the in-situ density
    (MOM_EOS.F90:2077-2079: WRIGHT at T=25, S=35, p=1e7 Pa -> 1027.54303596346 kg m-3; :2129-2131: LINEAR with
    Rho_T0_S0=1000, dRho_dT=-0.2, dRho_dS=0.8, dRho_dp=5e-7 -> 1028.0), tolerance 1000*epsilon relative (:2469-2473).
    The same routine also holds the analytic T and S derivatives to centred differences (:2401-2460): repeated here."""
    tol = 1000.0 * np.finfo(float).eps
    w = abi.eos_params_default(abi.WRIGHT)
    rho = orc.eos_density(w, 25.0, 35.0, 1.0e7)
    assert abs(rho - 1027.54303596346) < tol * rho
    lin = abi.eos_params_default(abi.LINEAR)
    lin.Rho_T0_S0 = 1000.0; lin.dRho_dT = -0.2; lin.dRho_dS = 0.8; lin.dRho_dp = 5.0e-7
    rho = orc.eos_density(lin, 25.0, 35.0, 1.0e7)
    assert abs(rho - 1028.0) < tol * rho
    for e in (w, lin):
        dT, dS = 0.1, 0.5
        a, b = orc.eos_density_derivs(e, 25.0, 35.0, 1.0e7)
        fd_T = (orc.eos_density(e, 25.0 + dT, 35.0, 1e7) - orc.eos_density(e, 25.0 - dT, 35.0, 1e7)) / (2 * dT)
        fd_S = (orc.eos_density(e, 25.0, 35.0 + dS, 1e7) - orc.eos_density(e, 25.0, 35.0 - dS, 1e7)) / (2 * dS)
        assert abs(a - fd_T) < 1e-4 * abs(a) and abs(b - fd_S) < 1e-4 * abs(b)


def _hv(orc, d, M, GV, P, u, v, h):
    planes = orc.hor_visc_init(d, M, P)
    du, dv = np.zeros_like(u), np.zeros_like(v)
    orc.horizontal_viscosity(d, M, GV, P, planes, u, v, h, du, dv)
    return du, dv, planes


def test_horizontal_viscosity_known_answers(orc):
    """hor_visc on a Cartesian channel with flat layers: (a) uniform flow feels no stress; (b) a zonal jet
    u = U sin(l y) is damped as -Kh l^2 s u (Laplacian) and -Ah l^4 s^2 u (biharmonic), with the second-difference
    factor s = (sin(l dy/2)/(l dy/2))^2 -- the discrete operator's exact answer (tolerance 1e-10 relative)."""
    gg, d, M = H.channel(nk=2, nj=36)
    GV = abi.vgrid_default()
    h = np.zeros((d.nk,) + d.shape2()); h[:] = 400.0 * M[abi.G["mask2dT"]] + GV.Angstrom_H
    h[:, M[abi.G["mask2dT"]] == 0] = GV.Angstrom_H
    dy = 2.0e4
    l = 2.0 * np.pi / (12.0 * dy)
    s = (np.sin(0.5 * l * dy) / (0.5 * l * dy)) ** 2
    jj = (np.arange(d.shape2()[0]) - d.joff + 0.5) * dy
    u = np.zeros_like(h); v = np.zeros_like(h)
    rows = d.sl(-1, d.ni - 1, 8, d.nj - 9)     # far enough from the walls for a 5-row stencil
    off = dict(bound_Kh=0, bound_Ah=0, better_bound_Kh=0, better_bound_Ah=0, use_land_mask=0)
    # (a)
    u[:] = 0.3 * M[abi.G["mask2dCu"]]
    P = abi.hor_visc_params_default(1200.0, Laplacian=True, biharmonic=True)
    P.Kh = 1.0e3; P.Ah = 1.0e11
    for k_, v_ in off.items():
        setattr(P, k_, v_)
    du, dv, _ = _hv(orc, d, M, GV, P, u, v, h)
    assert np.abs(du[(Ellipsis,) + tuple(rows)]).max() == 0.0 and np.abs(dv[(Ellipsis,) + tuple(rows)]).max() == 0.0
    # (b)
    u[:] = (0.3 * np.sin(l * jj))[None, :, None] * M[abi.G["mask2dCu"]]
    for lap, bih, expect in ((True, False, -1.0e3 * l ** 2 * s), (False, True, -1.0e11 * l ** 4 * s ** 2)):
        P = abi.hor_visc_params_default(1200.0, Laplacian=lap, biharmonic=bih)
        P.Kh = 1.0e3; P.Ah = 1.0e11
        for k_, v_ in off.items():
            setattr(P, k_, v_)
        du, dv, _ = _hv(orc, d, M, GV, P, u, v, h)
        a, b = du[(Ellipsis,) + tuple(rows)], expect * u[(Ellipsis,) + tuple(rows)]
        assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), (lap, bih, np.abs(a - b).max(), np.abs(b).max())
        assert np.abs(dv[(Ellipsis,) + tuple(rows)]).max() <= 1e-25


def test_leith_viscosity_known_answers(orc):
    """LEITH_KH / LEITH_AH (MOM_hor_visc.F90:987-1113, :1161-1167, :1317-1323, :1610-1620, :1761-1765) on the Cartesian channel
    with a zonal jet u = U sin(l y).  The discrete vorticity is -U l s1 cos(l y_J) (s1 = sin(l dy/2)/(l dy/2)), its
    gradient U l^2 s1^2 sin(l y_j), the Laplacian of the vorticity U l^3 s1^3 cos(l y_J); at the corner points
    Kh = C3 dx^3/pi^3 |grad vort| (averaged from the two faces: a factor cos(l dy/2)) and Ah = C6 dx^6/pi^6 |Del2 vort|,
    so the acceleration is d/dy (Kh du/dy) and -d/dy (Ah d/dy Del2 u) in their discrete forms -- evaluated here in closed
    form, row by row (1e-10 relative).  A uniform flow has no vorticity gradient: no stress.  Beta: with
    USE_BETA_IN_LEITH on an f-plane nothing changes; on a beta-plane grad f joins the gradient."""
    gg, d, M = H.channel(nk=2, nj=36, beta=0.0)
    GV = abi.vgrid_default()
    h = np.zeros((d.nk,) + d.shape2()); h[:] = 400.0 * M[abi.G["mask2dT"]] + GV.Angstrom_H
    h[:, M[abi.G["mask2dT"]] == 0] = GV.Angstrom_H
    dy = dx = 2.0e4
    assert M[abi.G["dyT"]][d.joff + 5, d.ioff + 5] == dy and M[abi.G["dxT"]][d.joff + 5, d.ioff + 5] == dx
    U = 0.3
    l = 2.0 * np.pi / (12.0 * dy)
    s1 = np.sin(0.5 * l * dy) / (0.5 * l * dy)
    cavg = np.cos(0.5 * l * dy)
    jrow = np.arange(d.shape2()[0]) - d.joff
    yc = (jrow + 0.5) * dy            # u rows
    yq = (jrow + 1.0) * dy            # the corner row J above u row j
    u = np.zeros_like(h); v = np.zeros_like(h)
    u[:] = (U * np.sin(l * yc))[None, :, None] * M[abi.G["mask2dCu"]]
    rows = d.sl(-1, d.ni - 1, 9, d.nj - 10)
    off = dict(bound_Kh=0, bound_Ah=0, better_bound_Kh=0, better_bound_Ah=0, use_land_mask=0)

    def params(**kw):
        P = abi.hor_visc_params_default(1200.0, Laplacian=kw.pop("Laplacian", False), biharmonic=kw.pop("biharmonic", False))
        for k_, v_ in {**off, **kw}.items():
            setattr(P, k_, v_)
        return P

    C3, C6 = 1.5, 0.8
    sh = U * l * s1 * np.cos(l * yq)                                  # sh_xy = du/dy at the corners
    Kh = C3 * dx ** 3 / np.pi ** 3 * (U * l ** 2 * s1 ** 2 * cavg * np.abs(np.sin(l * yq)))
    expect_K = (Kh * sh - np.roll(Kh * sh, 1)) / dy                  # row j: (F(J) - F(J-1)) / dy
    du, dv, _ = _hv(orc, d, M, GV, params(Laplacian=True, Leith_Kh=1, Leith_Lap_const=C3), u, v, h)
    a = du[(Ellipsis,) + tuple(rows)]
    b = np.broadcast_to(expect_K[None, :, None], du.shape)[(Ellipsis,) + tuple(rows)]
    assert np.abs(b).max() > 0 and np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())
    assert np.abs(dv[(Ellipsis,) + tuple(rows)]).max() <= 1e-25
    # biharmonic Leith
    Ah = C6 * dx ** 6 / np.pi ** 6 * np.abs(U * l ** 3 * s1 ** 3 * np.cos(l * yq))
    B = -U * l ** 3 * s1 ** 3 * np.cos(l * yq)                        # d/dy Del2 u at the corners
    expect_A = -(Ah * B - np.roll(Ah * B, 1)) / dy
    du, dv, _ = _hv(orc, d, M, GV, params(biharmonic=True, Leith_Ah=1, Leith_bi_const=C6), u, v, h)
    a = du[(Ellipsis,) + tuple(rows)]
    b = np.broadcast_to(expect_A[None, :, None], du.shape)[(Ellipsis,) + tuple(rows)]
    assert np.abs(b).max() > 0 and np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())
    # ADD_LES_VISCOSITY on a background: Kh_total = Kh_bg + Kh_Leith -> the two accelerations add
    Kb = 700.0
    s = s1 ** 2
    du2, _, _ = _hv(orc, d, M, GV, params(Laplacian=True, Leith_Kh=1, Leith_Lap_const=C3, Kh=Kb, add_LES_viscosity=1), u, v, h)
    b2 = (np.broadcast_to(expect_K[None, :, None], du.shape) - Kb * l ** 2 * s * u)[(Ellipsis,) + tuple(rows)]
    assert np.abs(du2[(Ellipsis,) + tuple(rows)] - b2).max() <= 1e-10 * np.abs(b2).max()
    # uniform flow: nothing
    u0 = np.zeros_like(u); u0[:] = U * M[abi.G["mask2dCu"]]
    du0, dv0, _ = _hv(orc, d, M, GV, params(Laplacian=True, biharmonic=True, Leith_Kh=1, Leith_Lap_const=C3, Leith_Ah=1, Leith_bi_const=C6,
                                             modified_Leith=1), u0, v, h)
    assert np.abs(du0[(Ellipsis,) + tuple(rows)]).max() == 0.0 and np.abs(dv0[(Ellipsis,) + tuple(rows)]).max() == 0.0
    # beta: f-plane -> identical; beta-plane -> |grad vort + beta| replaces |grad vort|
    P0 = params(Laplacian=True, Leith_Kh=1, Leith_Lap_const=C3)
    P1 = params(Laplacian=True, Leith_Kh=1, Leith_Lap_const=C3, use_beta_in_Leith=1)
    assert np.array_equal(_hv(orc, d, M, GV, P0, u, v, h)[0], _hv(orc, d, M, GV, P1, u, v, h)[0])
    beta = 2.0e-11
    ggb, db, Mb = H.channel(nk=2, nj=36, beta=beta)
    assert abs(np.diff(Mb[abi.G["CoriolisBu"]][d.joff + 10:d.joff + 12, d.ioff + 5])[0] / dy - beta) < 1e-6 * beta
    Khb = C3 * dx ** 3 / np.pi ** 3 * np.abs(U * l ** 2 * s1 ** 2 * cavg * np.sin(l * yq) + beta)
    expect_b = (Khb * sh - np.roll(Khb * sh, 1)) / dy
    dub, _, _ = _hv(orc, db, Mb, GV, P1, u, v, h)
    a = dub[(Ellipsis,) + tuple(rows)]
    b = np.broadcast_to(expect_b[None, :, None], dub.shape)[(Ellipsis,) + tuple(rows)]
    assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())


def test_horizontal_viscosity_better_bounds_are_stable(orc):
    """BETTER_BOUND_KH / BETTER_BOUND_AH (:3025-3114) exist so that a forward step with any requested viscosity
    cannot amplify grid-scale noise: with absurdly large KH and AH, one step of u + dt*diffu of a checkerboard must
    not grow, and the maximum-viscosity planes must be positive at wet points and zero on land."""
    gg, d, M = H.island_basin()
    M = H.partial_faces(d, M)
    GV = abi.vgrid_default()
    dt = 1200.0
    rng = np.random.default_rng(5)
    h, u, v = synth.make_state(d, M, thin_frac=0.2)
    sgn = (-1.0) ** (np.add.outer(np.arange(d.shape2()[0]), np.arange(d.shape2()[1])))
    u = 0.2 * sgn[None] * M[abi.G["mask2dCu"]] * (1.0 + 0.2 * rng.random(u.shape))
    v = -0.2 * sgn[None] * M[abi.G["mask2dCv"]] * (1.0 + 0.2 * rng.random(v.shape))
    for lap, bih in ((True, False), (False, True), (True, True)):
        P = abi.hor_visc_params_default(dt, Laplacian=lap, biharmonic=bih)
        P.Kh = 1.0e12; P.Ah = 1.0e24
        du, dv, planes = _hv(orc, d, M, GV, P, u, v, h)
        su, sv = H.interior(d, "u"), H.interior(d, "v")
        un, vn = (u + dt * du)[(Ellipsis,) + tuple(su)], (v + dt * dv)[(Ellipsis,) + tuple(sv)]
        ke0 = (u[(Ellipsis,) + tuple(su)] ** 2).sum() + (v[(Ellipsis,) + tuple(sv)] ** 2).sum()
        ke1 = (un ** 2).sum() + (vn ** 2).sum()
        assert np.isfinite(du).all() and ke1 < ke0, (lap, bih, ke1 / ke0)
        assert np.abs(un).max() <= 1.05 * np.abs(u).max() and np.abs(vn).max() <= 1.05 * np.abs(v).max()
        names = ["dx2h", "dy2h", "dx2q", "dy2q", "DX_dyT", "DY_dxT", "DX_dyBu", "DY_dxBu", "red_xx", "red_xy", "Kh_bg_xx", "Kh_bg_xy",
                 "Kh_Max_xx", "Kh_Max_xy", "Lap2_xx", "Lap2_xy", "Idx2dyCu", "Idxdy2u", "Idx2dyCv", "Idxdy2v", "Ah_bg_xx", "Ah_bg_xy",
                 "Ah_Max_xx", "Ah_Max_xy"]
        pl = dict(zip(names, planes))
        hs = H.interior(d, "h")
        wet = M[abi.G["mask2dT"]][tuple(hs)] > 0
        for nm, on in (("Kh_Max_xx", lap), ("Ah_Max_xx", bih)):
            if on:
                assert (pl[nm][tuple(hs)][wet] > 0).all()
        assert (pl["red_xx"][tuple(hs)] <= 1.0).all() and (pl["red_xx"][tuple(hs)] < 1.0).any()


def test_coriolis_en_dis_reduces_to_sadourny_for_centred_transports(orc):
    """CORIOLIS_EN_DIS (.testing/tc4) brackets each face transport between the continuity solver's value and the centred
    estimate 0.5*dy_Cu*u*(h_i + h_i+1).  When the two coincide the brackets collapse and the energy-dissipating scheme
    must return the energy-conserving SADOURNY75_ENERGY accelerations EXACTLY; when they differ it must not."""
    gg, d, M = H.double_gyre(nk=3)
    GV = abi.vgrid_default()
    h, u, v = synth.make_state(d, M, thin_frac=0.05)
    G = abi.G
    uh = np.ascontiguousarray(0.5 * ((M[G["dy_Cu"]][None] * 1.0) * u) * (h + np.roll(h, -1, axis=2)))
    vh = np.ascontiguousarray(0.5 * ((M[G["dx_Cv"]][None] * 1.0) * v) * (h + np.roll(h, -1, axis=1)))
    out = {}
    for en in (0, 1):
        CS = abi.coriolis_params_default(); CS.Coriolis_En_Dis = en
        CAu, CAv = np.zeros_like(h), np.zeros_like(h)
        orc.CorAdCalc(d, M, GV, CS, u, v, h, uh, vh, CAu, CAv)
        out[en] = (CAu, CAv)
    su, sv = H.interior(d, "u"), H.interior(d, "v")
    assert np.array_equal(out[0][0][(Ellipsis,) + tuple(su)], out[1][0][(Ellipsis,) + tuple(su)])
    assert np.array_equal(out[0][1][(Ellipsis,) + tuple(sv)], out[1][1][(Ellipsis,) + tuple(sv)])
    CS = abi.coriolis_params_default(); CS.Coriolis_En_Dis = 1
    CAu, CAv = np.zeros_like(h), np.zeros_like(h)
    orc.CorAdCalc(d, M, GV, CS, u, v, h, np.ascontiguousarray(1.7 * uh), np.ascontiguousarray(0.4 * vh), CAu, CAv)
    CS0 = abi.coriolis_params_default()
    CAu0, CAv0 = np.zeros_like(h), np.zeros_like(h)
    orc.CorAdCalc(d, M, GV, CS0, u, v, h, np.ascontiguousarray(1.7 * uh), np.ascontiguousarray(0.4 * vh), CAu0, CAv0)
    assert np.abs(CAu - CAu0)[(Ellipsis,) + tuple(su)].max() > 0


def test_direct_stress_known_answer(orc):
    """DIRECT_STRESS (:707-720): with 10 m layers and HMIX_STRESS = 25 m the body force of the wind is 1, 1 and 0.5 times
    stress/HMIX in the first three layers and nothing below; with zero viscosity the solve leaves exactly that."""
    gg, d, M = H.channel(nk=6)
    GV = abi.vgrid_default()
    h = np.full((d.nk,) + d.shape2(), 10.0)
    u = np.zeros_like(h); v = np.zeros_like(h)
    a = np.zeros((d.nk + 1,) + d.shape2())
    hu = np.full_like(h, 10.0)
    taux = np.ascontiguousarray(0.2 * M[abi.G["mask2dCu"]]); tauy = np.zeros(d.shape2())
    dt = 600.0
    orc.vertvisc(d, M, GV, u, v, a, a.copy(), hu, hu.copy(), None, None, taux, tauy, dt, Hmix_stress=25.0, h=h)
    x = (d.joff + 5, d.ioff + 7)
    assert M[abi.G["mask2dCu"]][x] > 0
    stress = dt / GV.H_to_RZ * 0.2
    col = u[(slice(None),) + x]
    assert np.allclose(col[:3], np.array([1.0, 1.0, 0.5]) * stress / 25.0, rtol=1e-15, atol=0)
    assert (col[3:] == 0.0).all() and (v == 0.0).all()
    # the momentum put into the column is the stress, whatever the depth it is spread over
    assert abs((col * 10.0).sum() - stress) <= 4e-16 * stress
    u2 = np.zeros_like(h); v2 = np.zeros_like(h)
    orc.vertvisc(d, M, GV, u2, v2, a, a.copy(), hu, hu.copy(), None, None, taux, tauy, dt)     # stress boundary condition
    assert abs((u2[(slice(None),) + x] * 10.0).sum() - stress) <= 4e-16 * stress and u2[(1,) + x] == 0.0
