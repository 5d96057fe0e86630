"""GPU parity of the whole split RK2 baroclinic step (step_MOM_dyn_split_RK2) against the oracle.

BT_STRONG_DRAG=True removes the only transcendental on the path (av_rem**(1/nstep)), so several
consecutive steps must stay BIT-IDENTICAL in every prognostic and restart field; on the default
path the fields must agree to 1e-11 of their range after 3 steps."""
import numpy as np
import pytest

from mom6_amd import abi, synth
from tests import helpers as H
from tests.test_dyn_gpu import visc_coefs

pytestmark = pytest.mark.gpu
G = abi.G


@pytest.fixture(autouse=True)
def _both_sum_orders(sums):
    """Every whole-step case runs twice: with the device's default order of the mass-flux column sums (TREE16, held to the
    oracle's restatement of that tree) and with MOM6X_SUMS=exact, i.e. the REFERENCE order of MOM_continuity_PPM.F90
    :1093-1242, :1293-1316 on both sides -- the device bit for bit against the sequential-k oracle (tests/conftest.py)."""
    return sums


STATE = ["u", "v", "h", "uh", "vh", "uhtr", "vhtr", "eta_av"]
STAG = dict(u="u", v="v", h="h", uh="u", vh="v", uhtr="u", vhtr="v", eta_av="h", CAu="u", CAv="v", CAu_pred="u", CAv_pred="v",
            PFu="u", PFv="v", diffu="u", diffv="v", visc_rem_u="u", visc_rem_v="v", u_accel_bt="u", v_accel_bt="v", u_av="u", v_av="v", h_av="h",
            eta="h", eta_PF="h", uhbt="u", vhbt="v", taux_bot="u", tauy_bot="v", BT_h_u="u", BT_h_v="v", pbce="h")


def run(orc, cfg, nsteps=3, bt_mod=None, rk2_mod=None, cor_mod=None, first_direction=0, per_stage=False, new_diff=False,
        exact=True, rtol=1e-11, eos_form=None, dev_vv=None, hv=None, Hmix_stress=0.0, recon=0, chk=False, cont_mod=None, ray=True,
        shear=True, hook=None):
    import torch
    from mom6_amd.dycore import Dycore
    from tests import cases
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg, per_stage, new_diff)
    GV, Rlay, gp, dt = inp["GV"], inp["Rlay"], inp["gp"], inp["dt"]
    h, u, v, coefs, taux, tauy, diff_new = (inp[k] for k in ("h", "u", "v", "coefs", "taux", "tauy", "diff_new"))

    def params():
        return cases.rk2_params(d, GV, bt_mod, rk2_mod, cor_mod, cont_mod)

    tv = None
    if eos_form is not None:   # tv%T, tv%S, tv%eqn_of_state: the PressureForce calls take the use_EOS branch
        Tt, St = cases.thermo_state(d, M)
        tv = (Tt, St, abi.eos_params_default(eos_form))
        if recon == "quadrature": tv[2].EOS_quadrature = 1   # int_density_dz_generic_pcm
        else: tv[2].Recon_Scheme = recon   # the ALE path of PressureForce: 1 PLM, 2 PPM reconstruction of T, S
        if Hmix_stress > 0.0:
            tv[2].MassWghtInterp = 1   # the tc4-like case also has MASS_WEIGHT_IN_PRESSURE_GRADIENT
    # ---------------- oracle
    vvset = None
    if dev_vv is not None:   # vertvisc_coef inside the step (no coefficient sets from outside)
        from tests.test_dyn_gpu import visc_inputs
        P = abi.vertvisc_params_default()
        for k_, v_ in dev_vv.items():
            setattr(P, k_, v_)
        vvset = (P,) + tuple(visc_inputs(d, M, with_shear=shear)) + ((coefs[0][4], coefs[0][5]) if ray else (None, None))   # Ray_u, Ray_v
    so, m = cases.oracle_rk2(orc, cfg, inp, nsteps, bt_mod, rk2_mod, cor_mod, first_direction, tv=tv, vv=vvset, hv=hv, Hmix_stress=Hmix_stress, cont_mod=cont_mod)

    # ---------------- device
    cont2, bt2, cor2, pgf2, rk22 = params()
    dyc = Dycore(d, M, GV, first_direction)
    dyc.continuity_init(cont2); dyc.barotropic_init(bt2); dyc.CoriolisAdv_init(cor2); dyc.PressureForce_init(pgf2, Rlay, gp)
    dyc.initialize_dyn_split_RK2(rk22)
    if tv is not None:
        tvd = (dyc.to_dev(tv[0]), dyc.to_dev(tv[1]))
        dyc.PressureForce_set_tv(tvd[0], tvd[1], tv[2])
    sg = dict(u=dyc.to_dev(u), v=dyc.to_dev(v), h=dyc.to_dev(h), uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=dyc.zeros3(),
              vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
    cdev = [tuple(dyc.to_dev(a) if a is not None else None for a in c) for c in (coefs if per_stage else coefs[:1])]
    if vvset is None:
        dyc.vertvisc_set_coef(*cdev[0])
    else:
        dyc.vertvisc_init(vvset[0])
        vdev = [dyc.to_dev(a) if a is not None else None for a in vvset[1:]]
        dyc.vertvisc_set_visc(*vdev)
    if hv is not None:   # horizontal_viscosity inside the step and in the new-run initialisation
        dyc.hor_visc_init(hv)
    if Hmix_stress > 0.0:
        dyc.vertvisc_set_direct_stress(Hmix_stress, sg["h"])
    txd, tyd = dyc.to_dev(taux), dyc.to_dev(tauy)
    dnew = tuple(dyc.to_dev(a) for a in diff_new) if diff_new else None
    torch.cuda.synchronize()
    dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)
    stages = []

    def coef_hook(stage, pu, pv, ph, dtt):
        stages.append((stage, dtt))
        dyc.vertvisc_set_coef(*cdev[stage])
        return 0

    def hv_hook(pu, pv, ph, puh, pvh, pdu, pdv):
        dyc.rk2_field("diffu").copy_(dnew[0]); dyc.rk2_field("diffv").copy_(dnew[1])
        torch.cuda.synchronize()
        return 0

    if hook: hook(dyc, "before")   # (a test that wants to see which kernels the steps launch)
    for n in range(nsteps):
        dyc.step_MOM_dyn_split_RK2(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], sg["uhtr"], sg["vhtr"], sg["eta_av"], txd, tyd,
                                   dt, calc_dtbt=(n == 0), vertvisc_coef=coef_hook if per_stage else None,
                                   horizontal_viscosity=hv_hook if new_diff else None)
    dyc.sync()
    if hook: hook(dyc, "after")
    if per_stage:
        assert [s for s, _ in stages[:3]] == [0, 1, 2] and stages[1][1] == dt * rk22.be

    def cmp(name, a, b):
        sl = H.interior(d, STAG[name])
        if exact:
            H.assert_bitwise(a, b, name, sl)
        else:
            H.assert_close(a, b, name, rtol, sl)

    for n in STATE:
        cmp(n, sg[n].cpu().numpy(), so[n])
    if hv is not None:
        cmp("diffu", dyc.rk2_field("diffu").cpu().numpy(), m["diffu"]); cmp("diffv", dyc.rk2_field("diffv").cpu().numpy(), m["diffv"])
        assert np.abs(m["diffu"]).max() > 0
    for n in ("CAu", "CAv", "CAu_pred", "CAv_pred", "PFu", "PFv", "visc_rem_u", "visc_rem_v", "u_accel_bt", "v_accel_bt",
              "u_av", "v_av", "h_av", "eta", "eta_PF", "uhbt", "vhbt", "taux_bot", "tauy_bot"):
        cmp(n, dyc.rk2_field(n).cpu().numpy(), m[n])
    cmp("BT_h_u", dyc.rk2_field("BT_h_u").cpu().numpy(), m.bt_cont["h_u"])
    assert np.isfinite(so["u"]).all() and np.abs(so["u"]).max() > 0
    # volume conservation of the device result (closed basins): sum(h*area) unchanged to round-off
    sl = H.interior(d, "h")
    A = M[G["areaT"]][sl]
    vol0 = (h[(Ellipsis,) + sl] * A).sum(); vol1 = (sg["h"].cpu().numpy()[(Ellipsis,) + sl] * A).sum()
    assert abs(vol1 / vol0 - 1.0) < 1e-13
    out = {n: sg[n].cpu().numpy() for n in STATE}
    if chk:   # the artefacts .testing compares: chksum lines (mean / min / max + bit counts) and the restart checksum attribute
        for n in STATE:
            a = sg[n]; st = STAG[n]
            got = dyc.chksum(a, st, haloshift=0); got.pop("kind")
            assert got == orc.chksum(d, so[n], st, haloshift=0), ("chksum", n, got)
            r = dict(h=(0, d.ni - 1, 0, d.nj - 1), u=(-1, d.ni - 1, 0, d.nj - 1), v=(0, d.ni - 1, -1, d.nj - 1))[st]
            assert dyc.field_chksum(a, *r) == orc.field_chksum(d, so[n], *r), ("restart checksum", n)
    dyc.close()
    return out


@pytest.mark.parametrize("first_direction", [0, 1])
def test_rk2_double_gyre_bitexact(orc, first_direction):
    run(orc, H.double_gyre(), nsteps=3, bt_mod=dict(strong_drag=1), first_direction=first_direction)


@pytest.mark.parametrize("cfg", ["double_gyre", "channel", "island_basin"])
def test_rk2_checksum_artefacts_equal_the_oracles(orc, cfg):
    """What the reference's regression tests actually diff are checksums: the chksum lines (reproducing mean, min, max and
    the BIT COUNT of the field, MOM_checksums.F90) and the restart files' `checksum` attribute (the integer sum of the bit
    patterns).  Both see the sign of a zero.  After three steps they equal the oracle's in either order of the sums --
    in the REFERENCE order with no allowance at all (assert_bitwise compares bit patterns there), in the TREE16 order
    (continuity_wave.hip is built with -fno-signed-zeros) because no transport of these cases is a zero of either sign
    off the masked faces, which this test is there to notice if it changes."""
    run(orc, getattr(H, cfg)(), nsteps=3, bt_mod=dict(strong_drag=1), chk=True)


def test_rk2_channel_bitexact_tc1_like(orc):
    # tc1-like switches: BT_PROJECT_VELOCITY, BEBT=0.2, BOUND_CORIOLIS; re-entrant channel exercises every halo pass
    run(orc, H.channel(), nsteps=3, bt_mod=dict(strong_drag=1, BT_project_velocity=1, bebt=0.2, dtbt_fraction=0.95),
        cor_mod=dict(bound_Coriolis=1))


@pytest.mark.parametrize("cfg", ["double_gyre", "channel", "benchmark_small"])
@pytest.mark.parametrize("nonlinear,period,thick", [(0, 1, 1), (1, 1, 1), (1, 0, 2), (1, 3, 3)])
def test_rk2_without_a_BT_cont_type(orc, cfg, nonlinear, period, thick):
    """USE_BT_CONT_TYPE = False (mom6x_rk2_params.no_BT_cont; NONLINEAR_BT_CONTINUITY only acts without one; .testing/tc1 sets it but keeps USE_BT_CONT_TYPE, where the reference ignores it): CS%BT_cont
    is not associated, so btcalc works from h before the barotropic mass source (RK2.F90:627, BT_THICK_SCHEME = HYBRID / HARMONIC /
    ARITHMETIC; the last case also with BOUND_BT_CORRECTION through eta_cor_bound), the
    first continuity call only makes the layer fluxes btstep adds (:644-648, BT_USE_LAYER_FLUXES), set_dtbt takes eta (:667,
    used by NONLINEAR_BT_CONTINUITY), both btstep calls find their face areas from the bathymetry (+ eta, recomputed every
    NONLIN_BT_CONT_UPDATE_PERIOD sub-steps) and the corrector keeps the predictor's thickness fractions (:867).  Three steps,
    every prognostic and restart field bit for bit."""
    run(orc, getattr(H, cfg)(), nsteps=3, rk2_mod=dict(no_BT_cont=1),
        bt_mod=dict(strong_drag=1, BT_project_velocity=1, bebt=0.2, nonlinear_continuity=nonlinear, nonlin_cont_update_period=period,
                    bt_thick_scheme=thick, bound_BT_corr=int(thick == 3), maxvel=3.0e8 if thick != 3 else 1.0e-4))


@pytest.mark.parametrize("cont_mod", [dict(vol_CFL=1), dict(aggress_adjust=1, vol_CFL=1)])
def test_rk2_with_aggress_adjust_and_volume_based_cfl(orc, cont_mod):
    """CONT_PPM_AGGRESS_ADJUST / CONT_PPM_VOLUME_BASED_CFL through the whole step (all four continuity calls of a step take the
    thread-per-column kernels then), on a grid whose open face widths differ from its cell widths."""
    gg, d, M = H.double_gyre()
    run(orc, (gg, d, H.narrowed_faces(d, M)), nsteps=2, bt_mod=dict(strong_drag=1), cont_mod=cont_mod)


def test_rk2_benchmark_small_hooks_and_flags(orc):
    run(orc, H.benchmark_small(), nsteps=2, bt_mod=dict(strong_drag=1), rk2_mod=dict(begw=0.25, split_bottom_stress=1, visc_rem_dt_bug=0),
        per_stage=True, new_diff=True)


def test_rk2_device_matches_committed_golden(orc):
    """The device result against the fixture committed under tests/golden/ (made by scripts/make_golden.py from the
    oracle; the CPU suite checks that today's oracle still reproduces it): bit for bit."""
    out = run(orc, H.double_gyre(), nsteps=3, bt_mod=dict(strong_drag=1))
    gold = H.load_golden("rk2_double_gyre_strong_drag_3steps")
    d = H.double_gyre()[1]
    for n in STATE:
        H.assert_bitwise(out[n][(Ellipsis,) + tuple(H.interior(d, STAG[n]))], gold[n], "golden:" + n)


@pytest.mark.parametrize("recon", [1, 2, "quadrature"])
@pytest.mark.parametrize("form", [abi.UNESCO, abi.ROQUET_RHO, abi.JACKETT06, abi.ROQUET_SPV], ids=["UNESCO", "ROQUET_RHO", "JACKETT_06", "ROQUET_SPV"])
def test_rk2_with_the_equations_of_state_without_analytic_integrals(orc, form, recon):
    """EQN_OF_STATE = UNESCO / ROQUET_RHO (NEMO), which have no analytic integrals: the whole step through the quadratures, bit for
    bit."""
    run(orc, H.benchmark_small(), nsteps=2, bt_mod=dict(strong_drag=1), eos_form=form, recon=recon)


@pytest.mark.parametrize("recon", [0, 1, 2, "quadrature"])
@pytest.mark.parametrize("form", [abi.LINEAR, abi.WRIGHT, abi.WRIGHT_FULL, abi.WRIGHT_REDUCED], ids=["LINEAR", "WRIGHT", "WRIGHT_FULL", "WRIGHT_REDUCED"])
def test_rk2_with_equation_of_state(orc, form, recon):
    """LINEAR and the Wright family: the analytic integrals (recon = 0) or the quadratures."""
    run(orc, H.benchmark_small(), nsteps=2, bt_mod=dict(strong_drag=1), rk2_mod=dict(begw=0.2), eos_form=form, recon=recon)


@pytest.mark.parametrize("mods", [dict(), dict(harmonic_visc=1, bottomdraglaw=0)])
def test_rk2_with_device_vertvisc_coef(orc, mods):
    """vertvisc_coef called by the step itself at :609, :738, :1003 (mom6x_vertvisc_init), no host callback."""
    run(orc, H.benchmark_small(), nsteps=2, bt_mod=dict(strong_drag=1), dev_vv=mods)


def test_rk2_default_path_tolerance(orc):
    run(orc, H.double_gyre(), nsteps=3, exact=False)


@pytest.mark.parametrize("cfg,mods", [("double_gyre", dict(Ah_vel_scale=0.02)),
                                      ("island_basin", dict(Laplacian=1, Kh=500.0, Smagorinsky_Kh=1, Smag_Lap_const=0.15, Smagorinsky_Ah=1,
                                                            Smag_bi_const=0.06, Ah_vel_scale=0.02)),
                                      ("island_basin", dict(Laplacian=1, Kh=50.0, Leith_Kh=1, Leith_Lap_const=1.0, use_beta_in_Leith=1,
                                                            Leith_Ah=1, Leith_bi_const=1.0, modified_Leith=1, Ah_vel_scale=0.01))])
def test_rk2_with_device_horizontal_viscosity(orc, cfg, mods):
    """hor_visc_init given: the step (:886) and the new-run initialisation (:1601) call horizontal_viscosity
    themselves (device vertvisc_coef as well: nothing comes from callbacks).  Bit for bit over 3 steps."""
    c = getattr(H, cfg)()
    P = abi.hor_visc_params_default(1200.0)
    for k_, v_ in mods.items():
        setattr(P, k_, v_)
    from tests import cases
    P.dt = cases.rk2_inputs(c, False, False)["dt"]
    run(orc, c, nsteps=3, bt_mod=dict(strong_drag=1), dev_vv=dict(), hv=P)


@pytest.mark.parametrize("ni,nj,nk,halo", [(17, 9, 3, 4), (33, 20, 2, 3), (16, 16, 1, 4), (50, 7, 5, 4), (64, 12, 17, 4)])
def test_rk2_ragged_tile_sizes(orc, ni, nj, nk, halo):
    """Tile extents that are not multiples of the 16-face work-group tile or the 64-lane wavefront, a single layer, the
    narrowest halo the stencils allow: two steps, bit for bit."""
    run(orc, H.double_gyre(nk=nk, ni=ni, nj=nj, halo=halo), nsteps=2, bt_mod=dict(strong_drag=1))


@pytest.mark.parametrize("ni,nj,nk", [(70, 10, 75), (24, 40, 75), (40, 12, 50), (40, 12, 63)])
def test_rk2_75_layers_on_chip_columns(orc, ni, nj, nk):
    """nk = 75 is the layer count the on-chip column solver (k_vertvisc_cols: c1 and u in registers, the remnant in
    LDS) and the 5-layer-per-lane mass-flux kernel are built for; rows that are not a multiple of the 64-lane
    work-group, with and without the bottom-stress hooks, two steps bit for bit."""
    run(orc, H.benchmark_small(nk=nk, ni=ni, nj=nj), nsteps=2, bt_mod=dict(strong_drag=1), rk2_mod=dict(split_bottom_stress=1),
        per_stage=True)
    run(orc, H.benchmark_small(nk=nk, ni=ni, nj=nj), nsteps=2, bt_mod=dict(strong_drag=1), dev_vv=dict())
    # VISC_REM_TIMESTEP_BUG = False: the velocity solve without the remnant (k_vertvisc_cols<UPD, !REM>) + k_vertvisc_remnant_cols
    run(orc, H.benchmark_small(nk=nk, ni=ni, nj=nj), nsteps=2, bt_mod=dict(strong_drag=1), rk2_mod=dict(visc_rem_dt_bug=0), dev_vv=dict())
    run(orc, H.benchmark_small(nk=nk, ni=ni, nj=nj), nsteps=2, bt_mod=dict(strong_drag=1), rk2_mod=dict(visc_rem_dt_bug=0), per_stage=True)


@pytest.mark.parametrize("ni,nj,nk", [(70, 10, 75), (24, 40, 75), (70, 10, 50), (24, 40, 63), (40, 12, 76), (24, 12, 33)])
@pytest.mark.parametrize("shear", [True, False])
def test_rk2_75_layers_one_kernel_vertical_viscosity(orc, ni, nj, nk, shear):
    """Without Rayleigh drag (and without the direct-stress and KV_ML_INVZ2 options) the three vertvisc_coef calls of a step and the
    solves that follow them run as ONE kernel per direction (k_vertvisc_coef_cols: coefficients bottom-up into registers and LDS, the
    Thomas sweeps top-down from there; MODE 1 with the remnant only for :602-610).  Two steps against the oracle, which calls
    vertvisc_coef, vertvisc and vertvisc_remnant one after the other: bit for bit, with and without visc%Kv_shear, with the remnant
    in the solve's sweep and (VISC_REM_TIMESTEP_BUG = False) in a kernel of its own after it -- and the kernel did run.  75 layers is
    the instantiation of the headline; every other count up to COLS_NK_BOUND = 76 runs the instantiation with 76 slots and a uniform
    test on the layer index (mom6x_dev.h), here 33, 50, 63 and 76 layers -- also the 3-, 4- and 5-slot mass-flux kernels."""
    for rk2_mod in (None, dict(visc_rem_dt_bug=0)):
        run(orc, H.benchmark_small(nk=nk, ni=ni, nj=nj), nsteps=2, bt_mod=dict(strong_drag=1), rk2_mod=rk2_mod, dev_vv=dict(), ray=False, shear=shear)
    import os
    if os.environ.get("MOM6X_VERTVISC") in (None, ""):   # (not under the switch tests that take the kernel away)
        import torch
        from mom6_amd.dycore import Dycore, prof_enable, prof_report
        from tests import cases
        from tests.test_dyn_gpu import visc_inputs
        cfg = H.benchmark_small(nk=nk, ni=ni, nj=nj); gg, d, M = cfg
        inp = cases.rk2_inputs(cfg)
        cont, bt, cor, pgf, rk2 = cases.rk2_params(d, inp["GV"], dict(strong_drag=1), None, None)
        dyc = Dycore(d, M, inp["GV"], 0)
        dyc.continuity_init(cont); dyc.barotropic_init(bt); dyc.CoriolisAdv_init(cor); dyc.PressureForce_init(pgf, inp["Rlay"], inp["gp"])
        dyc.initialize_dyn_split_RK2(rk2)
        dyc.vertvisc_init(abi.vertvisc_params_default())
        vis = [dyc.to_dev(a) if a is not None else None for a in visc_inputs(d, M, with_shear=shear)]
        dyc.vertvisc_set_visc(*vis, None, None)
        sg = dict(u=dyc.to_dev(inp["u"]), v=dyc.to_dev(inp["v"]), h=dyc.to_dev(inp["h"]), uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=dyc.zeros3(),
                  vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
        tx, ty = dyc.to_dev(inp["taux"]), dyc.to_dev(inp["tauy"])
        torch.cuda.synchronize()
        dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], inp["dt"])
        prof_enable(dyc, True)
        dyc.step_MOM_dyn_split_RK2(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], sg["uhtr"], sg["vhtr"], sg["eta_av"], tx, ty, inp["dt"], calc_dtbt=True)
        dyc.sync()
        rep = prof_report(dyc); prof_enable(dyc, False)
        assert rep.get("k_vertvisc_coef_cols<0>", (0, 0))[0] == 3 and rep.get("k_vertvisc_coef_cols<1>", (0, 0))[0] == 3, sorted(rep)
        dyc.close()


@pytest.mark.parametrize("ray", [True, False])
def test_rk2_90_layers_beyond_the_on_chip_columns(orc, ray):
    """More layers than the on-chip column kernels carry (COLS_NK_BOUND = 76): vertvisc_coef, the solves, btcalc and the mass-flux
    kernel's 8-slot instantiation take the paths that walk through HBM.  Two steps, bit for bit."""
    run(orc, H.benchmark_small(nk=90, ni=40, nj=12), nsteps=2, bt_mod=dict(strong_drag=1), dev_vv=dict(), ray=ray)


def test_rk2_75_layers_with_btcalc_written_out():
    """Inside the step at nk = 75 btcalc only notes its operands and btstep's column pass forms frhatu / frhatv from the face
    thicknesses (k_bt_col<., true>).  MOM6X_BTCALC=eager keeps btcalc's own kernels and the stored fractions: the 75-layer cases
    again in a process with the switch set, against the same oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_rk2_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "75_layers_on_chip"], env=dict(os.environ, MOM6X_BTCALC="eager"), capture_output=True, text=True, timeout=900,
                       cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_rk2_tc4_like_switches(orc):
    """The switches of .testing/tc4 that touch this path: CORIOLIS_EN_DIS, DIRECT_STRESS (HMIX_FIXED = 20 m), BE = 0.7,
    EQN_OF_STATE = LINEAR with MASS_WEIGHT_IN_PRESSURE_GRADIENT, SMAGORINSKY_AH with SMAG_BI_CONST = 0.03, BEBT = 0.2,
    KV_ML_INVZ2 -- three steps with every callee on the device, bit for bit."""
    from tests import cases
    c = H.island_basin()
    P = abi.hor_visc_params_default(1200.0)
    P.Smagorinsky_Ah = 1; P.Smag_bi_const = 0.03
    P.dt = cases.rk2_inputs(c, False, False)["dt"]
    run(orc, c, nsteps=3, bt_mod=dict(strong_drag=1, bebt=0.2), rk2_mod=dict(be=0.7), cor_mod=dict(Coriolis_En_Dis=1, bound_Coriolis=1),
        eos_form=abi.LINEAR, dev_vv=dict(Kvml_invZ2=0.01), hv=P, Hmix_stress=20.0)


@pytest.mark.parametrize("cor_mod", [dict(Coriolis_Scheme=abi.ARAKAWA_LAMB81), dict(Coriolis_Scheme=abi.AL_BLEND, F_eff_max_blend=2.5),
                                     dict(Coriolis_Scheme=abi.ROBUST_ENSTRO, PV_Adv_Scheme=abi.PV_ADV_UPWIND1)],
                         ids=["ARAKAWA_LAMB81", "ARAKAWA_LAMB_BLEND", "ROBUST_ENSTRO"])
def test_rk2_with_the_other_coriolis_schemes(orc, cor_mod):
    """Three steps of the whole RK2 step with CORIOLIS_SCHEME = ARAKAWA_LAMB81 / ARAKAWA_LAMB_BLEND / ROBUST_ENSTRO
    (MOM_CoriolisAdv.F90:534-588, :687-721): these read uh / vh one face beyond the faces the default scheme reads, i.e. the
    halo the pass_av_uvh group (RK2.F90:804) fills."""
    run(orc, H.island_basin(), nsteps=3, bt_mod=dict(strong_drag=1), cor_mod=cor_mod)


def test_btstep_warns_when_eta_drops_below_the_bottom_and_flags_nan(orc):
    """The two run-time error paths of the device: (1) btstep's "eta has dropped below bathyT" WARNING (MOM_barotropic.F90:2738-2745):
    a sea surface far below the bottom of a shallow basin is counted on the device, sub-step by sub-step, and the first
    offender is reported; a healthy state reports nothing.  (2) MOM6X_ENUMERIC: a NaN that reaches the thicknesses in
    continuity raises at the next context synchronisation."""
    import torch, warnings
    from mom6_amd.dycore import Dycore
    from tests import cases
    cfg = H.double_gyre()
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg, False, False)
    GV, Rlay, gp, dt = inp["GV"], inp["Rlay"], inp["gp"], inp["dt"]
    cont, bt, cor, pgf, rk2 = cases.rk2_params(d, GV, dict(strong_drag=1), None, None)
    dyc = Dycore(d, M, GV, 0)
    dyc.continuity_init(cont); dyc.barotropic_init(bt); dyc.CoriolisAdv_init(cor); dyc.PressureForce_init(pgf, Rlay, gp)
    dyc.initialize_dyn_split_RK2(rk2)
    coefs = inp["coefs"]
    st = dict(u=dyc.to_dev(inp["u"]), v=dyc.to_dev(inp["v"]), h=dyc.to_dev(inp["h"]), uh=dyc.zeros3(), vh=dyc.zeros3(),
              uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
    dyc.vertvisc_set_coef(*[dyc.to_dev(x) if x is not None else None for x in coefs[0]])
    tx, ty = dyc.to_dev(inp["taux"]), dyc.to_dev(inp["tauy"])
    torch.cuda.synchronize()

    def step():
        dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"], tx, ty, dt,
                                   calc_dtbt=True)

    dyc.dyn_split_RK2_new_run(st["u"], st["v"], st["h"], st["uh"], st["vh"], dt)
    step()
    dyc.sync()
    assert dyc.btstep_warnings() == (0, None)
    # (1) drain the basin: thicknesses of a millimetre leave eta = sum(h) - bathyT some 4 km below the bottom's mirror image
    st["h"].mul_(1.0e-6); torch.cuda.synchronize()
    dyc.dyn_split_RK2_new_run(st["u"], st["v"], st["h"], st["uh"], st["vh"], dt)
    step()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        n, first = dyc.btstep_warnings()
    assert n > 0 and first is not None and first["eta"] < first["minus_bathyT"]
    assert 0 <= first["i"] < d.ni and 0 <= first["j"] < d.nj and M[G["mask2dT"], d.joff + first["j"], d.ioff + first["i"]] > 0
    assert any("eta has dropped below bathyT" in str(x.message) for x in w)
    assert dyc.btstep_warnings() == (0, None)   # reset by the previous query
    # (2) a NaN thickness (a NaN velocity alone is swallowed by the upwind branches of the flux: neither u > 0 nor u < 0):
    # the flag is raised by the convergence kernel and reported by the synchronisation
    st["h"].copy_(dyc.to_dev(inp["h"])); st["u"].copy_(dyc.to_dev(inp["u"])); st["v"].copy_(dyc.to_dev(inp["v"]))
    st["h"][0, d.joff + 15, d.ioff + 16] = float("nan"); torch.cuda.synchronize()
    dyc.dyn_split_RK2_new_run(st["u"], st["v"], st["h"], st["uh"], st["vh"], dt)
    step()
    with pytest.raises(RuntimeError, match="numeric error flag"):
        dyc.sync()
    dyc.sync()   # the flag is cleared once reported
    dyc.close()
