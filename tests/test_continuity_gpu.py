"""GPU parity: HIP continuity_PPM (through the C ABI) vs the oracle, bit for bit.

FP64, no FMA contraction on either side, identical operation order => bit-exact is the bar
(integer-like strictness; any index or ordering bug shows up immediately).  Both orders of the column sums are
covered (mom6x_continuity_params.sum_order): the reference's sequential order on the three device paths that keep
it, and the 16-lane tree of the wave-owned kernel, which the oracle restates (oracle/orc_continuity.c::tree16_sum)
and which tests/test_oracle_cpu.py holds to the sequential order within 1e-13 of each field's range."""
import numpy as np
import pytest

from mom6_amd import abi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["wave", "lds", "lds_walk", "legacy", "wave_fma"])
def massflux_path(request, monkeypatch):
    """Every case runs on all device paths: the wave-owned kernel (sum_order = TREE16, the default), and with the
    reference's sequential sums the LDS-resident fused kernel, the same with the sequential duL/duR recurrence forced
    (lds_walk: the fall-back of the parallel min + certificate) and the thread-per-column kernels
    (MOM6X_MASSFLUX=legacy).  All must equal the oracle (run with the same sum_order) bit for bit.  wave_fma: the wave-owned kernel
    with fused multiply-adds at fixed sites (sum_order = MOM6X_SUM_TREE16_FMA, opt-in) against the oracle's restatement of the
    same sites -- also bit for bit."""
    monkeypatch.setenv("MOM6X_MASSFLUX", "wave" if request.param == "wave_fma" else request.param)
    monkeypatch.setenv("MOM6X_SUMS", {"wave": "tree", "wave_fma": "fma"}.get(request.param, "exact"))
    return request.param


def _run_case(orc, cfg, first_direction, mode, cs_mod=None, thin=0.0, u_scale=1.0, bt_pert=0.05, ties=False, stats=False):
    import torch
    from mom6_amd.dycore import Dycore, BTContDev
    gg, d, M = cfg
    GV = abi.vgrid_default()
    CS = abi.continuity_params_default(d.nk, GV.Angstrom_H)
    if cs_mod:
        for k, v in cs_mod.items():
            setattr(CS, k, v)
    h, u, v = synth.make_state(d, M, thin_frac=thin)
    u = np.ascontiguousarray(u * u_scale); v = np.ascontiguousarray(v * u_scale)
    dt = 1200.0
    rng = np.random.default_rng(5)
    vr_u = np.clip(0.5 + 0.6 * synth.smooth_field(d, 11, nk=d.nk, ox=1.0, oy=0.5), 0.0, 1.0)
    vr_v = np.clip(0.5 + 0.6 * synth.smooth_field(d, 12, nk=d.nk, ox=0.5, oy=1.0), 0.0, 1.0)
    if ties:   # depth-independent velocity and visc_rem over most of the column: the duL/duR quotients tie exactly
        kt = max(d.nk - 2, 1)
        u[:kt] = u[0]; v[:kt] = v[0]; vr_u[:kt] = vr_u[0]; vr_v[:kt] = vr_v[0]
        vr_u[:, ::3, :] = 0.0; vr_v[:, :, ::4] = 0.0   # and columns without any viscous remnant
        u = np.ascontiguousarray(u); v = np.ascontiguousarray(v)
    vr_u = np.ascontiguousarray(vr_u); vr_v = np.ascontiguousarray(vr_v)
    # reference transports to perturb into uhbt/vhbt
    h0 = np.zeros_like(h); uh0 = np.zeros_like(h); vh0 = np.zeros_like(h)
    orc.continuity_PPM(d, M, GV, CS, first_direction, u, v, h, h0, uh0, vh0, dt)
    uhbt = uh0.sum(0) * (1.0 + bt_pert * synth.smooth_field(d, 13, ox=1.0, oy=0.5))
    vhbt = vh0.sum(0) * (1.0 - bt_pert * synth.smooth_field(d, 14, ox=0.5, oy=1.0))

    kw_o, kw_g = {}, {}
    dyc = Dycore(d, M, GV, first_direction)
    dyc.continuity_init(CS)
    names3 = ["h", "uh", "vh"]
    out_o = {n: np.zeros_like(h) for n in names3}
    bt_o = None; bt_g = None
    if mode in ("bt_cont", "full"):
        bt_o = orc.new_bt_cont(d); bt_g = BTContDev(dyc)
        kw_o["BT_cont"] = bt_o; kw_g["BT_cont"] = bt_g
    if mode in ("visc", "bt_cont", "full", "adjust"):
        kw_o.update(visc_rem_u=vr_u, visc_rem_v=vr_v)
        kw_g.update(visc_rem_u=dyc.to_dev(vr_u), visc_rem_v=dyc.to_dev(vr_v))
    if mode in ("adjust", "full", "adjust_novisc"):
        out_o.update(u_cor=np.zeros_like(h), v_cor=np.zeros_like(h))
        kw_o.update(uhbt=uhbt, vhbt=vhbt, u_cor=out_o["u_cor"], v_cor=out_o["v_cor"])
        du_o = np.zeros(d.shape2()); dv_o = np.zeros(d.shape2())
        kw_o.update(du_cor=du_o, dv_cor=dv_o)
        out_o.update(du_cor=du_o, dv_cor=dv_o)
    orc.continuity_PPM(d, M, GV, CS, first_direction, u, v, h, out_o["h"], out_o["uh"], out_o["vh"], dt, **kw_o)

    out_g = {n: dyc.zeros3() for n in names3}
    if "u_cor" in out_o:
        out_g.update(u_cor=dyc.zeros3(), v_cor=dyc.zeros3(), du_cor=dyc.zeros2(), dv_cor=dyc.zeros2())
        kw_g.update(uhbt=dyc.to_dev(uhbt), vhbt=dyc.to_dev(vhbt), u_cor=out_g["u_cor"], v_cor=out_g["v_cor"],
                    du_cor=out_g["du_cor"], dv_cor=out_g["dv_cor"])
    ud, vd, hd = dyc.to_dev(u), dyc.to_dev(v), dyc.to_dev(h)
    torch.cuda.synchronize()
    if stats:
        dyc.continuity_stats(1)
    dyc.continuity_PPM(ud, vd, hd, out_g["h"], out_g["uh"], out_g["vh"], dt, **kw_g)
    dyc.sync()
    counts = dyc.continuity_stats(0) if stats else None

    sl = {"h": H.interior(d, "h"), "uh": H.interior(d, "u"), "vh": H.interior(d, "v"),
          "u_cor": H.interior(d, "u"), "v_cor": H.interior(d, "v"),
          "du_cor": H.interior(d, "u"), "dv_cor": H.interior(d, "v")}
    for n, a in out_o.items():
        H.assert_bitwise(out_g[n].cpu().numpy(), a, f"{mode}:{n}", sl[n])
    if bt_o is not None:
        for n in abi.BTCont._names:
            st = "u" if ("_u" in n or n.startswith("uBT")) else "v"
            H.assert_bitwise(bt_g[n].cpu().numpy(), bt_o[n], f"{mode}:BT_cont%{n}", H.interior(d, st))
    dyc.close()
    return counts


@pytest.mark.parametrize("mode", ["plain", "visc", "adjust_novisc", "adjust", "bt_cont", "full"])
@pytest.mark.parametrize("first_direction", [0, 1])
def test_continuity_double_gyre(orc, mode, first_direction):
    _run_case(orc, H.double_gyre(), first_direction, mode)


@pytest.mark.parametrize("mode", ["plain", "full"])
def test_continuity_channel_reentrant(orc, mode):
    _run_case(orc, H.channel(), 0, mode)


def test_continuity_thin_layers_and_tc1_tolerances(orc):
    # tc1 / p0 settings: ETA_TOLERANCE=1e-6, VELOCITY_TOLERANCE=1e-3 (.testing/tc1/MOM_input)
    _run_case(orc, H.benchmark_small(), 0, "full", cs_mod=dict(tol_eta=1e-6, tol_vel=1e-3), thin=0.15)


@pytest.mark.parametrize("flag", ["monotonic", "simple_2nd", "upwind_1st"])
def test_continuity_scheme_flags(orc, flag):
    _run_case(orc, H.benchmark_small(), 0, "full", cs_mod={flag: 1})


@pytest.mark.parametrize("flags", [dict(vol_CFL=1), dict(aggress_adjust=1, vol_CFL=1), dict(aggress_adjust=1, vol_CFL=0)])
@pytest.mark.parametrize("mode", ["adjust_novisc", "full"])
def test_continuity_aggress_adjust_and_volume_based_cfl(orc, flags, mode):
    """CONT_PPM_AGGRESS_ADJUST / CONT_PPM_VOLUME_BASED_CFL (MOM_continuity_PPM.F90:2725-2733; non-default): whatever path and sum
    order are asked for, the thread-per-column kernels run (reference order) -- velocities and adjustments strong enough for the
    limits on du to bind, faces narrower than their cells."""
    for cfg, fd in ((H.benchmark_small(), 0), (H.double_gyre(), 1)):
        gg, d, M = cfg
        _run_case(orc, (gg, d, H.narrowed_faces(d, M)), fd, mode, cs_mod=flags, u_scale=8.0, bt_pert=0.9)


@pytest.mark.parametrize("mode", ["bt_cont", "full"])
@pytest.mark.parametrize("cfg", ["double_gyre", "benchmark_small"])
def test_continuity_tied_quotients(orc, mode, cfg):
    """Barotropic columns: the candidates of the duL / duR recurrences of set_*_BT_cont tie exactly, the case in
    which the parallel min must reproduce the sequential loop's value (and zero visc_rem columns skip it)."""
    _run_case(orc, getattr(H, cfg)(nk=6), 0, mode, ties=True)


@pytest.mark.parametrize("nk", [20, 40, 48, 50, 63, 75, 90])
@pytest.mark.parametrize("mode", ["full", "bt_cont", "adjust"])
def test_continuity_many_layers(orc, nk, mode):
    # every layer lane carries ceil(nk/16) layers: the 2-, 3-, 4-, 5- and 8-slot instantiations, in the three shapes of a step's
    # launches (the 3-, 4- and 5-slot kernels have an instantiation compiled for each: struct Sw<SPEC>)
    _run_case(orc, H.benchmark_small(nk=nk), 1, mode, thin=0.1)


def test_continuity_device_matches_committed_golden(orc):
    """HIP continuity_PPM (corrector-call shape) against tests/golden/continuity_benchmark_small_corrector.npz."""
    import torch
    from mom6_amd.dycore import Dycore
    from tests import cases
    cfg = H.benchmark_small()
    gg, d, M = cfg
    inp = cases.continuity_inputs(cfg)
    _, uhbt, vhbt = cases.oracle_continuity(orc, cfg, inp)     # only for the uhbt/vhbt inputs
    dyc = Dycore(d, M, inp["GV"], 0)
    dyc.continuity_init(inp["CS"])
    out = {n: dyc.zeros3() for n in ("h", "uh", "vh", "u_cor", "v_cor")}
    dyc.continuity_PPM(dyc.to_dev(inp["u"]), dyc.to_dev(inp["v"]), dyc.to_dev(inp["h"]), out["h"], out["uh"], out["vh"], inp["dt"],
                       uhbt=dyc.to_dev(uhbt), vhbt=dyc.to_dev(vhbt), visc_rem_u=dyc.to_dev(inp["vr_u"]),
                       visc_rem_v=dyc.to_dev(inp["vr_v"]), u_cor=out["u_cor"], v_cor=out["v_cor"])
    dyc.sync()
    gold = H.load_golden("continuity_benchmark_small_corrector")
    stag = dict(h="h", uh="u", vh="v", u_cor="u", v_cor="v")
    for n in out:
        H.assert_bitwise(out[n].cpu().numpy()[(Ellipsis,) + tuple(H.interior(d, stag[n]))], gold[n], "golden:" + n)
    dyc.close()


@pytest.mark.parametrize("u_scale,bt_pert", [(8.0, 0.9), (20.0, 3.0)])
def test_continuity_newton_reaches_cfl_limits(orc, u_scale, bt_pert):
    """Fast flow and a barotropic transport far from the layer sum: the Newton steps of flux_adjust run into the CFL
    limits du_max_CFL / du_min_CFL (bisection branch :1198-1214), which on the LDS path forces the exact limit
    recurrence after the first attempt with cheap bounds."""
    _run_case(orc, H.benchmark_small(nk=20), 0, "full", u_scale=u_scale, bt_pert=bt_pert)

