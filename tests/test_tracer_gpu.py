"""GPU parity, bit for bit: advect_tracer (PLM / PPM:H3 / PPM, x- and y-first) and the tridiagonal solvers."""
import numpy as np
import pytest

from mom6_amd import abi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = abi.G


@pytest.fixture(autouse=True, params=["tiled", "legacy"])
def tracer_path(request, monkeypatch):
    """Every case runs on both device paths: one kernel per direction and pass (LDS tiles in x, LDS rings of rows in y; the
    default) and the face + cell kernel pairs (MOM6X_TRACER=legacy).  Both must equal the oracle."""
    monkeypatch.setenv("MOM6X_TRACER", request.param)
    return request.param


def transports(orc, d, M, GV, dt, scale, post=1.0):
    """Accumulated transports uhtr, vhtr consistent with a thickness change (one continuity step), scaled up so
    that some fluxes must be limited and several iterations are needed."""
    CS = abi.continuity_params_default(d.nk)
    h, u, v = synth.make_state(d, M, u_max=0.4, thin_frac=0.08)
    hn = np.zeros_like(h); uh = np.zeros_like(h); vh = np.zeros_like(h)
    orc.continuity_PPM(d, M, GV, CS, 0, u * scale, v * scale, h, hn, uh, vh, dt)
    for a, s in ((hn, 0), (uh, 1), (vh, 2)):
        orc.lib().orc_pass_var(__import__("ctypes").byref(d), a.ctypes.data_as(__import__("ctypes").c_void_p), s, d.nk)
    # `post` > 1 exaggerates the accumulated transports so that cells would be emptied in one pass: the flux
    # limiter (hup/hlos logic) and several iterations are exercised.
    return hn, np.ascontiguousarray(uh * dt * post), np.ascontiguousarray(vh * dt * post)


@pytest.mark.parametrize("cfg", ["double_gyre", "channel", "benchmark_small", "wide", "ragged", "narrow"])
@pytest.mark.parametrize("schemes,first,post", [([0, 0], 0, 1.0), ([1, 1, 2], 0, 60.0), ([2, 0, 1], 1, 60.0), ([1], 1, 1.0), ([0], 0, 60.0),
                                                ([0, 1, 2, 2, 1], 0, 60.0)])      # five tracers: the 8-tracer instantiations
def test_advect_tracer(orc, cfg, schemes, first, post):
    import torch
    from mom6_amd.dycore import Dycore
    # "wide": 600 x 300 cells, i.e. three 255-cell tiles along i and three 128-row segments along j on the tiled path, whose
    # boundaries the limiter and the 5-point stencils reach across; re-entrant in x
    if cfg == "wide":
        gg, d, M = H.channel(nk=2, ni=600, nj=300)
    elif cfg in ("ragged", "narrow"):      # tiles much smaller than a 240-cell tile / a 128-row segment
        gg, d, M = H.double_gyre(nk=3, ni=17, nj=9) if cfg == "ragged" else H.double_gyre(nk=2, ni=9, nj=33)
    else:
        gg, d, M = getattr(H, cfg)()
    GV = abi.vgrid_default()
    dt_dyn, dt = 900.0, 3600.0
    h_end, uhtr, vhtr = transports(orc, d, M, GV, dt, scale=3.0, post=post)
    trs = [np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 70 + m, nk=d.nk, ox=0.5, oy=0.5) * M[G["mask2dT"]][None])
           for m in range(len(schemes))]
    tro = [t.copy() for t in trs]
    uhr_o = np.zeros_like(h_end); vhr_o = np.zeros_like(h_end)
    it_o = orc.advect_tracer(d, M, GV, first, dt_dyn, 0, h_end, uhtr, vhtr, dt, tro, schemes, uhr_out=uhr_o, vhr_out=vhr_o)
    dyc = Dycore(d, M, GV, first)
    dyc.tracer_advect_init(dt_dyn, 0)
    trg = [dyc.to_dev(t) for t in trs]
    uhr_g, vhr_g = dyc.zeros3(), dyc.zeros3()
    hd, ud, vd = dyc.to_dev(h_end), dyc.to_dev(uhtr), dyc.to_dev(vhtr)
    torch.cuda.synchronize()
    it_g = dyc.advect_tracer(hd, ud, vd, dt, trg, schemes, uhr_out=uhr_g, vhr_out=vhr_g)
    dyc.sync()
    assert it_g == it_o, (it_g, it_o)
    if post > 1.0 and cfg not in ("ragged", "narrow"):
        assert it_o >= 3, it_o      # the limiter was active: more passes than the halo cycle alone needs
    sl = H.interior(d, "h")
    for m in range(len(schemes)):
        H.assert_bitwise(trg[m].cpu().numpy(), tro[m], f"tracer {m}", sl)
        assert np.abs(tro[m] - trs[m])[(Ellipsis,) + sl].max() > 1e-6
    H.assert_bitwise(uhr_g.cpu().numpy(), uhr_o, "uhr", H.interior(d, "u"))
    H.assert_bitwise(vhr_g.cpu().numpy(), vhr_o, "vhr", H.interior(d, "v"))
    dyc.close()


@pytest.mark.parametrize("cfg", ["double_gyre", "benchmark_small", "benchmark_75"])
def test_advect_more_tracers_than_one_pass_carries(orc, cfg):
    """The reference's tracer registry has no upper bound; the device kernels carry 8 tracers through a pass, a longer list goes
    eight at a time, each group through the whole iteration (mom6x_advect_tracer).  Eleven tracers of mixed schemes == the oracle,
    bit for bit, the leftover transports included."""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = H.benchmark_small(nk=75, ni=70, nj=10) if cfg == "benchmark_75" else getattr(H, cfg)()
    GV = abi.vgrid_default()
    dt_dyn, dt = 900.0, 3600.0
    schemes = [0, 1, 2, 2, 1, 0, 2, 1, 0, 2, 2]
    h_end, uhtr, vhtr = transports(orc, d, M, GV, dt, scale=3.0, post=60.0)
    trs = [np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 70 + m, nk=d.nk, ox=0.5, oy=0.5) * M[G["mask2dT"]][None]) for m in range(len(schemes))]
    tro = [t.copy() for t in trs]
    uhr_o = np.zeros_like(h_end); vhr_o = np.zeros_like(h_end)
    it_o = orc.advect_tracer(d, M, GV, 0, dt_dyn, 0, h_end, uhtr, vhtr, dt, tro, schemes, uhr_out=uhr_o, vhr_out=vhr_o)
    dyc = Dycore(d, M, GV, 0)
    dyc.tracer_advect_init(dt_dyn, 0)
    trg = [dyc.to_dev(t) for t in trs]
    uhr_g, vhr_g = dyc.zeros3(), dyc.zeros3()
    hd, ud, vd = dyc.to_dev(h_end), dyc.to_dev(uhtr), dyc.to_dev(vhtr)
    torch.cuda.synchronize()
    it_g = dyc.advect_tracer(hd, ud, vd, dt, trg, schemes, uhr_out=uhr_g, vhr_out=vhr_g)
    dyc.sync()
    assert it_g == it_o, (it_g, it_o)
    sl = H.interior(d, "h")
    for m in range(len(schemes)):
        H.assert_bitwise(trg[m].cpu().numpy(), tro[m], f"tracer {m} of 11", sl)
    H.assert_bitwise(uhr_g.cpu().numpy(), uhr_o, "uhr", H.interior(d, "u"))
    H.assert_bitwise(vhr_g.cpu().numpy(), vhr_o, "vhr", H.interior(d, "v"))
    dyc.close()


@pytest.mark.parametrize("cfg", ["double_gyre", "benchmark_small", "benchmark_75"])
def test_tridiagonal_solvers(orc, cfg):
    """(benchmark_75: nk = 75 is the layer count of the on-chip column kernel k_tridiag_cols; 70 columns per row = one full and
    one ragged wavefront)"""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = H.benchmark_small(nk=75, ni=70, nj=10) if cfg == "benchmark_75" else getattr(H, cfg)()
    GV = abi.vgrid_default()
    nk = d.nk
    h, _, _ = synth.make_state(d, M, thin_frac=0.05)
    ent = np.abs(synth.smooth_field(d, 81, nk=nk + 1, ox=0.5, oy=0.5)) * 5.0
    ent[0] = 0.0; ent[nk] = 0.0
    ent = np.ascontiguousarray(ent)
    ea = np.ascontiguousarray(ent[:nk]); eb = np.ascontiguousarray(ent[1:])
    T = np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 82, nk=nk, ox=0.5, oy=0.5))
    sfc = np.ascontiguousarray(1e-3 * synth.smooth_field(d, 83, ox=0.5, oy=0.5)); btm = np.ascontiguousarray(0.5 * sfc)
    dt = 3600.0
    o = {}
    o["ts"] = T.copy(); orc.triDiagTS(d, h, ea, eb, o["ts"], GV.H_subroundoff)
    o["tse"] = T.copy(); orc.triDiagTS_Eulerian(d, h, ent, o["tse"], GV.H_subroundoff)
    o["vd"] = T.copy(); orc.tracer_vertdiff(d, M, GV, h, ea, eb, dt, o["vd"], sfc, btm, True)
    o["vde"] = T.copy(); orc.tracer_vertdiff(d, M, GV, h, np.ascontiguousarray(ent[:nk]), np.ascontiguousarray(ent[1:]), dt, o["vde"], None, None, True)
    dyc = Dycore(d, M, GV)
    hd, ead, ebd, entd = dyc.to_dev(h), dyc.to_dev(ea), dyc.to_dev(eb), dyc.to_dev(ent)
    g = {k: dyc.to_dev(T) for k in o}
    S2 = dyc.to_dev(T)
    sd, bd = dyc.to_dev(sfc), dyc.to_dev(btm)
    torch.cuda.synchronize()
    dyc.triDiagTS(hd, ead, ebd, g["ts"], S2)
    dyc.triDiagTS_Eulerian(hd, entd, g["tse"])
    dyc.tracer_vertdiff(hd, ead, ebd, dt, g["vd"], sd, bd, True)
    dyc.tracer_vertdiff_Eulerian(hd, entd, dt, g["vde"])
    dyc.sync()
    sl = H.interior(d, "h")
    for k in o:
        H.assert_bitwise(g[k].cpu().numpy(), o[k], k, sl)
    H.assert_bitwise(S2.cpu().numpy(), o["ts"], "S", sl)
    np.testing.assert_array_equal(o["ts"], o["tse"])
    dyc.close()


@pytest.mark.parametrize("cfg", ["double_gyre", "benchmark_small", "benchmark_75"])
@pytest.mark.parametrize("reservoir", [False, True])
def test_tracer_vertdiff_with_sinking(orc, cfg, reservoir):
    """tracer_vertdiff / tracer_vertdiff_Eulerian with sink_rate (MOM_tracer_diabatic.F90:123-179 / :315-380): the limited sinking
    distances (all three branches of :134-146: the synthetic thin layers are thinner than the sinking distance, the thick ones not),
    the solve with the sinking flux on the lower diagonal, and -- with btm_reservoir -- the unlimited form that collects what leaves
    the bottom layer.  Device == oracle, bit for bit."""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = H.benchmark_small(nk=75, ni=70, nj=10) if cfg == "benchmark_75" else getattr(H, cfg)()
    GV = abi.vgrid_default()
    nk = d.nk
    h, _, _ = synth.make_state(d, M, thin_frac=0.15)
    ent = np.abs(synth.smooth_field(d, 81, nk=nk + 1, ox=0.5, oy=0.5)) * 5.0
    ent[0] = 0.0; ent[nk] = 0.0
    ent = np.ascontiguousarray(ent)
    ea = np.ascontiguousarray(ent[:nk]); eb = np.ascontiguousarray(ent[1:])
    T = np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 82, nk=nk, ox=0.5, oy=0.5))
    sfc = np.ascontiguousarray(1e-3 * synth.smooth_field(d, 83, ox=0.5, oy=0.5)); btm = np.ascontiguousarray(0.5 * sfc)
    dt = 3600.0
    hmean = float(h[(Ellipsis,) + tuple(H.interior(d, "h"))].mean())
    sink_rate = 0.7 * hmean / dt        # [Z T-1]: most layers are thicker than the sinking distance, the thin ones are not
    res_o = np.ascontiguousarray(0.1 + 0.05 * synth.smooth_field(d, 84, ox=0.5, oy=0.5)) if reservoir else None
    res_oe = res_o.copy() if reservoir else None
    o = {"vd": T.copy(), "vde": T.copy()}
    orc.tracer_vertdiff_sink(d, M, GV, h, ea, eb, dt, o["vd"], sink_rate, sfc, btm, res_o, True)
    orc.tracer_vertdiff_sink(d, M, GV, h, np.ascontiguousarray(ent[:nk]), np.ascontiguousarray(ent[1:]), dt, o["vde"], sink_rate, None, None, res_oe, False)
    assert np.abs(o["vd"] - T).max() > 1e-3
    dyc = Dycore(d, M, GV)
    hd, ead, ebd, entd = dyc.to_dev(h), dyc.to_dev(ea), dyc.to_dev(eb), dyc.to_dev(ent)
    g = {k: dyc.to_dev(T) for k in o}
    sd, bd = dyc.to_dev(sfc), dyc.to_dev(btm)
    res_g = dyc.to_dev(0.1 + 0.05 * synth.smooth_field(d, 84, ox=0.5, oy=0.5)) if reservoir else None
    res_ge = res_g.clone() if reservoir else None
    torch.cuda.synchronize()
    dyc.tracer_vertdiff_sink(hd, ead, ebd, dt, g["vd"], sink_rate, sd, bd, res_g, True)
    dyc.tracer_vertdiff_sink(hd, entd, None, dt, g["vde"], sink_rate, None, None, res_ge, False, eulerian=True)
    dyc.sync()
    sl = H.interior(d, "h")
    for k in o:
        H.assert_bitwise(g[k].cpu().numpy(), o[k], "sink:" + k, sl)
    if reservoir:
        H.assert_bitwise(res_g.cpu().numpy(), res_o, "sink:btm_reservoir", sl)
        H.assert_bitwise(res_ge.cpu().numpy(), res_oe, "sink:btm_reservoir (Eulerian)", sl)
        assert (res_o[sl] - (0.1 + 0.05 * synth.smooth_field(d, 84, ox=0.5, oy=0.5))[sl]).max() > 0.0
    dyc.close()
