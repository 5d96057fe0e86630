"""The reference's test.layout on ONE GPU: the same run on a 2 x 1 / 2 x 2 / 1 x 2 tile layout, one tile per host thread
(tests/transport: the halo exchanges and global reductions of mom6_amd/csrc/halo.hip go between threads instead of over
RCCL), against the one-tile run.  Every prognostic field of every tile must equal its part of the one-tile result bit for
bit: this is what the N-GPU runs rely on -- which rows and columns each kernel covers on an interior tile edge, the
wide-halo cycles of the barotropic solver, the all-reduces (dtbt, tracer iteration flags, reproducing sums)."""
import os
import threading

import numpy as np
import pytest

from mom6_amd import abi, parallel, sum_output as SO
from tests import helpers as H

pytestmark = pytest.mark.gpu

STATE = ["u", "v", "h", "uh", "vh", "uhtr", "vhtr", "eta_av"]
STAG = dict(u="u", v="v", h="h", uh="u", vh="v", uhtr="u", vhtr="v", eta_av="h", T="h")


def run_tile(cfg_fn, nk, layout, pe, uid, nsteps, bt_mod, out, errors, rich=False, halo=4, bt_tile=None):
    """One tile of the layout: a few baroclinic steps, a tracer advection with the accumulated transports, write_energy."""
    try:
        import torch
        from mom6_amd.dycore import Dycore
        from tests import cases
        gg, d1, M1 = cfg_fn(nk=nk)                                  # the one-tile grid: seeded inputs are made on it ...
        inp = cases.rk2_inputs((gg, d1, M1), False, False)
        T1 = cases.thermo_state(d1, M1)[0]
        d, M = gg.tile(nk, halo, layout, pe)                        # ... and cut to this tile (with its halos, which may be wider)

        def cut(a):
            j0, i0 = d.j_glob0 - d.halo, d.i_glob0 - d.halo         # global index of the tile's first memory row / column
            rows = np.arange(d.nrows) - d.joff + d.j_glob0
            cols = np.arange(d.pitch) - d.ioff + d.i_glob0
            if getattr(gg, "reentrant_x", False): cols = np.mod(cols, d1.ni)     # (a halo wider than the one-tile grid's own)
            if getattr(gg, "reentrant_y", False): rows = np.mod(rows, d1.nj)
            src_r = np.clip(rows + d1.joff, 0, d1.nrows - 1); src_c = np.clip(cols + d1.ioff, 0, d1.pitch - 1)
            return np.ascontiguousarray(a[..., src_r[:, None], src_c[None, :]])
        GV, Rlay, gp, dt = inp["GV"], inp["Rlay"], inp["gp"], inp["dt"]
        cont, bt, cor, pgf, rk2 = cases.rk2_params(d, GV, dict(bt_mod, **(bt_tile or {})) if layout is not None else bt_mod, None, None)
        dyc = Dycore(d, M, GV, 0)
        if halo > 4:
            dyc.set_dyn_pass_width(4)                               # NIHALO rows of the 3-D fields, the wide halo for the 2-D ones
        if layout != (1, 1):
            parallel.attach_comm(dyc, layout, pe, None, unique_id=uid)
        dyc.continuity_init(cont); dyc.barotropic_init(bt); dyc.CoriolisAdv_init(cor); dyc.PressureForce_init(pgf, Rlay, gp)
        dyc.initialize_dyn_split_RK2(rk2)
        if rich:   # every callee of the step on the device: vertvisc_coef, horizontal_viscosity, the EOS pressure force
            from tests.test_dyn_gpu import visc_inputs
            vis = list(visc_inputs(d1, M1)) + [inp["coefs"][0][4], inp["coefs"][0][5]]
            dyc.vertvisc_init(abi.vertvisc_params_default())
            dyc.vertvisc_set_visc(*[dyc.to_dev(cut(a)) if a is not None else None for a in vis])
            hvP = abi.hor_visc_params_default(dt)
            for k_, v_ in dict(Laplacian=1, Kh=500.0, Smagorinsky_Kh=1, Smag_Lap_const=0.15, Smagorinsky_Ah=1, Smag_bi_const=0.06,
                               Ah_vel_scale=0.02).items():
                setattr(hvP, k_, v_)
            dyc.hor_visc_init(hvP)
            Tt, St = cases.thermo_state(d1, M1)
            tvd = (dyc.to_dev(cut(Tt)), dyc.to_dev(cut(St)))
            eos = abi.eos_params_default(abi.WRIGHT); eos.MassWghtInterp = 1
            dyc.PressureForce_set_tv(tvd[0], tvd[1], eos)
            keep = (tvd, eos, hvP)                                    # noqa: F841 -- the context holds their addresses
        else:
            dyc.vertvisc_set_coef(*[dyc.to_dev(cut(a)) if a is not None else None for a in inp["coefs"][0]])
        dyc.tracer_advect_init(dt, 2)
        dyc.sum_output_init(abi.sum_output_params_default(dt), gp)
        sg = dict(u=dyc.to_dev(cut(inp["u"])), v=dyc.to_dev(cut(inp["v"])), h=dyc.to_dev(cut(inp["h"])), uh=dyc.zeros3(), vh=dyc.zeros3(),
                  uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2(), T=dyc.to_dev(cut(T1)))
        txd, tyd = dyc.to_dev(cut(inp["taux"])), dyc.to_dev(cut(inp["tauy"]))
        torch.cuda.synchronize()
        dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)
        lines = []
        stats = SO.SumOutput()
        for n in range(nsteps):
            dyc.step_MOM_dyn_split_RK2(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], sg["uhtr"], sg["vhtr"], sg["eta_av"], txd, tyd,
                                       dt, calc_dtbt=(n == 0))
            lines.append(stats.record(dyc.write_energy(sg["u"], sg["v"], sg["h"]), dt * (n + 1), n + 1)[1])
        dyc.advect_tracer(sg["h"], sg["uhtr"], sg["vhtr"], nsteps * dt, [sg["T"]])
        dyc.sync()
        res = {n: sg[n].cpu().numpy() for n in STATE + ["T"]}
        res["dtbt"] = dyc.barotropic_dtbt(); res["lines"] = lines; res["dims"] = d
        if os.environ.get("MOM6X_POISON_HALO") == "1" and layout != (1, 1):   # (the switch did poison: the step's last passes are 2-3 rows wide)
            assert np.isnan(res["u"]).any() and np.isnan(res["h"]).any(), "MOM6X_POISON_HALO left no NaN in the halos"
        # the debugging checksums (hchksum / uvchksum) and the restart checksum are sums over all tiles
        res["chk"] = [dyc.chksum(sg["h"], "h", haloshift=1), dyc.chksum(sg["u"], "u"), dyc.chksum(sg["v"], "v", haloshift=2, omit_corners=True),
                      dyc.field_chksum(sg["T"])]
        out[pe] = res
        dyc.close()
    except Exception as e:                                            # noqa: BLE001 -- reported by the main thread
        import traceback
        errors.append((pe, traceback.format_exc()))


@pytest.mark.parametrize("cfg_name,layout,rich", [("channel", (2, 1), False), ("double_gyre", (2, 2), False), ("benchmark_small", (1, 2), False),
                                                  ("island_basin", (2, 2), True), ("channel", (2, 1), True),
                                                  ("channel", (4, 2), False), ("benchmark_small", (4, 2), True)])   # the 8-GPU layout
def test_tile_layout_gives_the_one_tile_answer(cfg_name, layout, rich):
    _layout_case(cfg_name, layout, rich)


@pytest.mark.parametrize("cfg_name,layout,rich", [("double_gyre", (2, 2), False), ("benchmark_small", (4, 2), True)])
def test_tile_layout_in_the_reference_sum_order(cfg_name, layout, rich, monkeypatch):
    """The same with MOM6X_SUMS=exact: the LDS mass-flux kernel (sequential k sums) next to interior tile edges."""
    monkeypatch.setenv("MOM6X_SUMS", "exact")
    _layout_case(cfg_name, layout, rich)


@pytest.mark.parametrize("cfg_name,layout,rich,halo,bt", [("channel", (2, 1), False, 8, None), ("double_gyre", (2, 2), False, 6, None),
                                                          ("benchmark_small", (4, 2), True, 8, dict(BTHALO=8)),
                                                          ("island_basin", (1, 1), True, 8, None),
                                                          ("channel", (2, 1), False, 4, dict(use_wide_halos=0)),
                                                          ("benchmark_small", (2, 2), False, 8, dict(min_stencil=2, BTHALO=6))])
def test_wide_halos_give_the_same_answer(cfg_name, layout, rich, halo, bt):
    """BT_USE_WIDE_HALOS with BTHALO > NIHALO (MOM_barotropic.F90:5446-5461, :5717, the wide-halo cycle :2505-2512): the barotropic
    solver takes the number of sub-steps between two exchanges from the halo width of its domain.  The device context is created
    with that width (halo = 6 / 8 here): the sub-cycle then exchanges every 6 / 8 sub-steps instead of every 4 -- and every field
    of every tile still equals the one-tile, halo-4 run bit for bit.  Likewise without BT_USE_WIDE_HALOS (an exchange every
    sub-step) and with BT_WIDE_HALO_MIN_STENCIL = 2 (the valid range shrinks by two points per sub-step)."""
    _layout_case(cfg_name, layout, rich, halo, bt)


def test_bthalo_beyond_the_context_halo_is_refused():
    from mom6_amd.dycore import Dycore
    gg, d, M = H.double_gyre()
    dyc = Dycore(d, M, abi.vgrid_default(), 0)
    bt = abi.barotropic_params_default(30.0); bt.BTHALO = d.halo + 2
    with pytest.raises(abi.Mom6xError, match="BTHALO exceeds the halo of the tile context"):
        dyc.barotropic_init(bt)
    dyc.close()


def _layout_case(cfg_name, layout, rich, halo=4, bt_tile=None):
    from mom6_amd.abi import load_library
    H.use_threads_transport(load_library())
    try:
        _tile_layout_gives_the_one_tile_answer(cfg_name, layout, rich, halo, bt_tile)
    finally:
        H.use_threads_transport(load_library(), on=False)


def _tile_layout_gives_the_one_tile_answer(cfg_name, layout, rich, halo=4, bt_tile=None):
    from mom6_amd.abi import load_library
    cfg_fn = getattr(H, cfg_name)
    nk, nsteps, bt_mod = 3, 3, dict(strong_drag=1)
    errors = []
    ref = {}
    run_tile(cfg_fn, nk, (1, 1), (0, 0), None, nsteps, bt_mod, ref, errors, rich)
    assert not errors, errors[0][1]
    uid = parallel.unique_id(load_library())
    out = {}
    pes = [(px, py) for py in range(layout[1]) for px in range(layout[0])]
    threads = [threading.Thread(target=run_tile, args=(cfg_fn, nk, layout, pe, uid, nsteps, bt_mod, out, errors, rich, halo, bt_tile)) for pe in pes]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[0][1]
    assert len(out) == len(pes)
    whole = ref[(0, 0)]; d1 = whole["dims"]
    for pe in pes:
        r = out[pe]; d = r["dims"]
        assert r["dtbt"] == whole["dtbt"] and r["lines"] == whole["lines"], (pe, r["lines"], whole["lines"])
        if not (cfg_name == "channel"):      # (shifted bit counts look at the halos beyond closed boundaries, which tiles and one tile share)
            assert r["chk"] == whole["chk"], (pe, r["chk"], whole["chk"])
        else:
            assert r["chk"][3] == whole["chk"][3] and r["chk"][1] == whole["chk"][1] and r["chk"][0]["bc0"] == whole["chk"][0]["bc0"]
        for n in STATE + ["T"]:
            st = STAG[n]
            i0 = -1 if st == "u" else 0; j0 = -1 if st == "v" else 0
            mine = r[n][(Ellipsis,) + tuple(d.sl(i0, d.ni - 1, j0, d.nj - 1))]
            part = whole[n][(Ellipsis,) + tuple(d1.sl(d.i_glob0 + i0, d.i_glob0 + d.ni - 1, d.j_glob0 + j0, d.j_glob0 + d.nj - 1))]
            H.assert_bitwise(mine, part, f"{cfg_name} {layout} tile {pe}: {n}")
            assert np.isfinite(mine).all(), (cfg_name, layout, pe, n)      # (MOM6X_POISON_HALO: equal bits must not be equal NaNs)


@pytest.mark.parametrize("layout", ["4 2", "2 2", "2 1"])
def test_bench_model_layouts_at_full_size(layout):
    """bench.py's own 1440 x 1080 x 75 model on the 8-GPU (4 x 2), BASELINE.json configs[3]'s 4-GPU (2 x 2) and the 2-GPU (2 x 1) layout against one tile
    (scripts/check_layout_fullsize.py): messages of 13 MB, tiles of 360 x 540, the restart checksums of every field equal."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_layout_fullsize.py")] + layout.split() + ["2"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "every field checksum and dtbt identical" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
