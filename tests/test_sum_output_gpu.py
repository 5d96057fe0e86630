"""GPU parity of write_energy (mom6_amd/csrc/diag_sums.hip through the C ABI; MOM_sum_output.F90:321) with the oracle,
and the text of ocean.stats from device-resident fields.  Reproducing sums: every number must be identical."""
import numpy as np
import pytest

from mom6_amd import abi, synth, sum_output as SO
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = abi.G


def same(got, ref):
    for n in ("mass_tot", "KE_tot", "PE_tot"):
        assert got[n] == ref[n], (n, got[n], ref[n])
    assert tuple(got["max_CFL"]) == tuple(ref["max_CFL"])
    for n in ("mass_EFP", "salt_EFP", "heat_EFP", "mass_lay", "KE", "PE", "Z_0APE"):
        assert np.array_equal(got[n], ref[n]), (n, got[n], ref[n])


@pytest.mark.parametrize("cfg,nk,thermo", [("double_gyre", 2, False), ("benchmark_small", 7, True), ("island_basin", 5, True),
                                           ("channel", 3, False), ("ragged", 3, True)])
def test_write_energy_matches_oracle(orc, cfg, nk, thermo):
    from mom6_amd.dycore import Dycore
    from tests import cases
    gg, d, M = H.double_gyre(nk=nk, ni=17, nj=9, halo=3) if cfg == "ragged" else getattr(H, cfg)(nk=nk)
    GV = abi.vgrid_default()
    g_prime = np.concatenate(([GV.g_Earth], 0.01 + 0.002 * np.arange(nk - 1)))
    P = abi.sum_output_params_default(900.0, use_temperature=int(thermo))
    dyc = Dycore(d, M, GV)
    dyc.sum_output_init(P, g_prime)
    st = orc.SumOutputState(d, M, GV, g_prime, P)
    for a, b in zip(dyc.depth_list(), st.DL):
        assert np.array_equal(a, b)
    T = S = Td = Sd = None
    if thermo:
        T, S = cases.thermo_state(d, M)
        Td, Sd = dyc.to_dev(T), dyc.to_dev(S)
    # three different states in a row: the search hints CS%lH carry over from call to call
    for it, (thin, amp) in enumerate(((0.0, 1.0), (0.2, 3.0), (0.0, 0.0))):
        h, u, v = synth.make_state(d, M, thin_frac=thin)
        u = np.ascontiguousarray(u * amp); v = np.ascontiguousarray(v * amp)
        if it == 1:
            h[0] += 0.3 * (M[G["mask2dT"]] > 0) * (1.0 + synth.smooth_field(d, 21))
        got = dyc.write_energy(dyc.to_dev(u), dyc.to_dev(v), dyc.to_dev(h), Td, Sd)
        ref = orc.write_energy(st, u, v, h, T, S)
        same(got, ref)
        assert got["mass_tot"] > 0 and (amp == 0.0 or got["KE_tot"] > 0) and got["PE_tot"] != 0.0
    dyc.close()


def test_write_energy_without_APE_and_errors(orc):
    from mom6_amd.dycore import Dycore
    gg, d, M = H.double_gyre()
    GV = abi.vgrid_default()
    g_prime = np.array([9.8, 0.02])
    dyc = Dycore(d, M, GV)
    h, u, v = synth.make_state(d, M)
    with pytest.raises(RuntimeError, match="Module must be initialized"):
        dyc.write_energy(dyc.to_dev(u), dyc.to_dev(v), dyc.to_dev(h))
    P = abi.sum_output_params_default(1200.0, do_APE_calc=0)
    dyc.sum_output_init(P, g_prime)
    got = dyc.write_energy(dyc.to_dev(u), dyc.to_dev(v), dyc.to_dev(h))
    same(got, orc.write_energy(orc.SumOutputState(d, M, GV, g_prime, P), u, v, h))
    assert got["PE_tot"] == 0.0 and not got["Z_0APE"].any()
    h[1, d.joff + 5, d.ioff + 5] = np.nan
    with pytest.raises(RuntimeError, match="NaN in input field of reproducing_sum"):
        dyc.write_energy(dyc.to_dev(u), dyc.to_dev(v), dyc.to_dev(h))
    dyc.close()


def test_ocean_stats_of_a_run(orc, tmp_path):
    """Three baroclinic steps on the device with a line of ocean.stats after each: the file made from the device's sums
    equals the one made from the oracle's sums of the oracle's own run (BT_STRONG_DRAG: the runs are bit-identical),
    which is how the reference's regression tests compare two runs (.testing/Makefile:449-454)."""
    from mom6_amd.dycore import Dycore
    from tests import cases
    cfg = H.double_gyre()
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg, False, False)
    GV, Rlay, gp, dt = inp["GV"], inp["Rlay"], inp["gp"], inp["dt"]
    bt_mod = dict(strong_drag=1)
    P = abi.sum_output_params_default(dt)
    ref_lines = cases.oracle_ocean_stats(orc, cfg, 3, bt_mod)     # the oracle's run, a line after every step
    # device run
    cont2, bt2, cor2, pgf2, rk22 = cases.rk2_params(d, GV, bt_mod, None, None)
    dyc = Dycore(d, M, GV, 0)
    dyc.continuity_init(cont2); dyc.barotropic_init(bt2); dyc.CoriolisAdv_init(cor2); dyc.PressureForce_init(pgf2, Rlay, gp)
    dyc.initialize_dyn_split_RK2(rk22)
    dyc.sum_output_init(P, gp)
    sg = dict(u=dyc.to_dev(inp["u"]), v=dyc.to_dev(inp["v"]), h=dyc.to_dev(inp["h"]), uh=dyc.zeros3(), vh=dyc.zeros3(),
              uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
    dyc.vertvisc_set_coef(*[dyc.to_dev(a) if a is not None else None for a in inp["coefs"][0]])
    txd, tyd = dyc.to_dev(inp["taux"]), dyc.to_dev(inp["tauy"])
    dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)
    so_dev = SO.SumOutput()
    out0, _ = so_dev.record(dyc.write_energy(sg["u"], sg["v"], sg["h"]), 0.0, 0)
    for n in range(3):
        dyc.step_MOM_dyn_split_RK2(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], sg["uhtr"], sg["vhtr"], sg["eta_av"], txd, tyd,
                                   dt, calc_dtbt=(n == 0))
        so_dev.record(dyc.write_energy(sg["u"], sg["v"], sg["h"]), dt * (n + 1), n + 1)
    assert so_dev.lines == ref_lines and len(so_dev.lines) == 6
    import os
    golden = open(H.golden_path("ocean.stats.double_gyre_strong_drag_3steps" + H.golden_tag(), "")).read().splitlines()
    assert so_dev.lines == golden                                # the committed fixture (scripts/make_golden.py)
    assert out0.startswith("MOM Day       0.000      0: En ")
    # the volume-conserving continuity solver: the mass column and the fractional mass error stay put
    errs = [float(l.split("Me")[1]) for l in so_dev.lines[2:]]
    assert max(abs(e) for e in errs) < 1e-14
    so_dev.write(tmp_path / "ocean.stats")
    assert (tmp_path / "ocean.stats").read_text().splitlines() == ref_lines
    dyc.close()
