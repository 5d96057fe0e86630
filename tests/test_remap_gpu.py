"""MOM_remapping / ALE remapping on the device: the reference's own known answers (remapping_unit_tests) through the
C ABI, and bit-for-bit parity with the oracle (which tests/test_remap_cpu.py pins to the same known answers)."""
import numpy as np
import pytest

from mom6_amd import abi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = abi.G
H_NEGLECT = 1.0e-30


def dev_core_h(dyc, CS, h0, u0, h1):
    import torch
    t = lambda a: torch.tensor(np.atleast_2d(np.asarray(a, dtype=np.float64)), device=dyc.device)
    h0d, u0d, h1d = t(h0), t(u0), t(h1)
    u1d = torch.zeros_like(h1d)
    torch.cuda.synchronize()
    dyc.remapping_core_h(CS, h0d, u0d, h1d, u1d)
    dyc.sync()
    return u1d.cpu().numpy()


@pytest.fixture(scope="module")
def dyc():
    from mom6_amd.dycore import Dycore
    gg, d, M = H.double_gyre()
    x = Dycore(d, M, abi.vgrid_default())
    yield x
    x.close()


def test_reference_known_answers_on_the_device(dyc):
    """remapping_unit_tests :2125-2166 (PPM_H4 'remapping_core_h() 2/3/4') and :2484-2502 (PLM h=0110) through
    mom6x_remapping_core_h -- the expected values are the reference's."""
    CS = abi.remapping_params_default(abi.REMAP_PPM_H4, H_NEGLECT, answer_date=20190101)
    h0, u0 = [0.75] * 4, [9., 3., -3., -9.]
    assert np.array_equal(dev_core_h(dyc, CS, h0, u0, [0.5] * 6)[0], [10., 6., 2., -2., -6., -10.])
    assert np.array_equal(dev_core_h(dyc, CS, h0, u0, [.125] * 6)[0], [11.5, 10.5, 9.5, 8.5, 7.5, 6.5])
    assert np.array_equal(dev_core_h(dyc, CS, h0, u0, [2.25, 1.5, 1.])[0], [3., -10.5, -12.])
    for om4 in (1, 0):
        CS = abi.remapping_params_default(abi.REMAP_PLM, H_NEGLECT, answer_date=20190101, om4_remap_via_sub_cells=om4)
        assert np.array_equal(dev_core_h(dyc, CS, [0., 1., 1., 0.], [5., 4., 2., 1.], [1., 1.])[0], [4., 2.])
        assert np.array_equal(dev_core_h(dyc, CS, [0., 1., 1., 0.], [5., 4., 2., 1.], [1., 4.])[0], [4., 1.25])
    # sub-grid tests 3, 5, 6 (:2330-2478): PLM with boundary extrapolation, final target values
    CS = abi.remapping_params_default(abi.REMAP_PLM, H_NEGLECT, answer_date=20190101, force_bounds_in_target=0)
    assert np.array_equal(dev_core_h(dyc, CS, [2., 4.], [2., 5.], [2., 2., 2.])[0], [2., 4., 6.])
    assert np.array_equal(dev_core_h(dyc, CS, [2., 2., 1.], [2., 4., 5.5], [2., 4.])[0], [2., 4.875])
    assert np.array_equal(dev_core_h(dyc, CS, [2., 0., 2.], [2., 3., 4.], [1., 0., 1., 0., 2.])[0], [1.5, 2., 2.5, 3., 4.])


@pytest.mark.parametrize("scheme", [abi.REMAP_PCM, abi.REMAP_PLM, abi.REMAP_PPM_H4, abi.REMAP_PPM_IH4])
@pytest.mark.parametrize("mods", [dict(), dict(om4_remap_via_sub_cells=1), dict(boundary_extrapolation=0, force_bounds_in_subcell=1),
                                  dict(om4_remap_via_sub_cells=1, force_bounds_in_target=0, boundary_extrapolation=0)])
def test_random_columns_bitwise(orc, dyc, scheme, mods):
    """Ragged columns as the reference's brute-force tests draw them: vanished layers in both grids, target columns
    shallower and deeper than the source, n0 != n1 from 1 to 40 layers."""
    rng = np.random.default_rng(7 + scheme)
    CS = abi.remapping_params_default(scheme, H_NEGLECT, **mods)
    for n0, n1 in ((1, 3), (2, 2), (3, 7), (4, 4), (5, 2), (9, 12), (25, 40), (40, 17), (75, 75)):
        ncol = 193
        h0 = rng.random((ncol, n0)); h0[rng.random((ncol, n0)) < 0.15] = 0.0
        h1 = rng.random((ncol, n1)); h1[rng.random((ncol, n1)) < 0.15] = 0.0
        h0[:, 0] += 1e-3; h1[:, 0] += 1e-3
        scale = h0.sum(1) / h1.sum(1)
        h1 *= scale[:, None] * np.where(np.arange(ncol) % 3 == 0, 1.0, np.where(np.arange(ncol) % 3 == 1, 0.8, 1.3))[:, None]
        h1[::7] = h0[::7, :1] * 0 + h1[::7]          # (no-op; keeps the generator's stream aligned)
        if n0 == n1:
            h1[::5] = h0[::5]                          # unchanged grids
        u0 = rng.random((ncol, n0)) * 20 - 5
        u0[::11] = 3.25                                # uniform columns
        ref = orc.remapping_core_h_cols(CS, h0, u0, h1)
        got = dev_core_h(dyc, CS, h0, u0, h1)
        H.assert_bitwise(got, ref, f"remapping_core_h n0={n0} n1={n1}")


@pytest.mark.parametrize("cfg", ["island_basin", "benchmark_small"])
@pytest.mark.parametrize("scheme,mods", [(abi.REMAP_PPM_H4, dict(om4_remap_via_sub_cells=1, boundary_extrapolation=0)),
                                         (abi.REMAP_PPM_IH4, dict(boundary_extrapolation=0)),   # .testing/tc2, tc4
                                         (abi.REMAP_PLM, dict()), (abi.REMAP_PCM, dict())])
def test_ALE_remap_tracers_and_velocities(orc, cfg, scheme, mods):
    """The 3-D entry points on a basin with land: two tracers remapped in place from the model's layers to a z*-like
    grid with the same column thickness, then the velocities with ALE_remap_set_h_vel's face thicknesses."""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = getattr(H, cfg)(nk=12)
    GV = abi.vgrid_default()
    CS = abi.remapping_params_default(scheme, GV.H_subroundoff, **mods)
    h_old, u, v = synth.make_state(d, M, thin_frac=0.15)
    tot = h_old.sum(0)
    w = np.linspace(1.0, 3.0, d.nk)[:, None, None] * (1.0 + 0.3 * synth.smooth_field(d, 5, nk=d.nk, ox=0.5, oy=0.5))
    h_new = np.ascontiguousarray(w / w.sum(0) * tot)
    trs = [np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 70 + m, nk=d.nk, ox=0.5, oy=0.5)) for m in range(2)]
    tro = [t.copy() for t in trs]
    orc.ALE_remap_tracers(d, M, CS, h_old, h_new, tro)
    hu_o, hv_o, hu_n, hv_n = (np.full_like(h_old, 1.0e-3) for _ in range(4))
    orc.ALE_remap_set_h_vel(d, M, h_old, hu_o, hv_o); orc.ALE_remap_set_h_vel(d, M, h_new, hu_n, hv_n)
    uo, vo = u.copy(), v.copy()
    orc.ALE_remap_velocities(d, M, CS, hu_o, hv_o, hu_n, hv_n, uo, vo)

    dyc = Dycore(d, M, GV)
    hod, hnd = dyc.to_dev(h_old), dyc.to_dev(h_new)
    trg = [dyc.to_dev(t) for t in trs]
    ud, vd = dyc.to_dev(u), dyc.to_dev(v)
    g = [torch.full_like(hod, 1.0e-3) for _ in range(4)]
    torch.cuda.synchronize()
    dyc.ALE_remap_tracers(CS, hod, hnd, trg)
    dyc.ALE_remap_set_h_vel(hod, g[0], g[1]); dyc.ALE_remap_set_h_vel(hnd, g[2], g[3])
    dyc.ALE_remap_velocities(CS, g[0], g[1], g[2], g[3], ud, vd)
    dyc.sync()
    sl = H.interior(d, "h")
    for m in range(2):
        H.assert_bitwise(trg[m].cpu().numpy(), tro[m], f"tracer {m}", sl)
        wet = M[G["mask2dT"]][tuple(sl)] > 0
        a, b = tro[m][(Ellipsis,) + tuple(sl)][:, wet], trs[m][(Ellipsis,) + tuple(sl)][:, wet]
        assert np.abs(a - b).max() > 1e-3
        # the column integral of every wet column is conserved to round-off
        ho, hn = h_old[(Ellipsis,) + tuple(sl)][:, wet], h_new[(Ellipsis,) + tuple(sl)][:, wet]
        assert np.abs((a * hn).sum(0) - (b * ho).sum(0)).max() <= 1e-12 * np.abs(b * ho).sum(0).max()
    H.assert_bitwise(g[0].cpu().numpy(), hu_o, "h_u", H.interior(d, "u")); H.assert_bitwise(g[3].cpu().numpy(), hv_n, "h_v", H.interior(d, "v"))
    H.assert_bitwise(ud.cpu().numpy(), uo, "u", H.interior(d, "u")); H.assert_bitwise(vd.cpu().numpy(), vo, "v", H.interior(d, "v"))
    # the three calls as one, from the cells' thicknesses (OM4's switch set: h_u / h_v formed where they are read, never stored;
    # the other switch sets: made in work space): the same bits
    u2, v2 = dyc.to_dev(u), dyc.to_dev(v)
    torch.cuda.synchronize()
    dyc.ALE_remap_velocities_from_h(CS, hod, hnd, u2, v2)
    dyc.sync()
    H.assert_bitwise(u2.cpu().numpy(), uo, "u (from h)", H.interior(d, "u")); H.assert_bitwise(v2.cpu().numpy(), vo, "v (from h)", H.interior(d, "v"))
    dyc.close()


@pytest.mark.parametrize("cfg", ["double_gyre", "island_basin"])
def test_ALE_remap_velocities_conserving_ke(orc, cfg):
    """REMAP_VEL_CONSERVE_KE with allow_preserve_variance (MOM_ALE.F90:1166-1195, :1240-1270): the baroclinic part of every remapped
    velocity column rescaled so that its integrated square equals the source column's (by at most 25 %).  Device == oracle bit for
    bit; and the property itself on the device's result: where the cap is not hit, sum h2 (u - u_bt)^2 equals the source's."""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = getattr(H, cfg)(nk=12)
    GV = abi.vgrid_default()
    CS = abi.remapping_params_default(abi.REMAP_PPM_H4, GV.H_subroundoff, om4_remap_via_sub_cells=1)
    h_old, u, v = synth.make_state(d, M, thin_frac=0.15)
    tot = h_old.sum(0)
    w = np.linspace(1.0, 3.0, d.nk)[:, None, None] * (1.0 + 0.3 * synth.smooth_field(d, 5, nk=d.nk, ox=0.5, oy=0.5))
    h_new = np.ascontiguousarray(w / w.sum(0) * tot)
    hu_o, hv_o, hu_n, hv_n = (np.full_like(h_old, 1.0e-3) for _ in range(4))
    orc.ALE_remap_set_h_vel(d, M, h_old, hu_o, hv_o); orc.ALE_remap_set_h_vel(d, M, h_new, hu_n, hv_n)
    uo, vo = u.copy(), v.copy()
    orc.ALE_remap_velocities_conserve_ke(d, M, GV, CS, hu_o, hv_o, hu_n, hv_n, uo, vo)
    up, vp = u.copy(), v.copy()
    orc.ALE_remap_velocities(d, M, CS, hu_o, hv_o, hu_n, hv_n, up, vp)
    su = (Ellipsis,) + tuple(H.interior(d, "u"))
    assert np.abs(uo[su] - up[su]).max() > 1e-6          # the correction does something
    dyc = Dycore(d, M, GV)
    ud, vd = dyc.to_dev(u), dyc.to_dev(v)
    g = [dyc.to_dev(a) for a in (hu_o, hv_o, hu_n, hv_n)]
    torch.cuda.synchronize()
    dyc.ALE_remap_velocities(CS, g[0], g[1], g[2], g[3], ud, vd, conserve_ke=True)
    dyc.sync()
    H.assert_bitwise(ud.cpu().numpy(), uo, "u (KE-conserving)", H.interior(d, "u"))
    H.assert_bitwise(vd.cpu().numpy(), vo, "v (KE-conserving)", H.interior(d, "v"))
    # the property: baroclinic KE of the column conserved wherever the 25 % cap was not hit
    un = ud.cpu().numpy()[su]; h1 = hu_o[su]; h2 = hu_n[su]; us = u[su]
    wet = M[G["mask2dCu"]][su[1:]] > 0
    ubt = (h2 * un).sum(0) / (h2.sum(0) + GV.H_subroundoff)
    ke_s = (h1 * (us - ubt) ** 2).sum(0); ke_t = (h2 * (un - ubt) ** 2).sum(0)
    ke_p = (h2 * (up[su] - ubt) ** 2).sum(0)
    free = wet & (ke_s < 1.5625 * ke_p) & (ke_s > 0)
    assert free.sum() > 10
    assert (np.abs(ke_t - ke_s)[free] / ke_s[free]).max() < 1e-12
    dyc.close()


@pytest.mark.parametrize("nk", [5, 24, 75])
@pytest.mark.parametrize("scheme", [abi.REMAP_PPM_H4, abi.REMAP_PPM_IH4])
def test_ALE_remap_ragged_grids_through_the_shared_merge(orc, nk, scheme):
    """k_remap_merge (the OM4 switch set: one merge for the fields of a grid pair, target column through an LDS window, a
    cell's sub-cells kept as weights) on what it has special paths for: vanished layers in both grids, columns whose target
    index runs far ahead of or behind the source index (outside the window), source cells holding more sub-cells than the
    register buffer, target columns shallower and deeper than the source, unchanged grids; three tracers = one pair and a
    single field; then u and v.  Bit for bit against the oracle (which keeps the reference's arrays)."""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = H.island_basin(nk=nk)
    GV = abi.vgrid_default()
    CS = abi.remapping_params_default(scheme, GV.H_subroundoff, om4_remap_via_sub_cells=1, boundary_extrapolation=0)
    rng = np.random.default_rng(100 + nk)
    shp = (nk,) + d.shape2()
    col = np.arange(shp[1] * shp[2]).reshape(shp[1:])

    def grid(seed_shift):
        h = rng.random(shp)
        h[rng.random(shp) < 0.2] = 0.0
        h[0] += 1.0e-3
        return h
    h_old, h_new = grid(0), grid(1)
    # a third of the columns: the old grid's mass in the top few layers, the new one's in the bottom few (and the reverse)
    top = (np.arange(nk) < max(1, nk // 6))[:, None, None]
    h_old = np.where((col % 6 == 1)[None], np.where(top, h_old + 0.5, h_old * 1.0e-3), h_old)
    h_new = np.where((col % 6 == 1)[None], np.where(top[::-1], h_new + 0.5, h_new * 1.0e-3), h_new)
    h_old = np.where((col % 6 == 4)[None], np.where(top[::-1], h_old + 0.5, h_old * 1.0e-3), h_old)
    h_new = np.where((col % 6 == 4)[None], np.where(top, h_new + 0.5, h_new * 1.0e-3), h_new)
    scale = h_old.sum(0) / h_new.sum(0)
    h_new = h_new * scale[None] * np.where(col % 3 == 0, 1.0, np.where(col % 3 == 1, 0.8, 1.3))[None]
    h_new = np.where((col % 5 == 2)[None], h_old, h_new)                 # unchanged grids
    h_old, h_new = np.ascontiguousarray(h_old), np.ascontiguousarray(h_new)
    trs = [np.ascontiguousarray(rng.random(shp) * 20 - 5) for _ in range(3)]
    trs[1][:, col % 11 == 0] = 3.25                                       # uniform columns
    tro = [t.copy() for t in trs]
    orc.ALE_remap_tracers(d, M, CS, h_old, h_new, tro)
    u, v = np.ascontiguousarray(rng.random(shp) - 0.5), np.ascontiguousarray(rng.random(shp) - 0.5)
    hu_o, hv_o, hu_n, hv_n = (np.full_like(h_old, 1.0e-3) for _ in range(4))
    orc.ALE_remap_set_h_vel(d, M, h_old, hu_o, hv_o); orc.ALE_remap_set_h_vel(d, M, h_new, hu_n, hv_n)
    uo, vo = u.copy(), v.copy()
    orc.ALE_remap_velocities(d, M, CS, hu_o, hv_o, hu_n, hv_n, uo, vo)

    dyc = Dycore(d, M, GV)
    hod, hnd = dyc.to_dev(h_old), dyc.to_dev(h_new)
    trg = [dyc.to_dev(t) for t in trs]
    ud, vd = dyc.to_dev(u), dyc.to_dev(v)
    g = [dyc.to_dev(a) for a in (hu_o, hv_o, hu_n, hv_n)]
    torch.cuda.synchronize()
    dyc.ALE_remap_tracers(CS, hod, hnd, trg)
    dyc.ALE_remap_velocities(CS, g[0], g[1], g[2], g[3], ud, vd)
    dyc.sync()
    sl = H.interior(d, "h")
    for m in range(3):
        H.assert_bitwise(trg[m].cpu().numpy(), tro[m], f"tracer {m}", sl)
    H.assert_bitwise(ud.cpu().numpy(), uo, "u", H.interior(d, "u")); H.assert_bitwise(vd.cpu().numpy(), vo, "v", H.interior(d, "v"))
    wet = M[G["mask2dT"]][tuple(sl)] > 0
    assert np.abs(tro[0] - trs[0])[(Ellipsis,) + tuple(sl)][:, wet].max() > 1.0
    dyc.close()


def test_rejects_what_is_not_on_the_path(dyc):
    CS = abi.remapping_params_default(abi.REMAP_PPM_H4, H_NEGLECT, answer_date=20181231)
    with pytest.raises(RuntimeError, match="20190101"):
        dev_core_h(dyc, CS, [1., 1., 1., 1.], [1., 2., 3., 4.], [2., 2.])
    CS = abi.remapping_params_default(9, H_NEGLECT)       # PQM_IH6IH5
    with pytest.raises(RuntimeError, match="remapping method is invalid"):
        dev_core_h(dyc, CS, [1., 1., 1., 1.], [1., 2., 3., 4.], [2., 2.])


@pytest.mark.parametrize("cfg", ["island_basin", "benchmark_small", "channel", "benchmark_75"])
@pytest.mark.parametrize("mods", [dict(), dict(min_thickness=5.0),
                                  dict(old_grid_weight=0.4, depth_of_time_filter_shallow=200., depth_of_time_filter_deep=900.),
                                  dict(old_grid_weight=0.7)])
def test_ALE_regrid_zstar_then_remap(orc, cfg, mods):
    """A whole ALE step for the z* coordinate on the device: ALE_regrid (new grid and interface displacements), then the
    remapping of two tracers and the velocities onto it -- every array bit for bit against the oracle.  (benchmark_75: nk = 75 is
    the layer count of the on-chip column kernel k_regrid_zstar_cols.)"""
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = H.benchmark_small(nk=75, ni=70, nj=12) if cfg == "benchmark_75" else getattr(H, cfg)(nk=10)
    GV = abi.vgrid_default()
    h, u, v = synth.make_state(d, M, thin_frac=0.2)
    depth = float(M[G["bathyT"]].max())
    cr = np.linspace(1.0, 6.0, d.nk); cr *= depth / cr.sum()
    RP = abi.regrid_zstar_params_default(**mods)
    CS = abi.remapping_params_default(abi.REMAP_PPM_H4, GV.H_subroundoff, om4_remap_via_sub_cells=1, boundary_extrapolation=0)
    hn_o = np.zeros_like(h); dz_o = np.zeros((d.nk + 1,) + d.shape2())
    orc.ALE_regrid_zstar(d, M, GV, RP, cr, h, hn_o, dz_o)
    T = np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 70, nk=d.nk, ox=0.5, oy=0.5)); To = T.copy()
    orc.ALE_remap_tracers(d, M, CS, h, hn_o, [To])

    dyc = Dycore(d, M, GV)
    hd = dyc.to_dev(h)
    hn_g = torch.zeros_like(hd); dz_g = torch.zeros((d.nk + 1,) + d.shape2(), dtype=torch.float64, device=dyc.device)
    Tg = dyc.to_dev(T)
    torch.cuda.synchronize()
    dyc.ALE_regrid_zstar(RP, cr, hd, hn_g, dz_g)
    dyc.ALE_remap_tracers(CS, hd, hn_g, [Tg])
    dyc.sync()
    sl1 = H.interior(d, "h", 1)
    H.assert_bitwise(hn_g.cpu().numpy(), hn_o, "h_new", sl1)
    H.assert_bitwise(dz_g.cpu().numpy(), dz_o, "dzRegrid", sl1)
    H.assert_bitwise(Tg.cpu().numpy(), To, "T", H.interior(d, "h"))
    assert np.abs(dz_o).max() > 1.0 and np.abs(To - T).max() > 1e-3
    dyc.close()


@pytest.mark.parametrize("cfg", ["island_basin", "benchmark_small"])
@pytest.mark.parametrize("which", ["rho", "hycom1"])
@pytest.mark.parametrize("mods", [dict(), dict(interp_scheme=abi.INTERP_PLM, boundary_extrapolation=1),
                                  dict(interp_scheme=abi.INTERP_PPM_H4), dict(interp_scheme=abi.INTERP_PPM_H4, boundary_extrapolation=1),
                                  dict(boundary_extrapolation=1, min_thickness=2.0, old_grid_weight=0.5, depth_of_time_filter_shallow=100.,
                                       depth_of_time_filter_deep=700.),
                                  dict(integrate_downward_for_e=0, compressibility_fraction=0.3, ref_pressure=1.0e7)])
@pytest.mark.parametrize("form", [abi.LINEAR, abi.WRIGHT, abi.WRIGHT_FULL, abi.UNESCO, abi.ROQUET_RHO], ids=["LINEAR", "WRIGHT", "WRIGHT_FULL", "UNESCO", "ROQUET_RHO"])
def test_ALE_regrid_density_coordinates(orc, cfg, which, mods, form):
    """REGRIDDING_RHO (after convective_adjustment) and REGRIDDING_HYCOM1 on the device: the reordered column, the new
    thicknesses and the interface displacements bit for bit against the oracle -- every interpolation scheme on the path,
    with and without BOUNDARY_EXTRAPOLATION, the time filter, both EOS forms, HYCOM1's compressibility and caps."""
    import torch
    from mom6_amd.dycore import Dycore
    from tests import cases
    gg, d, M = getattr(H, cfg)(nk=10)
    GV = abi.vgrid_default()
    h, _, _ = synth.make_state(d, M, thin_frac=0.2)
    T, S = cases.thermo_state(d, M)
    rng = np.random.default_rng(3)
    T = np.ascontiguousarray(T + rng.normal(0.0, 2.0, T.shape))     # some static instability for convective_adjustment
    eos = abi.eos_params_default(form)
    CS = abi.regrid_rho_params_default(**mods)
    jj, ii = d.joff + d.nj // 2, d.ioff + d.ni // 2
    rr = np.sort(np.array([orc.eos_density(eos, float(t), float(s), CS.ref_pressure) for t, s in zip(T[:, jj, ii], S[:, jj, ii])]))
    tgt = np.concatenate(([rr[0] - 0.5], 0.5 * (rr[:-1] + rr[1:]), [rr[-1] + 0.5]))
    depth = float(M[G["bathyT"]].max())
    cr = np.linspace(1.0, 6.0, d.nk); cr *= depth / cr.sum()
    mid = np.concatenate(([0.0], np.cumsum(cr) * 1.5)) if "compressibility_fraction" in mods else None
    mlt = np.full(d.nk, 0.4 * depth) if "compressibility_fraction" in mods else None
    ho, To, So = h.copy(), T.copy(), S.copy()
    hn_o = np.zeros_like(h); dz_o = np.zeros((d.nk + 1,) + d.shape2())
    if which == "rho":
        orc.ALE_convective_adjustment(d, eos, ho, To, So)
        orc.ALE_regrid_rho(d, M, GV, CS, eos, tgt, ho, To, So, hn_o, dz_o)
    else:
        orc.ALE_regrid_hycom1(d, M, GV, CS, eos, cr, tgt, mid, mlt, ho, To, So, hn_o, dz_o)
    dyc = Dycore(d, M, GV)
    hd, Td, Sd = dyc.to_dev(h), dyc.to_dev(T), dyc.to_dev(S)
    hn_g = torch.zeros_like(hd); dz_g = torch.zeros((d.nk + 1,) + d.shape2(), dtype=torch.float64, device=dyc.device)
    torch.cuda.synchronize()
    if which == "rho":
        dyc.ALE_convective_adjustment(eos, hd, Td, Sd)
        dyc.ALE_regrid_rho(CS, eos, tgt, hd, Td, Sd, hn_g, dz_g)
    else:
        dyc.ALE_regrid_hycom1(CS, eos, cr, tgt, mid, mlt, hd, Td, Sd, hn_g, dz_g)
    dyc.sync()
    sl1 = H.interior(d, "h", 1)
    if which == "rho":
        for name, a, b in (("h", hd, ho), ("T", Td, To), ("S", Sd, So)):
            H.assert_bitwise(a.cpu().numpy(), b, name + " after convective_adjustment", sl1)
        assert not np.array_equal(To, T)
    H.assert_bitwise(hn_g.cpu().numpy(), hn_o, "h_new", sl1)
    H.assert_bitwise(dz_g.cpu().numpy(), dz_o, "dzRegrid", sl1)
    assert np.abs(dz_o).max() > 1.0
    dyc.close()


def test_ALE_regrid_density_rejects_what_is_not_carried():
    from mom6_amd.dycore import Dycore
    import torch
    gg, d, M = H.double_gyre(nk=4)
    dyc = Dycore(d, M, abi.vgrid_default())
    z = dyc.zeros3(); dz = torch.zeros((d.nk + 1,) + d.shape2(), dtype=torch.float64, device=dyc.device)
    with pytest.raises(RuntimeError, match="INTERPOLATION_SCHEME"):
        dyc.ALE_regrid_rho(abi.regrid_rho_params_default(interp_scheme=9), abi.eos_params_default(), np.linspace(1020., 1030., 5), z, z, z,
                           dyc.zeros3(), dz)
    dyc.close()
