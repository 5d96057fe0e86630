"""The density-following coordinate generators of the oracle (oracle/orc_remap.c: REGRIDDING_RHO, REGRIDDING_HYCOM1, the
interpolation machinery of regrid_interp.F90, convective_adjustment, adjust_interface_motion) -- the reference holds no numbers
for them (their files need MOM_error_handler / the EOS module tree), so they are held to what the algorithm must do on
columns whose answer is known in closed form."""
import numpy as np
import pytest

from mom6_amd import abi, grid, synth
from tests import helpers as H

G = abi.G


def flat(nk, depth=4000.0, ni=8, nj=6):
    gg = grid.GlobalGrid(ni, nj, kind="cartesian", dx=1.0e4, dy=1.0e4, f0=1e-4, beta=0.0, depth_fn=grid.flat_depth(ni, nj, depth))
    d, M = gg.tile(nk)
    return gg, d, M


def linear_eos():
    e = abi.eos_params_default(abi.LINEAR)
    e.Rho_T0_S0 = 1000.0; e.dRho_dT = -0.2; e.dRho_dS = 0.8; e.dRho_dp = 0.0
    return e


def linear_column(d, M, nk, depth, rho_top=1022.0, rho_bot=1028.0):
    """uniform layers; T such that the (LINEAR, S = 35) density at layer centres is linear in depth"""
    h = np.zeros((nk,) + d.shape2()); h[:] = depth / nk
    zc = (np.arange(nk) + 0.5) * depth / nk
    rho = rho_top + (rho_bot - rho_top) * zc / depth
    S = np.full_like(h, 35.0)
    T = np.zeros_like(h); T[:] = ((1000.0 + 0.8 * 35.0 - rho) / 0.2)[:, None, None]
    return h, T, S, rho


def wet_point(d, M):
    x = (d.joff + 2, d.ioff + 3)
    assert M[G["mask2dT"]][x] > 0
    return x


@pytest.mark.parametrize("scheme", [abi.INTERP_P1M_H2, abi.INTERP_P1M_H4, abi.INTERP_PLM, abi.INTERP_PPM_H4])
def test_rho_coordinate_on_a_linear_density_profile(orc, scheme):
    """(a) A column that already sits on its target interface densities does not move (1e-9 m).  (b) Other targets inside
    the profile: the new interfaces are where the linear profile takes those values, z = (target - rho_0) / N, to 1e-9 m in
    4000 m (the two end cells are not linear to their outer edge without BOUNDARY_EXTRAPOLATION: targets inside
    cells 2 .. nk-1).  (c) the column's thickness is kept, surface and bottom do not move."""
    nk, depth = 12, 4000.0
    gg, d, M = flat(nk, depth)
    GV = abi.vgrid_default()
    eos = linear_eos()
    h, T, S, rho = linear_column(d, M, nk, depth)
    x = wet_point(d, M)
    N = (rho[-1] - rho[0]) / (depth - depth / nk)            # d rho / dz of the profile
    rho0 = rho[0] - N * 0.5 * depth / nk                     # its value at the surface
    CS = abi.regrid_rho_params_default(interp_scheme=scheme, ref_pressure=2.0e7)
    # (a)
    tgt = np.concatenate(([rho0 - 5.0], 0.5 * (rho[:-1] + rho[1:]), [rho[-1] + 5.0]))
    hn = np.zeros_like(h); dz = np.zeros((nk + 1,) + d.shape2())
    orc.ALE_regrid_rho(d, M, GV, CS, eos, tgt, h, T, S, hn, dz)
    col = (slice(None),) + x
    assert np.abs(dz[col]).max() <= 1e-9 and np.abs(hn[col] - h[col]).max() <= 1e-9   # (the edge values are exact to round-off)
    # (b)
    zt = np.linspace(0.0, depth, nk + 1)
    zt[1:-1] = np.linspace(1.6, nk - 1.6, nk - 1) * depth / nk      # inside cells 2 .. nk-1
    tgt = rho0 + N * zt
    orc.ALE_regrid_rho(d, M, GV, CS, eos, tgt, h, T, S, hn, dz)
    z_new = np.concatenate(([0.0], np.cumsum(hn[col])))
    assert np.abs(z_new - zt).max() <= 1e-9, (scheme, np.abs(z_new - zt).max())
    # (c)
    assert dz[col][0] == 0.0 and abs(dz[col][-1]) <= 1e-9 and abs(hn[col].sum() - depth) <= 1e-9
    # land stays as it was
    reg = np.zeros(d.shape2(), dtype=bool); reg[tuple(H.interior(d, "h", 1))] = True
    land = np.argwhere((M[G["mask2dT"]] == 0) & reg)
    if len(land):
        yl = tuple(land[0])
        assert np.array_equal(hn[(slice(None),) + yl], h[(slice(None),) + yl]) and np.abs(dz[(slice(None),) + yl]).max() == 0.0


def test_rho_coordinate_vanished_layers_and_out_of_range_targets(orc):
    """Targets lighter than the whole column collapse onto the surface, heavier ones onto the bottom: the layers between them
    come out with MIN_THICKNESS (old_inflate_layers_1d), the thickest layer pays for it, the column is conserved; layers
    thinner than MIN_THICKNESS in the source do not take part in the density profile (copy_finite_thicknesses)."""
    nk, depth = 10, 3000.0
    gg, d, M = flat(nk, depth)
    GV = abi.vgrid_default()
    eos = linear_eos()
    h, T, S, rho = linear_column(d, M, nk, depth)
    x = wet_point(d, M); col = (slice(None),) + x
    CS = abi.regrid_rho_params_default(min_thickness=0.5)
    tgt = np.array([1000.0, 1001.0, 1002.0] + list(np.linspace(rho[2], rho[-3], nk - 5)) + [1040.0, 1041.0, 1042.0])
    hn = np.zeros_like(h); dz = np.zeros((nk + 1,) + d.shape2())
    orc.ALE_regrid_rho(d, M, GV, CS, eos, tgt, h, T, S, hn, dz)
    a = hn[col]
    assert abs(a.sum() - depth) <= 1e-9 and a.min() >= 0.5 * (1 - 1e-12)
    assert np.allclose(a[:2], 0.5, rtol=0, atol=1e-12) and np.allclose(a[-2:], 0.5, rtol=0, atol=1e-12)
    # a vanished source layer with a wild temperature changes nothing
    h2, T2 = h.copy(), T.copy()
    h2[4] = 1.0e-4; h2[5] += h[4] - 1.0e-4; T2[4] = 40.0
    hn2 = np.zeros_like(h); dz2 = np.zeros_like(dz)
    orc.ALE_regrid_rho(d, M, GV, CS, eos, tgt, h2, T2, S, hn2, dz2)
    assert np.isfinite(hn2[col]).all() and abs(hn2[col].sum() - depth) <= 1e-9 and hn2[col].min() >= 0.5 * (1 - 1e-12)


def test_hycom1_is_zstar_where_the_targets_are_out_of_reach_and_isopycnal_below(orc):
    """HYCOM1 = max(depth of the target density, nominal z* depth) (Bleck 2002).  (a) Targets lighter than the surface water:
    every interface falls back to the z* grid -- the thicknesses of ALE_regrid_zstar with the same resolution.  (b) A
    profile whose upper targets are too light and whose lower ones lie deeper than z*: z* above, the positions of the target
    densities below.  (c) MAXIMUM_INT_DEPTH_CONFIG and MAX_LAYER_THICKNESS_CONFIG cap the interfaces.  The column is kept."""
    nk, depth = 10, 4000.0
    gg, d, M = flat(nk, depth)
    GV = abi.vgrid_default()
    eos = linear_eos()
    h, T, S, rho = linear_column(d, M, nk, depth)
    h[0] += 1.5      # a sea surface 1.5 m above the resting level: z* stretches
    x = wet_point(d, M); col = (slice(None),) + x
    cr = np.array([10.0, 20.0, 40.0, 80.0, 150.0, 300.0, 500.0, 700.0, 1000.0, 1200.0])
    CS = abi.regrid_rho_params_default(min_thickness=1.0e-3)
    hn = np.zeros_like(h); dz = np.zeros((nk + 1,) + d.shape2())
    # (a)
    orc.ALE_regrid_hycom1(d, M, GV, CS, eos, cr, np.linspace(990.0, 999.0, nk + 1), None, None, h, T, S, hn, dz)
    hz = np.zeros_like(h); dzz = np.zeros_like(dz)
    orc.ALE_regrid_zstar(d, M, GV, CS.f, cr, h, hz, dzz)
    assert np.abs(hn[col] - hz[col]).max() <= 1e-9 and abs(hn[col].sum() - (depth + 1.5)) <= 1e-9
    # (b)
    tot = depth + 1.5
    zi = np.concatenate(([0.0], np.cumsum(h[col])))
    N = (rho[-1] - rho[0]) / (0.5 * (zi[-2] + zi[-1]) - 0.5 * (zi[0] + zi[1]))
    zstar = np.concatenate(([0.0], np.cumsum(cr))) * (tot / depth)
    z_want = zstar.copy(); z_want[6:10] = [1500.0, 2200.0, 2900.0, 3500.0]       # deeper than z* there (1101.., 1601.., 2301.., 3301..)
    tgt = np.full(nk + 1, 990.0)
    zc = 0.5 * (zi[:-1] + zi[1:])
    tgt[6:10] = np.interp(z_want[6:10], zc, rho)
    tgt[10] = 1050.0
    orc.ALE_regrid_hycom1(d, M, GV, CS, eos, cr, tgt, None, None, h, T, S, hn, dz)
    z_new = np.concatenate(([0.0], np.cumsum(hn[col])))
    assert np.abs(z_new[:6] - zstar[:6]).max() <= 1e-9
    assert np.abs(z_new[6:10] - z_want[6:10]).max() <= 2e-2     # (the layer-mean profile of P1M_H2 is the interpolant: centimetres)
    assert abs(z_new[-1] - tot) <= 1e-9
    # (c)
    mid = np.concatenate(([0.0], np.cumsum(cr) * 1.2)); mid[7] = 1800.0
    mlt = np.full(nk, 650.0)
    orc.ALE_regrid_hycom1(d, M, GV, CS, eos, cr, tgt, mid, mlt, h, T, S, hn, dz)
    z_cap = np.concatenate(([0.0], np.cumsum(hn[col])))
    assert z_cap[7] <= 1800.0 + 1e-9 and (np.diff(z_cap)[:8] <= 650.0 + 1e-9).all() and abs(z_cap[-1] - tot) <= 1e-9
    assert not np.allclose(z_cap, z_new)


@pytest.mark.parametrize("scheme", [abi.INTERP_P1M_H2, abi.INTERP_P1M_H4, abi.INTERP_PLM, abi.INTERP_PPM_H4])
@pytest.mark.parametrize("form", [abi.LINEAR, abi.WRIGHT])
def test_density_coordinates_on_random_stacks_over_a_bowl(orc, scheme, form):
    """Random stratified stacks over the bowl (vanished layers at the rim): both generators keep each column's thickness to
    round-off, leave surface and bottom where they are, return no negative thickness, honour MIN_THICKNESS (HYCOM1, through
    adjust_interface_motion), and the time filter moves the interfaces less far than the unfiltered regridding."""
    gg, d, M = H.benchmark_small(nk=10)
    GV = abi.vgrid_default()
    h, _, _ = synth.make_state(d, M, thin_frac=0.2)
    from tests import cases
    T, S = cases.thermo_state(d, M)
    eos = abi.eos_params_default(form)
    sl = H.interior(d, "h", 1)
    wet = M[G["mask2dT"]][tuple(sl)] > 0
    p_ref = 2.0e7
    rr = np.sort(np.array([orc.eos_density(eos, float(t), float(s), p_ref) for t, s in zip(T[:, d.joff + 8, d.ioff + 12], S[:, d.joff + 8, d.ioff + 12])]))
    tgt = np.concatenate(([rr[0] - 1.0], 0.5 * (rr[:-1] + rr[1:]), [rr[-1] + 1.0]))
    cr = np.linspace(50., 800., d.nk); cr *= 4000. / cr.sum()
    for which in ("rho", "hycom1"):
        moved = []
        for CS in (abi.regrid_rho_params_default(interp_scheme=scheme, min_thickness=1.0e-3),
                   abi.regrid_rho_params_default(interp_scheme=scheme, min_thickness=1.0e-3, old_grid_weight=0.5,
                                                 depth_of_time_filter_shallow=100., depth_of_time_filter_deep=600.)):
            hn = np.zeros_like(h); dz = np.zeros((d.nk + 1,) + d.shape2())
            hh, TT, SS = h.copy(), T.copy(), S.copy()
            if which == "rho":
                orc.ALE_convective_adjustment(d, eos, hh, TT, SS)
                orc.ALE_regrid_rho(d, M, GV, CS, eos, tgt, hh, TT, SS, hn, dz)
            else:
                orc.ALE_regrid_hycom1(d, M, GV, CS, eos, cr, tgt, None, None, hh, TT, SS, hn, dz)
            a, b = hn[(Ellipsis,) + tuple(sl)][:, wet], hh[(Ellipsis,) + tuple(sl)][:, wet]
            assert np.isfinite(a).all() and a.min() >= 0.0
            assert np.abs(a.sum(0) - b.sum(0)).max() <= 1e-11 * b.sum(0).max(), (which, np.abs(a.sum(0) - b.sum(0)).max())
            assert np.abs(dz[0][tuple(sl)]).max() == 0.0 and np.abs(dz[-1][tuple(sl)][wet]).max() <= 1e-8
            if which == "hycom1" and CS.f.old_grid_weight == 0.0:
                deep = b.sum(0) > d.nk * 1.0e-3
                assert a[1:, deep].min() >= 1.0e-3 * (1 - 1e-9)
            moved.append(np.abs(dz[(Ellipsis,) + tuple(sl)][:, wet]).sum())
        assert moved[0] > 0 and moved[1] < moved[0]


def test_convective_adjustment_sorts_the_column(orc):
    """convective_adjustment :1905: afterwards the density at the surface pressure does not decrease downward, and every column
    holds the same (h, T, S) triples as before, reordered."""
    gg, d, M = H.benchmark_small(nk=9)
    rng = np.random.default_rng(5)
    shp = (d.nk,) + d.shape2()
    h = rng.uniform(1.0, 100.0, shp); T = rng.uniform(0.0, 25.0, shp); S = rng.uniform(33.0, 36.0, shp)
    for form in (abi.LINEAR, abi.WRIGHT):
        eos = abi.eos_params_default(form)
        hh, TT, SS = h.copy(), T.copy(), S.copy()
        orc.ALE_convective_adjustment(d, eos, hh, TT, SS)
        sl = H.interior(d, "h", 1)
        for (jj, ii) in ((d.joff + 3, d.ioff + 4), (d.joff - 1, d.ioff + 7), (d.joff + d.nj, d.ioff + d.ni)):
            r = np.array([orc.eos_density(eos, float(t), float(s), 0.0) for t, s in zip(TT[:, jj, ii], SS[:, jj, ii])])
            assert (np.diff(r) >= 0).all()
            before = sorted(zip(h[:, jj, ii], T[:, jj, ii], S[:, jj, ii])); after = sorted(zip(hh[:, jj, ii], TT[:, jj, ii], SS[:, jj, ii]))
            assert before == after
        assert not np.array_equal(TT[(Ellipsis,) + tuple(sl)], T[(Ellipsis,) + tuple(sl)])
