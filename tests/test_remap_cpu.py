"""The oracle's restatement of MOM_remapping replayed against the reference's OWN known answers: every value below
is copied from remapping_unit_tests (src/ALE/MOM_remapping.F90:2072-2943, the tests that `MOM6 unit_tests` runs), as
data -- inputs and expected outputs -- with the line it comes from.  This pins orc_remap.c (PCM / PLM / PPM_H4,
answer date 20190101) to the reference's numbers; the GPU tests then hold the HIP kernels to orc_remap.c bit for bit."""
import numpy as np
import pytest

from mom6_amd import abi

H_NEGLECT = 1.0e-30       # :2121
A = np.array


def eq(a, b, tol=0.0, what=""):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, what
    assert np.abs(a - b).max() <= tol, f"{what}: {a} vs {b}"


def test_remapping_core_h_PPM_H4_first_generation(orc):
    """:2125-2166 'remapping_core_h() 2', '3', '4' (4 layers of 0.75 with du/dz = 8)."""
    CS = abi.remapping_params_default(abi.REMAP_PPM_H4, H_NEGLECT, answer_date=20190101)
    h0, u0 = [0.75] * 4, [9., 3., -3., -9.]
    eq(orc.remapping_core_h(CS, h0, u0, [0.5] * 6)[0], [10., 6., 2., -2., -6., -10.], what="core_h 2")
    eq(orc.remapping_core_h(CS, h0, u0, [.125] * 6)[0], [11.5, 10.5, 9.5, 8.5, 7.5, 6.5], what="core_h 3")
    eq(orc.remapping_core_h(CS, h0, u0, [2.25, 1.5, 1.])[0], [3., -10.5, -12.], what="core_h 4")


def test_reconstructions_PLM(orc):
    """:2184-2208 'Unlim PLM', 'Left lim PLM', 'Right lim PLM', 'Non-uniform line PLM'."""
    for h, u, eL, eR, p1 in (([1., 1., 1.], [1., 3., 5.], [1., 2., 5.], [1., 4., 5.], [0., 2., 0.]),
                             ([1., 1., 1.], [1., 2., 7.], [1., 1., 7.], [1., 3., 7.], [0., 2., 0.]),
                             ([1., 1., 1.], [1., 6., 7.], [1., 5., 7.], [1., 7., 7.], [0., 2., 0.]),
                             ([1., 2., 3.], [1., 4., 9.], [1., 2., 9.], [1., 6., 9.], [0., 4., 0.])):
        E1, E2, C1, C2 = orc.PLM_reconstruction(h, u, H_NEGLECT)
        eq(E1, eL, what="PLM left edges"); eq(E2, eR, what="PLM right edges"); eq(C1, eL, what="PLM P0"); eq(C2, p1, what="PLM P1")


def test_reconstructions_H4_and_PPM(orc):
    """:2210-2248 'Line H4' (tol 8e-15 / 1e-14), 'Line PPM', 'Parabola H4' (2.7e-14 / 4.8e-14), 'Parabola PPM', 'Limits PPM'."""
    one = [1.] * 5
    E1, E2 = orc.edge_values_explicit_h4(one, [1., 3., 5., 7., 9.], 1e-10)
    eq(E1, [0., 2., 4., 6., 8.], 8.0e-15, "Line H4: left edges"); eq(E2, [2., 4., 6., 8., 10.], 1.0e-14, "Line H4: right edges")
    e1, e2, c1, c2, c3 = orc.PPM_reconstruction(one, [1., 3., 5., 7., 9.], [0., 2., 4., 6., 8.], [2., 4., 6., 8., 10.], H_NEGLECT)
    eq(c1, [1., 2., 4., 6., 9.], what="Line PPM: P0"); eq(c2, [0., 2., 2., 2., 0.], what="Line PPM: P1"); eq(c3, [0.] * 5, what="Line PPM: P2")
    E1, E2 = orc.edge_values_explicit_h4(one, [1., 1., 7., 19., 37.], 1e-10)
    eq(E1, [3., 0., 3., 12., 27.], 2.7e-14, "Parabola H4: left edges"); eq(E2, [0., 3., 12., 27., 48.], 4.8e-14, "Parabola H4: right edges")
    e1, e2, c1, c2, c3 = orc.PPM_reconstruction(one, [0., 1., 7., 19., 37.], [0., 0., 3., 12., 27.], [0., 3., 12., 27., 48.], H_NEGLECT)
    eq(e1, [0., 0., 3., 12., 37.], what="Parabola PPM: left edges"); eq(e2, [0., 3., 12., 27., 37.], what="Parabola PPM: right edges")
    eq(c1, [0., 0., 3., 12., 37.], what="Parabola PPM: P0"); eq(c2, [0., 0., 6., 12., 0.], what="P1"); eq(c3, [0., 3., 3., 3., 0.], what="P2")
    e1, e2, c1, c2, c3 = orc.PPM_reconstruction(one, [0., 5., 7., 16., 15.], [0., 0., 6., 10., 15.], [0., 6., 12., 17., 15.], H_NEGLECT)
    eq(e1, [0., 3., 6., 16., 15.], what="Limits PPM: left edges"); eq(e2, [0., 6., 9., 16., 15.], what="Limits PPM: right edges")
    eq(c1, [0., 3., 6., 16., 15.], what="Limits PPM: P0"); eq(c2, [0., 6., 0., 0., 0.], what="P1"); eq(c3, [0., -3., 3., 0., 0.], what="P2")


INTERSECT = [   # :2255-2478 tests 1-6: (h0, h1, h_sub, h0_eff, isrc_start, isrc_end, isrc_max, itgt_start, itgt_end, isub_src)
    ([3., 3.], [2., 2., 2.], [0., 2., 1., 1., 2., 0.], [3., 3.], [1, 4], [3, 5], [2, 5], [1, 3, 5], [2, 4, 6], [1, 1, 1, 2, 2, 2]),
    ([2., 2., 2.], [3., 3.], [0., 2., 1., 1., 2., 0.], [2., 2., 2.], [1, 3, 5], [2, 4, 5], [2, 4, 5], [1, 4], [3, 6], [1, 1, 2, 2, 3, 3]),
    ([2., 4.], [2., 2., 2.], [0., 2., 0., 2., 2., 0.], [2., 4.], [1, 3], [2, 5], [2, 5], [1, 4, 5], [3, 4, 6], [1, 1, 2, 2, 2, 2]),
    ([2., 4.], [2., 2., 1.], [0., 2., 0., 2., 1., 1.], [2., 3.], [1, 3], [2, 6], [2, 4], [1, 4, 5], [3, 4, 5], [1, 1, 2, 2, 2, 2]),
    ([2., 2., 1.], [2., 4.], [0., 2., 0., 2., 1., 1.], [2., 2., 1.], [1, 3, 5], [2, 4, 5], [2, 4, 5], [1, 4], [3, 6], [1, 1, 2, 2, 3, 3]),
    ([2., 0., 2.], [1., 0., 1., 0., 2.], [0., 1., 0., 1., 0., 0., 0., 2., 0.], [2., 0., 2.], [1, 5, 6], [4, 5, 8], [4, 5, 8],
     [1, 3, 4, 7, 8], [2, 3, 6, 7, 9], [1, 1, 1, 1, 2, 3, 3, 3, 3]),
]


@pytest.mark.parametrize("case", range(len(INTERSECT)))
def test_intersect_src_tgt_grids(orc, case):
    h0, h1, h_sub, h0_eff, s0, e0, m0, s1, e1, src = INTERSECT[case]
    r = orc.intersect_src_tgt_grids(h0, h1)
    eq(r["h_sub"], h_sub, what="h_sub"); eq(r["h0_eff"], h0_eff, what="h0_eff")
    for n, v in (("isrc_start", s0), ("isrc_end", e0), ("isrc_max", m0), ("itgt_start", s1), ("itgt_end", e1), ("isub_src", src)):
        assert list(r[n]) == v, (n, list(r[n]), v)


SUBGRID = [   # :2330-2478 tests 3-6: (h0, u0, h1, u_sub, u1 or None, which integrators the reference checks)
    ([2., 4.], [2., 5.], [2., 2., 2.], [1., 2., 3., 4., 6., 7.], [2., 4., 6.], (1, 0)),
    ([2., 4.], [2., 5.], [2., 2., 1.], [1., 2., 3., 4., 5.5, 6.5], None, (0,)),
    ([2., 2., 1.], [2., 4., 5.5], [2., 4.], [1., 2., 3., 4., 5.5, 6.], [2., 4.875], (1, 0)),
    ([2., 0., 2.], [2., 3., 4.], [1., 0., 1., 0., 2.], [1., 1.5, 2., 2.5, 3., 3., 3., 4., 5.], [1.5, 2., 2.5, 3., 4.], (1, 0)),
]


@pytest.mark.parametrize("case", range(len(SUBGRID)))
def test_remap_src_to_sub_grid_and_back(orc, case):
    h0, u0, h1, u_sub, u1, which = SUBGRID[case]
    for om4 in which:
        for bounds in (False, True):   # 'u1' and 'u1.b'
            us, ut = orc.remap_src_to_sub_grid_plm(h0, u0, h1, om4, H_NEGLECT, force_bounds=False)
            eq(us, u_sub, what=f"u_sub om4={om4}")
            if u1 is not None:
                eq(ut, u1, what=f"u1 om4={om4}")


def test_remapping_core_h_PLM(orc):
    """:2484-2502 'PLM: remapped h=0110->h=11' and 'h=0110->h=14' (interior layers between vanished ones)."""
    for om4 in (1, 0):
        CS = abi.remapping_params_default(abi.REMAP_PLM, H_NEGLECT, answer_date=20190101, om4_remap_via_sub_cells=om4)
        eq(orc.remapping_core_h(CS, [0., 1., 1., 0.], [5., 4., 2., 1.], [1., 1.])[0], [4., 2.], what="h=0110->h=11")
        eq(orc.remapping_core_h(CS, [0., 1., 1., 0.], [5., 4., 2., 1.], [1., 4.])[0], [4., 1.25], what="h=0110->h=14")


@pytest.mark.parametrize("scheme", [abi.REMAP_PCM, abi.REMAP_PLM, abi.REMAP_PPM_H4, abi.REMAP_PPM_IH4])
@pytest.mark.parametrize("om4", [0, 1])
def test_invariants_of_the_reference_brute_force_tests(orc, scheme, om4):
    """test_preserve_uniform :1900, test_unchanged_grid :1961 and conservation (check_remapped_values :1498) on random
    columns, as the reference's brute-force loops do: with the bounds forced in sub-cells and targets (the switches
    test_preserve_uniform sets) a uniform profile stays uniform (exactly for PCM; to 4 ulp with the
    boundary-extrapolated edge values of PLM / PPM_H4, whose weighted means of equal numbers may round); an unchanged grid returns the source values to
    round-off; the column integral is conserved to the routine's own error bound; no new extrema."""
    rng = np.random.default_rng(42 + scheme + 10 * om4)
    CS = abi.remapping_params_default(scheme, H_NEGLECT, om4_remap_via_sub_cells=om4)
    CSb = abi.remapping_params_default(scheme, H_NEGLECT, om4_remap_via_sub_cells=om4, force_bounds_in_subcell=1)
    CSn = abi.remapping_params_default(scheme, H_NEGLECT, om4_remap_via_sub_cells=om4, boundary_extrapolation=0)
    for it in range(200):
        n0, n1 = rng.integers(4, 12), rng.integers(2, 12)
        h0 = rng.random(n0); h0[rng.random(n0) < 0.2] = 0.0
        h1 = rng.random(n1); h1[rng.random(n1) < 0.2] = 0.0
        if h0.sum() == 0 or h1.sum() == 0:
            continue
        h1 *= h0.sum() / h1.sum()
        u1, err = orc.remapping_core_h(CSb, h0, np.full(n0, 3.25), h1)
        # (the implicit edge values of PPM_IH4 come out of a linear solve, uniform only to its round-off; the reference
        #  itself marks the non-PCM schemes as failing its exact test_preserve_uniform, MOM_remapping.F90:2775-2782)
        tol = 0.0 if scheme == abi.REMAP_PCM else (64 if scheme == abi.REMAP_PPM_IH4 else 4) * np.finfo(float).eps * 3.25
        assert np.abs(u1 - 3.25).max() <= tol, (scheme, om4, u1 - 3.25)
        u0 = rng.random(n0) * 10 - 5
        u1, err = orc.remapping_core_h(CS, h0, u0, h0)
        assert np.abs(u1 - u0)[h0 > 0].max() < 1e-13
        u1, err = orc.remapping_core_h(CS, h0, u0, h1)
        assert abs((u1 * h1).sum() - (u0 * h0).sum()) <= max(err, 1e-15) * 4 + 1e-14
        u1, err = orc.remapping_core_h(CSn, h0, u0, h1)     # without boundary extrapolation the schemes are monotone
        assert u1.min() >= u0.min() - 1e-12 and u1.max() <= u0.max() + 1e-12


def test_edge_values_implicit_h4_is_fourth_order(orc):
    """edge_values_implicit_h4 (regrid_edge_values.F90:473; PPM_IH4 of .testing/tc2 and tc4): the reference holds no
    numbers for it; by construction the compact scheme and the end_value_h4 closure reproduce cubic profiles."""
    h = np.array([1., 2., 1.5, 0.5, 3., 1., 2.])
    z = np.concatenate(([0.], np.cumsum(h)))
    for f, F in ((lambda x: 2 * x + 1, lambda x: x * x + x), (lambda x: x ** 3 - 2 * x, lambda x: x ** 4 / 4 - x * x)):
        u = (F(z[1:]) - F(z[:-1])) / h
        E1, E2 = orc.edge_values_implicit_h4(h, u, 1e-30)
        assert np.abs(E1 - f(z[:-1])).max() < 1e-11 and np.abs(E2 - f(z[1:])).max() < 1e-11
        assert np.array_equal(E1[1:], E2[:-1])          # one value per interface


# ---- regridding (z*): the reference holds no numbers for it; invariants and an exactly representable known answer ----
def test_regrid_zstar_known_answer_and_invariants(orc):
    from tests import helpers as H
    from mom6_amd import grid, synth
    G = abi.G
    # (a) flat bottom of 100 m, four nominal layers of 25 m, SSH = +2 m, layers squeezed into the top:
    #     z* puts the interfaces at eta - k * 25 * (102/100) and the bottom at -100 -- exactly representable quarters
    gg = grid.GlobalGrid(12, 10, kind="cartesian", dx=1.0e4, dy=1.0e4, f0=1e-4, beta=0.0, depth_fn=grid.flat_depth(12, 10, 100.0))
    d, M = gg.tile(4)
    GV = abi.vgrid_default()
    h = np.zeros((4,) + d.shape2()); h[0] = 99.0; h[1:] = 1.0
    hn = np.zeros_like(h); dz = np.zeros((5,) + d.shape2())
    orc.ALE_regrid_zstar(d, M, GV, abi.regrid_zstar_params_default(), [25.0] * 4, h, hn, dz)
    x = (d.joff + 3, d.ioff + 4)
    assert M[G["mask2dT"]][x] > 0
    assert np.array_equal(hn[(slice(None),) + x], [25.5, 25.5, 25.5, 25.5])
    assert np.array_equal(dz[(slice(None),) + x], [0.0, 73.5, 49.0, 24.5, 0.0])
    # (b) random stacks over a bowl: total thickness kept to round-off, no layer thinner than MIN_THICKNESS (or H/nk),
    #     the surface and the bottom do not move, and regridding the regridded column changes nothing (fixed point)
    gg, d, M = H.benchmark_small(nk=10)
    h, _, _ = synth.make_state(d, M, thin_frac=0.2)
    cr = np.linspace(50., 800., d.nk); cr *= 4000. / cr.sum()
    for CS in (abi.regrid_zstar_params_default(), abi.regrid_zstar_params_default(min_thickness=5.0),
               abi.regrid_zstar_params_default(old_grid_weight=0.4, depth_of_time_filter_shallow=200., depth_of_time_filter_deep=900.)):
        hn = np.zeros_like(h); dz = np.zeros((d.nk + 1,) + d.shape2())
        orc.ALE_regrid_zstar(d, M, GV, CS, cr, h, hn, dz)
        sl = H.interior(d, "h", 1)
        wet = M[G["mask2dT"]][tuple(sl)] > 0
        a, b = hn[(Ellipsis,) + tuple(sl)][:, wet], h[(Ellipsis,) + tuple(sl)][:, wet]
        assert np.abs(a.sum(0) - b.sum(0)).max() <= 1e-12 * b.sum(0).max()
        assert np.abs(dz[0][tuple(sl)]).max() == 0.0 and np.abs(dz[-1][tuple(sl)][wet]).max() <= 1e-9
        if CS.old_grid_weight == 0.0:
            assert a.min() >= min(CS.min_thickness, (b.sum(0) / d.nk).min()) * (1 - 1e-9)
            hn2 = np.zeros_like(h); dz2 = np.zeros_like(dz)
            orc.ALE_regrid_zstar(d, M, GV, CS, cr, hn, hn2, dz2)
            assert np.abs(dz2[(Ellipsis,) + tuple(sl)][:, wet]).max() <= 1e-9
        else:     # the time filter moves the deep interfaces only part of the way
            dz0 = np.zeros_like(dz); orc.ALE_regrid_zstar(d, M, GV, abi.regrid_zstar_params_default(), cr, h, np.zeros_like(h), dz0)
            assert np.abs(dz).sum() < np.abs(dz0).sum()


# ---- the REAL reference code, where it compiles from its own files (oracle/_ref, built by `make -C oracle ref`) ----------
def _ref_lib():
    import ctypes as C
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_ale.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/libref_ale.so not built (needs /root/reference and amdflang: make -C oracle ref)")
    L = C.CDLL(p)
    for f in ("ref_PLM_slope_wa", "ref_PLM_monotonized_slope", "ref_PLM_extrapolate_slope"):
        getattr(L, f).restype = C.c_double
    return L


def test_oracle_PLM_against_the_compiled_reference(orc):
    """src/ALE/PLM_functions.F90 and PCM_functions.F90 have no dependencies, so oracle/Makefile compiles them as they
    lie in the reference tree (amdflang, -fdefault-real-8 as the reference's builds, no FMA contraction) behind the
    bind(C) doors of oracle/ref_shim.F90.  The oracle's PLM reconstruction -- edge values and coefficients, with and
    without boundary extrapolation -- must equal the reference's own code BIT FOR BIT on random ragged columns."""
    import ctypes as C
    L = _ref_lib()
    rng = np.random.default_rng(20250808)
    dbl = C.c_double
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    nchk = 0
    for it in range(400):
        n = int(rng.integers(2, 40))
        h = rng.random(n); h[rng.random(n) < 0.2] = 0.0
        u = rng.random(n) * 30 - 10
        if it % 7 == 0:
            u = np.round(u)                      # ties and exact extrema
        if it % 11 == 0:
            u[:] = u[0]
        hn = (1e-30, 1e-10)[it % 2]
        for extrap in (0, 1):
            E1, E2, C1, C2 = orc.PLM_reconstruction(h, u, hn, extrapolate=bool(extrap))
            edges = np.zeros((2, n)); coefs = np.zeros((2, n))          # Fortran (n,2) = C [2][n]
            L.ref_PLM_reconstruction(n, ptr(np.ascontiguousarray(h)), ptr(np.ascontiguousarray(u)), ptr(edges), ptr(coefs), dbl(hn), extrap)
            for a, b, what in ((E1, edges[0], "left edges"), (E2, edges[1], "right edges"), (C1, coefs[0], "P0"), (C2, coefs[1], "P1")):
                assert np.array_equal(a, b) and np.isfinite(a).all(), (what, it, extrap, a - b)
            nchk += 1
    assert nchk == 800
    # PCM :16-35
    u = rng.random(9); edges = np.zeros((2, 9)); coefs = np.zeros((1, 9))
    L.ref_PCM_reconstruction(9, ptr(u), ptr(edges), ptr(coefs))
    CS = abi.remapping_params_default(abi.REMAP_PCM, H_NEGLECT)
    assert np.array_equal(edges[0], u) and np.array_equal(edges[1], u) and np.array_equal(coefs[0], u)
    u1, _ = orc.remapping_core_h(CS, np.ones(9), u, np.ones(9))
    assert np.array_equal(u1, u)


def test_oracle_PPM_H4_against_the_compiled_reference(orc):
    """The reference ships a second, dependency-free implementation of the same PPM_H4 (2019 expressions) as a class,
    src/ALE/Recon1d_PPM_H4_2019.F90 (+ Recon1d_type.F90, numerical_testing_type.F90); oracle/_ref compiles it as it lies.
    The oracle's edge_values_explicit_h4 -> bound_edge_values -> check_discontinuous_edge_values -> PPM limiter chain
    (build_reconstructions_1d's REMAPPING_PPM_H4 branch without boundary extrapolation) must give that code's edge
    values BIT FOR BIT on random ragged columns, vanished layers and ties included."""
    import ctypes as C
    L = _ref_lib()
    if not hasattr(L, "ref_PPM_H4_2019"):
        pytest.skip("oracle/_ref built without Recon1d_PPM_H4_2019")
    rng = np.random.default_rng(20250809)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    for it in range(600):
        n = int(rng.integers(4, 60))
        h = rng.random(n); h[rng.random(n) < 0.2] = 0.0
        if it % 13 == 0:
            h[:] = 0.0; h[rng.integers(0, n)] = 1.0      # a column that is almost entirely vanished
        u = rng.random(n) * 30 - 10
        if it % 7 == 0:
            u = np.round(u)
        if it % 11 == 0:
            u[:] = u[0]
        hn = (1e-30, 1e-10)[it % 2]
        E1, E2 = orc.edge_values_explicit_h4(h, u, hn)
        e1, e2, c1, c2, c3 = orc.PPM_reconstruction(h, u, E1, E2, hn)
        ul, ur = np.zeros(n), np.zeros(n)
        L.ref_PPM_H4_2019(n, ptr(np.ascontiguousarray(h)), ptr(np.ascontiguousarray(u)), C.c_double(hn), ptr(ul), ptr(ur))
        assert np.isfinite(ul).all() and np.isfinite(ur).all()
        assert np.array_equal(e1, ul), (it, n, e1 - ul)
        assert np.array_equal(e2, ur), (it, n, e2 - ur)


def test_oracle_sub_grid_integration_against_the_compiled_reference(orc):
    """Recon1d_type.F90:173 remap_to_sub_grid (the reference's class-based twin of remap_src_to_sub_grid :962) with the
    compiled PPM_H4_2019 class, fed with the oracle's intersect_src_tgt_grids: the sub-cell averages of every sub-cell of
    positive width and ALL sub-cell integrals -- including the thickest-sub-cell conservation fix -- must be bit-identical.
    (Zero-width sub-cells are point values; the class evaluates those with the interval formula, the OM4-era function
    with a separate one, so they agree to round-off only and carry no weight.)"""
    import ctypes as C
    L = _ref_lib()
    if not hasattr(L, "ref_PPM_H4_2019_to_sub_grid"):
        pytest.skip("oracle/_ref built without Recon1d_PPM_H4_2019")
    rng = np.random.default_rng(20250810)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    lib = orc.lib()
    for it in range(400):
        n0, n1 = int(rng.integers(4, 30)), int(rng.integers(1, 30))
        h0 = rng.random(n0); h0[rng.random(n0) < 0.15] = 0.0
        h1 = rng.random(n1); h1[rng.random(n1) < 0.15] = 0.0
        if h0.sum() == 0 or h1.sum() == 0:
            continue
        h1 *= h0.sum() / h1.sum() * (1.0, 0.8, 1.3)[it % 3]
        u0 = rng.random(n0) * 30 - 10
        hn = 1e-30
        r = orc.intersect_src_tgt_grids(h0, h1)
        ns = n0 + n1 + 1
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        us_ref, uhs_ref = np.zeros(ns), np.zeros(ns)
        L.ref_PPM_H4_2019_to_sub_grid(n0, ptr(np.ascontiguousarray(h0)), ptr(np.ascontiguousarray(u0)), C.c_double(hn), n1,
                                      ptr(np.ascontiguousarray(r["h_sub"])), ptr(i32(r["isrc_start"])), ptr(i32(r["isrc_end"])),
                                      ptr(i32(r["isrc_max"])), ptr(i32(r["isub_src"])), ptr(us_ref), ptr(uhs_ref))
        # the oracle's chain: PPM_H4 reconstruction (no boundary extrapolation) + remap_src_to_sub_grid (non-OM4)
        E1, E2 = orc.edge_values_explicit_h4(h0, u0, hn)
        e1, e2, c1, c2, c3 = orc.PPM_reconstruction(h0, u0, E1, E2, hn)
        one = lambda a, n: np.concatenate(([0.0], np.asarray(a, dtype=np.float64), [0.0] * (n + 1 - len(a))))
        ione = lambda a, n: np.concatenate(([0], np.asarray(a, dtype=np.int32), [0] * (n + 1 - len(a)))).astype(np.int32)
        u_sub, uh_sub = np.zeros(ns + 2), np.zeros(ns + 2); err = C.c_double(0.0)
        lib.orc_remap_src_to_sub_grid(0, n0, ptr(one(h0, n0)), ptr(one(u0, n0)), ptr(one(e1, n0)), ptr(one(e2, n0)), ptr(one(c1, n0)),
                                      ptr(one(c2, n0)), n1, ptr(one(r["h_sub"], ns)), ptr(one(r["h0_eff"], n0)), ptr(ione(r["isrc_start"], n0)),
                                      ptr(ione(r["isrc_end"], n0)), ptr(ione(r["isrc_max"], n0)), ptr(ione(r["isub_src"], ns + 1)), 3, 0,
                                      ptr(u_sub), ptr(uh_sub), C.byref(err))
        pos = r["h_sub"] > 0
        assert np.array_equal(u_sub[1:ns + 1][pos], us_ref[pos]), (it, u_sub[1:ns + 1][pos] - us_ref[pos])
        assert np.array_equal(uh_sub[1:ns + 1], uhs_ref), (it, uh_sub[1:ns + 1] - uhs_ref)
        assert np.abs(u_sub[1:ns + 1] - us_ref).max() <= 1e-12 * max(np.abs(u0).max(), 1.0)
