"""GPU parity of the reproducing sums and checksums (mom6_amd/csrc/diag_sums.hip, through the C ABI) with the oracle
(oracle/orc_sums.c; MOM_coms.F90, MOM_checksums.F90).  Integer work: every number must be identical."""
import numpy as np
import pytest

from mom6_amd import abi
from tests import helpers as H
from tests.test_sums_cpu import wide_range_field

pytestmark = pytest.mark.gpu


def make(cfg):
    from mom6_amd.dycore import Dycore
    gg, d, M = cfg
    return d, Dycore(d, M, abi.vgrid_default())


def same_sum(got, ref):
    assert got["sum"] == ref["sum"] and np.array_equal(got["EFP"], ref["EFP"])
    for n in ("sums", "EFP_lay"):
        if n in ref:
            assert np.array_equal(got[n], ref[n]), n


@pytest.mark.parametrize("cfg", ["double_gyre", "benchmark_small", "wide"])
def test_reproducing_sum_bitwise(orc, cfg):
    # "wide": 600 x 300 points per layer, more than max_count_prec = 131071 -- the reference carries row by row there
    d, dyc = make(H.channel(nk=3, ni=600, nj=300) if cfg == "wide" else getattr(H, cfg)(nk=5))
    nk = d.nk
    a = wide_range_field(d, nk, 11)
    a[0] = np.abs(a[0])                                   # a layer of one sign: limbs far above 2**53 before the carry
    ad = dyc.to_dev(a)
    same_sum(dyc.reproducing_sum(ad), orc.reproducing_sum(d, a))
    same_sum(dyc.reproducing_sum(ad, layer_sums=True), orc.reproducing_sum(d, a, layer_sums=True))
    same_sum(dyc.reproducing_sum(ad, unscale=0.001, layer_sums=True), orc.reproducing_sum(d, a, unscale=0.001, layer_sums=True))
    same_sum(dyc.reproducing_sum(ad, unscale=1024.0), orc.reproducing_sum(d, a, unscale=1024.0))
    same_sum(dyc.reproducing_sum(ad[1]), orc.reproducing_sum(d, a[1]))
    same_sum(dyc.reproducing_sum(ad[2], unscale=3.0), orc.reproducing_sum(d, a[2], unscale=3.0))
    r = (-1, d.ni - 1, 2, d.nj - 3)                        # a u-point range that starts on the western face
    same_sum(dyc.reproducing_sum(ad, *r, layer_sums=True), orc.reproducing_sum(d, a, *r, layer_sums=True))
    same_sum(dyc.reproducing_sum(ad[0], 5, 5, 7, 7), orc.reproducing_sum(d, a[0], 5, 5, 7, 7))
    dyc.close()


def test_reproducing_sum_order_invariance_on_the_device(orc):
    """The same numbers laid out differently (reversed layers, mirrored rows) give the same integers."""
    d, dyc = make(H.benchmark_small(nk=4))
    a = wide_range_field(d, 4, 12)
    r0 = dyc.reproducing_sum(dyc.to_dev(a))
    b = np.ascontiguousarray(a[::-1, ::-1, :])
    sl = H.interior(d, "h")
    c = np.zeros_like(a); c[(Ellipsis,) + tuple(sl)] = a[::-1][(Ellipsis,) + tuple(sl)][:, ::-1, ::-1]
    r1 = dyc.reproducing_sum(dyc.to_dev(c))
    assert r0["sum"] == r1["sum"] and np.array_equal(r0["EFP"], r1["EFP"])
    dyc.close()


def test_reproducing_sum_error_codes(orc):
    d, dyc = make(H.double_gyre())
    a = np.ones((2,) + tuple(d.shape2()))
    a[1, d.joff + 3, d.ioff + 3] = np.nan
    with pytest.raises(RuntimeError, match="NaN in input field of reproducing_sum"):
        dyc.reproducing_sum(dyc.to_dev(a))
    assert dyc.reproducing_sum(dyc.to_dev(a), want_err=True)["err"] == orc.reproducing_sum(d, a, want_err=True)["err"] == 2
    assert dyc.reproducing_sum(dyc.to_dev(a)[1], want_err=True)["err"] == orc.reproducing_sum(d, a[1], want_err=True)["err"] == 4
    a[1, d.joff + 3, d.ioff + 3] = 1.0e60
    with pytest.raises(RuntimeError, match="Overflow"):
        dyc.reproducing_sum(dyc.to_dev(a))
    assert dyc.reproducing_sum(dyc.to_dev(a), want_err=True)["err"] == orc.reproducing_sum(d, a, want_err=True)["err"]
    a[1, d.joff + 3, d.ioff + 3] = 2.0
    assert dyc.reproducing_sum(dyc.to_dev(a))["sum"] == 2.0 * d.ni * d.nj + 1.0
    dyc.close()


@pytest.mark.parametrize("stagger", ["h", "u", "v", "B"])
def test_chksum_matches_oracle(orc, stagger):
    d, dyc = make(H.benchmark_small(nk=3))
    a = wide_range_field(d, 3, 13)
    ad = dyc.to_dev(a)
    cases = [dict(), dict(haloshift=1), dict(haloshift=2, omit_corners=True), dict(scale=0.125), dict(haloshift=1, scale=3.0)]
    if stagger != "h":
        cases += [dict(symmetric=True), dict(haloshift=1, symmetric=True), dict(haloshift=2, symmetric=True, omit_corners=True)]
    for kw in cases:
        for arr_d, arr_h in ((ad, a), (ad[1], a[1])):                 # the 3-d and the 2-d routine
            got = dyc.chksum(arr_d, stagger, **kw); ref = orc.chksum(d, arr_h, stagger, **kw)
            kind = got.pop("kind")
            assert got == ref, (stagger, kw, arr_h.ndim, got, ref)
            assert len(got["bc"]) == {abi.CHK_NONE: 0, abi.CHK_CORNERS: 4, abi.CHK_NSEW: 4, abi.CHK_W: 1, abi.CHK_S: 1}[kind]
    l1, l2 = dyc.chksum_lines(ad, stagger, "test field", haloshift=1)
    assert l1.startswith(stagger + "-point: mean=") and l1.endswith(" test field") and len(l1) == len(stagger + "-point:") + 6 + 26 + 4 + 26 + 4 + 26 + 10
    assert l2.startswith(stagger + "-point: c=") and " sw=" in l2 and l2.endswith(" test field")
    a[2, d.joff + 2, d.ioff + 2] = np.nan
    with pytest.raises(RuntimeError, match="NaN detected"):
        dyc.chksum(dyc.to_dev(a), stagger)
    dyc.close()


def test_chksum_bitcount_wraps_like_a_default_integer(orc):
    """More than 2**31 set bits in one array: the reference's default-integer running sum wraps before the mod."""
    d, dyc = make(H.channel(nk=40, ni=1200, nj=800))
    a = np.full((40,) + tuple(d.shape2()), -np.nextafter(2.0, 1.0))      # 62 set bits in |x| (sign cleared)
    got = dyc.chksum(dyc.to_dev(a), "h")
    total = 40 * 1200 * 800 * 62
    wrapped = ((total + 2 ** 31) % 2 ** 32) - 2 ** 31
    want = int(np.fmod(wrapped, 1000000000))
    assert total > 2 ** 31 and got["bc0"] == want
    assert got["min"] == got["max"] == -np.nextafter(2.0, 1.0)
    ref = orc.chksum(d, a, "h")
    assert got["bc0"] == ref["bc0"] and got["mean"] == ref["mean"]
    dyc.close()


def test_field_chksum_matches_oracle(orc):
    d, dyc = make(H.benchmark_small(nk=6))
    a = wide_range_field(d, 6, 14)
    ad = dyc.to_dev(a)
    for r, un in (((0, d.ni - 1, 0, d.nj - 1), 1.0), ((-1, d.ni - 1, 0, d.nj - 1), 1.0), ((0, d.ni - 1, -1, d.nj - 1), 0.01)):
        assert dyc.field_chksum(ad, *r, unscale=un) == orc.field_chksum(d, a, *r, unscale=un)
    assert dyc.field_chksum(ad[3]) == orc.field_chksum(d, a[3], 0, d.ni - 1, 0, d.nj - 1)
    dyc.close()


def test_reproducing_sum_is_layout_invariant(orc):
    """The reference's test.layout property for the sums: the field cut into the tiles of a 2 x 2 layout, each tile summed
    by its own context (only_on_PE), the integers added as sum_across_PEs adds them -- the result is the one-tile sum, bit
    for bit, for every layer."""
    from mom6_amd.dycore import Dycore
    from mom6_amd import sum_output as SO
    gg, d, M = H.benchmark_small(nk=3)
    a = wide_range_field(d, 3, 21)
    dyc = Dycore(d, M, abi.vgrid_default())
    whole = dyc.reproducing_sum(dyc.to_dev(a), layer_sums=True)
    dyc.close()
    tot = np.zeros((3, 6), dtype=object)
    for pe in ((0, 0), (1, 0), (0, 1), (1, 1)):
        dt_, Mt = gg.tile(3, 4, (2, 2), pe)
        loc = np.zeros((3,) + tuple(dt_.shape2()))
        gl = d.sl(dt_.i_glob0, dt_.i_glob0 + dt_.ni - 1, dt_.j_glob0, dt_.j_glob0 + dt_.nj - 1)
        loc[(Ellipsis,) + tuple(dt_.sl(0, dt_.ni - 1, 0, dt_.nj - 1))] = a[(Ellipsis,) + tuple(gl)]
        dy = Dycore(dt_, Mt, abi.vgrid_default())
        r = dy.reproducing_sum(dy.to_dev(loc), layer_sums=True, only_on_PE=True)
        tot = tot + np.array([[int(x) for x in row] for row in r["EFP_lay"]], dtype=object)
        dy.close()
    for k in range(3):
        assert SO.EFP_to_real(list(tot[k])) == whole["sums"][k]


@pytest.mark.parametrize("ni,nj,nk,halo", [(7, 5, 1, 3), (17, 9, 3, 4), (257, 3, 2, 4), (3, 300, 2, 4)])
def test_sums_and_checksums_on_ragged_tiles(orc, ni, nj, nk, halo):
    """Tile extents far from the 256 x 8 blocks of the reduction kernel, one layer, the narrowest halo."""
    d, dyc = make(H.double_gyre(nk=nk, ni=ni, nj=nj, halo=halo))
    a = wide_range_field(d, nk, 31)
    ad = dyc.to_dev(a)
    same_sum(dyc.reproducing_sum(ad, layer_sums=True), orc.reproducing_sum(d, a, layer_sums=True))
    same_sum(dyc.reproducing_sum(ad[0], -1, d.ni - 1, -1, d.nj - 1), orc.reproducing_sum(d, a[0], -1, d.ni - 1, -1, d.nj - 1))
    for stg in "huvB":
        for kw in (dict(), dict(haloshift=halo - 1, symmetric=(stg != "h")) if stg != "h" else dict(haloshift=halo - 1)):
            got = dyc.chksum(ad, stg, **kw); got.pop("kind")
            assert got == orc.chksum(d, a, stg, **kw), (stg, kw)
    assert dyc.field_chksum(ad) == orc.field_chksum(d, a, 0, d.ni - 1, 0, d.nj - 1)
    dyc.close()
