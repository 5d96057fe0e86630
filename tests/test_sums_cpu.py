"""The oracle's reproducing sums and checksums (oracle/orc_sums.c; MOM_coms.F90, MOM_checksums.F90) on the CPU.

The reference holds no numbers for these routines, so the restatement is checked by what the construction promises:
the extended-fixed-point integers ARE the exact sum (compared with rational arithmetic), whatever the order and the
decomposition of the domain; exactly representable sums come out exactly; the bit counts follow from popcounts."""
from fractions import Fraction

import numpy as np
import pytest

from tests import helpers as H


def efp_value(v):
    return sum(Fraction(int(v[i])) * Fraction(2) ** (46 * (2 - i)) for i in range(6))


def wide_range_field(d, nk, seed):
    rng = np.random.default_rng(seed)
    a = np.zeros((nk,) + tuple(d.shape2()))
    a[:] = rng.standard_normal(a.shape) * 10.0 ** rng.integers(-8, 12, a.shape)
    return a


def test_reproducing_sum_is_the_exact_sum(orc):
    gg, d, M = H.benchmark_small(nk=4)
    a = wide_range_field(d, 4, 1)
    sl = H.interior(d, "h")
    r = orc.reproducing_sum(d, a, layer_sums=True)
    exact = [sum(Fraction(x) for x in a[k][tuple(sl)].ravel()) for k in range(4)]
    for k in range(4):
        # every term is truncated at 2**-138; nothing else is lost
        assert abs(efp_value(r["EFP_lay"][k]) - exact[k]) <= a[k][tuple(sl)].size * Fraction(2) ** -138
        assert r["sums"][k] == pytest.approx(float(exact[k]), rel=4e-16)
        assert all(abs(int(x)) < 2 ** 46 for x in r["EFP_lay"][k][1:])            # regularized: canonical digits
        assert len({np.sign(x) for x in r["EFP_lay"][k] if x != 0}) <= 1           # ... of one sign
    tot = 0.0
    for k in range(4):
        tot = tot + r["sums"][k]
    assert r["sum"] == tot                      # with layer sums the total is their floating-point sum (MOM_coms.F90:470-477)
    # without the layer sums the total is ONE conversion of the summed integers (MOM_coms.F90:541-542)
    r1 = orc.reproducing_sum(d, a)
    assert abs(efp_value(r1["EFP"]) - sum(exact)) <= a.size * Fraction(2) ** -138
    assert r1["sum"] == pytest.approx(float(sum(exact)), rel=4e-16)
    # 2-d entry point, sub-rectangles, unscale
    r2 = orc.reproducing_sum(d, a[1], 3, d.ni - 5, 2, d.nj - 2)
    ex2 = sum(Fraction(x) for x in a[1][d.joff + 2:d.joff + d.nj - 1, d.ioff + 3:d.ioff + d.ni - 4].ravel())
    assert abs(efp_value(r2["EFP"]) - ex2) <= a[1].size * Fraction(2) ** -138
    r3 = orc.reproducing_sum(d, a, unscale=0.25, layer_sums=True)
    assert np.array_equal(r3["sums"], orc.reproducing_sum(d, 0.25 * a, layer_sums=True)["sums"] * 4.0)


def test_reproducing_sum_order_and_decomposition_invariance(orc):
    """The point of the routine (.testing test.layout): sums over sub-domains, added as integers, give the same bits."""
    gg, d, M = H.benchmark_small(nk=3)
    a = wide_range_field(d, 3, 2)
    whole = orc.reproducing_sum(d, a)
    parts = [orc.reproducing_sum(d, a, i0, i1, j0, j1)["EFP"] for (i0, i1) in ((0, 17), (18, d.ni - 1))
             for (j0, j1) in ((0, 9), (10, d.nj - 1))]
    tot = np.zeros(6, dtype=object)
    for p in parts:
        tot = tot + np.array([int(x) for x in p], dtype=object)
    assert efp_value(tot) == efp_value(whole["EFP"])
    assert orc.EFP_to_real(np.array([int(x) for x in tot], dtype=np.int64)) == whole["sum"]
    # mirrored and transposed-in-k copies: a different order of the additions
    b = np.ascontiguousarray(a[::-1])
    assert orc.reproducing_sum(d, b)["sum"] == whole["sum"]


def test_reproducing_sum_errors(orc):
    gg, d, M = H.double_gyre()
    a = np.ones((2,) + tuple(d.shape2()))
    a[1, d.joff + 3, d.ioff + 3] = np.nan
    with pytest.raises(RuntimeError, match="NaN"):
        orc.reproducing_sum(d, a)
    assert orc.reproducing_sum(d, a, want_err=True)["err"] == 2
    assert orc.reproducing_sum(d, a[1], want_err=True)["err"] == 4
    a[1, d.joff + 3, d.ioff + 3] = 1.0e60                        # beyond max_efp_float
    with pytest.raises(RuntimeError, match="Overflow"):
        orc.reproducing_sum(d, a)
    a[1, d.joff + 3, d.ioff + 3] = 2.0
    assert orc.reproducing_sum(d, a)["sum"] == 2.0 * d.ni * d.nj + 1.0


def test_EFP_arithmetic(orc):
    a, b = orc.real_to_EFP(1.0e15 + 0.375), orc.real_to_EFP(-3.0e-20)
    assert orc.EFP_to_real(a) == 1.0e15 + 0.375
    diff = orc.EFP_minus(a, b)
    assert efp_value(diff) == efp_value(a) - efp_value(b)


def test_chksum_bitcounts_and_stats(orc):
    gg, d, M = H.benchmark_small(nk=3)
    a = wide_range_field(d, 3, 3)
    sl = H.interior(d, "h")

    def popsum(di, dj, arr=a, scale=1.0):
        s = 0
        for k in range(arr.shape[0] if arr.ndim == 3 else 1):
            pl = arr[k] if arr.ndim == 3 else arr
            v = np.abs(scale * pl[d.joff + dj:d.joff + dj + d.nj, d.ioff + di:d.ioff + di + d.ni])
            s += int(np.sum([bin(x).count("1") for x in v.view(np.uint64).ravel()]))
        return s % 1000000000
    r = orc.chksum(d, a, "h", haloshift=2)
    assert r["bc0"] == popsum(0, 0) and r["bc"] == [popsum(-2, -2), popsum(2, -2), popsum(-2, 2), popsum(2, 2)]
    assert r["min"] == a[(Ellipsis,) + tuple(sl)].min() and r["max"] == a[(Ellipsis,) + tuple(sl)].max()
    assert r["mean"] == orc.reproducing_sum(d, a)["sum"] / (3 * d.ni * d.nj)
    r = orc.chksum(d, a, "h", haloshift=1, omit_corners=True)
    assert r["bc"] == [popsum(0, 1), popsum(0, -1), popsum(1, 0), popsum(-1, 0)]
    r = orc.chksum(d, a, "u", symmetric=True)
    assert r["bc"] == [popsum(-1, 0)]
    assert r["min"] == a[:, d.joff:d.joff + d.nj, d.ioff - 1:d.ioff + d.ni].min()
    r = orc.chksum(d, a, "v", haloshift=1, symmetric=True)
    assert r["bc"] == [popsum(-1, -2), popsum(1, -2), popsum(-1, 1), popsum(1, 1)]
    # the 2-d and 3-d B-point routines of the reference shift differently without `symmetric`
    assert orc.chksum(d, a[:1], "B", haloshift=1)["bc"] == [popsum(-2, -2, a[:1]), popsum(1, -2, a[:1]), popsum(-2, 1, a[:1]), popsum(1, 1, a[:1])]
    assert orc.chksum(d, a[0], "B", haloshift=1)["bc"] == [popsum(-1, -1, a[0]), popsum(1, -1, a[0]), popsum(-1, 1, a[0]), popsum(1, 1, a[0])]
    r = orc.chksum(d, a, "h", scale=0.5)
    assert r["bc0"] == popsum(0, 0, scale=0.5) and r["max"] == 0.5 * a[(Ellipsis,) + tuple(sl)].max()
    a[1, d.joff + 1, d.ioff + 1] = np.nan
    with pytest.raises(RuntimeError, match="NaN"):
        orc.chksum(d, a, "h")


def test_field_chksum_is_the_wrapping_sum_of_bit_patterns(orc):
    gg, d, M = H.double_gyre()
    a = wide_range_field(d, 2, 4)
    sl = H.interior(d, "h")
    want = int(np.sum(a[(Ellipsis,) + tuple(sl)].view(np.uint64).astype(object))) % 2 ** 64
    assert orc.field_chksum(d, a, 0, d.ni - 1, 0, d.nj - 1) % 2 ** 64 == want
    assert "%016X" % want == "%016X" % (orc.field_chksum(d, a, 0, d.ni - 1, 0, d.nj - 1) % 2 ** 64)
