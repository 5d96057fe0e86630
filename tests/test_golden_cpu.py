"""The oracle against the fixtures committed under tests/golden/ (made by scripts/make_golden.py).

The fixtures are regression data of the oracle itself, not reference output (parity unpinned);
this test makes sure the checker the GPU tests rely on has not drifted."""
import numpy as np
import pytest

from tests import cases
from tests import helpers as H

@pytest.fixture(autouse=True, params=["exact", "tree", "fma"])
def sum_order(request, monkeypatch):
    """Both orders of the mass-flux column sums (mom6x_continuity_params.sum_order) have their own fixtures."""
    monkeypatch.setenv("MOM6X_SUMS", request.param)


STAG = dict(u="u", v="v", h="h", uh="u", vh="v", uhtr="u", vhtr="v", eta_av="h", u_cor="u", v_cor="v")


def test_oracle_rk2_reproduces_golden(orc):
    cfg = H.double_gyre()
    so, _ = cases.oracle_rk2(orc, cfg, cases.rk2_inputs(cfg), 3, bt_mod=dict(strong_drag=1))
    gold = H.load_golden("rk2_double_gyre_strong_drag_3steps")
    assert set(gold) == set(so)
    for n in so:
        H.assert_bitwise(so[n][(Ellipsis,) + tuple(H.interior(cfg[1], STAG[n]))], gold[n], n)
    assert np.abs(gold["u"]).max() > 1e-3 and np.isfinite(gold["h"]).all()


def test_oracle_continuity_reproduces_golden(orc):
    cfg = H.benchmark_small()
    out, _, _ = cases.oracle_continuity(orc, cfg, cases.continuity_inputs(cfg))
    gold = H.load_golden("continuity_benchmark_small_corrector")
    for n in out:
        H.assert_bitwise(out[n][(Ellipsis,) + tuple(H.interior(cfg[1], STAG[n]))], gold[n], n)
