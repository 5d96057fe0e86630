import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The oracle (test infrastructure only)."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(params=["fma", "tree", "exact"])
def sums(request, monkeypatch):
    """The three arithmetics of the mass-flux kernels (mom6x_continuity_params.sum_order, abi.default_sum_order): "fma" = the
    default of the device (MOM6X_SUM_TREE16_FMA: the wave-owned kernel's 16-lane tree + fused multiply-adds at fixed sites),
    "tree" = the same tree un-fused (MOM6X_SUM_TREE16), "exact" = the reference's sequential k order
    (MOM6X_SUM_REFERENCE, the LDS kernel).  The oracle follows the same switch, so "exact" cases are held to the
    REFERENCE-order restatement bit for bit."""
    monkeypatch.setenv("MOM6X_SUMS", request.param)
    return request.param


def pytest_terminal_summary(terminalreporter):
    from tests import helpers as H
    if H.SIGNED_ZERO_LOG:
        tot = sum(H.SIGNED_ZERO_LOG.values())
        terminalreporter.write_line(f"assert_bitwise: {tot} zeros of opposite sign accepted (sum_order TREE16 only): "
                                    + ", ".join(f"{k}={v}" for k, v in sorted(H.SIGNED_ZERO_LOG.items())[:12]))
