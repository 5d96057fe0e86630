"""mom6x_continuity_stats (what bench.py prints as newton_evals_per_solve) on the wave-owned mass-flux kernel, in both of its
arithmetics: the counting instantiation gives the oracle's bits like the plain one, and it counts.  (Round 6's first FMA default had
no counting instantiation: the bench line said 0.0 Newton sweeps per solve while the kernel did 1.95.)"""
import pytest

from tests import helpers as H
from tests.test_continuity_gpu import _run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sums", ["tree", "fma"])
def test_continuity_stats_count_the_newton_sweeps(orc, sums, monkeypatch):
    monkeypatch.setenv("MOM6X_MASSFLUX", "wave")
    monkeypatch.setenv("MOM6X_SUMS", sums)
    # uhbt 5 % off the layers' sum: every wavefront's solve takes at least one more sweep than the first evaluation
    evals, solves, redos = _run_case(orc, H.benchmark_small(nk=20), 0, "full", stats=True)
    assert solves > 0 and evals >= solves, (evals, solves, redos)
