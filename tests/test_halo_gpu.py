"""GPU checks of the halo-exchange path (mom6_amd/csrc/halo.hip) that can run on ONE GPU:
pack -> (device copy | ncclSend/ncclRecv to self) -> unpack must reproduce the single-tile wrap bit for bit,
and the RK2 step with a communicator attached must equal the step without one."""
import numpy as np
import pytest

from mom6_amd import abi, parallel, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _fields(dyc, d, seed=0):
    import torch
    rng = np.random.default_rng(seed)
    shapes = [d.shape3(), d.shape3(), d.shape3(), d.shape2(), d.shape2(), d.shape3()]
    stg = [0, 1, 2, 3, 0, 3]
    return [dyc.to_dev(rng.standard_normal(s)) for s in shapes], stg


@pytest.mark.parametrize("force_nccl", [False, True])
@pytest.mark.parametrize("cfg", ["channel", "double_gyre"])
def test_exchange_equals_wrap(cfg, force_nccl):
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = getattr(H, cfg)()
    ref = Dycore(d, M)
    f_ref, stg = _fields(ref, d)
    torch.cuda.synchronize()
    parallel.pass_fields(ref, f_ref, stg)          # no communicator: wrap kernels (or nothing for a closed basin)
    ref.sync()
    dyc = Dycore(d, M)
    parallel.attach_comm(dyc, (1, 1), (0, 0), None, force_nccl_self=force_nccl)
    f, _ = _fields(dyc, d)
    torch.cuda.synchronize()
    parallel.pass_fields(dyc, f, stg)
    dyc.sync()
    for a, b in zip(f, f_ref):
        assert torch.equal(a, b)
    ref.close(); dyc.close()


def test_rk2_step_with_comm_attached(orc):
    from tests import test_rk2_gpu as T
    import mom6_amd.dycore as D
    orig = D.Dycore.initialize_dyn_split_RK2

    def patched(self, params=None):
        orig(self, params)
        parallel.attach_comm(self, (1, 1), (0, 0), None, force_nccl_self=True)
    D.Dycore.initialize_dyn_split_RK2 = patched
    try:
        T.run(orc, H.channel(), nsteps=2, bt_mod=dict(strong_drag=1))
    finally:
        D.Dycore.initialize_dyn_split_RK2 = orig


@pytest.mark.parametrize("halo", [4, 8])
def test_rk2_steps_on_a_torus_with_every_pass_through_rccl(orc, halo):
    """A doubly re-entrant tile whose eight neighbours are the tile itself, every group pass through ncclSend / ncclRecv on the second
    stream: the 3-D passes of the RK2 step around the split mass-flux launches, and -- the barotropic solver's own pass of eta, ubt,
    vbt (MOM_barotropic.F90:2505-2512) -- packed on the compute stream and travelling while the own-points half of the next sub-step
    runs (k_bt_substep / k_bt_pred sel 1, then sel 2).  Three steps equal the oracle's bit for bit; and the split did happen (the
    sub-step kernels ran more often than there are sub-steps)."""
    from tests import test_rk2_gpu as T
    import mom6_amd.dycore as D
    from mom6_amd.dycore import prof_enable, prof_report
    orig = D.Dycore.initialize_dyn_split_RK2
    seen = {}

    def patched(self, params=None):
        orig(self, params)
        parallel.attach_comm(self, (1, 1), (0, 0), None, force_nccl_self=True)
        self.comm_overlap_btstep(True)

    def watch(dyc, when):
        if when == "before":
            prof_enable(dyc, True)
        else:
            dyc.sync(); seen.update(prof_report(dyc)); prof_enable(dyc, False)
    D.Dycore.initialize_dyn_split_RK2 = patched
    try:
        T.run(orc, H.torus(halo=halo), nsteps=3, bt_mod=dict(strong_drag=1), hook=watch)
    finally:
        D.Dycore.initialize_dyn_split_RK2 = orig
    sub = seen.get("k_bt_substep<u>", (0, 0))[0] + seen.get("k_bt_substep<v>", (0, 0))[0]
    pred = seen.get("k_bt_pred", (0, 0))[0]
    assert sub > 0 and pred >= 2 * 6, sorted(seen)       # (every travelling pass of the loop makes two k_bt_pred launches)
