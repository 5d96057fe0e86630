"""Several longer runs of the split RK2 step with a communicator attached whose messages all go through RCCL (send / receive
to self) -- the overlapped pass on the halo stream included -- against the oracle, bit for bit.  One GPU.
Usage: python scripts/stress_overlap.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc
orc.build()
from tests import test_rk2_gpu as T, helpers as H
from mom6_amd import parallel
import mom6_amd.dycore as D
orig = D.Dycore.initialize_dyn_split_RK2
def patched(self, params=None):
    orig(self, params)
    parallel.attach_comm(self, (1, 1), (0, 0), None, force_nccl_self=True)
D.Dycore.initialize_dyn_split_RK2 = patched
for rep in range(3):
    for cfg in (H.channel(nk=6, ni=96, nj=64), H.benchmark_small(nk=10)):
        T.run(orc, cfg, nsteps=8, bt_mod=dict(strong_drag=1))
print("stress ok")
