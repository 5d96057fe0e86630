import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mom6_amd import abi, synth
from mom6_amd.dycore import Dycore
from tests import helpers as H
gg, d, M = H.channel(nk=20, ni=256, nj=128)
GV = abi.vgrid_default()
h, u, v = synth.make_state(d, M)
gp = np.concatenate(([9.8], np.full(d.nk - 1, 0.01)))
def used():
    free, tot = torch.cuda.mem_get_info(); return (tot - free) / 2**20
base = None
for it in range(30):
    dyc = Dycore(d, M, GV)
    dyc.sum_output_init(abi.sum_output_params_default(900.0), gp)
    hd, ud, vd = dyc.to_dev(h), dyc.to_dev(u), dyc.to_dev(v)
    torch.cuda.synchronize()
    dyc.write_energy(ud, vd, hd)
    dyc.chksum(hd, "h", haloshift=1)
    dyc.tracer_advect_init(900.0, 2)
    T = dyc.to_dev(h * 0 + 10.0)
    torch.cuda.synchronize()
    dyc.advect_tracer(hd, dyc.zeros3(), dyc.zeros3(), 900.0, [T])
    dyc.sync(); dyc.close()
    del hd, ud, vd, T
    torch.cuda.empty_cache()
    if it == 4: base = used()
print("MB used after 5 / 30 contexts:", base, used())
assert used() - base < 64, "device memory grows with every context"
print("leak check ok")
