"""Per-kernel timing of horizontal_viscosity at a chosen size (dev tool)."""
import sys
import torch
sys.path.insert(0, ".")
from mom6_amd import abi, grid, synth_dev
from mom6_amd.dycore import Dycore, prof_enable, prof_report, prof_reset
import bench

ni, nj, nk = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1440, 1080, 75))]
gg = bench.global_grid(ni, nj)
d, M = gg.tile(nk)
GV = abi.vgrid_default()
dyc = Dycore(d, M, GV, 0)
Md = dyc.to_dev(M)
h, u, v = synth_dev.make_state(d, Md, u_max=0.05, h_pert=0.001)
du, dv = dyc.zeros3(), dyc.zeros3()
torch.cuda.synchronize()
N3 = ni * nj * nk
for name, P in (("bench (Lap + Smag biharm, better bounds)", bench.hor_visc_params(abi, 900.0)),
                ("biharmonic only (defaults)", abi.hor_visc_params_default(900.0)),
                ("Laplacian only", abi.hor_visc_params_default(900.0, Laplacian=True, biharmonic=False))):
    if "only" in name:
        P.Kh_vel_scale = 0.01; P.Ah_vel_scale = 0.01
    dyc.hor_visc_init(P)
    dyc.horizontal_viscosity(u, v, h, du, dv); dyc.sync()
    prof_enable(dyc, True); prof_reset(dyc)
    for _ in range(3):
        dyc.horizontal_viscosity(u, v, h, du, dv)
    dyc.sync()
    rep = prof_report(dyc); prof_enable(dyc, False)
    tot = sum(v_[1] for v_ in rep.values()) / 3
    print(name, " ".join(f"{k}={v_[1] / 3:.2f}" for k, v_ in sorted(rep.items())), f"sum={tot:.2f} ms  ({40.0 * N3 / tot / 1e6:.0f} GB/s on the 40 B/cell-layer model)", flush=True)
