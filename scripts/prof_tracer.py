"""Per-kernel timing of advect_tracer / tridiagonal solvers at a chosen size (dev tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from mom6_amd import abi, grid, synth_dev
from mom6_amd.dycore import Dycore, prof_enable, prof_report, prof_reset

ni, nj, nk = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1440, 1080, 75))]
ntr = int(sys.argv[4]) if len(sys.argv) > 4 else 2
gg = grid.GlobalGrid(ni, nj, kind="spherical", lon0=0.0, lat0=-65.0, dlon=360.0 / ni, dlat=130.0 / nj,
                     reentrant_x=True, depth_fn=grid.bowl_depth(ni, nj, 4000.0, rim=2))
d, M = gg.tile(nk)
GV = abi.vgrid_default()
dyc = Dycore(d, M, GV, 0)
dyc.continuity_init(abi.continuity_params_default(nk, GV.Angstrom_H))
Md = dyc.to_dev(M)
h, u, v = synth_dev.make_state(d, Md, u_max=float(os.environ.get("PROF_UMAX", "0.05")), h_pert=float(os.environ.get("PROF_HPERT", "0.001")))   # (bench.py: 0.5, 0.01)
hp, uh, vh = (dyc.zeros3() for _ in range(3))
dt = 900.0
dyc.continuity_PPM(u, v, h, hp, uh, vh, dt)
dyc.sync()   # (the dycore has its own stream: what follows runs on torch's)
ndt = float(os.environ.get("PROF_NDT", "2"))   # dynamics steps the transports have accumulated over (bench.py: 4)
uhtr = (uh * (ndt * dt)).contiguous(); vhtr = (vh * (ndt * dt)).contiguous()
dyc.tracer_advect_init(dt, scheme=2)
tr0 = [(10.0 + 5.0 * synth_dev.smooth_field(d, dyc.device, 71 + m, nk=nk, ox=0.5, oy=0.5)).contiguous() for m in range(ntr)]
for rep in range(3):
    tr = [t.clone() for t in tr0]
    dyc.sync(); torch.cuda.synchronize()
    if rep == 2:
        prof_enable(dyc, True); prof_reset(dyc)
    t0 = time.perf_counter()
    it = dyc.advect_tracer(hp, uhtr, vhtr, ndt * dt, tr)
    dyc.sync(); t1 = time.perf_counter()
    print("advect_tracer rep", rep, "iters", it, "ms", (t1 - t0) * 1e3, flush=True)
rep = prof_report(dyc); prof_enable(dyc, False)
tot = sum(v[1] for v in rep.values())
for name, (cnt, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} n={cnt:5d} total={ms:9.3f} ms avg={ms/cnt*1e3:9.1f} us  ({100*ms/tot:5.1f}%)")
print("sum kernels ms", tot)
