"""Dev tool: N dynamics steps of the tile an MI355X carries in the 4 x 2 layout of the headline grid (360 x 540 x 75, doubly
re-entrant, all eight neighbours = this rank), to be run under `rocprofv3 --kernel-trace --stats` -- the per-kernel table of
that tile (profiles/r04_tile_*).  python scripts/prof_tile.py [local_wrap|rccl_self] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "local_wrap"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10


class A:
    pass


a = A(); a.ni, a.nj, a.nk, a.dt, a.tracers = 1440 // 4, 1080 // 2, 75, 900.0, 0
dyc, d, st, taux, tauy, keep = bench.build_model(a, (1, 1), (0, 0), 0, reentrant_y=True, force_nccl_self=(mode == "rccl_self"), res_of=(1440, 1080))
torch.cuda.set_stream(dyc.torch_stream())


def step(calc=False):
    dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"], taux, tauy, a.dt, calc_dtbt=calc)


step(True); step(); step()
dyc.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
dyc.sync(); torch.cuda.synchronize()
print("tile", mode, "ms_per_step", 1e3 * (time.perf_counter() - t0) / steps, flush=True)
