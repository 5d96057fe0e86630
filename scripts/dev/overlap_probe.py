"""Dev probe: does a bandwidth-bound kernel with few registers hide under the mass-flux kernels when it runs beside them on a second
stream?  (k_bt_col<first direction> could run beside the second direction's mass-flux kernel of the predictor's first continuity
call: DESIGN.md section 7.)  Stand-in for k_bt_col: three triads over 945-MB arrays (8.5 GB, ~1.45 ms alone) through torch."""
import sys, time
import torch
sys.path.insert(0, ".")
from mom6_amd import abi, grid, synth_dev
from mom6_amd.dycore import Dycore, BTContDev

ni, nj, nk = 1440, 1080, 75
gg = grid.GlobalGrid(ni, nj, kind="spherical", lon0=0.0, lat0=-65.0, dlon=360.0 / ni, dlat=130.0 / nj,
                     reentrant_x=True, depth_fn=grid.bowl_depth(ni, nj, 4000.0, rim=2))
d, M = gg.tile(nk)
GV = abi.vgrid_default()
dyc = Dycore(d, M, GV, 0)
dyc.continuity_init(abi.continuity_params_default(nk, GV.Angstrom_H))
Md = dyc.to_dev(M)
h, u, v = synth_dev.make_state(d, Md, u_max=0.5, h_pert=0.01)
vr = torch.clamp(0.85 + 0.2 * synth_dev.smooth_field(d, dyc.device, 11, nk=nk), 0, 1)
vru = (vr * Md[abi.G["mask2dCu"]][None]).contiguous(); vrv = (vr * Md[abi.G["mask2dCv"]][None]).contiguous(); del vr
bt = BTContDev(dyc)
hp, uh, vh = (dyc.zeros3() for _ in range(3))
a, b, c = (dyc.zeros3() for _ in range(3))
side = torch.cuda.Stream()
kw = dict(visc_rem_u=vru, visc_rem_v=vrv, BT_cont=bt)

def cont():
    dyc.continuity_PPM(u, v, h, hp, uh, vh, 900.0, **kw)   # on the context's stream

def bw():
    with torch.cuda.stream(side):
        for _ in range(3):
            torch.add(b, c, alpha=3.0, out=a)

def wall(fn, n=6):
    t = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    return min(t[1:])

ta, tb = wall(cont), wall(bw)
tab = wall(lambda: (cont(), bw()))
tba = wall(lambda: (bw(), cont()))
print("continuity (4 kernels: 2 mass flux + 2 convergence) alone %.3f ms; 3 triads alone %.3f ms; together %.3f / %.3f ms (sum %.3f)" % (ta, tb, tab, tba, ta + tb))
