"""Quick per-kernel timing of the kernels that exist so far at a chosen size (dev tool)."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from mom6_amd import abi, grid, synth_dev
from mom6_amd.dycore import Dycore, BTContDev, prof_enable, prof_report, prof_reset

ni, nj, nk = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1440, 1080, 75))]
gg = grid.GlobalGrid(ni, nj, kind="spherical", lon0=0.0, lat0=-65.0, dlon=360.0 / ni, dlat=130.0 / nj,
                     reentrant_x=True, depth_fn=grid.bowl_depth(ni, nj, 4000.0, rim=2))
d, M = gg.tile(nk)
GV = abi.vgrid_default()
dyc = Dycore(d, M, GV, 0)
dyc.continuity_init(abi.continuity_params_default(nk, GV.Angstrom_H))
P = abi.barotropic_params_default(20.0)
dyc.barotropic_init(P)
Md = dyc.to_dev(M)
h, u, v = synth_dev.make_state(d, Md, u_max=0.3)
vr = torch.clamp(0.85 + 0.2 * synth_dev.smooth_field(d, dyc.device, 11, nk=nk), 0, 1)
vru = (vr * Md[abi.G["mask2dCu"]][None]).contiguous(); vrv = (vr * Md[abi.G["mask2dCv"]][None]).contiguous(); del vr
bt = BTContDev(dyc)
hp, uh, vh, ucor, vcor = (dyc.zeros3() for _ in range(5))
pbce = torch.empty_like(h)
for k in range(nk):
    pbce[k] = 9.8 + 0.01 * k
eta = ((h.sum(0) - Md[abi.G["bathyT"]]) * Md[abi.G["mask2dT"]]).contiguous()
taux = (0.1 * synth_dev.smooth_field(d, dyc.device, 41) * Md[abi.G["mask2dCu"]]).contiguous(); tauy = torch.zeros_like(taux)
bcu = (1e-6 * u).contiguous(); bcv = (1e-6 * v).contiguous()
alu, alv = dyc.zeros3(), dyc.zeros3(); eta_out, uhbtav, vhbtav, etaav = (dyc.zeros2() for _ in range(4))
dt = 900.0
torch.cuda.synchronize()
dyc.continuity_PPM(u, v, h, hp, uh, vh, dt, visc_rem_u=vru, visc_rem_v=vrv, BT_cont=bt)
dyc.btcalc(h, bt["h_u"], bt["h_v"])
dtbt = dyc.set_dtbt(gtot_est=9.8 + 0.01 * nk, SSH_add=10.0)
dyc.bt_mass_source(h, eta, True)
print("dtbt", dtbt, "nstep", int(np.ceil(dt / dtbt)), flush=True)

def step():
    dyc.continuity_PPM(u, v, h, hp, uh, vh, dt, visc_rem_u=vru, visc_rem_v=vrv, BT_cont=bt)
    dyc.btstep(u, v, eta, dt, bcu, bcv, taux, tauy, pbce, eta, u, v, alu, alv, eta_out, uhbtav, vhbtav, vru, vrv, bt,
               uh0=uh, vh0=vh, u_uh0=u, v_vh0=v, etaav=etaav)
    dyc.continuity_PPM(u, v, h, hp, uh, vh, dt, uhbt=uhbtav, vhbt=vhbtav, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor, v_cor=vcor, BT_cont=bt)
step(); dyc.sync()
t0 = time.time(); step(); dyc.sync(); t1 = time.time()
print("partial step wall ms", (t1 - t0) * 1e3, flush=True)
prof_enable(dyc, True); prof_reset(dyc)
step(); dyc.sync()
rep = prof_report(dyc)
tot = sum(v[1] for v in rep.values())
N3 = ni * nj * nk; N2 = ni * nj
for name, (cnt, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} n={cnt:5d} total={ms:9.3f} ms avg={ms/cnt*1e3:9.1f} us  ({100*ms/tot:5.1f}%)")
print("sum kernels ms", tot)
print("eta_out range", float(eta_out.min()), float(eta_out.max()), "h min", float(hp.min()))
