import sys, os
sys.path.insert(0, ".")
import torch
from mom6_amd import abi, synth_dev
from mom6_amd.dycore import Dycore, prof_enable, prof_report, prof_reset
import bench
ni, nj, nk = 1440, 1080, 75
gg = bench.global_grid(ni, nj); d, M = gg.tile(nk); GV = abi.vgrid_default()
dyc = Dycore(d, M, GV, 0); Md = dyc.to_dev(M)
h, u, v = synth_dev.make_state(d, Md, u_max=0.05, h_pert=0.02)
Hcol = h.sum(0); jm, im = divmod(int(torch.argmax(Hcol)), Hcol.shape[1])
cr = (h[:, jm, im] / GV.Z_to_H).cpu().numpy().copy()
RP = abi.regrid_zstar_params_default()
h_new = torch.zeros_like(h); dzI = torch.zeros((nk + 1,) + tuple(h.shape[1:]), dtype=h.dtype, device=h.device)
dyc.ALE_regrid_zstar(RP, cr, h, h_new, dzI)
hu_o, hv_o, hu_n, hv_n = (torch.full_like(h, 1e-3) for _ in range(4))
CS = abi.remapping_params_default(abi.REMAP_PPM_H4, GV.H_subroundoff, om4_remap_via_sub_cells=1, boundary_extrapolation=0)
torch.cuda.synchronize()
for mode in ("separate", "from_h", "separate", "from_h"):
    uc, vc = u.clone(), v.clone()
    dyc.sync(); torch.cuda.synchronize()
    prof_enable(dyc, True); prof_reset(dyc)
    if mode == "separate":
        dyc.ALE_remap_set_h_vel(h, hu_o, hv_o); dyc.ALE_remap_set_h_vel(h_new, hu_n, hv_n)
        dyc.ALE_remap_velocities(CS, hu_o, hv_o, hu_n, hv_n, uc, vc)
    else:
        dyc.ALE_remap_velocities_from_h(CS, h, h_new, uc, vc)
    dyc.sync()
    rep = prof_report(dyc); prof_enable(dyc, False)
    print(mode, " ".join(f"{k}={v_[1]:.2f}ms/{v_[0]}" for k, v_ in sorted(rep.items())), "sum=%.2f" % sum(v_[1] for v_ in rep.values()), flush=True)
