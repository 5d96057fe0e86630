"""Per-kernel timing of the ALE remapping entry points at a chosen size (dev tool)."""
import sys
import torch
sys.path.insert(0, ".")
from mom6_amd import abi, synth_dev
from mom6_amd.dycore import Dycore, prof_enable, prof_report, prof_reset
import bench

ni, nj, nk = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1440, 1080, 75))]
gg = bench.global_grid(ni, nj)
d, M = gg.tile(nk)
GV = abi.vgrid_default()
dyc = Dycore(d, M, GV, 0)
Md = dyc.to_dev(M)
h, u, v = synth_dev.make_state(d, Md, u_max=0.05, h_pert=0.02)
import os
GRID = os.environ.get("PROF_REMAP_GRID", "smooth")   # smooth | first
if GRID == "smooth":
    # the regridded column: same total thickness, interfaces moved by a few per cent of a layer (a z*-like adjustment)
    w = (1.0 + 0.05 * synth_dev.smooth_field(d, dyc.device, 5, nk=nk, ox=0.5, oy=0.5))
    h_new = (h * w); h_new = (h_new * (h.sum(0) / h_new.sum(0))[None]).contiguous()
else:
    # "first": the bench's ALE step (terrain-following layers onto z* with the nominal layers of the deepest column: the target
    # index of a shallow column falls far behind its source index)
    Hcol = h.sum(0)
    jm, im = divmod(int(torch.argmax(Hcol)), Hcol.shape[1])
    cr = (h[:, jm, im] / GV.Z_to_H).cpu().numpy().copy()
    RP = abi.regrid_zstar_params_default()
    h_new = torch.zeros_like(h); dzI = torch.zeros((nk + 1,) + tuple(h.shape[1:]), dtype=h.dtype, device=h.device)
    dyc.ALE_regrid_zstar(RP, cr, h, h_new, dzI)
    print("grid:", GRID, "vanished target layers:", float((h_new[:, d.joff:d.joff + d.nj, d.ioff:d.ioff + d.ni] <= 1e-9).double().mean()))
T = (10.0 + 5.0 * synth_dev.smooth_field(d, dyc.device, 71, nk=nk, ox=0.5, oy=0.5)).contiguous()
S = (35.0 + synth_dev.smooth_field(d, dyc.device, 72, nk=nk, ox=0.5, oy=0.5)).contiguous()
hu_o, hv_o, hu_n, hv_n = (torch.full_like(h, 1e-3) for _ in range(4))
torch.cuda.synchronize()
N3 = ni * nj * nk
for name, CS in (("PPM_H4 (OM4: sub-cells om4, no boundary extrapolation)", abi.remapping_params_default(abi.REMAP_PPM_H4, GV.H_subroundoff, om4_remap_via_sub_cells=1, boundary_extrapolation=0)),
                 ("PLM", abi.remapping_params_default(abi.REMAP_PLM, GV.H_subroundoff)),
                 ("PCM", abi.remapping_params_default(abi.REMAP_PCM, GV.H_subroundoff))):
    Tc, Sc, uc, vc = T.clone(), S.clone(), u.clone(), v.clone()
    dyc.ALE_remap_tracers(CS, h, h_new, [Tc.clone()]); dyc.sync()
    prof_enable(dyc, True); prof_reset(dyc)
    dyc.ALE_remap_tracers(CS, h, h_new, [Tc, Sc])
    dyc.ALE_remap_set_h_vel(h, hu_o, hv_o); dyc.ALE_remap_set_h_vel(h_new, hu_n, hv_n)
    dyc.ALE_remap_velocities(CS, hu_o, hv_o, hu_n, hv_n, uc, vc)
    dyc.sync()
    rep = prof_report(dyc); prof_enable(dyc, False)
    tot = sum(v_[1] for v_ in rep.values())
    print(name, " ".join(f"{k}={v_[1]:.2f}ms/{v_[0]}" for k, v_ in sorted(rep.items())), f"sum={tot:.2f} ms for 4 fields "
          f"({4 * 32.0 * N3 / tot / 1e6:.0f} GB/s on 32 B per cell-layer and field)", flush=True)
    assert bool(torch.isfinite(Tc).all())
    sl = (slice(d.joff, d.joff + d.nj), slice(d.ioff, d.ioff + d.ni))     # the computational domain (halos are not remapped)
    wet = Md[abi.G["mask2dT"]][sl] > 0
    a, b = (Tc * h_new).sum(0)[sl], (T * h).sum(0)[sl]
    cons = (a - b)[wet].abs().max().item() / b[wet].abs().max().item()
    print("   column-integral change (relative):", cons)
    print("   checksums T S u v:", " ".join("%016X" % (dyc.field_chksum(a) % 2 ** 64) for a in (Tc, Sc, uc, vc)))
