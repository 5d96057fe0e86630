"""Per-mode timing of continuity_PPM's kernels at a chosen size, on both device paths (dev tool)."""
import os, sys
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
Md = dyc.to_dev(M)
if os.environ.get("PROF_STATE", "") == "coherent":   # the labelled, vertically coherent state (synth_dev.make_state_coherent)
    h, u, v = synth_dev.make_state_coherent(d, Md, u_max=float(os.environ.get("PROF_UMAX", "0.05")), h_pert=0.001)
else:
    h, u, v = synth_dev.make_state(d, Md, u_max=float(os.environ.get("PROF_UMAX", "0.05")), h_pert=0.001)
vr = torch.clamp(0.85 + 0.2 * synth_dev.smooth_field(d, dyc.device, 11, nk=nk), 0, 1)
vru = (vr * Md[abi.G["mask2dCu"]][None]).contiguous(); vrv = (vr * Md[abi.G["mask2dCv"]][None]).contiguous(); del vr
bt = BTContDev(dyc)
hp, uh, vh, ucor, vcor = (dyc.zeros3() for _ in range(5))
dt = 900.0
dyc.continuity_PPM(u, v, h, hp, uh, vh, dt)
uhbt = (uh.sum(0) * (1.0 + 0.02 * synth_dev.smooth_field(d, dyc.device, 13))).contiguous()
vhbt = (vh.sum(0) * (1.0 - 0.02 * synth_dev.smooth_field(d, dyc.device, 14))).contiguous()
torch.cuda.synchronize()   # uhbt, vhbt are made on torch's stream, the dycore has its own
modes = {
    "plain": dict(),
    "visc": dict(visc_rem_u=vru, visc_rem_v=vrv),
    "bt_cont": dict(visc_rem_u=vru, visc_rem_v=vrv, BT_cont=bt),
    "adjust": dict(visc_rem_u=vru, visc_rem_v=vrv, uhbt=uhbt, vhbt=vhbt, u_cor=ucor, v_cor=vcor),
    "full": dict(visc_rem_u=vru, visc_rem_v=vrv, uhbt=uhbt, vhbt=vhbt, u_cor=ucor, v_cor=vcor, BT_cont=bt),
}
import ctypes
wave = (dyc.cont_params.sum_order != abi.SUM_REFERENCE)   # the wave-owned kernel (default); MOM6X_SUMS=exact: the LDS kernel
timing = hasattr(dyc.lib, "mom6x_debug_mfw_timing" if wave else "mom6x_debug_mfl_timing")   # that file built with -DMOM6X_MFL_TIMING
tfun = (dyc.lib.mom6x_debug_mfw_timing if wave else dyc.lib.mom6x_debug_mfl_timing) if timing else None
PH = ["load+PPM", "bounds", "sweep0+sum", "adjust(uhbt)", "store+h_face", "adjust(du0)", "duL/duR rec", "3 trial sweeps",
      "-", "-", "row-top wait", "LDS->reg+PPM", "DMA issue", "BT stores", "-", "-"]
if wave:   # continuity_wave.hip's phases (slot 0 is not used by it)
    PH[0] = "-"; PH[4] = "h_face"; PH[3] = "adjust(uhbt)+stores"; PH[13] = "BT_cont tail"
only = os.environ.get("PROF_MODES")   # e.g. PROF_MODES=full,adjust
if only:
    modes = {k: v for k, v in modes.items() if k in only.split(",")}
for path in ("lds",) if (timing or only or wave) else ("lds", "legacy"):
    os.environ["MOM6X_MASSFLUX"] = path
    for name, kw in modes.items():
        dyc.continuity_PPM(u, v, h, hp, uh, vh, dt, **kw); dyc.sync()
        if timing:
            buf = (ctypes.c_ulonglong * 32)(); tfun(buf, 1)
        prof_enable(dyc, True); prof_reset(dyc)
        dyc.continuity_PPM(u, v, h, hp, uh, vh, dt, **kw); dyc.sync()
        rep = prof_report(dyc); prof_enable(dyc, False)
        if timing:
            tfun(buf, 1)
            for dr in (0, 1):
                sel = [q for q in range(14) if q not in (8, 9) and PH[q] != "-"]
                tot = float(sum(buf[dr * 16 + q] for q in sel)) or 1.0
                print("   phases dir", dr, " ".join(f"{PH[q]}={100 * buf[dr * 16 + q] / tot:.1f}%" for q in sel), f"total={tot:.3e} cyc")
                if wave:
                    print("   flux re-evaluations per wavefront solve, dir %d: towards uhbt %.2f, towards zero transport %.2f" %
                          (dr, buf[dr * 16 + 8] / max(buf[dr * 16 + 14], 1), buf[dr * 16 + 9] / max(buf[dr * 16 + 15], 1)))
        if hasattr(dyc.lib, "mom6x_debug_mfw_oneway"):   # built with -DMOM6X_MFW_ONEWAY_ALL=2: the launches above (warm-up + timed)
            ob = (ctypes.c_ulonglong * 2)(); dyc.lib.mom6x_debug_mfw_oneway(ob, 1)
            print("   one-way sweeps (first + Newton, wavefront-uniform): %d of %d = %.1f %%" % (ob[1], ob[0], 100.0 * ob[1] / max(ob[0], 1)))
        print(path, name, " ".join(f"{k}={v[1]:.2f}" for k, v in sorted(rep.items())), "sum=%.2f ms" % sum(v[1] for v in rep.values()), flush=True)
