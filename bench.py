#!/usr/bin/env python
"""bench.py -- the headline benchmark of BASELINE.json: simulated days per wall-second of the MOM6
split-explicit dynamical core (step_MOM_dyn_split_RK2) on a synthetic 0.25-degree-class
1440 x 1080 x 75 grid, at 1 / 2 / 4 / 8 MI355X (strong scaling: the global grid is fixed and cut into
MOM6's 2-D tile layout, one tile per GPU).

A "step" is ONE baroclinic step (DT = 900 s) of the headline configuration of SURVEY.md 8(d): step_MOM_dyn_split_RK2
(PressureForce, CorAdCalc x2, continuity_PPM x3, btstep x2 -- each a full barotropic sub-cycle --, horizontal_viscosity,
vertvisc_coef x3, vertvisc x2, vertvisc_remnant x3 and the RK2 glue) PLUS its share of the thermodynamic step that
follows every DT_THERM / DT = 4 of them: advect_tracer of T, S and two passive tracers (PPM) with the transports
accumulated over the four steps, tracer_vertdiff of the passive tracers and triDiagTS of T, S.  The thermodynamic
step runs INSIDE the timed region after every fourth dynamics step, so `value` = simulated time / wall time of the
whole cycle (K a multiple of 4 amortises exactly).  State resides in HBM; every callee runs on the device; only the
set_viscous_BBL inputs of vertvisc_coef and the diffusive exchanges ea, eb of the tridiagonal solves are synthetic
constants (stated in `config`).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel -- the one with the largest
total time in the last warm-up step (k_mass_flux_lds: PPM reconstruction + zonal/meridional mass flux +
Newton flux adjustment + BT_cont fits) -- timed live with HIP events on the compute stream inside the
timed region; `cpu_baseline` is the oracle (plain-C port of the reference algorithm, OpenMP over the loops the
reference threads, all host cores) timed on a bounded 360x180x75 tile of the same workload and scaled by the cell count.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAYOUTS = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak (6.3 TB/s achievable)


def global_grid(ni, nj, reentrant_y=False, res_of=None):
    """res_of = (NI, NJ): the grid spacing of an NI x NJ global grid on this (smaller) ni x nj patch -- the comm_model leg's tile has
    the headline's resolution, hence its barotropic time step and number of sub-steps."""
    from mom6_amd import grid
    NI, NJ = res_of if res_of else (ni, nj)
    return grid.GlobalGrid(ni, nj, kind="spherical", lon0=0.0, lat0=-65.0 if not res_of else -32.5, dlon=360.0 / NI, dlat=130.0 / NJ,
                           reentrant_x=True, reentrant_y=reentrant_y, depth_fn=grid.bowl_depth(ni, nj, 4000.0, rim=0 if reentrant_y else 2))


def algorithmic_bytes_per_step(N3, N2, nsub_total):
    """SURVEY.md section 8(d) / BASELINE.md section 3: compulsory FP64 traffic of one baroclinic step."""
    per_N3 = {"continuity_PPM x3": 256, "CorAdCalc x3": 168, "PressureForce": 32, "btstep 3-D setup/teardown x2": 272,
              "btcalc + bt_mass_source": 24, "vertvisc x2 + remnant x3": 480, "RK2 pointwise glue": 384,
              # not in SURVEY 8(d) (a "next" row there): u, h in; a_u, h_u out, per direction and call
              "vertvisc_coef x3": 192,
              # also a "next" row: u, v, h in; diffu, diffv out
              "horizontal_viscosity": 40}
    total = sum(per_N3.values()) * N3 + 570.0 * N2 * nsub_total
    return total, per_N3


def hor_visc_params(abi, dt):
    """OM4_025-class lateral friction switches: LAPLACIAN with KH_VEL_SCALE, BIHARMONIC with AH_VEL_SCALE and
    SMAGORINSKY_AH (SMAG_BI_CONST = 0.06), the better bounds on both (defaults)."""
    P = abi.hor_visc_params_default(dt, Laplacian=True, biharmonic=True)
    P.Kh_vel_scale = 0.01; P.Ah_vel_scale = 0.01; P.Smagorinsky_Ah = 1; P.Smag_bi_const = 0.06
    return P


def build_model(args, layout, pe, device, dist=None, unique_id=None, reentrant_y=False, force_nccl_self=False, bthalo=0, res_of=None):
    """Create the device model for tile `pe` of `layout` and a synthetic state in HBM.  With more than one tile the
    communicator is attached before anything exchanges halos (the new-run initialisation does)."""
    import torch
    from mom6_amd import abi, synth_dev
    from mom6_amd.dycore import Dycore
    G = abi.G
    gg = global_grid(args.ni, args.nj, reentrant_y, res_of)
    # BTHALO > NIHALO = 4 (MOM_barotropic.F90:5446-5461): the context carries the wide halo, the barotropic solver exchanges every
    # BTHALO sub-steps, the 3-D passes of the step stay at 4 rows
    halo = max(4, int(bthalo or getattr(args, "bthalo", 0) or 0))
    d, M = gg.tile(args.nk, halo, layout, pe)
    GV = abi.vgrid_default()
    dyc = Dycore(d, M, GV, 0, device)
    if halo > 4:
        dyc.set_dyn_pass_width(4)
    if layout != (1, 1) or force_nccl_self:
        from mom6_amd.parallel import attach_comm
        attach_comm(dyc, layout, pe, dist, unique_id=unique_id, force_nccl_self=force_nccl_self)
    dyc.continuity_init(abi.continuity_params_default(args.nk, GV.Angstrom_H))
    bt = abi.barotropic_params_default(20.0)
    bt.BTHALO = halo if halo > 4 else 0
    dyc.barotropic_init(bt)
    dyc.CoriolisAdv_init(abi.coriolis_params_default())
    Rlay, gp = abi.layer_densities(args.nk, GV.Rho0, GV.g_Earth)
    dyc.PressureForce_init(abi.pgf_params_default(GV.Rho0), Rlay, gp)
    dyc.initialize_dyn_split_RK2(abi.rk2_params_default())
    Md = dyc.to_dev(M)
    # |u| <= 0.5 m/s (the seeded smooth fields have values in about [-1, 1]), thicknesses perturbed by 1 %
    h, u, v = synth_dev.make_state(d, Md, u_max=float(os.environ.get("MOM6X_BENCH_UMAX", "0.5")), h_pert=0.01)
    st = dict(u=u, v=v, h=h, uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
    # vertvisc_coef runs on the device three times per step (RK2.F90:609, :738, :1003); its vertvisc_type inputs
    # (set_viscous_BBL outputs: a drag-law bottom viscosity over a 10 m boundary layer) are synthetic and constant
    dyc.vertvisc_init(abi.vertvisc_params_default(Kv=1.0e-4, Hmix=20.0, Hbbl=10.0))
    Kv_bbl_u = (2.0e-3 * (1.0 + 0.5 * synth_dev.smooth_field(d, dyc.device, 91, ox=1.0, oy=0.5)) * Md[G["mask2dCu"]]).contiguous()
    Kv_bbl_v = (2.0e-3 * (1.0 + 0.5 * synth_dev.smooth_field(d, dyc.device, 92, ox=0.5, oy=1.0)) * Md[G["mask2dCv"]]).contiguous()
    bbl_u = torch.full_like(Kv_bbl_u, 10.0); bbl_v = torch.full_like(Kv_bbl_v, 10.0)
    dyc.vertvisc_set_visc(Kv_bbl_u, Kv_bbl_v, bbl_u, bbl_v)
    dyc.vertvisc_coef(u, v, h, args.dt)
    dyc.hor_visc_init(hor_visc_params(abi, args.dt))   # the step and the new-run initialisation call horizontal_viscosity
    taux = (0.1 * synth_dev.smooth_field(d, dyc.device, 41, ox=1.0, oy=0.5) * Md[G["mask2dCu"]]).contiguous()
    tauy = torch.zeros_like(taux)
    torch.cuda.synchronize()
    dyc.dyn_split_RK2_new_run(st["u"], st["v"], st["h"], st["uh"], st["vh"], args.dt)
    dyc.sync()
    keep = (Kv_bbl_u, Kv_bbl_v, bbl_u, bbl_v, Md)
    return dyc, d, st, taux, tauy, keep


# 8-byte words of unavoidable HBM traffic per cell-layer and launch of the kernels that can dominate a step
# (reads + writes of 3-D arrays; 2-D planes are noise at nk = 75).  DESIGN.md section 5 derives them.
KERNEL_WORDS = {
    # h, u, visc_rem in; uh out; plus u_cor (calls with uhbt) or BT_cont%h_u (the call that sets BT_cont): 5 either way
    "k_mass_flux_lds": 5.0,
    # the same routine with one wavefront row per face column (sum_order TREE16, the default).  Mean over a step's three launches
    # (continuity x3 = (8+0+2) + (8+2+2) + (8+2+0) words for both directions): h, u, visc_rem in + uh out, + h_u (the
    # call that sets BT_cont) = 5; + u_cor AND h_u (the predictor's second call) = 6; + u_cor (the corrector's) = 5
    "k_mass_flux_wave": 16.0 / 3.0,
    # thread-per-column path (MOM6X_MASSFLUX=legacy): u, visc_rem, h, (h_L, h_R from k_edge) in; uh [+ u_cor] out
    "k_mass_flux<": 14.0 / 3.0,
    "k_vertvisc_coef_cols": 22.0 / 3.0,   # the step's three calls: u, u_bc, h in + visc_rem out = 4; + pbce in, the estimate out and in, u out = 8; + a_u, h_u out = 10
    "k_vertvisc_remnant": 3.0,   # a(k), h in; visc_rem out
    "k_vertvisc<": 4.0,          # u, a(k), h in; u out
    "k_bc_accel": 7.0,           # CAu, PFu, diffu in + u_bc_accel out, both directions less shared reads
    "k_vel_update": 6.0,
    "k_layer_accel": 5.0,
    "k_convergence": 3.0,        # h_in, uh in; h out
    "k_hv_stress": 7.0,          # h, sh_xx, sh_xy, Del2u, Del2v in; str_xx, str_xy out
}


def synthetic_entrainment(h):
    """ea, eb of triDiagTS / tracer_vertdiff [H]: a diffusive exchange of 1e-3 h across every interior interface -- what layer k
    takes from below (eb(k)) is what layer k+1 gives up to above (ea(k+1)), nothing crosses the surface or the bottom, so the
    solves conserve the column integrals."""
    import torch
    ea = (1.0e-3 * h).contiguous()
    ea[0].zero_()
    eb = torch.roll(ea, -1, 0).contiguous()
    eb[-1].zero_()
    return ea, eb


def make_thermo(args, dyc, d, st, nth):
    """The thermodynamic step of the headline configuration (DT_THERM = 3600 s = 4 DT, T and S plus two
    passive tracers): advect_tracer (PPM) moves all of them with the transports uhtr / vhtr accumulated over the last nth
    dynamics steps (and clears the accumulators, as step_MOM does), tracer_vertdiff solves the passive tracers' vertical
    tridiagonal systems and triDiagTS those of T and S.  Returns (the step, a function that reports what it did)."""
    import torch
    from mom6_amd import synth_dev
    ntr, nk = max(args.tracers, 0), args.nk
    dt_th = nth * args.dt
    dyc.tracer_advect_init(args.dt, scheme=2)                          # TRACER_ADVECTION_SCHEME = "PPM"
    T = (10.0 + synth_dev.smooth_field(d, dyc.device, 81, nk=nk)).contiguous()
    S = (35.0 + 0.5 * synth_dev.smooth_field(d, dyc.device, 82, nk=nk)).contiguous()
    tr = [(10.0 + 5.0 * synth_dev.smooth_field(d, dyc.device, 71 + m, nk=nk, ox=0.5, oy=0.5)).contiguous() for m in range(ntr)]
    ea, eb = synthetic_entrainment(st["h"])
    stat = {"calls": 0, "iters": 0}

    def thermo():
        h = st["h"]
        stat["iters"] += dyc.advect_tracer(h, st["uhtr"], st["vhtr"], dt_th, [T, S] + tr)
        for t in tr:
            dyc.tracer_vertdiff(h, ea, eb, dt_th, t)
        dyc.triDiagTS(h, ea, eb, T, S)
        st["uhtr"].zero_(); st["vhtr"].zero_()                            # (on the context's stream: see main)
        stat["calls"] += 1

    def info():
        # timed on its own after the run, for the breakdown only (the headline already contains it)
        for _ in range(nth):
            dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"],
                                       *info.forcing, args.dt)
        dyc.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        it = dyc.advect_tracer(st["h"], st["uhtr"], st["vhtr"], dt_th, [T, S] + tr)
        dyc.sync(); t_adv = time.perf_counter() - t0; t0 = time.perf_counter()
        for t in tr:
            dyc.tracer_vertdiff(st["h"], ea, eb, dt_th, t)
        dyc.triDiagTS(st["h"], ea, eb, T, S)
        dyc.sync(); t_tri = time.perf_counter() - t0
        if getattr(args, "breakdown", False):   # the kernels of one more thermodynamic call (HIP events around every launch)
            from mom6_amd.dycore import prof_enable, prof_report, prof_reset
            prof_enable(dyc, True); prof_reset(dyc)
            dyc.advect_tracer(st["h"], st["uhtr"], st["vhtr"], dt_th, [T, S] + tr)
            for t in tr:
                dyc.tracer_vertdiff(st["h"], ea, eb, dt_th, t)
            dyc.triDiagTS(st["h"], ea, eb, T, S)
            dyc.sync()
            for name, (cnt, ms) in sorted(prof_report(dyc).items(), key=lambda kv: -kv[1][1]):
                print(f"thermo: {name:28s} n={cnt:5d} total={ms:9.3f} ms", file=sys.stderr)
            prof_enable(dyc, False)
        st["uhtr"].zero_(); st["vhtr"].zero_()
        N3 = args.ni * args.nj * args.nk
        nf = ntr + 2
        # compulsory words per cell-layer: set-up (h_end, uhtr, vhtr in; hprev, uhr, vhr out) + per iteration and direction
        # (uhr, hprev in/out + every tracer in/out); tridiagonal: h, ea, eb + each field in/out (T and S share one sweep)
        b_adv = 8.0 * N3 * (6 + it * 2 * (4 + 2 * nf)); b_tri = 8.0 * N3 * (ntr * 5 + 7)
        return {"fields": "T, S + %d passive tracers" % ntr, "scheme": "PPM", "dynamics_steps_per_thermo_step": nth,
                "calls_in_run": stat["calls"], "advect_iterations_mean": round(stat["iters"] / max(stat["calls"], 1), 2),
                "advect_tracer_ms": round(1e3 * t_adv, 3), "advect_algorithmic_GB": round(b_adv / 1e9, 2),
                "advect_frac_of_hbm_peak": round(b_adv / 1e9 / t_adv / (HBM_PEAK_GBS * args.gpus), 4),
                "tridiag_ms": round(1e3 * t_tri, 3), "tridiag_algorithmic_GB": round(b_tri / 1e9, 2),
                "tridiag_frac_of_hbm_peak": round(b_tri / 1e9 / t_tri / (HBM_PEAK_GBS * args.gpus), 4),
                "ms_per_dynamics_step_amortised": round(1e3 * (t_adv + t_tri) / nth, 3),
                "note": "INSIDE the headline metric (one call per %d dynamics steps); this breakdown is one more call timed after the run" % nth}
    return thermo, info


def ale_remap_leg(args, dyc, d, st, barrier, dist):
    """An ALE step (BASELINE.json configs[4]: "ALE remap included") after the timed region, NOT part of `value`:
    ALE_regrid for the z* coordinate, then T and S remapped with OM4's switches (PPM_H4, OM4 sub-cells, no boundary
    extrapolation) from the model's layers to the new grid, then u and v with ALE_remap_set_h_vel's face thicknesses."""
    import torch
    from mom6_amd import abi, synth_dev
    nk = args.nk
    CS = abi.remapping_params_default(abi.REMAP_PPM_H4, dyc.GV.H_subroundoff, om4_remap_via_sub_cells=1, boundary_extrapolation=0)
    h = st["h"]
    # the z* coordinate whose nominal layers are those of the deepest column (regridding then moves the interfaces of the
    # shallower columns and follows the free surface everywhere)
    Hcol = h.sum(0)
    jm, im = divmod(int(torch.argmax(Hcol)), Hcol.shape[1])
    cr = (h[:, jm, im] / dyc.GV.Z_to_H).cpu().numpy().copy()
    RP = abi.regrid_zstar_params_default()
    h_new = torch.zeros_like(h); dzI = torch.zeros((nk + 1,) + tuple(h.shape[1:]), dtype=h.dtype, device=h.device)
    T = (10.0 + 5.0 * synth_dev.smooth_field(d, dyc.device, 71, nk=nk, ox=0.5, oy=0.5)).contiguous()
    S = (35.0 + synth_dev.smooth_field(d, dyc.device, 72, nk=nk, ox=0.5, oy=0.5)).contiguous()
    u, v = st["u"].clone(), st["v"].clone()
    dyc.ALE_regrid_zstar(RP, cr, h, h_new, dzI)
    dyc.ALE_remap_tracers(CS, h, h_new, [T.clone(), S.clone()])         # untimed: allocates the work arrays (two fields share a merge)
    barrier(); t0 = time.perf_counter()
    dyc.ALE_regrid_zstar(RP, cr, h, h_new, dzI)
    dyc.ALE_remap_tracers(CS, h, h_new, [T, S])
    dyc.ALE_remap_velocities_from_h(CS, h, h_new, u, v)      # ALE_remap_set_h_vel x 2 + ALE_remap_velocities (MOM.F90's three calls in a row)
    barrier(); t = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([t], dtype=torch.float64, device=dyc.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    N3 = args.ni * args.nj * args.nk
    # regrid: h in, h_new and dzRegrid out; per field: h_old, h_new, field in, field out; set_h_vel x2: h in, h_u, h_v out (the
    # routines' own un-fused count, as SURVEY.md 8(d) counts: the device no longer writes h_u / h_v at all)
    b = 8.0 * N3 * (3 + 4 * 4 + 2 * 3)
    return {"fields": "T, S, u, v", "scheme": "PPM_H4 (OM4 sub-cells, no boundary extrapolation)", "remap_ms": round(1e3 * t, 3),
            "algorithmic_GB": round(b / 1e9, 2), "GBps": round(b / 1e9 / t, 1), "frac_of_hbm_peak": round(b / 1e9 / t / (HBM_PEAK_GBS * args.gpus), 4),
            "regrid": "ZSTAR, nominal layers of the deepest column",
            "note": "reported next to, not inside, the headline metric: ALE_regrid (z*) + remapping of T, S, u, v, all on the device"}


def diag_leg(args, dyc, d, st, barrier, dist):
    """The regression artefacts of one output interval, after the timed region and NOT part of `value`: write_energy (the
    line of ocean.stats: 3-D reproducing sums of mass, APE and KE, the CFL maxima) and the checksum lines of u, v, h
    (uvchksum + hchksum) plus their restart checksums, all formed on the device-resident state."""
    import torch
    from mom6_amd import abi, sum_output
    GV = dyc.GV
    Rlay, gp = abi.layer_densities(args.nk, GV.Rho0, GV.g_Earth)
    dyc.sum_output_init(abi.sum_output_params_default(args.dt), gp)
    so = sum_output.SumOutput()
    dyc.write_energy(st["u"], st["v"], st["h"])                          # untimed: allocates the work arrays
    barrier(); t0 = time.perf_counter()
    sums = dyc.write_energy(st["u"], st["v"], st["h"])
    barrier(); t_en = time.perf_counter() - t0
    out_line, _ = so.record(sums, 0.0, 0)
    barrier(); t0 = time.perf_counter()
    lines = [dyc.chksum_lines(st["u"], "u", "u", symmetric=True)[1], dyc.chksum_lines(st["v"], "v", "v", symmetric=True)[1],
             dyc.chksum_lines(st["h"], "h", "h")[1]]
    chk = [dyc.field_chksum(st[n]) for n in ("u", "v", "h")]
    barrier(); t_ck = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([t_en, t_ck], dtype=torch.float64, device=dyc.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_en, t_ck = float(tt[0].item()), float(tt[1].item())
    N3 = args.ni * args.nj * args.nk
    b_en = 8.0 * N3 * (1 + 2 + 3 + 2)        # mass: h; APE: h in, PE_pt out and in; KE: h, u, v; CFL: u, v
    b_ck = 8.0 * N3 * (3 * 3 + 3)            # per field: sum + min/max pass, bit count(s); restart checksum
    return {"write_energy_ms": round(1e3 * t_en, 3), "write_energy_GBps": round(b_en / 1e9 / t_en, 1),
            "write_energy_frac_of_hbm_peak": round(b_en / 1e9 / t_en / (HBM_PEAK_GBS * args.gpus), 4),
            "chksum_ms": round(1e3 * t_ck, 3), "chksum_GBps": round(b_ck / 1e9 / t_ck, 1), "stdout_line": out_line,
            "h_chksum_line": lines[2], "restart_checksum_h": "%016X" % (chk[2] % 2 ** 64),
            "note": "reported next to, not inside, the headline metric: one ocean.stats line (reproducing sums) and the debugging / "
                    "restart checksums of u, v, h from the device-resident state; host round trips included"}


# Which of the 1616 x N3 bytes of the un-fused model the device does NOT move, by routine (8-byte words per cell-layer and step;
# DESIGN.md section 4).  The model keeps them so that `frac_of_peak` stays comparable between rounds; the measured
# FETCH_SIZE + WRITE_SIZE of the same command is printed next to it.
FUSED_WORDS = {
    "vertvisc x2 + remnant x3 and the velocity updates of the RK2 glue": "k_vertvisc_fused does the RK2 velocity update, vertvisc and vertvisc_remnant "
        "in one column sweep: 11 words per face-layer and call where the separate routines move 17 (DESIGN.md section 4)",
    "vertvisc_coef x3": "round 5: k_vertvisc_coef_cols forms the coefficients in the kernel that solves with them (a_u in registers, h_u in LDS): "
        "4 / 8 / 10 words per face-layer for the step's three calls where coefficient kernel + solve moved 8 / 12 / 12",
    "continuity_PPM": "the PPM edge values h_W/h_E/h_S/h_N, the Newton iterations' layer transports and the flux thicknesses never leave the chip "
        "(they were never part of the 256 B x N3 count either): 5 words per face-layer and launch",
    "everything else": "moved as counted; the 2-D metric planes a kernel re-reads per layer or per 15-layer chunk come on top (not in the model)",
}


def ale_cycle(args, device=0, steps=4, warm=1, check=None, layout=(1, 1), pe=(0, 0), unique_id=None, env=None):
    """BASELINE.json configs[4] on ONE of its 4 x 2 tiles (1080 x 1620 x 75 of the 4320 x 3240 grid: the per-GPU share of the
    8-GPU run, as one stand-alone grid -- eight tiles at 53 GB each do not fit one GPU): an ALE cycle of step_MOM, i.e.
    `steps` dynamics steps with the pressure force of an ALE grid (tv%T, tv%S, LINEAR equation of state, PLM reconstruction:
    PRESSURE_RECONSTRUCTION_SCHEME = 1), then advect_tracer of T, S and two passive tracers with the accumulated transports,
    their vertical tridiagonal solves, ALE_regrid to z* and the remapping of T, S, the tracers, u, v (PPM_H4) and of the
    auxiliary restart variables.  Returns (seconds per cycle, dict of what was measured); `check(state)` is called after the
    last cycle with the live fields (tests/test_configs_gpu.py)."""
    import torch
    from mom6_amd import abi, parallel, synth_dev

    class A:
        pass
    a = A()
    a.ni, a.nj, a.nk, a.dt, a.tracers = args.ale_ni, args.ale_nj, args.nk, args.dt, 2
    dyc, d, st, taux, tauy, keep = build_model(a, layout, pe, device, None, unique_id)
    torch.cuda.set_stream(dyc.torch_stream())
    nk = a.nk
    T = (20.0 - 15.0 * torch.arange(nk, device=dyc.device, dtype=torch.float64)[:, None, None] / max(nk - 1, 1) +
         0.8 * synth_dev.smooth_field(d, dyc.device, 7, nk=nk, ox=0.5, oy=0.5)).contiguous()
    S = (34.0 + 1.0 * torch.arange(nk, device=dyc.device, dtype=torch.float64)[:, None, None] / max(nk - 1, 1) +
         0.2 * synth_dev.smooth_field(d, dyc.device, 8, nk=nk, ox=0.5, oy=0.5)).contiguous()
    eos = abi.eos_params_default(abi.LINEAR); eos.Recon_Scheme = 1
    dyc.PressureForce_set_tv(T, S, eos)
    tr = [(10.0 + 5.0 * synth_dev.smooth_field(d, dyc.device, 71 + m, nk=nk, ox=0.5, oy=0.5)).contiguous() for m in range(2)]
    ea, eb = synthetic_entrainment(st["h"])
    dyc.tracer_advect_init(a.dt, scheme=2)
    CSr = abi.remapping_params_default(abi.REMAP_PPM_H4, dyc.GV.H_subroundoff, om4_remap_via_sub_cells=1, boundary_extrapolation=0)
    RP = abi.regrid_zstar_params_default()
    # the z* coordinate whose nominal layers are nk equal parts of the basin's maximum depth (the same on every tile of a layout)
    cr = np.full(nk, 4000.0 / nk)
    h_new = torch.zeros_like(st["h"]); dzI = torch.zeros((nk + 1,) + tuple(st["h"].shape[1:]), dtype=torch.float64, device=dyc.device)
    info = {}

    def cycle(first=False):
        for n in range(steps):
            dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"], taux, tauy,
                                       a.dt, calc_dtbt=(first and n == 0))
        info["advect_iterations"] = dyc.advect_tracer(st["h"], st["uhtr"], st["vhtr"], steps * a.dt, [T, S] + tr)
        for t in tr:
            dyc.tracer_vertdiff(st["h"], ea, eb, steps * a.dt, t)
        dyc.triDiagTS(st["h"], ea, eb, T, S)
        if check is not None:   # the column sums of the accumulated transports: what the cycle may change a column's thickness by
            U, V = st["uhtr"].sum(0), st["vhtr"].sum(0)
            info["_col_transport_div"] = (U - torch.roll(U, 1, 1)) + (V - torch.roll(V, 1, 0))
        st["uhtr"].zero_(); st["vhtr"].zero_()
        # ALE_regridding_and_remapping (MOM.F90:1751): new z* grid, remap everything that lives on the old one
        dyc.ALE_regrid_zstar(RP, cr, st["h"], h_new, dzI)
        dyc.ALE_remap_tracers(CSr, st["h"], h_new, [T, S] + tr)
        dyc.ALE_remap_velocities_from_h(CSr, st["h"], h_new, st["u"], st["v"])   # ALE_remap_set_h_vel x 2 + ALE_remap_velocities
        st["h"].copy_(h_new)
        # step_MOM_thermo ends with the group pass of everything the thermodynamics and the remapping changed (MOM.F90:
        # do_group_pass(pass_uv_T_S_h)): the next dynamics step reads T, S, h of the neighbouring tiles' edge columns
        parallel.pass_fields(dyc, [st["u"], st["v"], st["h"], T, S] + tr, [1, 2, 0, 0, 0] + [0] * len(tr))

    def meet():
        dyc.sync(); torch.cuda.synchronize()
        if env is not None:
            env.barrier()

    torch.cuda.synchronize()
    for w in range(warm):
        cycle(first=(w == 0))
    meet()
    pre = dict(T=T.clone(), h=st["h"].clone(), tr0_min=tr[0].min().item(), tr0_max=tr[0].max().item()) if check is not None else None
    t0 = time.perf_counter()
    cycle()
    meet()
    sec = time.perf_counter() - t0
    # the restart checksums of the remapped state (collective: summed over the tiles of the layout)
    info["restart_checksums"] = {n: "%016X" % (dyc.field_chksum(x) % 2 ** 64) for n, x in
                                 (("u", st["u"]), ("v", st["v"]), ("h", st["h"]), ("T", T), ("S", S), ("tr1", tr[0]), ("tr2", tr[1]))}
    info.update(tile=[d.ni, d.nj, d.nk], dynamics_steps_per_cycle=steps, ms_per_cycle=round(1e3 * sec, 2),
                ms_per_dynamics_step=round(1e3 * sec / steps, 2),
                simulated_days_per_wall_sec=round((steps * a.dt / 86400.0) / sec, 5),
                hbm_GB_resident=round(torch.cuda.memory_allocated(dyc.device) / 1e9, 1),
                note="BASELINE.json configs[4]: one of the 4 x 2 tiles of the 4320 x 3240 x 75 grid as a stand-alone grid on one GPU; "
                     "dynamics with the ALE pressure force (PLM reconstruction of T, S), PPM tracer advection of T, S + 2 tracers, "
                     "tridiagonal solves, z* regridding and PPM_H4 remapping of T, S, the tracers, u, v; not part of `value`")
    if check is not None:
        check(dict(dyc=dyc, d=d, st=st, T=T, S=S, tr=tr, Md=keep[-1], pre=pre, info=info))
    torch.cuda.set_stream(torch.cuda.default_stream())   # (torch must not keep a stream the context is about to destroy)
    dyc.close()
    del st, T, S, tr, ea, eb, h_new, dzI, keep
    torch.cuda.empty_cache()
    return sec, info


def comm_model_leg(args, device):
    """What the halo exchanges add to a step at 8 GPUs, measured on ONE: the tile an MI355X carries in the 4 x 2 layout of the
    headline grid (1440 x 1080 -> 360 x 540 x 75), doubly re-entrant so that all eight neighbours exist -- and are this rank --
    with every group pass packed and sent through RCCL (ncclSend / ncclRecv to self, force_nccl_self) on the halo stream, and
    the same tile with local wrap copies instead.  exposed_exchange_ms = the difference of the two step times: packing, the
    RCCL launches and the stream dependencies that the overlap (start_group_pass ... own rows ... complete ... halo rows)
    does not hide; the wire itself is absent (a self send is a device copy), so this is the floor the xGMI latency comes on
    top of.  sent_MB_per_step: what this tile packs for its eight neighbours in a step, with the reference's own per-pass halo
    widths (RK2.F90:484-495; MOM6X_PASS_WIDTHS=full: NIHALO rows of everything, round 3's behaviour)."""
    import torch
    from mom6_amd.dycore import prof_enable, prof_report, prof_reset

    class A:
        pass
    out = {}
    for mode in ("local_wrap", "rccl_self"):
        a = A(); a.ni, a.nj, a.nk, a.dt, a.tracers = args.ni // 4, args.nj // 2, args.nk, args.dt, 0
        dyc, d, st, taux, tauy, keep = build_model(a, (1, 1), (0, 0), device, reentrant_y=True, force_nccl_self=(mode == "rccl_self"),
                                                   res_of=(args.ni, args.nj))
        torch.cuda.set_stream(dyc.torch_stream())

        def step(calc=False):
            dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"], taux, tauy,
                                       a.dt, calc_dtbt=calc)
        step(True); step(); step()
        dyc.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        dyc.sync(); torch.cuda.synchronize()
        ms = 1e2 * (time.perf_counter() - t0)
        prof_enable(dyc, True); prof_reset(dyc)
        dyc.comm_exchange_count(reset=True); dyc.comm_exchange_bytes(reset=True)
        step(); dyc.sync()
        nex = dyc.comm_exchange_count(); nbytes = dyc.comm_exchange_bytes()
        rep = prof_report(dyc); prof_enable(dyc, False)
        out[mode] = {"ms_per_step": round(ms, 3), "kernel_sum_ms": round(sum(v[1] for v in rep.values()), 3),
                     "launches_per_step": int(sum(v[0] for v in rep.values())), "exchanges_per_step": nex,
                     "sent_MB_per_step": round(nbytes / 1e6, 2)}
        # ... and the headline's thermodynamic step on the tile (advect_tracer of T, S + 2 tracers with its per-iteration halo pass and the
        # all-reduce of the layer flags, tracer_advect.F90:229-262, :331; the tridiagonal solves): once per nth dynamics steps, as in the
        # headline.  Three cycles of nth steps + one thermodynamic step; its own time from a second, separately timed call.
        nth = max(1, int(round(args.dt_therm / args.dt)))
        a.tracers = max(args.tracers, 0); a.breakdown = False
        thermo, _info = make_thermo(a, dyc, d, st, nth)
        for _ in range(nth):
            step()
        thermo(); dyc.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for _ in range(nth):
                step()
            thermo()
        dyc.sync(); torch.cuda.synchronize()
        ms_cycle = 1e3 * (time.perf_counter() - t0) / 3.0
        for _ in range(nth):
            step()
        dyc.sync(); torch.cuda.synchronize()
        dyc.comm_exchange_count(reset=True)
        t0 = time.perf_counter()
        thermo(); dyc.sync(); torch.cuda.synchronize()
        ms_th = 1e3 * (time.perf_counter() - t0)
        out[mode].update({"ms_per_step_with_thermo": round(ms_cycle / nth, 3), "thermo_ms_per_call": round(ms_th, 3),
                          "thermo_exchanges_per_call": dyc.comm_exchange_count()})
        torch.cuda.set_stream(torch.cuda.default_stream())
        dyc.close()
        del st, keep
        torch.cuda.empty_cache()
    out["rccl_self"]["btstep_pass"] = "blocking (the default: mom6x_comm_overlap_btstep off)"
    # BTHALO = 8, 12: the same tile with a wider barotropic halo (the reference's own lever against the latency of the sub-cycle's
    # exchanges): exchanges per step (packed group messages, of which the sub-cycle's are 2 x sub-steps / BTHALO) and the step time
    out["bthalo"] = {"4": {"ms_per_step": out["rccl_self"]["ms_per_step"], "exchanges_per_step": out["rccl_self"].get("exchanges_per_step")}}
    for bh in (8, 12, -4):   # (-4: BTHALO = 4 with btstep's own pass overlapped with the own-points half of the next sub-step)
        a = A(); a.ni, a.nj, a.nk, a.dt, a.tracers = args.ni // 4, args.nj // 2, args.nk, args.dt, 0
        dyc, d, st, taux, tauy, keep = build_model(a, (1, 1), (0, 0), device, reentrant_y=True, force_nccl_self=True, bthalo=max(bh, 0),
                                                   res_of=(args.ni, args.nj))
        if bh < 0:
            dyc.comm_overlap_btstep(True)
        torch.cuda.set_stream(dyc.torch_stream())

        def step(calc=False):
            dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"], taux, tauy,
                                       a.dt, calc_dtbt=calc)
        step(True); step(); step()
        dyc.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        dyc.sync(); torch.cuda.synchronize()
        ms = 1e2 * (time.perf_counter() - t0)
        dyc.comm_exchange_count(reset=True)
        step(); dyc.sync()
        if bh < 0:
            out["btstep_pass_overlapped"] = {"ms_per_step": round(ms, 3), "exchanges_per_step": dyc.comm_exchange_count(),
                                             "note": "BTHALO = 4, mom6x_comm_overlap_btstep on: the sub-cycle's pass packed on the compute stream, messages and "
                                                     "unpack on the second stream while the own-points half of the next sub-step runs; against rccl_self.ms_per_step"}
        else:
            out["bthalo"][str(bh)] = {"ms_per_step": round(ms, 3), "exchanges_per_step": dyc.comm_exchange_count()}
        torch.cuda.set_stream(torch.cuda.default_stream())
        dyc.close()
        del st, keep
        torch.cuda.empty_cache()
    out["tile"] = [args.ni // 4, args.nj // 2, args.nk]
    out["exposed_exchange_ms"] = round(out["rccl_self"]["ms_per_step"] - out["local_wrap"]["ms_per_step"], 3)
    out["exposed_exchange_ms_with_thermo"] = round(out["rccl_self"]["ms_per_step_with_thermo"] - out["local_wrap"]["ms_per_step_with_thermo"], 3)
    out["exposed_exchange_frac_of_step"] = round(out["exposed_exchange_ms"] / out["rccl_self"]["ms_per_step"], 4)
    # (no "launch gap": the per-kernel events serialise the two streams and add their own cost, so wall time minus their sum came out
    #  NEGATIVE in round 3; the idle time of the compute stream is read from a kernel trace instead: profiles/r04_tile_*.txt)
    out["note"] = ("one tile of the 4 x 2 layout on one GPU, all eight neighbours = this rank; every group pass through RCCL send/recv to self "
                   "on the halo stream (rccl_self) against local wrap copies (local_wrap); kernel_sum_ms is measured with HIP events around "
                   "every launch (which itself serialises the streams)")
    return out


def pmc_step_traffic():
    """Measured HBM-side GB per step (sum over all kernels of one step) from the newest committed PMC summary, or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_pmc.json")), key=_profile_order, reverse=True):
        try:
            j = json.load(open(path))
            if "bytes_per_step" in j:
                return round(j["bytes_per_step"] / 1e9, 1), os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def _variants_mean(tab, kernel, field=None, launches=None):
    """`kernel` is a launch label ("k_mass_flux_wave<0>"); a counter table names the template instantiations that ran under it
    ("k_mass_flux_wave<0, 5, false, 2, false>").  The step's launches of the mass-flux kernel are three instantiations compiled
    for their switches (SPEC 1, 2, 3: continuity_wave.hip struct Sw); the general one (SPEC 0) only runs at initialisation and the
    one with the Newton statistics (third argument true) only in the step after the timed region.  Returns the mean over the step's
    instantiations (weighted with their launches per step where the table has them) of tab[name] or tab[name][field]."""
    import re
    key = lambda n: (re.match(r"\w+(<\d+)?", n.replace(" ", "")) or [n])[0]
    names = [n for n in tab if key(n) == key(kernel)]
    if not names:
        return None
    args = {n: [a.strip() for a in n[n.index("<") + 1:n.rindex(">")].split(",")] if "<" in n else [] for n in names}
    step = [n for n in names if not (len(args[n]) >= 5 and (args[n][2] == "true" or args[n][3] == "0"))] or names
    val = lambda n: (tab[n] if field is None else tab[n].get(field))
    step = [n for n in step if val(n)]
    if not step:
        return None
    w = {n: float((launches or {}).get(n, 1.0)) for n in step}
    return sum(w[n] * float(val(n)) for n in step) / sum(w.values())


def _profile_order(path):
    """Newest last: profiles/rNN_<tag><n>_*.json ordered by round, then v1 < v2 < ... < final < final2 < ... (a plain sort of the
    names would put final2 before final)."""
    import re
    m = re.match(r"r(\d+)_([a-z]+?)(\d*)_", os.path.basename(path))
    if not m:
        return (-1, -1, 0, os.path.basename(path))
    return (int(m.group(1)), {"v": 0, "final": 1}.get(m.group(2), -1), int(m.group(3) or 1), os.path.basename(path))


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*_hbm_pmc.json,
    written by scripts/rocprof_summary.py from separate FETCH_SIZE / WRITE_SIZE passes of this same command; the
    counters cannot be collected from inside the timed run).  None if there is no summary for this kernel."""
    import glob
    import re
    key = lambda n: (re.match(r"\w+(<\d+)?", n.replace(" ", "")) or [n])[0]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_pmc.json")), key=_profile_order, reverse=True):
        try:
            j = json.load(open(path))
            tab = j["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            continue
        v = _variants_mean(tab, kernel, launches=j.get("launches_per_step"))
        if v is not None:
            return float(v), os.path.relpath(path, ROOT)
    return None, None


FP64_VALU_PEAK_TLANE_S = 256 * 4 * 16 * 2.4e9 / 1e12   # lane-instructions per second: 256 CUs x 4 SIMDs x 16 FP64 lanes x 2.4 GHz
                                                       # (= the 78.6 TFLOP/s FP64 vector peak of the MI355X counted without FMA)


def valu_counters(kernel, N3_tile, avg_ms):
    """The instruction side of the dominant kernel from the newest committed SQ-counter summary (profiles/*_sq_pmc.json,
    scripts/profile_bench.sh pass 4): wave-instructions issued per launch -> lane-instructions per face-layer, and the rate
    against the FP64 vector peak.  The counters cannot be collected inside the timed run; the launch time is this run's."""
    import glob
    import re
    key = lambda n: (re.match(r"\w+(<\d+)?", n.replace(" ", "")) or [n])[0]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_pmc.json")), key=_profile_order, reverse=True):
        try:
            tab = json.load(open(path))["per_launch"]
        except (OSError, ValueError, KeyError):
            continue
        m = {f: _variants_mean(tab, kernel, field=f) for f in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES")}
        if m["SQ_INSTS_VALU"]:
            lane = 64.0 * m["SQ_INSTS_VALU"]
            rate = lane / (avg_ms * 1e-3) / 1e12
            return {"SQ_INSTS_VALU_per_launch": round(m["SQ_INSTS_VALU"], 1), "SQ_ACTIVE_INST_VALU_per_launch": m["SQ_ACTIVE_INST_VALU"],
                    "SQ_WAVES_per_launch": m["SQ_WAVES"], "SQ_BUSY_CYCLES_per_launch": m["SQ_BUSY_CYCLES"],
                    "SQ_WAVE_CYCLES_per_launch": m["SQ_WAVE_CYCLES"],
                    "lane_instructions_per_face_layer": round(lane / N3_tile, 1),
                    "achieved_Tlane_instr_per_s": round(rate, 2), "peak_Tlane_instr_per_s": round(FP64_VALU_PEAK_TLANE_S, 2),
                    "frac_of_fp64_vector_issue_peak": round(rate / FP64_VALU_PEAK_TLANE_S, 3),
                    "averaged_over": "the step's three launches (the instantiations compiled for their switches, SPEC 1-3)",
                    "occupancy": "2 wavefronts per SIMD (190-240 VGPRs)", "source": os.path.relpath(path, ROOT) + " sha256:" + sha256_of(path)}
    return None


def sha256_of(path):
    """The first 16 hex digits of a file's SHA-256 (names a cited measurement whether or not the tree has a .git)."""
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return "unreadable"



def box_calibration(device):
    """What THIS box's memory system gives the plainest streams (torch ops, outside the timed region): a device-to-device copy
    (1 read + 1 write) and a triad a = b + s c (2 reads + 1 write) of 1 GiB arrays, best of five.  The boxes of the pool differ by
    several per cent in exactly this (profiles/README.md), and every bandwidth-bound kernel of the step follows it."""
    import torch
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float64, device=device); b = torch.ones_like(a); c = torch.ones_like(a)
    def best(fn, nbytes):
        t = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1))
        return round(nbytes / (min(t[1:]) * 1e-3) / 1e9, 1)
    out = {"copy_GBps": best(lambda: a.copy_(b), 2 * 8 * n), "triad_GBps": best(lambda: torch.add(b, c, alpha=3.0, out=a), 3 * 8 * n),
           "arrays": "3 x 1 GiB, FP64", "note": "torch device ops, best of five, not part of any other number in this line"}
    del a, b, c
    torch.cuda.empty_cache()
    return out

def cpu_baseline(args, full_size=False):
    """The oracle (kind='port': plain-C restatement of the reference Fortran with OpenMP over the loops the reference
    threads -- !$OMP parallel do over j or k, e.g. MOM_continuity_PPM.F90:370/:615, MOM_barotropic.F90:868 -- on all host
    cores) on a bounded tile: BASELINE.json configs[2]'s 360 x 180 x 75, scaled by the cell count.  full_size: ONE dynamics step
    of the headline's own grid instead (no scaling; the thermodynamic step is left out: `--cpu-full-size`)."""
    from mom6_amd import abi, grid, synth
    from oracle import orc
    import ctypes
    # the host cores this process may use: the cgroup CPU quota where there is one (a box that shows 256 hardware threads
    # but grants 16 CPUs runs the loops slower with 128 threads than with one), else what the scheduler allows
    cores = orc.set_threads(orc.usable_cores())
    ni, nj, nk = (args.ni, args.nj, args.nk) if full_size else (360, 180, args.nk)
    gg = grid.GlobalGrid(ni, nj, kind="spherical", lon0=0.0, lat0=-65.0, dlon=360.0 / args.ni, dlat=130.0 / args.nj,
                         depth_fn=grid.bowl_depth(ni, nj, 4000.0, rim=2))
    d, M = gg.tile(nk)
    GV = abi.vgrid_default()
    Rlay, gp = abi.layer_densities(nk)
    bt = abi.barotropic_params_default(20.0)
    m = orc.OrcModel(d, M, GV, abi.continuity_params_default(nk), bt, abi.coriolis_params_default(), abi.pgf_params_default(),
                     abi.rk2_params_default(), Rlay, gp)
    h, u, v = synth.make_state(d, M, u_max=float(os.environ.get("MOM6X_BENCH_UMAX", "0.5")), h_pert=0.01)
    # the same vertvisc_coef set-up as the device run: the oracle's step calls orc_vertvisc_coef three times
    kbu = np.ascontiguousarray(2.0e-3 * (1.0 + 0.5 * synth.smooth_field(d, 91, ox=1.0, oy=0.5)) * M[abi.G["mask2dCu"]])
    kbv = np.ascontiguousarray(2.0e-3 * (1.0 + 0.5 * synth.smooth_field(d, 92, ox=0.5, oy=1.0)) * M[abi.G["mask2dCv"]])
    bbl = np.full(d.shape2(), 10.0)
    m.set_vertvisc(abi.vertvisc_params_default(Kv=1.0e-4, Hmix=20.0, Hbbl=10.0), kbu, kbv, bbl, bbl.copy())
    m.set_hor_visc(hor_visc_params(abi, args.dt))
    coefs = (None,) * 6
    z3 = lambda: np.zeros_like(h)
    uh, vh, uhtr, vhtr, eta_av = z3(), z3(), z3(), z3(), np.zeros(d.shape2())
    taux = np.ascontiguousarray(0.1 * synth.smooth_field(d, 41, ox=1, oy=.5) * M[abi.G["mask2dCu"]]); tauy = np.zeros(d.shape2())
    # the thermodynamic step of the headline workload (make_thermo): T, S + the passive tracers, PPM, every nth dynamics step
    nth = max(int(round(args.dt_therm / args.dt)), 1)
    ntr = max(args.tracers, 0)
    do_thermo = args.tracers >= 0
    T = np.ascontiguousarray(10.0 + synth.smooth_field(d, 81, nk=nk)); S = np.ascontiguousarray(35.0 + 0.5 * synth.smooth_field(d, 82, nk=nk))
    tr = [np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 71 + q, nk=nk, ox=0.5, oy=0.5)) for q in range(ntr)]
    ea = np.ascontiguousarray(1.0e-3 * h); ea[0] = 0.0
    eb = np.ascontiguousarray(np.roll(ea, -1, 0)); eb[-1] = 0.0

    def thermo():
        orc.advect_tracer(d, M, GV, 0, args.dt, 2, h, uhtr, vhtr, nth * args.dt, [T, S] + tr)
        for t_ in tr:
            orc.tracer_vertdiff(d, M, GV, h, ea, eb, nth * args.dt, t_)
        orc.triDiagTS(d, h, ea, eb, T, GV.H_subroundoff); orc.triDiagTS(d, h, ea, eb, S, GV.H_subroundoff)
        uhtr[...] = 0.0; vhtr[...] = 0.0

    m.initialize(u, v, h, uh, vh, args.dt)
    if full_size:   # one step, timed as it is (it also sets dtbt: set_dtbt is a 2-D pass, 0.1 % of a step)
        t0 = time.time()
        m.step(u, v, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy, args.dt, coefs, calc_dtbt=True)
        t = time.time() - t0
        return {"value": (args.dt / 86400.0) / t, "unit": "simulated-days/wall-sec", "cores": cores, "kind": "port",
                "sample": f"ONE oracle step of step_MOM_dyn_split_RK2 (dynamics only, no thermodynamic step) at the headline's own "
                          f"{ni}x{nj}x{nk} with {cores} OpenMP threads ({t:.1f} s/step), not scaled"}
    m.step(u, v, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy, args.dt, coefs, calc_dtbt=True)   # warm-up, sets dtbt
    uhtr[...] = 0.0; vhtr[...] = 0.0
    nst = max((args.cpu_steps // nth) * nth, nth) if do_thermo else args.cpu_steps   # whole cycles: the thermodynamic share amortises exactly
    t0 = time.time()
    for n in range(nst):
        m.step(u, v, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy, args.dt, coefs)
        if do_thermo and (n + 1) % nth == 0:
            thermo()
    t = (time.time() - t0) / nst
    scale = (args.ni * args.nj) / float(ni * nj)
    t_full = t * scale
    what = (f"step_MOM_dyn_split_RK2 + every {nth} steps advect_tracer (PPM) of T, S and {ntr} passive tracers and their tridiagonal solves "
            "(the headline's workload)") if do_thermo else "step_MOM_dyn_split_RK2 (dynamics only)"
    return {"value": (args.dt / 86400.0) / t_full, "unit": "simulated-days/wall-sec", "cores": cores, "kind": "port",
            "sample": f"{nst} oracle steps of {what} on a {ni}x{nj}x{nk} tile with {cores} OpenMP threads "
                      f"({t:.2f} s/step), scaled x{scale:.1f} to {args.ni}x{args.nj}x{nk}"}


PMC_PASS_TIMEOUT = 40   # s; a pass that works takes about 4 s, one whose rocprofv3 hangs at start-up never ends


def run_group(cmd, cwd, env, timeout):
    """Run `cmd` in a process group of its own; on timeout the whole group is killed (rocprofv3's child -- this script again --
    would otherwise live on and share the GPU with the legs that follow).  Returns (exit code or None on timeout, stderr)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        _, err = p.communicate(timeout=timeout)
        return p.returncode, err or ""
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        try:
            p.communicate(timeout=10)
        except Exception:   # noqa: BLE001
            pass
        return None, ""


def measure_traffic_inrun(kernel, args):
    """HBM-side bytes of THIS state of the code, measured now: two more runs of this script under rocprofv3 (one counter per
    pass -- FETCH_SIZE, WRITE_SIZE -- and nothing else, as the MI355X guide prescribes; a step of the dynamics each), condensed by
    scripts/rocprof_summary.py (unit KB, calibrated on a kernel whose traffic is known exactly).  Returns ((bytes per launch of
    `kernel`, GB per step over all kernels, description), None) or (None, why it could not be done): rocprofv3 absent, a pass that
    timed out (each pass is tried three times, 40 s each; rocprofv3's start-up has hung on some boxes of the pool), exited non-zero or wrote no
    counter file -- with the tail of its stderr, so that a line that has to cite profiles/ says why."""
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("MOM6X_BENCH_NO_PMC"):
        return None, "MOM6X_BENCH_NO_PMC is set (this run is itself a profiled pass)"
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on the PATH"
    import re
    key = lambda n: (re.match(r"\w+(<\d+)?", n.replace(" ", "")) or [n])[0]
    tmp = tempfile.mkdtemp(prefix="mom6x_pmc_")
    env = dict(os.environ, MOM6X_BENCH_NO_PMC="1", TMPDIR="/tmp")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-config4",
           "--no-comm-model", "--tracers", "-1", "--ni", str(args.ni), "--nj", str(args.nj), "--nk", str(args.nk)]
    try:
        for tag, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            why = None
            for attempt in (1, 2, 3):
                shutil.rmtree(os.path.join(tmp, "prof_" + tag), ignore_errors=True)
                rc, err = run_group(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, "prof_" + tag), "-o", tag, "--"] + cmd,
                                    cwd="/tmp", env=env, timeout=PMC_PASS_TIMEOUT)
                if rc is None:
                    why = f"the {ctr} pass of rocprofv3 did not finish in {PMC_PASS_TIMEOUT} s ({attempt} attempts)"
                    continue
                if rc != 0:
                    why = f"the {ctr} pass of rocprofv3 exited with {rc}: " + " | ".join(err.strip().splitlines()[-2:])[:300]
                    continue
                if not os.path.exists(os.path.join(tmp, "prof_" + tag, tag + "_counter_collection.csv")):
                    why = f"the {ctr} pass of rocprofv3 wrote no counter file: " + " | ".join(err.strip().splitlines()[-2:])[:300]
                    continue
                why = None
                break
            if why:
                return None, why
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rocprof_summary.py"), tmp, os.path.join(tmp, "inrun"),
                            str(args.ni), str(args.nj), str(args.nk)], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            return None, "scripts/rocprof_summary.py failed: " + " | ".join((r.stderr or "").strip().splitlines()[-2:])[:300]
        j = json.load(open(os.path.join(tmp, "inrun_hbm_pmc.json")))
        per = _variants_mean(j["traffic_bytes_per_launch"], kernel, launches=j.get("launches_per_step"))
        return (per, round(j["bytes_per_step"] / 1e9, 1), ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one dynamics step, "
                                                          f"FETCH x{j['fetch_cal']:.3f}, WRITE x{j['write_cal']:.3f} (calibrated on {j.get('calibration_kernel')})")), None
    except Exception as e:   # noqa: BLE001  (a hung or missing profiler must not cost the benchmark line)
        return None, "the in-run counter passes failed: %s: %s" % (type(e).__name__, str(e)[:200])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ni", type=int, default=1440)
    ap.add_argument("--nj", type=int, default=1080)
    ap.add_argument("--nk", type=int, default=75)
    ap.add_argument("--dt", type=float, default=900.0)
    ap.add_argument("--dt-therm", type=float, default=3600.0, help="DT_THERM: a thermodynamic step follows every DT_THERM / DT dynamics steps")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--no-cpu-full-size", action="store_true", help="skip the one oracle step at the headline's own size (cpu_baseline.full_size)")
    ap.add_argument("--ale-ni", type=int, default=1080, help="configs[4] leg: the tile of the 4320 x 3240 grid on a 4 x 2 layout")
    ap.add_argument("--ale-nj", type=int, default=1620)
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[4] tile leg")
    ap.add_argument("--no-comm-model", action="store_true", help="skip the 1-GPU exchange-overhead leg")
    ap.add_argument("--bthalo", type=int, default=None, help="BTHALO of the barotropic solver (> 4: the tile context carries that halo, the 3-D "
                    "passes of the step stay at 4 rows; the answers do not depend on it).  Default: 0 (= NIHALO) on one GPU, 12 on more "
                    "(BT_USE_WIDE_HALOS: 15 instead of 21 exchanges per step; comm_model on the 8-GPU tile: 9.5-9.6 against 9.8-10.1 ms per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 (two extra one-step runs); cite profiles/ instead")
    ap.add_argument("--workload", choices=["headline", "config4"], default="headline", help="--transport threads only: config4 = the ALE cycle of BASELINE.json configs[4]")
    ap.add_argument("--transport", choices=["rccl", "threads"], default="rccl",
                    help="threads: run the --gpus N ranks as host threads on one GPU and check the layout against N = 1 (not a performance run)")
    ap.add_argument("--tracers", type=int, default=2, help="passive PPM tracers next to T and S in the thermodynamic step; -1 = dynamics only")
    args = ap.parse_args()
    if args.bthalo is None:
        args.bthalo = 12 if args.gpus > 1 else 0

    if args.transport == "threads":
        return run_threads(args)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if world > ndev:
        # RCCL cannot build a communicator with two ranks on one device, and a strong-scaling number from shared devices would be bogus
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPU(s) visible: one rank per GPU is required")
    local_rank %= max(ndev, 1)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if args.gpus not in LAYOUTS:
        raise SystemExit(f"--gpus must be one of {sorted(LAYOUTS)}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    out = run_rank(args, RankEnv(rank, world, local_rank, dist))
    if rank == 0:
        _REAL_STDOUT.write(json.dumps(out) + "\n"); _REAL_STDOUT.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


class RankEnv:
    """What one rank of the job knows about the others: its number, the size of the job, its device, and how to meet the
    other ranks -- torch.distributed over RCCL (the real N-GPU run: one process per GPU), or a group of host threads sharing one
    GPU (--transport threads: the same code path with the in-process halo transport of tests/transport, for checking the
    N > 1 plumbing where only one GPU exists)."""

    def __init__(self, rank, world, local_rank, dist=None, group=None, unique_id=None):
        self.rank, self.world, self.local_rank, self.dist, self.group, self.unique_id = rank, world, local_rank, dist, group, unique_id

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        elif self.group is not None:
            self.group.barrier.wait()

    def max(self, x, device):
        """The maximum of a Python float over the ranks."""
        if self.dist is not None:
            import torch
            t = torch.tensor([x], dtype=torch.float64, device=device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        if self.group is not None:
            return self.group.max(self.rank, x)
        return x


class ThreadGroup:
    def __init__(self, n):
        import threading
        self.n, self.barrier, self.vals = n, threading.Barrier(n), [0.0] * n

    def max(self, rank, x):
        self.vals[rank] = x
        self.barrier.wait()
        m = max(self.vals)
        self.barrier.wait()
        return m


RESTART_FIELDS = ["u", "v", "h", "uh", "vh", "uhtr", "vhtr", "eta_av"]


def run_threads(args):
    """--transport threads: the N ranks of `--gpus N` as host threads of this process on ONE GPU, every exchange through the
    in-process transport (mom6x_comm_set_transport; tests/transport/threads_transport.cpp).  The same run_rank() the real job
    executes -- rank -> tile, communicator attached before the new-run initialisation, barriers, the max over ranks -- followed
    by the one check a single GPU allows: after the timed steps the restart checksums of every prognostic field (summed over
    the tiles by the job's own all-reduce) and dtbt equal those of the N = 1 run of the same workload.  The timing of such a run
    says nothing (N tiles share a GPU): the line is marked invalid as a performance number."""
    import threading
    import torch
    from mom6_amd import parallel
    from mom6_amd.abi import load_library
    from tests import helpers as TH
    if args.gpus not in LAYOUTS:
        raise SystemExit(f"--gpus must be one of {sorted(LAYOUTS)}")
    args.no_pmc = True; args.no_cpu_baseline = True; args.no_comm_model = True; args.no_config4 = True; args.skip_legs = True
    layout = LAYOUTS[args.gpus]
    if args.workload == "config4":   # BASELINE.json configs[4]'s ALE cycle on the layout (at the size --ale-ni / --ale-nj give)
        one = lambda r, env, uid: {"restart_checksums": ale_cycle(args, 0, 4, 1, None, layout if env else (1, 1), parallel.rank_to_pe(r, layout) if env else (0, 0), uid, env)[1]["restart_checksums"],
                                   "workload": f"configs[4] ALE cycle (4 dynamics steps with the PLM pressure force, PPM tracer advection, tridiagonal solves, z* regridding, PPM_H4 remapping) on {args.ale_ni}x{args.ale_nj}x{args.nk}"}
    else:
        one = lambda r, env, uid: run_rank(args if env else args_with(args, gpus=1), env or RankEnv(0, 1, 0))
    ref = one(0, None, None)                                                      # the one-tile reference, same process
    TH.use_threads_transport(load_library())
    try:
        uid = parallel.unique_id(load_library())
        grp = ThreadGroup(args.gpus)
        outs, errors = [None] * args.gpus, []

        def work(r):
            try:
                outs[r] = one(r, RankEnv(r, args.gpus, 0, None, grp, uid), uid)
            except BaseException:   # noqa: BLE001
                import traceback
                errors.append((r, traceback.format_exc()))
                grp.barrier.abort()
        th = [threading.Thread(target=work, args=(r,)) for r in range(args.gpus)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        TH.use_threads_transport(load_library(), on=False)
    if errors:
        raise SystemExit(errors[0][1])
    out = outs[0]
    same = {k: all(o["restart_checksums"][k] == ref["restart_checksums"][k] for o in outs) for k in ref["restart_checksums"]}
    out["transport"] = "threads"; out["n_gpus_emulated"] = args.gpus; out["layout"] = list(layout)
    out["invalid_as_performance"] = "N tiles share one GPU: this mode checks the N > 1 code path, not its speed"
    out["layout_check"] = {"identical": all(same.values()), "fields": same, "reference": "the N = 1 run of the same workload in the same process",
                           "checksums": ref["restart_checksums"]}
    _REAL_STDOUT.write(json.dumps(out) + "\n"); _REAL_STDOUT.flush()
    if not out["layout_check"]["identical"]:
        raise SystemExit("bench.py --transport threads: the layout does not reproduce the one-tile run: " + str(same))


def args_with(args, **kw):
    import copy
    a = copy.copy(args)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


class Phases:
    """Wall time of the phases of a run (seconds since the last mark): `wall_s` in the result line, so that the time the command
    takes outside its timed region is accounted for."""
    def __init__(self):
        self.t = time.perf_counter(); self.out = {}

    def mark(self, name):
        now = time.perf_counter(); self.out[name] = round(self.out.get(name, 0.0) + now - self.t, 2); self.t = now


def run_rank(args, env):
    """One rank of the benchmark: everything `python bench.py` does once it knows which rank of how many it is."""
    import torch
    ph = Phases()
    from mom6_amd.dycore import prof_enable, prof_report, prof_reset
    rank, local_rank, dist = env.rank, env.local_rank, env.dist
    layout = LAYOUTS[args.gpus]
    from mom6_amd.parallel import rank_to_pe
    pe = rank_to_pe(rank, layout)
    dyc, d, st, taux, tauy, keep = build_model(args, layout, pe, local_rank, dist, env.unique_id)

    def step(calc_dtbt=False):
        dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"],
                                   taux, tauy, args.dt, calc_dtbt=calc_dtbt)

    def barrier():
        dyc.sync()
        torch.cuda.synchronize()
        env.barrier()

    # every torch operation of this script is issued on the context's own stream (ordered with its kernels)
    torch.cuda.set_stream(dyc.torch_stream())
    nth = max(int(round(args.dt_therm / args.dt)), 1)          # dynamics steps per thermodynamic step
    thermo, thermo_info = make_thermo(args, dyc, d, st, nth) if args.tracers >= 0 else (None, None)
    if thermo_info is not None:
        thermo_info.forcing = (taux, tauy)
    n_dyn = [0]

    def cycle_step():
        """One baroclinic step of the headline configuration: the dynamics, and after every nth of them the thermodynamic step."""
        step()
        n_dyn[0] += 1
        if thermo is not None and n_dyn[0] % nth == 0:
            thermo()

    ph.mark("build_model")
    step(calc_dtbt=True)                    # sets dtbt (untimed; part of warm-up)
    if thermo is not None:
        thermo()                            # untimed: allocates the work arrays of advect_tracer
    for _ in range(max(args.warmup - 2, 0)):
        cycle_step()
    # last warm-up step, with HIP events around EVERY kernel: the per-kernel breakdown, and which kernel dominates
    dyc.lib.mom6x_prof_filter(dyc.ctx, None)
    prof_enable(dyc, True); prof_reset(dyc)
    step(); dyc.sync()
    full = prof_report(dyc)
    prof_enable(dyc, False)
    dom_name = max(full.items(), key=lambda kv: kv[1][1])[0]
    # time EXACTLY K steps; HIP events only around the dominant kernel inside the timed region
    dyc.lib.mom6x_prof_filter(dyc.ctx, dom_name.encode())
    prof_enable(dyc, True); prof_reset(dyc)
    n_dyn[0] = 0
    st["uhtr"].zero_(); st["vhtr"].zero_()
    barrier()
    ph.mark("warmup")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle_step()
    barrier()
    elapsed = env.max(time.perf_counter() - t0, dyc.device)
    ph.mark("timed_region")
    restart_checksums = {n: "%016X" % (dyc.field_chksum(st[n]) % 2 ** 64) for n in RESTART_FIELDS}   # (collective: summed over the tiles)
    restart_checksums["dtbt"] = repr(dyc.barotropic_dtbt())
    dom = prof_report(dyc)
    prof_enable(dyc, False)
    dyc.lib.mom6x_prof_filter(dyc.ctx, None)
    n_thermo = args.steps // nth if thermo is not None else 0
    # the Newton statistics of the mass-flux kernel: ONE more step after the timed region with the collecting variant of the kernel
    dyc.continuity_stats(1); step(); nw_evals, nw_solves, nw_redos = dyc.continuity_stats(0)

    ms_per_step = 1e3 * elapsed / args.steps
    value = (args.steps * args.dt / 86400.0) / elapsed
    N3g, N2g = args.ni * args.nj * args.nk, args.ni * args.nj
    nsub = sum(v[0] for k, v in full.items() if k == "k_bt_eta")   # barotropic sub-steps per baroclinic step (both btstep calls)
    bytes_step, per_N3 = algorithmic_bytes_per_step(N3g, N2g, nsub)
    # dominant kernel: algorithmic bytes per launch = KERNEL_WORDS (8-byte words per cell-layer of the LOCAL tile,
    # DESIGN.md section 5) x 8 B x N3_tile
    N3_tile = d.ni * d.nj * d.nk
    n_dom = sum(v[0] for v in dom.values()); ms_dom = sum(v[1] for v in dom.values())
    roofline = None
    traffic, traffic_src, measured_live = None, None, None
    traffic_why = None
    if args.gpus == 1 and rank == 0 and not args.no_pmc:
        measured_live, traffic_why = measure_traffic_inrun(dom_name, args)
        ph.mark("pmc_passes")
    if measured_live is not None:
        traffic, traffic_src = measured_live[0], measured_live[2]
    elif args.gpus == 1 and (args.ni, args.nj, args.nk) == (1440, 1080, 75):
        traffic, traffic_src = pmc_traffic(dom_name)
        if traffic_src:
            traffic_src += " sha256:" + sha256_of(os.path.join(ROOT, traffic_src)) + " (cited: " + (traffic_why or "the in-run counter passes were switched off (--no-pmc)") + ")"
    words = next((w for pre, w in KERNEL_WORDS.items() if dom_name.startswith(pre)), None)
    if n_dom and words is not None:
        avg_ms = ms_dom / n_dom
        bytes_launch = words * 8.0 * N3_tile
        ach = bytes_launch / (avg_ms * 1e-3) / 1e9
        # achieved / peak / frac: the contract's numbers (algorithmic bytes over the measured launch time against the HBM peak).
        # `limited_by` says what actually limits the kernel: the mass-flux kernel issues FP64 vector instructions at more than half
        # of the chip's rate while it moves a fifth of the HBM peak -- `valu` holds the counters (scripts/profile_bench.sh, SQ pass)
        # (`bound` is the contract's word for the roof the three numbers are priced against: "hbm"; `limited_by` says what the counters say)
        bound = "valu_fp64" if dom_name.startswith("k_mass_flux_wave") else "hbm"
        roofline = {"bound": "hbm", "limited_by": bound, "kernel": dom_name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 4),
                    "launches_per_step": n_dom / args.steps, "algorithmic_bytes_per_launch": bytes_launch,
                    "words_per_cell_layer": words}
        if bound == "valu_fp64":
            roofline["valu"] = valu_counters(dom_name, N3_tile, avg_ms)
    out = {
        "metric": "simulated-days/wall-sec", "value": value, "unit": "simulated-days/wall-sec", "n_gpus": args.gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"step_MOM_dyn_split_RK2 + every {nth} steps advect_tracer (PPM) of T, S and {max(args.tracers, 0)} passive tracers "
                               f"and their vertical tridiagonal solves, on {args.ni}x{args.nj}x{args.nk} (0.25-degree-class synthetic global, "
                               f"BASELINE.json configs[3] grid), DT={args.dt:g} s, DT_THERM={args.dt_therm:g} s, layout {layout[0]}x{layout[1]}, "
                               f"{nsub} barotropic sub-steps per step",
                   "thermo_steps_in_timed_region": n_thermo,
                   "state": "seeded smooth fields (SURVEY.md 8d): |u|, |v| <= %.2g m/s per layer, thicknesses perturbed by 1 %%" % float(os.environ.get("MOM6X_BENCH_UMAX", "0.5")),
                   "newton_evals_per_solve": round(nw_evals / max(nw_solves, 1), 3), "newton_solves_repeated_with_exact_limits": nw_redos,
                   "sum_order": {0: "REFERENCE (sequential in k, bit-identical to the Fortran loop nest)",
                                 1: "TREE16 (column sums of the mass-flux kernels as a 16-lane tree; MOM6X_SUMS=exact: the reference's k order)",
                                 2: "TREE16_FMA (column sums of the mass-flux kernels as a 16-lane tree, fused multiply-adds at fixed sites that the "
                                    "oracle restates; 3e-12 of range from the reference's order after 10 steps; MOM6X_SUMS=tree: un-fused, "
                                    "MOM6X_SUMS=exact: the reference's k order)"}[int(dyc.cont_params.sum_order)],
                   "frozen_inputs": "none of the step's callees; vertvisc_coef and horizontal_viscosity run on the device inside the step (the set_viscous_BBL inputs of vertvisc_coef are constant synthetic fields)",
                   "tile": [d.ni, d.nj, d.nk], "halo": d.halo, "BTHALO": args.bthalo},
        "roofline": roofline, "restart_checksums": restart_checksums,
    }
    th = thermo_info() if thermo_info is not None else None
    if th is not None:
        out["thermo"] = th
    th_bytes = (th["advect_algorithmic_GB"] + th["tridiag_algorithmic_GB"]) * 1e9 / nth if th else 0.0   # amortised per dynamics step
    tot_bytes = bytes_step + th_bytes
    measured, measured_src = (measured_live[1], measured_live[2]) if measured_live is not None else \
        (pmc_step_traffic() if args.gpus == 1 and (args.ni, args.nj, args.nk) == (1440, 1080, 75) else (None, None))
    if measured_live is None and measured_src:
        measured_src += " sha256:" + sha256_of(os.path.join(ROOT, measured_src))
    out["hbm_step"] = {
        "algorithmic_GB_per_step": round(tot_bytes / 1e9, 2), "dynamics_GB": round(bytes_step / 1e9, 2), "thermo_GB_amortised": round(th_bytes / 1e9, 2),
        "achieved_GBps": round(tot_bytes / 1e9 / (ms_per_step * 1e-3), 1),
        "frac_of_peak": round(tot_bytes / 1e9 / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * args.gpus), 4),
        "model": "SURVEY.md 8(d), un-fused: 1616 B x N3 + 570 B x N2 x sub-steps, + 192 B x N3 for vertvisc_coef x3 + 40 B x N3 for horizontal_viscosity"
                 " (+ the thermodynamic step's words / 4)",
        "fused_away": FUSED_WORDS,
        "measured_FETCH_plus_WRITE_GB_per_step": measured, "measured_source": measured_src}
    if args.tracers >= 0 and args.gpus == 1 and not getattr(args, "skip_legs", False):
        # the legs reported next to the headline are measured on one GPU; the scaling runs (N > 1) keep to the headline path
        out["ale_remap_leg"] = ale_remap_leg(args, dyc, d, st, barrier, dist)
        out["diag_leg"] = diag_leg(args, dyc, d, st, barrier, dist)
        ph.mark("ale_and_diag_legs")
        if not args.no_comm_model and (args.ni, args.nj) == (1440, 1080):
            torch.cuda.set_stream(torch.cuda.default_stream()); dyc.close(); st.clear()
            torch.cuda.empty_cache()
            out["comm_model"] = comm_model_leg(args, local_rank)
            ph.mark("comm_model_leg")
        if not args.no_config4 and (args.ni, args.nj) == (1440, 1080):
            torch.cuda.set_stream(torch.cuda.default_stream()); dyc.close(); st.clear()                  # the headline model makes room for the larger tile
            torch.cuda.empty_cache()
            out["config4_tile_leg"] = ale_cycle(args, local_rank)[1]
            ph.mark("config4_tile_leg")
    if rank == 0:
        try:
            out["box_calibration"] = box_calibration(torch.device("cuda", local_rank))
        except Exception as e:   # noqa: BLE001  (a calibration that fails must not cost the line)
            out["box_calibration"] = {"error": str(e)[:120]}
        tot = sum(v[1] for v in full.values())
        out["kernel_ms_per_step"] = {k: round(v[1], 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1][1])[:12]}
        out["kernel_sum_ms"] = round(tot, 2)
        if args.breakdown:
            for k, (cnt, ms) in sorted(full.items(), key=lambda kv: -kv[1][1]):
                if cnt == 0:
                    continue
                print(f"{k:28s} n={cnt:5d} total={ms:9.3f} ms avg={ms / cnt * 1e3:9.1f} us ({100 * ms / tot:5.1f}%)", file=sys.stderr)
        if not args.no_cpu_baseline and args.gpus == 1:   # (rank 0 at N = 1 only)
            ph.mark("other")
            out["cpu_baseline"] = cpu_baseline(args)
            if not args.no_cpu_full_size and not os.environ.get("MOM6X_BENCH_NO_CPU_FULL"):
                # ... and ONE dynamics step at the headline's own size, unscaled (the model of this run is closed first: the oracle's
                # ~45 arrays of 0.95 GB live in host memory, nothing on the device)
                try:
                    out["cpu_baseline"]["full_size"] = cpu_baseline(args, full_size=True)
                except MemoryError as e:   # noqa: BLE001
                    out["cpu_baseline"]["full_size"] = {"error": "not enough host memory: " + str(e)[:100]}
            ph.mark("cpu_baseline")
    if st:                                   # (the legs that need the memory have closed the model already)
        torch.cuda.set_stream(torch.cuda.default_stream()); dyc.close(); st.clear()
    ph.mark("other")
    out["wall_s"] = ph.out
    return out


# The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout when a communicator is made
# (the comm_model leg makes one even at N = 1), so everything but the result line is sent to stderr: file descriptor 1 is
# kept aside for the line and pointed at stderr for the rest of the process (C stdio buffers included).
_REAL_STDOUT = sys.stdout
if __name__ == "__main__":
    sys.stdout.flush()
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    main()
