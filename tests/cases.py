"""Seeded inputs and oracle runs shared by the GPU parity tests, the CPU golden-fixture tests and
scripts/make_golden.py (so that the three always talk about the same case)."""
import numpy as np

from mom6_amd import abi, synth

G = abi.G


def visc_coefs(d, M):
    from tests.test_dyn_gpu import visc_coefs as vc
    return vc(d, M)


def rk2_params(d, GV, bt_mod=None, rk2_mod=None, cor_mod=None, cont_mod=None):
    bt = abi.barotropic_params_default(30.0)
    for k, v in (bt_mod or {}).items():
        setattr(bt, k, v)
    rk2 = abi.rk2_params_default()
    for k, v in (rk2_mod or {}).items():
        setattr(rk2, k, v)
    cor = abi.coriolis_params_default()
    for k, v in (cor_mod or {}).items():
        setattr(cor, k, v)
    cont = abi.continuity_params_default(d.nk, GV.Angstrom_H)
    for k, v in (cont_mod or {}).items():
        setattr(cont, k, v)
    return cont, bt, cor, abi.pgf_params_default(GV.Rho0), rk2


def rk2_inputs(cfg, per_stage=False, new_diff=False):
    gg, d, M = cfg
    GV = abi.vgrid_default()
    Rlay, gp = abi.layer_densities(d.nk)
    h, u, v = synth.make_state(d, M, u_max=0.05, h_pert=0.001)
    base = visc_coefs(d, M)
    if per_stage:
        coefs = [tuple(base), tuple([base[0] * 1.1, base[1] * 1.1] + base[2:]), tuple([base[0] * 0.9, base[1] * 0.9] + base[2:])]
        coefs = [tuple(np.ascontiguousarray(a) for a in c) for c in coefs]
    else:
        coefs = [tuple(base)] * 3
    taux = np.ascontiguousarray(0.1 * synth.smooth_field(d, 41, ox=1, oy=.5) * M[G["mask2dCu"]])
    tauy = np.ascontiguousarray(0.05 * synth.smooth_field(d, 42, ox=.5, oy=1) * M[G["mask2dCv"]])
    diff_new = None
    if new_diff:
        diff_new = (np.ascontiguousarray(1e-7 * synth.smooth_field(d, 61, nk=d.nk, ox=1, oy=.5) * M[G["mask2dCu"]][None]),
                    np.ascontiguousarray(1e-7 * synth.smooth_field(d, 62, nk=d.nk, ox=.5, oy=1) * M[G["mask2dCv"]][None]))
    return dict(GV=GV, Rlay=Rlay, gp=gp, dt=1200.0, h=h, u=u, v=v, coefs=coefs, taux=taux, tauy=tauy, diff_new=diff_new)


def oracle_rk2(orc, cfg, inp, nsteps, bt_mod=None, rk2_mod=None, cor_mod=None, first_direction=0, tv=None, vv=None, hv=None, Hmix_stress=0.0,
               cont_mod=None):
    """nsteps of orc_step_dyn_split_RK2 from the seeded state; returns (final state dict, OrcModel)."""
    gg, d, M = cfg
    GV, dt, h, u, v = inp["GV"], inp["dt"], inp["h"], inp["u"], inp["v"]
    cont, bt, cor, pgf, rk2 = rk2_params(d, GV, bt_mod, rk2_mod, cor_mod, cont_mod)
    m = orc.OrcModel(d, M, GV, cont, bt, cor, pgf, rk2, inp["Rlay"], inp["gp"], first_direction)
    if tv is not None:
        m.set_tv(*tv)
    if vv is not None:
        m.set_vertvisc(*vv)
    if hv is not None:
        m.set_hor_visc(hv)
    if Hmix_stress > 0.0:
        m.set_direct_stress(Hmix_stress)
    so = dict(u=u.copy(), v=v.copy(), h=h.copy(), uh=np.zeros_like(h), vh=np.zeros_like(h), uhtr=np.zeros_like(h),
              vhtr=np.zeros_like(h), eta_av=np.zeros(d.shape2()))
    m.initialize(so["u"], so["v"], so["h"], so["uh"], so["vh"], dt)
    dn = inp["diff_new"]
    for n in range(nsteps):
        m.step(so["u"], so["v"], so["h"], so["uh"], so["vh"], so["uhtr"], so["vhtr"], so["eta_av"], inp["taux"], inp["tauy"], dt,
               inp["coefs"], calc_dtbt=(n == 0), diffu_new=dn[0] if dn else None, diffv_new=dn[1] if dn else None)
    return so, m


def continuity_inputs(cfg):
    gg, d, M = cfg
    GV = abi.vgrid_default()
    CS = abi.continuity_params_default(d.nk, GV.Angstrom_H)
    h, u, v = synth.make_state(d, M, thin_frac=0.1)
    vr_u = np.clip(0.5 + 0.6 * synth.smooth_field(d, 11, nk=d.nk, ox=1.0, oy=0.5), 0.0, 1.0)
    vr_v = np.clip(0.5 + 0.6 * synth.smooth_field(d, 12, nk=d.nk, ox=0.5, oy=1.0), 0.0, 1.0)
    return dict(GV=GV, CS=CS, h=h, u=u, v=v, vr_u=np.ascontiguousarray(vr_u), vr_v=np.ascontiguousarray(vr_v), dt=1200.0)


def oracle_continuity(orc, cfg, inp):
    """continuity_PPM with uhbt/vhbt (5 % off the unadjusted sums), visc_rem, u_cor/v_cor: the corrector-call shape."""
    gg, d, M = cfg
    GV, CS, h, u, v, dt = (inp[k] for k in ("GV", "CS", "h", "u", "v", "dt"))
    z = lambda: np.zeros_like(h)
    h0, uh0, vh0 = z(), z(), z()
    orc.continuity_PPM(d, M, GV, CS, 0, u, v, h, h0, uh0, vh0, dt)
    uhbt = np.ascontiguousarray(uh0.sum(0) * (1.0 + 0.05 * synth.smooth_field(d, 13, ox=1.0, oy=0.5)))
    vhbt = np.ascontiguousarray(vh0.sum(0) * (1.0 - 0.05 * synth.smooth_field(d, 14, ox=0.5, oy=1.0)))
    out = dict(h=z(), uh=z(), vh=z(), u_cor=z(), v_cor=z())
    orc.continuity_PPM(d, M, GV, CS, 0, u, v, h, out["h"], out["uh"], out["vh"], dt, uhbt=uhbt, vhbt=vhbt,
                       visc_rem_u=inp["vr_u"], visc_rem_v=inp["vr_v"], u_cor=out["u_cor"], v_cor=out["v_cor"])
    return out, uhbt, vhbt


def thermo_state(d, M, seed=7):
    """A stably stratified T, S pair with horizontal structure (tv%T, tv%S)."""
    h, _, _ = synth.make_state(d, M)
    T = np.zeros_like(h); S = np.zeros_like(h)
    for k in range(d.nk):
        T[k] = 20.0 - 15.0 * k / max(d.nk - 1, 1) + 0.8 * synth.smooth_field(d, seed + k, ox=0.5, oy=0.5)
        S[k] = 34.0 + 1.0 * k / max(d.nk - 1, 1) + 0.2 * synth.smooth_field(d, seed + 100 + k, ox=0.5, oy=0.5)
    return np.ascontiguousarray(T), np.ascontiguousarray(S)


def oracle_ocean_stats(orc, cfg, nsteps=3, bt_mod=None):
    """The ocean.stats lines (header + one line at the start + one after each of nsteps baroclinic steps) of the oracle's
    run of the seeded double-gyre-type case: the oracle's write_energy sums through mom6_amd.sum_output."""
    from mom6_amd import sum_output as SO
    gg, d, M = cfg
    inp = rk2_inputs(cfg, False, False)
    dt = inp["dt"]
    P = abi.sum_output_params_default(dt)
    st = orc.SumOutputState(d, M, inp["GV"], inp["gp"], P)
    so = SO.SumOutput()
    so.record(orc.write_energy(st, inp["u"], inp["v"], inp["h"]), 0.0, 0)
    for n in range(1, nsteps + 1):
        s = oracle_rk2(orc, cfg, inp, n, bt_mod, None, None, 0)[0]
        so.record(orc.write_energy(st, s["u"], s["v"], s["h"]), dt * n, n)
    return so.lines


# ---- the case of tests/fortran_stubs/drive_shims.F90: the split-RK2 step behind the Fortran shim modules.  The shims always
# run vertvisc_coef and horizontal_viscosity on the device, so the oracle run does too; the MOM_input table below is what the
# driver's stand-in for the parameter file answers get_param with (everything else takes the reference's default).
SHIM_NSTEPS, SHIM_SAVE_AFTER = 4, 2


def shim_case_params(dt, tag_tree):   # tag_tree: helpers.golden_tag()
    # (DTBT keeps its default -0.98: a fraction of the stability limit, set by the first step's set_dtbt as in the oracle's run)
    return {"DT": repr(dt), "BT_STRONG_DRAG": "True", "KV": "1.0e-4", "HMIX_FIXED": "20.0", "HBBL": "10.0",
            "AH_VEL_SCALE": "0.02", "SMAGORINSKY_AH": "True", "SMAG_BI_CONST": "0.06", "REENTRANT_X": "False",
            "ENABLE_THERMODYNAMICS": "False", "MOM6X_CONTINUITY_SUMS": {"": "REFERENCE", "_tree16": "TREE16", "_tree16_fma": "TREE16_FMA"}[tag_tree]}


def oracle_shim_case(orc, cfg, nsteps=SHIM_NSTEPS, no_bt_cont=False):
    """The oracle's run of the case above: (final state, OrcModel, inputs, visc inputs).  no_bt_cont: the variant
    USE_BT_CONT_TYPE = False, NONLINEAR_BT_CONTINUITY = True."""
    from tests.test_dyn_gpu import visc_inputs
    gg, d, M = cfg
    inp = rk2_inputs(cfg, False, False)
    P = abi.vertvisc_params_default(Kv=1.0e-4, Hmix=20.0, Hbbl=10.0)
    vis = visc_inputs(d, M)
    hv = abi.hor_visc_params_default(inp["dt"])
    hv.Ah_vel_scale = 0.02; hv.Smagorinsky_Ah = 1; hv.Smag_bi_const = 0.06
    bt_mod, rk2_mod = dict(strong_drag=1), None
    if no_bt_cont:
        bt_mod.update(nonlinear_continuity=1, bt_thick_scheme=abi.BT_THICK_HYBRID); rk2_mod = dict(no_BT_cont=1)
    so, m = oracle_rk2(orc, cfg, inp, nsteps, bt_mod=bt_mod, rk2_mod=rk2_mod, vv=(P,) + tuple(vis) + (None, None), hv=hv)
    return so, m, inp, vis
