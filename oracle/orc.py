"""ORACLE python wrapper -- test infrastructure only.  PARITY UNPINNED (see orc_common.h).

Thin ctypes binding of oracle/liborc.so operating on HOST numpy arrays in the pitched
tile layout of include/mom6x.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from mom6_amd.abi import BTCont, Dims, ContinuityParams, VGrid, BarotropicParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def usable_cores():
    """The cores this process may really use: the affinity mask capped by the cgroup CPU quota (a box can show 256 hardware
    threads behind a quota of 16: an OpenMP team of 256 on such a box is slower than one thread)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return cores


def set_threads(n=None):
    """Size of the oracle's OpenMP teams (default: usable_cores(), or OMP_NUM_THREADS when that is set)."""
    if n is None:
        n = int(os.environ["OMP_NUM_THREADS"]) if os.environ.get("OMP_NUM_THREADS", "").isdigit() else usable_cores()
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        return 1
    return int(n)


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(p):
            build()
        _LIB = C.CDLL(p)
        set_threads()
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.dtype in (np.float64, np.int32, np.int64) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def bt_cont_struct(bt):
    """dict name -> ndarray (or None)  ->  BTCont of host pointers."""
    if bt is None:
        return None
    s = BTCont()
    for n in BTCont._names:
        a = bt.get(n)
        setattr(s, n, a.ctypes.data if a is not None else None)
    return s


def new_bt_cont(d, with_h=True):
    bt = {n: np.zeros(d.shape2()) for n in BTCont._names[:12]}
    bt["h_u"] = np.zeros(d.shape3()) if with_h else None
    bt["h_v"] = np.zeros(d.shape3()) if with_h else None
    return bt


def continuity_PPM(d, G, GV, CS, first_direction, u, v, hin, h, uh, vh, dt, uhbt=None, vhbt=None,
                   visc_rem_u=None, visc_rem_v=None, u_cor=None, v_cor=None, BT_cont=None,
                   du_cor=None, dv_cor=None):
    bts = bt_cont_struct(BT_cont)
    rc = lib().orc_continuity_PPM(C.byref(d), _p(G), C.byref(GV), C.byref(CS), C.c_int(first_direction),
                                  _p(u), _p(v), _p(hin), _p(h), _p(uh), _p(vh), C.c_double(dt),
                                  _p(uhbt), _p(vhbt), _p(visc_rem_u), _p(visc_rem_v), _p(u_cor), _p(v_cor),
                                  C.byref(bts) if bts is not None else None, _p(du_cor), _p(dv_cor))
    if rc != 0:
        raise RuntimeError(f"orc_continuity_PPM rc={rc}")


# ------------------------------------------------------------------------------------------
# MOM_barotropic
class BtCS(C.Structure):
    """orc_bt_cs: the persistent part of barotropic_CS (host arrays)."""
    _names = ["frhatu", "frhatv", "IDatu", "IDatv", "ubtav", "vbtav", "eta_cor", "q_D", "D_u_Cor", "D_v_Cor"]
    _fields_ = [(n, C.c_void_p) for n in _names]


class BtState:
    def __init__(self, d):
        self.f = {}
        for n in BtCS._names:
            self.f[n] = np.zeros(d.shape3() if n.startswith("frhat") else d.shape2())
        self.struct = BtCS()
        for n in BtCS._names:
            setattr(self.struct, n, self.f[n].ctypes.data)

    def __getitem__(self, n):
        return self.f[n]


def barotropic_init(d, G, GV, P, cs):
    rc = lib().orc_barotropic_init(C.byref(d), _p(G), C.byref(GV), C.byref(P), C.byref(cs.struct))
    assert rc == 0, rc


def btcalc(d, G, GV, h, h_u, h_v, cs, scheme=0):
    rc = lib().orc_btcalc(C.byref(d), _p(G), C.byref(GV), _p(h), _p(h_u), _p(h_v), C.byref(cs.struct), C.c_int(int(scheme)))
    assert rc == 0, rc


def bt_mass_source(d, G, GV, h, eta, set_cor, cs):
    rc = lib().orc_bt_mass_source(C.byref(d), _p(G), C.byref(GV), _p(h), _p(eta), C.c_int(int(set_cor)), C.byref(cs.struct))
    assert rc == 0, rc


def set_dtbt(d, G, GV, P, cs, pbce=None, gtot_est=0.0, SSH_add=0.0):
    dtbt = C.c_double(0.0); dtbt_max = C.c_double(0.0)
    rc = lib().orc_set_dtbt(C.byref(d), _p(G), C.byref(GV), C.byref(P), C.byref(cs.struct), _p(pbce),
                            C.c_double(gtot_est), C.c_double(SSH_add), C.byref(dtbt), C.byref(dtbt_max))
    assert rc == 0, rc
    return dtbt.value, dtbt_max.value


def set_dtbt_pbce_eta(d, G, GV, P, cs, pbce, eta=None):
    """set_dtbt(pbce, eta=eta) as step_MOM_dyn_split_RK2 calls it (:667); returns (and leaves in P.dtbt) the new dtbt."""
    rc = lib().orc_set_dtbt_pbce_eta(C.byref(d), _p(G), C.byref(GV), C.byref(P), C.byref(cs.struct), _p(pbce), _p(eta))
    assert rc == 0, rc
    return P.dtbt


def btstep(d, G, GV, P, cs, first_direction, U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, taux, tauy, pbce,
           eta_PF_in, U_Cor, V_Cor, accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v,
           BT_cont, taux_bot=None, tauy_bot=None, uh0=None, vh0=None, u_uh0=None, v_vh0=None, etaav=None):
    bts = bt_cont_struct(BT_cont) if BT_cont is not None else None   # None: USE_BT_CONT_TYPE = False (linear barotropic continuity)
    nstep = C.c_int(0)
    rc = lib().orc_btstep(C.byref(d), _p(G), C.byref(GV), C.byref(P), C.byref(cs.struct), C.c_int(first_direction),
                          _p(U_in), _p(V_in), _p(eta_in), C.c_double(dt), _p(bc_accel_u), _p(bc_accel_v),
                          _p(taux), _p(tauy), _p(pbce), _p(eta_PF_in), _p(U_Cor), _p(V_Cor),
                          _p(accel_layer_u), _p(accel_layer_v), _p(eta_out), _p(uhbtav), _p(vhbtav),
                          _p(visc_rem_u), _p(visc_rem_v), C.byref(bts) if bts is not None else None, _p(taux_bot), _p(tauy_bot),
                          _p(uh0), _p(vh0), _p(u_uh0), _p(v_vh0), _p(etaav), C.byref(nstep))
    if rc != 0:
        raise RuntimeError(f"orc_btstep rc={rc}")
    return nstep.value


# ------------------------------------------------------------------------------------------
# MOM_CoriolisAdv / MOM_PressureForce_FV / MOM_vert_friction
def CorAdCalc(d, G, GV, CS, u, v, h, uh, vh, CAu, CAv):
    rc = lib().orc_CorAdCalc(C.byref(d), _p(G), C.byref(GV), C.byref(CS), _p(u), _p(v), _p(h), _p(uh), _p(vh), _p(CAu), _p(CAv))
    if rc != 0:
        raise RuntimeError(f"orc_CorAdCalc rc={rc}")


def PressureForce(d, G, GV, CS, Rlay, g_prime, h, PFu, PFv, pbce=None, eta=None, T=None, S=None, eos=None):
    rc = lib().orc_PressureForce_FV_Bouss(C.byref(d), _p(G), C.byref(GV), C.byref(CS), _p(Rlay), _p(g_prime), _p(h),
                                          _p(PFu), _p(PFv), _p(pbce), _p(eta), _p(T), _p(S),
                                          C.byref(eos) if eos is not None else None)
    if rc != 0:
        raise RuntimeError(f"orc_PressureForce rc={rc}")


def ALE_PLM_edge_values(d, GV, h, Q, bdry_extrap, Q_t, Q_b):
    """ALE_PLM_edge_values (MOM_ALE.F90:1520)."""
    lib().orc_ALE_PLM_edge_values(C.byref(d), C.byref(GV), _p(h), _p(Q), C.c_int(int(bdry_extrap)), _p(Q_t), _p(Q_b))


def ALE_PPM_edge_values(d, GV, h, Q, bdry_extrap, Q_t, Q_b):
    """One field of TS_PPM_edge_values (MOM_ALE.F90:1581)."""
    lib().orc_ALE_PPM_edge_values(C.byref(d), C.byref(GV), _p(h), _p(Q), C.c_int(int(bdry_extrap)), _p(Q_t), _p(Q_b))


def eos_density_anomaly(eos, T, S, p, rho_ref):
    """calculate_density(T, S, p, rho, EOS, rho_ref=rho_ref) for one point."""
    f = lib().orc_eos_density_anomaly
    f.restype = C.c_double
    return f(C.byref(eos), C.c_double(T), C.c_double(S), C.c_double(p), C.c_double(rho_ref))


def hor_visc_init(d, G, CS):
    """The 2-D coefficient planes of hor_visc_CS as one block [nplanes][slab]."""
    n = lib().orc_hor_visc_nplanes()
    P = np.zeros((n,) + d.shape2())
    rc = lib().orc_hor_visc_init(C.byref(d), _p(G), C.byref(CS), _p(P))
    if rc != 0:
        raise RuntimeError(f"orc_hor_visc_init rc={rc}")
    return P


def horizontal_viscosity(d, G, GV, CS, P, u, v, h, diffu, diffv):
    rc = lib().orc_horizontal_viscosity(C.byref(d), _p(G), C.byref(GV), C.byref(CS), _p(P), _p(u), _p(v), _p(h), _p(diffu), _p(diffv))
    if rc != 0:
        raise RuntimeError(f"orc_horizontal_viscosity rc={rc}")


# ---- MOM_remapping / MOM_ALE (orc_remap.c: 1-based arrays inside, element 0 unused) ----------------------------------
def _one(a, n=None):
    a = np.asarray(a, dtype=np.float64)
    out = np.zeros((len(a) if n is None else n) + 2)
    out[1:len(a) + 1] = a
    return out


def remapping_core_h(CS, h0, u0, h1):
    h0, u0, h1 = (np.ascontiguousarray(a, dtype=np.float64) for a in (h0, u0, h1))
    u1 = np.zeros(len(h1)); err = C.c_double(0.0)
    rc = lib().orc_remapping_core_h(C.byref(CS), len(h0), _p(h0), _p(u0), len(h1), _p(h1), _p(u1), C.byref(err))
    if rc != 0:
        raise RuntimeError(f"orc_remapping_core_h rc={rc}")
    return u1, err.value


def remapping_core_h_cols(CS, h0, u0, h1):
    """h0, u0: [ncol][n0]; h1: [ncol][n1]"""
    h0, u0, h1 = (np.ascontiguousarray(a, dtype=np.float64) for a in (h0, u0, h1))
    u1 = np.zeros_like(h1)
    rc = lib().orc_remapping_core_h_cols(C.byref(CS), h0.shape[0], h0.shape[1], _p(h0), _p(u0), h1.shape[1], _p(h1), _p(u1))
    if rc != 0:
        raise RuntimeError(f"orc_remapping_core_h_cols rc={rc}")
    return u1


def PLM_reconstruction(h, u, h_neglect, extrapolate=False):
    n = len(h); h1, u1 = _one(h), _one(u)
    E1, E2, C1, C2 = (np.zeros(n + 2) for _ in range(4))
    lib().orc_PLM_reconstruction(n, _p(h1), _p(u1), _p(E1), _p(E2), _p(C1), _p(C2), C.c_double(h_neglect))
    if extrapolate:
        lib().orc_PLM_boundary_extrapolation(n, _p(h1), _p(u1), _p(E1), _p(E2), _p(C1), _p(C2), C.c_double(h_neglect))
    return E1[1:n + 1], E2[1:n + 1], C1[1:n + 1], C2[1:n + 1]


def edge_values_explicit_h4(h, u, h_neglect):
    n = len(h); h1, u1 = _one(h), _one(u)
    E1, E2 = np.zeros(n + 2), np.zeros(n + 2)
    lib().orc_edge_values_explicit_h4(n, _p(h1), _p(u1), _p(E1), _p(E2), C.c_double(h_neglect))
    return E1[1:n + 1], E2[1:n + 1]


def edge_values_implicit_h4(h, u, h_neglect):
    n = len(h); h1, u1 = _one(h), _one(u)
    E1, E2 = np.zeros(n + 2), np.zeros(n + 2)
    lib().orc_edge_values_implicit_h4(n, _p(h1), _p(u1), _p(E1), _p(E2), C.c_double(h_neglect))
    return E1[1:n + 1], E2[1:n + 1]


def PPM_reconstruction(h, u, E1, E2, h_neglect):
    n = len(h); h1, u1 = _one(h), _one(u)
    e1, e2 = _one(E1), _one(E2)
    C1, C2, C3 = (np.zeros(n + 2) for _ in range(3))
    lib().orc_PPM_reconstruction(n, _p(h1), _p(u1), _p(e1), _p(e2), _p(C1), _p(C2), _p(C3), C.c_double(h_neglect))
    return e1[1:n + 1], e2[1:n + 1], C1[1:n + 1], C2[1:n + 1], C3[1:n + 1]


def intersect_src_tgt_grids(h0, h1):
    n0, n1 = len(h0), len(h1); ns = n0 + n1 + 1
    a0, a1 = _one(h0), _one(h1)
    h_sub, h0_eff = np.zeros(ns + 2), np.zeros(n0 + 2)
    I = lambda n: np.zeros(n + 2, dtype=np.int32)
    isrc_start, isrc_end, isrc_max, itgt_start, itgt_end, isub_src = I(n0), I(n0), I(n0), I(n1), I(n1), I(ns)
    lib().orc_intersect_src_tgt_grids(n0, _p(a0), n1, _p(a1), _p(h_sub), _p(h0_eff), _p(isrc_start), _p(isrc_end), _p(isrc_max),
                                      _p(itgt_start), _p(itgt_end), _p(isub_src))
    return dict(h_sub=h_sub[1:ns + 1], h0_eff=h0_eff[1:n0 + 1], isrc_start=isrc_start[1:n0 + 1], isrc_end=isrc_end[1:n0 + 1],
                isrc_max=isrc_max[1:n0 + 1], itgt_start=itgt_start[1:n1 + 1], itgt_end=itgt_end[1:n1 + 1], isub_src=isub_src[1:ns + 1])


def remap_src_to_sub_grid_plm(h0, u0, h1, om4, h_neglect, force_bounds=False):
    """PLM_reconstruction + PLM_boundary_extrapolation + remap_src_to_sub_grid(_om4) + remap_sub_to_tgt_grid_om4, as the
    reference's unit tests chain them; returns (u_sub, u1)."""
    n0, n1 = len(h0), len(h1); ns = n0 + n1 + 1
    a0, a1, b0 = _one(h0), _one(h1), _one(u0)
    h_sub, h0_eff = np.zeros(ns + 2), np.zeros(n0 + 2)
    I = lambda n: np.zeros(n + 2, dtype=np.int32)
    isrc_start, isrc_end, isrc_max, itgt_start, itgt_end, isub_src = I(n0), I(n0), I(n0), I(n1), I(n1), I(ns + 1)
    lib().orc_intersect_src_tgt_grids(n0, _p(a0), n1, _p(a1), _p(h_sub), _p(h0_eff), _p(isrc_start), _p(isrc_end), _p(isrc_max),
                                      _p(itgt_start), _p(itgt_end), _p(isub_src))
    E1, E2, C1, C2 = (np.zeros(n0 + 2) for _ in range(4))
    lib().orc_PLM_reconstruction(n0, _p(a0), _p(b0), _p(E1), _p(E2), _p(C1), _p(C2), C.c_double(h_neglect))
    lib().orc_PLM_boundary_extrapolation(n0, _p(a0), _p(b0), _p(E1), _p(E2), _p(C1), _p(C2), C.c_double(h_neglect))
    u_sub, uh_sub = np.zeros(ns + 2), np.zeros(ns + 2); err = C.c_double(0.0)
    lib().orc_remap_src_to_sub_grid(int(om4), n0, _p(a0), _p(b0), _p(E1), _p(E2), _p(C1), _p(C2), n1, _p(h_sub), _p(h0_eff), _p(isrc_start),
                                    _p(isrc_end), _p(isrc_max), _p(isub_src), 1, int(force_bounds), _p(u_sub), _p(uh_sub), C.byref(err))
    u1 = np.zeros(n1 + 2); err2 = C.c_double(0.0)
    lib().orc_remap_sub_to_tgt_grid_om4(n0, n1, _p(a1), _p(h_sub), _p(u_sub), _p(uh_sub), _p(itgt_start), _p(itgt_end), int(force_bounds),
                                        _p(u1), C.byref(err2))
    return u_sub[1:ns + 1], u1[1:n1 + 1]


def ALE_remap_tracers(d, G, CS, h_old, h_new, fields):
    ptrs = (C.c_void_p * len(fields))(*[f.ctypes.data for f in fields])
    rc = lib().orc_ALE_remap_tracers(C.byref(d), _p(G), C.byref(CS), _p(h_old), _p(h_new), ptrs, len(fields))
    if rc != 0:
        raise RuntimeError(f"orc_ALE_remap_tracers rc={rc}")


def ALE_remap_set_h_vel(d, G, h_new, h_u, h_v):
    assert lib().orc_ALE_remap_set_h_vel(C.byref(d), _p(G), _p(h_new), _p(h_u), _p(h_v)) == 0


def ALE_remap_velocities(d, G, CS, h_old_u, h_old_v, h_new_u, h_new_v, u, v):
    rc = lib().orc_ALE_remap_velocities(C.byref(d), _p(G), C.byref(CS), _p(h_old_u), _p(h_old_v), _p(h_new_u), _p(h_new_v), _p(u), _p(v))
    if rc != 0:
        raise RuntimeError(f"orc_ALE_remap_velocities rc={rc}")


def ALE_remap_velocities_conserve_ke(d, G, GV, CS, h_old_u, h_old_v, h_new_u, h_new_v, u, v):
    """ALE_remap_velocities with REMAP_VEL_CONSERVE_KE and allow_preserve_variance (MOM_ALE.F90:1166-1195)."""
    assert lib().orc_ALE_remap_velocities_conserve_ke(C.byref(d), _p(G), C.byref(CS), C.c_double(GV.H_subroundoff), _p(h_old_u), _p(h_old_v),
                                                      _p(h_new_u), _p(h_new_v), _p(u), _p(v)) == 0


def ALE_regrid_zstar(d, G, GV, CS, coordinateResolution, h, h_new, dzRegrid):
    cr = np.ascontiguousarray(coordinateResolution, dtype=np.float64)
    rc = lib().orc_ALE_regrid_zstar(C.byref(d), _p(G), C.byref(GV), C.byref(CS), _p(cr), _p(h), _p(h_new), _p(dzRegrid))
    if rc != 0:
        raise RuntimeError(f"orc_ALE_regrid_zstar rc={rc}")


def _opt(a):
    return None if a is None else _p(np.ascontiguousarray(a, dtype=np.float64))


def ALE_regrid_rho(d, G, GV, CS, eos, target_density, h, T, S, h_new, dzRegrid):
    td = np.ascontiguousarray(target_density, dtype=np.float64)
    rc = lib().orc_ALE_regrid_rho(C.byref(d), _p(G), C.byref(GV), C.byref(CS), C.byref(eos), _p(td), _p(h), _p(T), _p(S), _p(h_new),
                                  _p(dzRegrid))
    if rc != 0:
        raise RuntimeError(f"orc_ALE_regrid_rho rc={rc}")


def ALE_regrid_hycom1(d, G, GV, CS, eos, coordinateResolution, target_density, max_interface_depths, max_layer_thickness, h, T, S,
                      h_new, dzRegrid):
    cr = np.ascontiguousarray(coordinateResolution, dtype=np.float64)
    td = np.ascontiguousarray(target_density, dtype=np.float64)
    mid = None if max_interface_depths is None else np.ascontiguousarray(max_interface_depths, dtype=np.float64)
    mlt = None if max_layer_thickness is None else np.ascontiguousarray(max_layer_thickness, dtype=np.float64)
    rc = lib().orc_ALE_regrid_hycom1(C.byref(d), _p(G), C.byref(GV), C.byref(CS), C.byref(eos), _p(cr), _p(td),
                                     None if mid is None else _p(mid), None if mlt is None else _p(mlt), _p(h), _p(T), _p(S), _p(h_new),
                                     _p(dzRegrid))
    if rc != 0:
        raise RuntimeError(f"orc_ALE_regrid_hycom1 rc={rc}")


def ALE_convective_adjustment(d, eos, h, T, S):
    rc = lib().orc_ALE_convective_adjustment(C.byref(d), C.byref(eos), _p(h), _p(T), _p(S))
    if rc != 0:
        raise RuntimeError(f"orc_ALE_convective_adjustment rc={rc}")


def eos_density(eos, T, S, p):
    L = lib(); L.orc_eos_density.restype = C.c_double
    return L.orc_eos_density(C.byref(eos), C.c_double(T), C.c_double(S), C.c_double(p))


def eos_density_derivs(eos, T, S, p):
    a, b = C.c_double(), C.c_double()
    lib().orc_eos_density_derivs(C.byref(eos), C.c_double(T), C.c_double(S), C.c_double(p), C.byref(a), C.byref(b))
    return a.value, b.value


def vertvisc_coef(d, G, GV, CS, u, v, h, dt, a_u, a_v, h_u, h_v, Kv_bbl_u=None, Kv_bbl_v=None, bbl_thick_u=None,
                  bbl_thick_v=None, Kv_shear=None):
    rc = lib().orc_vertvisc_coef(C.byref(d), _p(G), C.byref(GV), C.byref(CS), _p(u), _p(v), _p(h), C.c_double(dt), _p(Kv_bbl_u),
                                 _p(Kv_bbl_v), _p(bbl_thick_u), _p(bbl_thick_v), _p(Kv_shear), _p(a_u), _p(a_v), _p(h_u), _p(h_v))
    if rc != 0:
        raise RuntimeError(f"orc_vertvisc_coef rc={rc}")


def vertvisc(d, G, GV, u, v, a_u, a_v, h_u, h_v, Ray_u, Ray_v, taux, tauy, dt, taux_bot=None, tauy_bot=None, Hmix_stress=0.0, h=None):
    rc = lib().orc_vertvisc(C.byref(d), _p(G), C.byref(GV), _p(u), _p(v), _p(a_u), _p(a_v), _p(h_u), _p(h_v), _p(Ray_u),
                            _p(Ray_v), _p(taux), _p(tauy), C.c_double(dt), _p(taux_bot), _p(tauy_bot), C.c_double(Hmix_stress), _p(h))
    assert rc == 0, rc


def vertvisc_remnant(d, G, visc_rem_u, visc_rem_v, a_u, a_v, h_u, h_v, Ray_u, Ray_v, dt):
    rc = lib().orc_vertvisc_remnant(C.byref(d), _p(G), _p(visc_rem_u), _p(visc_rem_v), _p(a_u), _p(a_v), _p(h_u), _p(h_v),
                                    _p(Ray_u), _p(Ray_v), C.c_double(dt))
    assert rc == 0, rc


# ------------------------------------------------------------------------------------------
# MOM_dynamics_split_RK2
from mom6_amd import abi as _abi


class ViscCoef(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("a_u", "a_v", "h_u", "h_v", "Ray_u", "Ray_v")]


class Rk2CS(C.Structure):
    _names3 = ["CAu", "CAv", "CAu_pred", "CAv_pred", "PFu", "PFv", "diffu", "diffv", "visc_rem_u", "visc_rem_v",
               "u_accel_bt", "v_accel_bt", "u_av", "v_av", "h_av", "pbce"]
    _names2 = ["eta", "eta_PF", "uhbt", "vhbt", "taux_bot", "tauy_bot"]
    _fields_ = [(n, C.c_void_p) for n in _names3 + _names2] + [("CAu_pred_stored", C.c_int)]


class Rk2All(C.Structure):
    _fields_ = [("d", C.c_void_p), ("G", C.c_void_p), ("GV", C.c_void_p), ("cont", C.c_void_p), ("bt", C.c_void_p),
                ("cor", C.c_void_p), ("pgf", C.c_void_p), ("rk2", C.c_void_p), ("Rlay", C.c_void_p), ("g_prime", C.c_void_p),
                ("CS", C.c_void_p), ("BTCS", C.c_void_p), ("BT_cont", C.c_void_p), ("first_direction", C.c_int),
                ("T", C.c_void_p), ("S", C.c_void_p), ("eos", C.c_void_p), ("vv", C.c_void_p),
                ("Kv_bbl_u", C.c_void_p), ("Kv_bbl_v", C.c_void_p), ("bbl_thick_u", C.c_void_p), ("bbl_thick_v", C.c_void_p),
                ("Kv_shear", C.c_void_p), ("Ray_u", C.c_void_p), ("Ray_v", C.c_void_p),
                ("vv_a_u", C.c_void_p), ("vv_a_v", C.c_void_p), ("vv_h_u", C.c_void_p), ("vv_h_v", C.c_void_p),
                ("hv", C.c_void_p), ("hv_planes", C.c_void_p), ("Hmix_stress", C.c_double)]


class OrcModel:
    """The oracle's model instance: every control structure step_MOM_dyn_split_RK2 reaches."""

    def __init__(self, d, M, GV, cont, bt, cor, pgf, rk2, Rlay, g_prime, first_direction=0):
        self.d, self.M, self.GV = d, M, GV
        self.cont, self.bt, self.cor, self.pgf, self.rk2 = cont, bt, cor, pgf, rk2
        self.Rlay = np.ascontiguousarray(Rlay); self.g_prime = np.ascontiguousarray(g_prime)
        self.f = {n: np.zeros(d.shape3()) for n in Rk2CS._names3}
        self.f.update({n: np.zeros(d.shape2()) for n in Rk2CS._names2})
        self.cs = Rk2CS()
        for n in Rk2CS._names3 + Rk2CS._names2:
            setattr(self.cs, n, self.f[n].ctypes.data)
        self.cs.CAu_pred_stored = 0
        self.btcs = BtState(d)
        barotropic_init(d, M, GV, bt, self.btcs)
        self.bt_cont = new_bt_cont(d)
        self.bt_cont_s = bt_cont_struct(self.bt_cont)
        A = Rk2All()
        A.d = C.addressof(d); A.G = M.ctypes.data; A.GV = C.addressof(GV); A.cont = C.addressof(cont); A.bt = C.addressof(bt)
        A.cor = C.addressof(cor); A.pgf = C.addressof(pgf); A.rk2 = C.addressof(rk2)
        A.Rlay = self.Rlay.ctypes.data; A.g_prime = self.g_prime.ctypes.data
        A.CS = C.addressof(self.cs); A.BTCS = C.addressof(self.btcs.struct); A.BT_cont = C.addressof(self.bt_cont_s)
        A.first_direction = first_direction
        A.T = None; A.S = None; A.eos = None; A.vv = None; A.hv = None; A.hv_planes = None; A.Hmix_stress = 0.0
        self.A = A

    def set_vertvisc(self, vv, Kv_bbl_u=None, Kv_bbl_v=None, bbl_thick_u=None, bbl_thick_v=None, Kv_shear=None,
                     Ray_u=None, Ray_v=None):
        """vertvisc_init + the vertvisc_type inputs: the step then calls vertvisc_coef itself (coefs argument ignored)."""
        d = self.d
        self._vv = (vv, Kv_bbl_u, Kv_bbl_v, bbl_thick_u, bbl_thick_v, Kv_shear, Ray_u, Ray_v)
        self.vv_out = dict(a_u=np.zeros((d.nk + 1,) + d.shape2()), a_v=np.zeros((d.nk + 1,) + d.shape2()),
                           h_u=np.zeros(d.shape3()), h_v=np.zeros(d.shape3()))
        A = self.A
        A.vv = C.addressof(vv)
        for n, a in (("Kv_bbl_u", Kv_bbl_u), ("Kv_bbl_v", Kv_bbl_v), ("bbl_thick_u", bbl_thick_u), ("bbl_thick_v", bbl_thick_v),
                     ("Kv_shear", Kv_shear), ("Ray_u", Ray_u), ("Ray_v", Ray_v)):
            setattr(A, n, a.ctypes.data if a is not None else None)
        A.vv_a_u = self.vv_out["a_u"].ctypes.data; A.vv_a_v = self.vv_out["a_v"].ctypes.data
        A.vv_h_u = self.vv_out["h_u"].ctypes.data; A.vv_h_v = self.vv_out["h_v"].ctypes.data

    def set_direct_stress(self, Hmix_stress):
        """DIRECT_STRESS with HMIX_STRESS = Hmix_stress [H] in the step's vertvisc calls (0: off)."""
        self.A.Hmix_stress = float(Hmix_stress)

    def set_hor_visc(self, hv):
        """hor_visc_init: the step and the new-run initialisation then call horizontal_viscosity themselves."""
        self._hv = hv
        self.hv_planes = hor_visc_init(self.d, self.M, hv)
        self.A.hv = C.addressof(hv); self.A.hv_planes = self.hv_planes.ctypes.data

    def set_tv(self, T, S, eos):
        """tv%T, tv%S, tv%eqn_of_state for the PressureForce calls of the step (None: layered path)."""
        self._tv = (T, S, eos)
        self.A.T = T.ctypes.data if T is not None else None
        self.A.S = S.ctypes.data if S is not None else None
        self.A.eos = C.addressof(eos) if eos is not None else None

    def __getitem__(self, n):
        return self.f[n]

    def remap_aux_vars(self, CS, h_old_u, h_old_v, h_new_u, h_new_v):
        """remap_dyn_split_RK2_aux_vars (MOM_dynamics_split_RK2.F90:1302-1330); the pass_vector calls are the caller's
        (one closed tile: nothing to exchange)."""
        if not self.rk2.remap_aux:
            return
        if self.rk2.store_CAu:
            ALE_remap_velocities(self.d, self.M, CS, h_old_u, h_old_v, h_new_u, h_new_v, self.f["u_av"], self.f["v_av"])
            ALE_remap_velocities(self.d, self.M, CS, h_old_u, h_old_v, h_new_u, h_new_v, self.f["CAu_pred"], self.f["CAv_pred"])
        ALE_remap_velocities(self.d, self.M, CS, h_old_u, h_old_v, h_new_u, h_new_v, self.f["diffu"], self.f["diffv"])

    def initialize(self, u, v, h, uh, vh, dt):
        rc = lib().orc_initialize_dyn_split_RK2(C.byref(self.A), _p(u), _p(v), _p(h), _p(uh), _p(vh), C.c_double(dt))
        if rc != 0:
            raise RuntimeError(f"orc_initialize_dyn_split_RK2 rc={rc}")

    def restart_fills(self, u, v, h, uh, vh, dt, have):
        """initialize_dyn_split_RK2 :1577-1668 for a restarted run (have: abi.RK2_HAVE_* bits of what the file held)."""
        rc = lib().orc_restart_fills_dyn_split_RK2(C.byref(self.A), _p(u), _p(v), _p(h), _p(uh), _p(vh), C.c_double(dt), C.c_int(have))
        if rc != 0:
            raise RuntimeError(f"orc_restart_fills_dyn_split_RK2 rc={rc}")

    def step(self, u, v, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy, dt, coefs, calc_dtbt=False, diffu_new=None, diffv_new=None):
        """coefs: list of 3 tuples (a_u, a_v, h_u, h_v, Ray_u, Ray_v) or a single tuple used for all stages."""
        if not isinstance(coefs, list):
            coefs = [coefs] * 3
        arr = (ViscCoef * 3)()
        for s in range(3):
            for n, a in zip(("a_u", "a_v", "h_u", "h_v", "Ray_u", "Ray_v"), coefs[s]):
                setattr(arr[s], n, a.ctypes.data if a is not None else None)
        rc = lib().orc_step_dyn_split_RK2(C.byref(self.A), _p(u), _p(v), _p(h), _p(uh), _p(vh), _p(uhtr), _p(vhtr), _p(eta_av),
                                          _p(taux), _p(tauy), C.c_double(dt), C.c_int(int(calc_dtbt)), arr, _p(diffu_new), _p(diffv_new))
        if rc != 0:
            raise RuntimeError(f"orc_step_dyn_split_RK2 rc={rc}")


# ------------------------------------------------------------------------------------------
# MOM_tracer_advect / MOM_diabatic_aux / MOM_tracer_diabatic
def advect_tracer(d, G, GV, first_direction, dt_dyn, default_scheme, h_end, uhtr, vhtr, dt, tracers, schemes=None,
                  useHuynhStencilBug=False, x_first_in=-1, max_iter_in=0, uhr_out=None, vhr_out=None):
    ntr = len(tracers)
    ptrs = (C.c_void_p * ntr)(*[t.ctypes.data for t in tracers])
    sch = (C.c_int * ntr)(*(schemes if schemes is not None else [-1] * ntr))
    iters = C.c_int(0)
    rc = lib().orc_advect_tracer(C.byref(d), _p(G), C.byref(GV), C.c_int(first_direction), C.c_double(dt_dyn),
                                 C.c_int(default_scheme), C.c_int(int(useHuynhStencilBug)), _p(h_end), _p(uhtr), _p(vhtr),
                                 C.c_double(dt), ptrs, sch, C.c_int(ntr), C.c_int(x_first_in), C.c_int(max_iter_in),
                                 _p(uhr_out), _p(vhr_out), C.byref(iters))
    if rc != 0:
        raise RuntimeError(f"orc_advect_tracer rc={rc}")
    return iters.value


def triDiagTS(d, hold, ea, eb, T, h_neglect):
    assert lib().orc_triDiagTS(C.byref(d), _p(hold), _p(ea), _p(eb), _p(T), C.c_double(h_neglect)) == 0


def triDiagTS_Eulerian(d, hold, ent, T, h_neglect):
    assert lib().orc_triDiagTS_Eulerian(C.byref(d), _p(hold), _p(ent), _p(T), C.c_double(h_neglect)) == 0


def tracer_vertdiff(d, G, GV, h_old, ea, eb, dt, tr, sfc_flux=None, btm_flux=None, convert_flux=True):
    assert lib().orc_tracer_vertdiff(C.byref(d), _p(G), C.byref(GV), _p(h_old), _p(ea), _p(eb), C.c_double(dt), _p(tr),
                                     _p(sfc_flux), _p(btm_flux), C.c_int(int(convert_flux))) == 0


# ---- MOM_coms reproducing sums, MOM_checksums (orc_sums.c) -----------------------------------------------------------
_EFP_FATAL = {1: "NaN in input field of reproducing_sum", 2: "Overflow in reproducing_sum conversion", 3: "Overflow in reproducing_sum"}


def tracer_vertdiff_sink(d, G, GV, h_old, ea, eb, dt, tr, sink_rate, sfc_flux=None, btm_flux=None, btm_reservoir=None, convert_flux=True):
    """tracer_vertdiff / tracer_vertdiff_Eulerian with sink_rate (MOM_tracer_diabatic.F90:123-179 / :315-380)."""
    assert lib().orc_tracer_vertdiff_sink(C.byref(d), _p(G), C.byref(GV), _p(h_old), _p(ea), _p(eb), C.c_double(dt), _p(tr), _p(sfc_flux),
                                          _p(btm_flux), _p(btm_reservoir), C.c_double(sink_rate), C.c_int(int(convert_flux))) == 0


def reproducing_sum(d, array, is_=None, ie=None, js=None, je=None, unscale=1.0, layer_sums=False, want_err=False):
    """reproducing_sum_3d (array of nk planes) or reproducing_sum_2d (one plane).  Returns a dict: sum, [sums], EFP
    (int64[6]), [EFP_lay (int64[nk,6])], [err]."""
    a = np.ascontiguousarray(array)
    is_ = 0 if is_ is None else is_; js = 0 if js is None else js
    ie = d.ni - 1 if ie is None else ie; je = d.nj - 1 if je is None else je
    out = {}
    s = C.c_double(0.0); efp = np.zeros(6, dtype=np.int64); err = C.c_int(0)
    perr = C.byref(err) if want_err else None
    if a.ndim == 2:
        lib().orc_reproducing_sum_2d.restype = C.c_int
        rc = lib().orc_reproducing_sum_2d(C.byref(d), _p(a), is_, ie, js, je, C.c_double(unscale), C.byref(s), _p(efp), perr)
    else:
        nk = a.shape[0]
        sums = np.zeros(nk) if layer_sums else None
        lay = np.zeros((nk, 6), dtype=np.int64) if layer_sums else None
        rc = lib().orc_reproducing_sum_3d(C.byref(d), _p(a), nk, is_, ie, js, je, C.c_double(unscale), C.byref(s), _p(sums),
                                          _p(efp), _p(lay), perr)
        if layer_sums:
            out.update(sums=sums, EFP_lay=lay)
    if rc:
        raise RuntimeError(_EFP_FATAL[rc])
    out.update(sum=s.value, EFP=efp)
    if want_err:
        out["err"] = err.value
    return out


def reproducing_EFP_sum_2d(d, array, is_, ie, js, je, unscale=1.0, overflow_check=True):
    efp = np.zeros(6, dtype=np.int64)
    rc = lib().orc_reproducing_EFP_sum_2d(C.byref(d), _p(np.ascontiguousarray(array)), is_, ie, js, je, int(overflow_check),
                                          C.c_double(unscale), _p(efp), None)
    if rc:
        raise RuntimeError(_EFP_FATAL[rc])
    return efp


def EFP_to_real(efp):
    lib().orc_EFP_to_real.restype = C.c_double
    e = np.array(efp, dtype=np.int64)
    return lib().orc_EFP_to_real(_p(e))


def EFP_minus(a, b):
    out = np.zeros(6, dtype=np.int64)
    lib().orc_EFP_minus(_p(np.array(a, dtype=np.int64)), _p(np.array(b, dtype=np.int64)), _p(out))
    return out


def real_to_EFP(r):
    out = np.zeros(6, dtype=np.int64); over = C.c_int(0)
    lib().orc_real_to_ints(C.c_double(r), C.c_int64(-1), C.byref(over), _p(out))
    if over.value:
        raise RuntimeError("Overflow in real_to_EFP conversion")
    return out


def chksum(d, array, stagger, haloshift=0, symmetric=False, omit_corners=False, scale=None):
    """chksum_{h,u,v,B}_{2d,3d}: dict(mean, min, max, bc0, bc=[shifted bitcounts in message order])."""
    a = np.ascontiguousarray(array)
    nk = 1 if a.ndim == 2 else a.shape[0]
    stats = np.zeros(3); bc = np.zeros(5, dtype=np.int32)
    n = lib().orc_chksum(C.byref(d), _p(a), nk, a.ndim, "huvB".index(stagger), haloshift, int(symmetric), int(omit_corners),
                         int(scale is not None), C.c_double(1.0 if scale is None else scale), _p(stats), _p(bc))
    if n < 0:
        raise RuntimeError("NaN detected")
    return dict(mean=stats[0], min=stats[1], max=stats[2], bc0=int(bc[0]), bc=[int(x) for x in bc[1:1 + n]])


def field_chksum(d, array, is_, ie, js, je, unscale=1.0):
    a = np.ascontiguousarray(array)
    nk = 1 if a.ndim == 2 else a.shape[0]
    lib().orc_field_chksum.restype = C.c_int64
    return lib().orc_field_chksum(C.byref(d), _p(a), nk, is_, ie, js, je, C.c_double(unscale))


# ---- MOM_sum_output (orc_sums.c) -------------------------------------------------------------------------------------
def create_depth_list(d, G, Z_ref=0.0, min_depth_inc=1.0e-10):
    n = d.ni_glob * d.nj_glob + 3
    dep, area, vol = np.zeros(n), np.zeros(n), np.zeros(n)
    ls = lib().orc_create_depth_list(C.byref(d), _p(G), C.c_double(Z_ref), C.c_double(min_depth_inc), _p(dep), _p(area), _p(vol))
    return dep[1:ls + 1].copy(), area[1:ls + 1].copy(), vol[1:ls + 1].copy()


class SumOutputState:
    """The part of Sum_output_CS the sums of write_energy read and update: the depth list and the search hints lH."""

    def __init__(self, d, G, GV, g_prime, P):
        self.d, self.G, self.GV, self.P = d, G, GV, P
        self.g_prime = np.ascontiguousarray(g_prime, dtype=np.float64)
        self.DL = create_depth_list(d, G, P.Z_ref, P.D_list_min_inc) if P.do_APE_calc else (np.zeros(1), np.zeros(1), np.zeros(1))
        self.listsize = len(self.DL[0]) if P.do_APE_calc else 0
        self.lH = np.full(d.nk, self.listsize - 1, dtype=np.int32)


def write_energy(st, u, v, h, T=None, S=None):
    d = st.d; nk = d.nk
    one = [np.concatenate(([0.0], a, [0.0])) for a in st.DL]            # 1-based inside
    vec = dict(mass_lay=np.zeros(nk), KE=np.zeros(nk), PE=np.zeros(nk + 1), Z_0APE=np.zeros(nk + 1))
    scal = np.zeros(5); efp = [np.zeros(6, dtype=np.int64) for _ in range(3)]
    rc = lib().orc_write_energy(C.byref(d), _p(st.G), C.byref(st.GV), _p(st.g_prime), int(st.P.do_APE_calc), C.c_double(st.P.dt_in_T),
                                C.c_double(st.P.Z_ref), C.c_double(st.P.C_p), st.listsize, _p(one[0]), _p(one[1]), _p(one[2]),
                                _p(st.lH), _p(u), _p(v), _p(h), _p(T), _p(S), _p(vec["mass_lay"]), _p(vec["KE"]), _p(vec["PE"]),
                                _p(vec["Z_0APE"]), _p(scal), _p(efp[0]), _p(efp[1]), _p(efp[2]))
    if rc:
        raise RuntimeError(_EFP_FATAL[rc])
    out = dict(mass_tot=scal[0], KE_tot=scal[1], PE_tot=scal[2], max_CFL=(scal[3], scal[4]), mass_EFP=efp[0], salt_EFP=efp[1],
               heat_EFP=efp[2])
    out.update(vec)
    return out
