"""Host-side mirror of the reference's Fortran interface for the split-RK2 hot path.

Each method has the name and argument meaning of the reference module procedure it
stands in for and forwards to the C ABI (include/mom6x.h) with DEVICE pointers taken
from torch tensors (torch is only used for HBM allocation and streams).  There is no
CPU fallback: constructing a `Dycore` without the HIP library or without a GPU raises.

Fields are torch.float64 CUDA tensors in the pitched tile layout:
  2-D: shape (nj+2*halo+1, pitch); 3-D: shape (nk, nj+2*halo+1, pitch).

Streams: every call is enqueued on the context's own HIP stream (Dycore.stream_ptr), not on torch's.  A torch
operation on an array the context is still working on (zero_(), copy_(), .cpu()) must be preceded by Dycore.sync(), and a
call into the context after torch wrote an array by torch.cuda.synchronize().
"""
import ctypes as C

import numpy as np
import torch

from . import abi
from .abi import BTCont, check, load_library


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), "need contiguous float64 CUDA tensor"
    return C.c_void_p(t.data_ptr())


class BTContDev:
    """BT_cont_type (MOM_variables.F90:315-350) on the device."""

    def __init__(self, dyc, with_h=True):
        self.f = {n: dyc.zeros2() for n in BTCont._names[:12]}
        self.f["h_u"] = dyc.zeros3() if with_h else None
        self.f["h_v"] = dyc.zeros3() if with_h else None
        self.struct = BTCont()
        for n in BTCont._names:
            t = self.f[n]
            setattr(self.struct, n, t.data_ptr() if t is not None else None)

    def __getitem__(self, n):
        return self.f[n]


class Dycore:
    """One tile of the dynamical core on one MI355X."""

    def __init__(self, dims, metrics, GV=None, first_direction=0, device=0):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("mom6_amd.Dycore needs a GPU (no CPU fallback on the product path)")
        self.dims = dims
        self.GV = GV if GV is not None else abi.vgrid_default()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.metrics_host = np.ascontiguousarray(metrics, dtype=np.float64)
        assert self.metrics_host.shape == (abi.G_COUNT,) + dims.shape2()
        self.ctx = C.c_void_p()
        check(self.lib, self.lib.mom6x_ctx_create(C.byref(self.ctx), C.byref(dims), C.c_int(device),
                                                  self.metrics_host.ctypes.data_as(C.c_void_p),
                                                  C.byref(self.GV), C.c_int(first_direction)))
        self._keep = []

    def close(self):
        if self.ctx:
            try:   # torch must not keep the context's stream as its current one (see torch_stream) once it is destroyed
                if torch.cuda.current_stream(self.device).cuda_stream == self.stream_ptr:
                    torch.cuda.synchronize(self.device)
                    torch.cuda.set_stream(torch.cuda.default_stream(self.device))
            except Exception:
                pass
            self.lib.mom6x_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ------------------------------------------------------------------------------
    def zeros2(self):
        # (filled on the context's stream: a kernel of the context launched right after it must not overtake the fill)
        with torch.cuda.stream(self.torch_stream()):
            return torch.zeros(self.dims.shape2(), dtype=torch.float64, device=self.device)

    def zeros3(self, nk=None):
        with torch.cuda.stream(self.torch_stream()):
            return torch.zeros(self.dims.shape3(nk), dtype=torch.float64, device=self.device)

    def to_dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def sync(self):
        """Wait for the context's streams; raises on a device-side numeric error flag."""
        check(self.lib, self.lib.mom6x_ctx_sync(self.ctx))

    @property
    def stream_ptr(self):
        return self.lib.mom6x_ctx_stream(self.ctx)

    def torch_stream(self):
        """The context's compute stream as a torch stream: torch work issued under `with torch.cuda.stream(dyc.torch_stream())`
        is ordered with the context's kernels, so no synchronisation is needed between the two (see the module docstring)."""
        return torch.cuda.ExternalStream(self.stream_ptr, device=self.device)

    # -- MOM_continuity_PPM ------------------------------------------------------------------
    def continuity_init(self, params=None):
        """continuity_PPM_init (MOM_continuity_PPM.F90:2674)."""
        self.cont_params = params if params is not None else abi.continuity_params_default(self.dims.nk, self.GV.Angstrom_H)
        check(self.lib, self.lib.mom6x_continuity_init(self.ctx, C.byref(self.cont_params)))

    def continuity_PPM(self, u, v, hin, h, uh, vh, dt, uhbt=None, vhbt=None, visc_rem_u=None,
                       visc_rem_v=None, u_cor=None, v_cor=None, BT_cont=None, du_cor=None, dv_cor=None):
        """continuity_PPM (MOM_continuity_PPM.F90:86): optional arguments are None when absent."""
        bt = C.byref(BT_cont.struct) if BT_cont is not None else None
        check(self.lib, self.lib.mom6x_continuity_PPM(
            self.ctx, _ptr(u), _ptr(v), _ptr(hin), _ptr(h), _ptr(uh), _ptr(vh), C.c_double(dt),
            _ptr(uhbt), _ptr(vhbt), _ptr(visc_rem_u), _ptr(visc_rem_v), _ptr(u_cor), _ptr(v_cor), bt,
            _ptr(du_cor), _ptr(dv_cor)))

    continuity = continuity_PPM  # MOM_continuity.F90:6-28 pass-through name

    # -- MOM_barotropic ----------------------------------------------------------------------
    def barotropic_init(self, params):
        """barotropic_init (MOM_barotropic.F90:5301): static fields q_D, D_[uv]_Cor, IDat[uv]."""
        self.bt_params = params
        check(self.lib, self.lib.mom6x_barotropic_init(self.ctx, C.byref(params)))

    _BT_FIELDS = {"ubtav": (0, 2), "vbtav": (1, 2), "eta_cor": (2, 2), "frhatu": (3, 3), "frhatv": (4, 3),
                  "IDatu": (5, 2), "IDatv": (6, 2), "q_D": (7, 2), "D_u_Cor": (8, 2), "D_v_Cor": (9, 2)}

    def barotropic_field(self, name):
        """A zero-copy torch view of a barotropic_CS array owned by the context."""
        which, nd = self._BT_FIELDS[name]
        ptr = self.lib.mom6x_barotropic_field(self.ctx, C.c_int(which))
        shape = self.dims.shape2() if nd == 2 else self.dims.shape3()
        return _view(ptr, shape, self.device)

    def btstep_warnings(self, reset=True, warn=True):
        """The count of btstep's "eta has dropped below bathyT" warnings (MOM_barotropic.F90:2738-2745) over all sub-steps
        since the last reset, and the first offender (eta, -bathyT, i, j).  With `warn`, the message of the reference goes
        out as a Python warning."""
        n = C.c_longlong(0)
        info = (C.c_double * 4)()
        check(self.lib, self.lib.mom6x_btstep_warnings(self.ctx, C.c_int(int(reset)), C.byref(n), info))
        first = dict(eta=info[0], minus_bathyT=info[1], i=int(info[2]), j=int(info[3])) if n.value else None
        if n.value and warn:
            import warnings
            warnings.warn("btstep: eta has dropped below bathyT: %24.16E vs. %24.16E at i, j = %d, %d (%d occurrences)" %
                          (info[0], info[1], int(info[2]), int(info[3]), n.value), RuntimeWarning)
        return n.value, first

    def continuity_stats(self, mode=-1):
        """mom6x_continuity_stats: (flux re-evaluations, Newton solves, exact-limit redos) of the wave-owned mass-flux kernel,
        counted per wavefront of four face columns while collection is on (mode 1: on + reset, 0: off, -1: read)."""
        out = (C.c_ulonglong * 3)()
        check(self.lib, self.lib.mom6x_continuity_stats(self.ctx, int(mode), out))
        return int(out[0]), int(out[1]), int(out[2])

    def barotropic_dtbt(self, value=None):
        """CS%dtbt (the restart scalar DTBT); with a value, set it (a restarted run)."""
        out = C.c_double(0.0)
        check(self.lib, self.lib.mom6x_barotropic_dtbt(self.ctx, C.byref(out), None if value is None else C.byref(C.c_double(value))))
        return out.value

    def btcalc(self, h, h_u=None, h_v=None, may_use_default=True):
        """btcalc (MOM_barotropic.F90:4360)."""
        f = self.lib.mom6x_btcalc if may_use_default else self.lib.mom6x_btcalc_strict
        check(self.lib, f(self.ctx, _ptr(h), _ptr(h_u), _ptr(h_v)))

    def set_dtbt_pbce(self, pbce, eta=None, SSH_add=0.0):
        """set_dtbt(G, GV, US, CS, pbce, eta=eta, SSH_add=) without BT_cont (MOM_barotropic.F90:3576-3582); returns CS%dtbt."""
        out = C.c_double(0.0)
        check(self.lib, self.lib.mom6x_set_dtbt_pbce_eta(self.ctx, _ptr(pbce), _ptr(eta), C.c_double(SSH_add), C.byref(out)))
        return out.value

    def bt_mass_source(self, h, eta, set_cor):
        """bt_mass_source (MOM_barotropic.F90:5243)."""
        check(self.lib, self.lib.mom6x_bt_mass_source(self.ctx, _ptr(h), _ptr(eta), C.c_int(int(set_cor))))

    def set_dtbt(self, pbce=None, gtot_est=0.0, SSH_add=0.0):
        """set_dtbt (MOM_barotropic.F90:3509); returns CS%dtbt."""
        out = C.c_double(0.0)
        check(self.lib, self.lib.mom6x_set_dtbt(self.ctx, _ptr(pbce), C.c_double(gtot_est), C.c_double(SSH_add), C.byref(out)))
        return out.value

    def btstep(self, U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, taux, tauy, pbce, eta_PF_in, U_Cor, V_Cor,
               accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v, BT_cont,
               taux_bot=None, tauy_bot=None, uh0=None, vh0=None, u_uh0=None, v_vh0=None, etaav=None):
        """btstep (MOM_barotropic.F90:455); forces%taux/tauy are passed as planes.  BT_cont = None: USE_BT_CONT_TYPE = False (the
        barotropic continuity equation linear in the velocities, with the face areas of find_face_areas)."""
        check(self.lib, self.lib.mom6x_btstep(
            self.ctx, _ptr(U_in), _ptr(V_in), _ptr(eta_in), C.c_double(dt), _ptr(bc_accel_u), _ptr(bc_accel_v),
            _ptr(taux), _ptr(tauy), _ptr(pbce), _ptr(eta_PF_in), _ptr(U_Cor), _ptr(V_Cor), _ptr(accel_layer_u),
            _ptr(accel_layer_v), _ptr(eta_out), _ptr(uhbtav), _ptr(vhbtav), _ptr(visc_rem_u), _ptr(visc_rem_v),
            C.byref(BT_cont.struct) if BT_cont is not None else None, _ptr(taux_bot), _ptr(tauy_bot), _ptr(uh0), _ptr(vh0), _ptr(u_uh0),
            _ptr(v_vh0), _ptr(etaav)))

    # -- MOM_CoriolisAdv ---------------------------------------------------------------------
    def CoriolisAdv_init(self, params=None):
        """CoriolisAdv_init (MOM_CoriolisAdv.F90:1054)."""
        self.cor_params = params if params is not None else abi.coriolis_params_default()
        check(self.lib, self.lib.mom6x_CoriolisAdv_init(self.ctx, C.byref(self.cor_params)))

    def CorAdCalc(self, u, v, h, uh, vh, CAu, CAv):
        """CorAdCalc (MOM_CoriolisAdv.F90:125)."""
        check(self.lib, self.lib.mom6x_CorAdCalc(self.ctx, _ptr(u), _ptr(v), _ptr(h), _ptr(uh), _ptr(vh), _ptr(CAu), _ptr(CAv)))

    # -- MOM_PressureForce -------------------------------------------------------------------
    def PressureForce_init(self, params, Rlay, g_prime):
        """PressureForce_init (MOM_PressureForce.F90:85) for the analytic FV Boussinesq PGF."""
        self.pgf_params = params
        self._Rlay = np.ascontiguousarray(Rlay, dtype=np.float64)
        self._g_prime = np.ascontiguousarray(g_prime, dtype=np.float64)
        check(self.lib, self.lib.mom6x_PressureForce_init(self.ctx, C.byref(params), self._Rlay.ctypes.data_as(C.c_void_p),
                                                          self._g_prime.ctypes.data_as(C.c_void_p)))

    def PressureForce(self, h, PFu, PFv, pbce=None, eta=None):
        """PressureForce (MOM_PressureForce.F90:41) -> PressureForce_FV_Bouss (FV.F90:947)."""
        check(self.lib, self.lib.mom6x_PressureForce(self.ctx, _ptr(h), _ptr(PFu), _ptr(PFv), _ptr(pbce), _ptr(eta)))

    def ALE_PLM_edge_values(self, h, Q, bdry_extrap, Q_t, Q_b):
        """ALE_PLM_edge_values (MOM_ALE.F90:1520): top and bottom values of the PLM reconstruction of Q in every layer."""
        check(self.lib, self.lib.mom6x_ALE_PLM_edge_values(self.ctx, _ptr(h), _ptr(Q), C.c_int(int(bdry_extrap)), _ptr(Q_t), _ptr(Q_b)))

    def set_dyn_pass_width(self, width):
        """NIHALO rows for the 3-D group passes of the RK2 step in a context whose halo was widened for BTHALO (0: the context's)."""
        check(self.lib, self.lib.mom6x_set_dyn_pass_width(self.ctx, C.c_int(int(width))))

    def comm_overlap_btstep(self, on=True):
        """btstep's own group pass overlapped with the own-points half of the next sub-step (mom6x_comm_overlap_btstep; off by default)."""
        check(self.lib, self.lib.mom6x_comm_overlap_btstep(self.ctx, C.c_int(1 if on else 0)))

    def comm_exchange_count(self, reset=False):
        """Packed group exchanges since the last reset (mom6x_comm_exchange_count)."""
        f = self.lib.mom6x_comm_exchange_count
        f.restype = C.c_longlong
        return int(f(self.ctx, C.c_int(1 if reset else 0)))

    def comm_exchange_bytes(self, reset=False):
        """Bytes this tile sent in packed group exchanges since the last reset (mom6x_comm_exchange_bytes)."""
        f = self.lib.mom6x_comm_exchange_bytes
        f.restype = C.c_longlong
        return int(f(self.ctx, C.c_int(1 if reset else 0)))

    def ALE_PPM_edge_values(self, h, Q, bdry_extrap, Q_t, Q_b):
        """One field of TS_PPM_edge_values (MOM_ALE.F90:1581): edge_values_implicit_h4 + PPM_reconstruction edge values."""
        check(self.lib, self.lib.mom6x_ALE_PPM_edge_values(self.ctx, _ptr(h), _ptr(Q), C.c_int(int(bdry_extrap)), _ptr(Q_t), _ptr(Q_b)))

    def PressureForce_set_tv(self, T, S, eos):
        """tv%T, tv%S, tv%eqn_of_state of PressureForce's thermo_var_ptrs argument; T=None: layered path."""
        self._tv = (T, S, eos)
        check(self.lib, self.lib.mom6x_PressureForce_set_tv(self.ctx, _ptr(T), _ptr(S), C.byref(eos) if eos is not None else None))

    # -- MOM_vert_friction -------------------------------------------------------------------
    def vertvisc_set_coef(self, a_u, a_v, h_u, h_v, Ray_u=None, Ray_v=None):
        """Hand CS%a_u, CS%a_v, CS%h_u, CS%h_v (and visc%Ray_u/v) of vertvisc_coef to the context."""
        self._vv = (a_u, a_v, h_u, h_v, Ray_u, Ray_v)
        check(self.lib, self.lib.mom6x_vertvisc_set_coef(self.ctx, _ptr(a_u), _ptr(a_v), _ptr(h_u), _ptr(h_v), _ptr(Ray_u), _ptr(Ray_v)))

    def vertvisc_init(self, params):
        """vertvisc_init (MOM_vert_friction.F90:3135): CS%a_u, a_v, h_u, h_v now live in the context."""
        self.vv_params = params
        check(self.lib, self.lib.mom6x_vertvisc_init(self.ctx, C.byref(params)))

    def vertvisc_set_visc(self, Kv_bbl_u=None, Kv_bbl_v=None, bbl_thick_u=None, bbl_thick_v=None, Kv_shear=None, Ray_u=None, Ray_v=None):
        """The vertvisc_type members vertvisc_coef / vertvisc read (set_viscous_BBL etc. stay on the host)."""
        self._visc = (Kv_bbl_u, Kv_bbl_v, bbl_thick_u, bbl_thick_v, Kv_shear, Ray_u, Ray_v)
        check(self.lib, self.lib.mom6x_vertvisc_set_visc(self.ctx, *[_ptr(a) for a in self._visc]))

    def vertvisc_field(self, name):
        """CS%a_u / a_v / h_u / h_v of the device vertvisc_CS as a torch view."""
        which = ["a_u", "a_v", "h_u", "h_v"].index(name)
        self.lib.mom6x_vertvisc_field.restype = C.c_void_p
        p = self.lib.mom6x_vertvisc_field(self.ctx, which)
        nlev = self.dims.nk + (1 if which < 2 else 0)
        return _view(p, (nlev,) + self.dims.shape2(), self.device)

    def vertvisc_coef(self, u, v, h, dt):
        """vertvisc_coef (MOM_vert_friction.F90:1357) with dz = H_to_Z*h."""
        check(self.lib, self.lib.mom6x_vertvisc_coef(self.ctx, _ptr(u), _ptr(v), _ptr(h), C.c_double(dt)))

    def vertvisc_set_direct_stress(self, Hmix_stress, h=None):
        """DIRECT_STRESS / HMIX_STRESS (MOM_vert_friction.F90:3208, :707); h = vertvisc's thickness argument."""
        self._ds_h = h
        check(self.lib, self.lib.mom6x_vertvisc_set_direct_stress(self.ctx, C.c_double(Hmix_stress), _ptr(h)))

    def hor_visc_init(self, params):
        """hor_visc_init (MOM_hor_visc.F90:2322): the 2-D viscosity planes are computed on the device.  From here on
        step_dyn_split_RK2 calls horizontal_viscosity itself unless a host callback is given."""
        self.hv_params = params
        check(self.lib, self.lib.mom6x_hor_visc_init(self.ctx, C.byref(params)))

    def horizontal_viscosity(self, u, v, h, diffu, diffv):
        """horizontal_viscosity (MOM_hor_visc.F90:266)."""
        check(self.lib, self.lib.mom6x_horizontal_viscosity(self.ctx, _ptr(u), _ptr(v), _ptr(h), _ptr(diffu), _ptr(diffv)))

    # ---- MOM_remapping / MOM_ALE
    def ALE_remap_tracers(self, CS, h_old, h_new, fields):
        """ALE_remap_tracers (MOM_ALE.F90:760): remapping_core_h of every wet column of every field, in place."""
        ptrs = (C.c_void_p * len(fields))(*[f.data_ptr() for f in fields])
        check(self.lib, self.lib.mom6x_ALE_remap_tracers(self.ctx, C.byref(CS), _ptr(h_old), _ptr(h_new), ptrs, len(fields)))

    def ALE_remap_set_h_vel(self, h_new, h_u, h_v):
        """ALE_remap_set_h_vel (MOM_ALE.F90:882)."""
        check(self.lib, self.lib.mom6x_ALE_remap_set_h_vel(self.ctx, _ptr(h_new), _ptr(h_u), _ptr(h_v)))

    def ALE_remap_velocities(self, CS, h_old_u, h_old_v, h_new_u, h_new_v, u, v, conserve_ke=False):
        """ALE_remap_velocities (MOM_ALE.F90:1089); conserve_ke: REMAP_VEL_CONSERVE_KE with allow_preserve_variance (:1166-1195)."""
        f = self.lib.mom6x_ALE_remap_velocities_conserve_ke if conserve_ke else self.lib.mom6x_ALE_remap_velocities
        check(self.lib, f(self.ctx, C.byref(CS), _ptr(h_old_u), _ptr(h_old_v), _ptr(h_new_u), _ptr(h_new_v), _ptr(u), _ptr(v)))

    def ALE_remap_velocities_from_h(self, CS, h_old, h_new, u, v):
        """ALE_remap_set_h_vel (old grid), ALE_remap_set_h_vel (new grid) and ALE_remap_velocities (MOM_ALE.F90:882, :1089) in one call."""
        check(self.lib, self.lib.mom6x_ALE_remap_velocities_from_h(self.ctx, C.byref(CS), _ptr(h_old), _ptr(h_new), _ptr(u), _ptr(v)))

    def ALE_regrid_zstar(self, CS, coordinateResolution, h, h_new, dzRegrid):
        """ALE_regrid (MOM_ALE.F90:518) for the z* coordinate; coordinateResolution: nk host values [Z]."""
        cr = np.ascontiguousarray(coordinateResolution, dtype=np.float64)
        assert cr.shape == (self.dims.nk,)
        check(self.lib, self.lib.mom6x_ALE_regrid_zstar(self.ctx, C.byref(CS), cr.ctypes.data_as(C.c_void_p), _ptr(h), _ptr(h_new),
                                                        _ptr(dzRegrid)))

    @staticmethod
    def _hostvec(a, n):
        if a is None:
            return None, None
        v = np.ascontiguousarray(a, dtype=np.float64)
        assert v.shape == (n,)
        return v, v.ctypes.data_as(C.c_void_p)

    def ALE_regrid_rho(self, CS, eos, target_density, h, T, S, h_new, dzRegrid):
        """regridding_main (MOM_regridding.F90:862) for REGRIDDING_RHO; target_density: nk+1 host values (interfaces).
        REGRIDDING_RHO expects ALE_convective_adjustment first (regridding_preadjust_reqs :966)."""
        td, ptd = self._hostvec(target_density, self.dims.nk + 1)
        check(self.lib, self.lib.mom6x_ALE_regrid_rho(self.ctx, C.byref(CS), C.byref(eos), ptd, _ptr(h), _ptr(T), _ptr(S), _ptr(h_new),
                                                      _ptr(dzRegrid)))

    def ALE_regrid_hycom1(self, CS, eos, coordinateResolution, target_density, max_interface_depths, max_layer_thickness, h, T, S,
                          h_new, dzRegrid):
        """regridding_main (MOM_regridding.F90:862) for REGRIDDING_HYCOM1 (build_grid_HyCOM1 :1638)."""
        nk = self.dims.nk
        cr, pcr = self._hostvec(coordinateResolution, nk)
        td, ptd = self._hostvec(target_density, nk + 1)
        mid, pmid = self._hostvec(max_interface_depths, nk + 1)
        mlt, pmlt = self._hostvec(max_layer_thickness, nk)
        check(self.lib, self.lib.mom6x_ALE_regrid_hycom1(self.ctx, C.byref(CS), C.byref(eos), pcr, ptd, pmid, pmlt, _ptr(h), _ptr(T), _ptr(S),
                                                         _ptr(h_new), _ptr(dzRegrid)))

    def ALE_convective_adjustment(self, eos, h, T, S):
        """convective_adjustment (MOM_regridding.F90:1905): h, T, S are reordered in place."""
        check(self.lib, self.lib.mom6x_ALE_convective_adjustment(self.ctx, C.byref(eos), _ptr(h), _ptr(T), _ptr(S)))

    def remapping_core_h(self, CS, h0, u0, h1, u1):
        """remapping_core_h (MOM_remapping.F90:234) for [ncol][n0] | [ncol][n1] device arrays."""
        ncol, n0 = h0.shape; n1 = h1.shape[1]
        check(self.lib, self.lib.mom6x_remapping_core_h(self.ctx, C.byref(CS), ncol, n0, _ptr(h0), _ptr(u0), n1, _ptr(h1), _ptr(u1)))

    # -- MOM_coms / MOM_checksums: the reproducing sums and checksums of the regression artefacts --------------
    def _rect(self, is_, ie, js, je):
        d = self.dims
        return (0 if is_ is None else is_, d.ni - 1 if ie is None else ie, 0 if js is None else js, d.nj - 1 if je is None else je)

    def reproducing_sum(self, array, is_=None, ie=None, js=None, je=None, unscale=1.0, layer_sums=False, only_on_PE=False,
                        want_err=False):
        """reproducing_sum (MOM_coms.F90:235 for a 2-D plane, :349 for nk planes) of a device array over the local
        index range (default: the h-point computational domain).  Returns dict(sum, EFP[, sums, EFP_lay][, err])."""
        r = self._rect(is_, ie, js, je)
        s = C.c_double(0.0); efp = (C.c_int64 * 6)(); err = C.c_int(0)
        perr = C.byref(err) if want_err else None
        out = {}
        if array.dim() == 2:
            check(self.lib, self.lib.mom6x_reproducing_sum_2d(self.ctx, _ptr(array), *r, C.c_double(unscale), int(only_on_PE),
                                                              C.byref(s), efp, perr))
        else:
            nk = array.shape[0]
            sums = (C.c_double * nk)() if layer_sums else None
            lay = (C.c_int64 * (6 * nk))() if layer_sums else None
            check(self.lib, self.lib.mom6x_reproducing_sum_3d(self.ctx, _ptr(array), nk, *r, C.c_double(unscale), int(only_on_PE),
                                                              C.byref(s), sums, efp, lay, perr))
            if layer_sums:
                out.update(sums=np.array(sums[:]), EFP_lay=np.array(lay[:], dtype=np.int64).reshape(nk, 6))
        out.update(sum=s.value, EFP=np.array(efp[:], dtype=np.int64))
        if want_err:
            out["err"] = err.value
        return out

    def chksum(self, array, stagger, haloshift=0, symmetric=False, omit_corners=False, scale=None):
        """chksum_{h,u,v,B}_{2d,3d} (MOM_checksums.F90); stagger in 'huvB'.  dict(mean, min, max, bc0, bc, kind)."""
        res = abi.ChksumResult()
        nk = 1 if array.dim() == 2 else array.shape[0]
        sc = None if scale is None else C.byref(C.c_double(scale))
        check(self.lib, self.lib.mom6x_chksum(self.ctx, _ptr(array), nk, array.dim(), "huvB".index(stagger), haloshift,
                                              int(symmetric), int(omit_corners), sc, C.byref(res)))
        return dict(mean=res.mean, min=res.amin, max=res.amax, bc0=res.bc0, bc=[res.bc[n] for n in range(res.nbc)], kind=res.bc_kind)

    def chksum_lines(self, array, stagger, mesg, **kw):
        """The two lines hchksum / uchksum / vchksum / Bchksum write (chk_sum_msg3 :2638, chk_sum_msg1/5/_NSEW/_W/_S)."""
        r = self.chksum(array, stagger, **kw)
        pt = {"h": "h-point:", "u": "u-point:", "v": "v-point:", "B": "B-point:"}[stagger]
        l1 = pt + " mean=" + _es25_16(r["mean"]) + "min=" + _es25_16(r["min"]) + "max=" + _es25_16(r["max"]) + mesg
        tags = {abi.CHK_NONE: [], abi.CHK_CORNERS: ["sw=", "se=", "nw=", "ne="], abi.CHK_NSEW: ["N=", "S=", "E=", "W="],
                abi.CHK_W: ["W="], abi.CHK_S: ["S="]}[r["kind"]]
        l2 = pt + " c=%10d " % r["bc0"] + "".join("%s%10d " % (t, b) for t, b in zip(tags, r["bc"])) + mesg
        return l1, l2

    def field_chksum(self, array, is_=None, ie=None, js=None, je=None, unscale=1.0):
        """The restart files' `checksum` attribute (MOM_restart.F90:1741; FMS mpp_chksum) as a signed 64-bit integer."""
        nk = 1 if array.dim() == 2 else array.shape[0]
        v = C.c_int64(0)
        check(self.lib, self.lib.mom6x_field_chksum(self.ctx, _ptr(array), nk, *self._rect(is_, ie, js, je), C.c_double(unscale),
                                                    C.byref(v)))
        return v.value

    # -- MOM_sum_output ----------------------------------------------------------------------
    def sum_output_init(self, params, g_prime):
        """MOM_sum_output_init (MOM_sum_output.F90:147) + depth_list_setup (:1161)."""
        gp = np.ascontiguousarray(g_prime, dtype=np.float64)
        assert gp.size >= self.dims.nk
        self.sum_output_params = params
        check(self.lib, self.lib.mom6x_sum_output_init(self.ctx, C.byref(params), gp.ctypes.data_as(C.c_void_p)))

    def depth_list(self):
        """The Depth_List of create_depth_list (:1203): depth, area, vol_below."""
        n = C.c_int(0)
        check(self.lib, self.lib.mom6x_depth_list(self.ctx, C.byref(n), None, None, None))
        out = [np.zeros(n.value) for _ in range(3)]
        check(self.lib, self.lib.mom6x_depth_list(self.ctx, C.byref(n), *[a.ctypes.data_as(C.c_void_p) for a in out]))
        return out

    def write_energy(self, u, v, h, T=None, S=None):
        """The sums of write_energy (:321): dict(mass_tot, KE_tot, PE_tot, max_CFL, mass_EFP, salt_EFP, heat_EFP, mass_lay, KE,
        PE, Z_0APE).  mom6_amd.sum_output.SumOutput turns them into the ocean.stats line."""
        nk = self.dims.nk
        res = abi.EnergySums()
        vec = dict(mass_lay=np.zeros(nk), KE=np.zeros(nk), PE=np.zeros(nk + 1), Z_0APE=np.zeros(nk + 1))
        check(self.lib, self.lib.mom6x_write_energy(self.ctx, _ptr(u), _ptr(v), _ptr(h), _ptr(T), _ptr(S), C.byref(res),
                                                    *[vec[n].ctypes.data_as(C.c_void_p) for n in ("mass_lay", "KE", "PE", "Z_0APE")]))
        out = dict(mass_tot=res.mass_tot, KE_tot=res.KE_tot, PE_tot=res.PE_tot, max_CFL=(res.max_CFL[0], res.max_CFL[1]),
                   mass_EFP=np.array(res.mass_EFP[:], dtype=np.int64), salt_EFP=np.array(res.salt_EFP[:], dtype=np.int64),
                   heat_EFP=np.array(res.heat_EFP[:], dtype=np.int64))
        out.update(vec)
        return out

    def vertvisc(self, u, v, taux, tauy, dt, taux_bot=None, tauy_bot=None):
        """vertvisc (MOM_vert_friction.F90:557)."""
        check(self.lib, self.lib.mom6x_vertvisc(self.ctx, _ptr(u), _ptr(v), _ptr(taux), _ptr(tauy), C.c_double(dt),
                                                _ptr(taux_bot), _ptr(tauy_bot)))

    def vertvisc_remnant(self, visc_rem_u, visc_rem_v, dt):
        """vertvisc_remnant (MOM_vert_friction.F90:1229)."""
        check(self.lib, self.lib.mom6x_vertvisc_remnant(self.ctx, _ptr(visc_rem_u), _ptr(visc_rem_v), C.c_double(dt)))

    # -- MOM_dynamics_split_RK2 --------------------------------------------------------------
    def initialize_dyn_split_RK2(self, params=None):
        """initialize_dyn_split_RK2 (MOM_dynamics_split_RK2.F90:1346): allocate the CS on the device."""
        self.rk2_params = params if params is not None else abi.rk2_params_default()
        check(self.lib, self.lib.mom6x_initialize_dyn_split_RK2(self.ctx, C.byref(self.rk2_params)))

    def dyn_split_RK2_new_run(self, u, v, h, uh, vh, dt):
        """The new-run fills of initialize_dyn_split_RK2 (:1577-1650)."""
        check(self.lib, self.lib.mom6x_dyn_split_RK2_new_run(self.ctx, _ptr(u), _ptr(v), _ptr(h), _ptr(uh), _ptr(vh), C.c_double(dt)))

    def dyn_split_RK2_restart_fills(self, u, v, h, uh, vh, dt, have):
        """initialize_dyn_split_RK2 (:1577-1668) for a restarted run: `have` = the abi.RK2_HAVE_* bits of the restart variables the
        caller has uploaded into their mom6x_rk2_field arrays; the others are formed as the reference forms them."""
        check(self.lib, self.lib.mom6x_dyn_split_RK2_restart_fills(self.ctx, _ptr(u), _ptr(v), _ptr(h), _ptr(uh), _ptr(vh), C.c_double(dt),
                                                                   C.c_int(have)))

    def remap_dyn_split_RK2_aux_vars(self, CS, h_old_u, h_old_v, h_new_u, h_new_v):
        """remap_dyn_split_RK2_aux_vars (MOM_dynamics_split_RK2.F90:1302); CS = ALE_CSp%vel_remapCS."""
        check(self.lib, self.lib.mom6x_remap_dyn_split_RK2_aux_vars(self.ctx, C.byref(CS), _ptr(h_old_u), _ptr(h_old_v), _ptr(h_new_u),
                                                                    _ptr(h_new_v)))

    def rk2_set_CAu_pred_stored(self, stored=True):
        """A restarted run whose file held CAu, CAv (query_initialized, MOM_dynamics_split_RK2.F90:1616) skips the
        CorAdCalc of :552-557 at its first step."""
        check(self.lib, self.lib.mom6x_rk2_set_CAu_pred_stored(self.ctx, int(bool(stored))))

    def rk2_field(self, name):
        """Zero-copy torch view of a MOM_dyn_split_RK2_CS array (restart / diagnostics access)."""
        which = abi.RK2_FIELDS.index(name)
        ptr = self.lib.mom6x_rk2_field(self.ctx, C.c_int(which))
        shape = self.dims.shape2() if name in abi.RK2_FIELDS_2D else self.dims.shape3()
        return _view(ptr, shape, self.device)

    def step_MOM_dyn_split_RK2(self, u, v, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy, dt, calc_dtbt=False,
                               vertvisc_coef=None, horizontal_viscosity=None):
        """step_MOM_dyn_split_RK2 (MOM_dynamics_split_RK2.F90:294).

        vertvisc_coef(stage, u_ptr, v_ptr, h_ptr, dt) / horizontal_viscosity(u_av, v_av, h_av, uh, vh, diffu, diffv)
        are optional host callbacks (device pointers as ints) for the un-ported callees."""
        hooks = None
        if vertvisc_coef is not None or horizontal_viscosity is not None:
            hk = abi.RK2Hooks()
            if vertvisc_coef is not None:
                hk.vertvisc_coef = abi.VERTVISC_COEF_HOOK(lambda user, stage, pu, pv, ph, dtt: int(vertvisc_coef(stage, pu, pv, ph, dtt) or 0))
            if horizontal_viscosity is not None:
                hk.horizontal_viscosity = abi.HOR_VISC_HOOK(lambda user, a, b2, c2, d2, e2, f2, g2: int(horizontal_viscosity(a, b2, c2, d2, e2, f2, g2) or 0))
            self._hooks = hk
            hooks = C.byref(hk)
        check(self.lib, self.lib.mom6x_step_dyn_split_RK2(
            self.ctx, _ptr(u), _ptr(v), _ptr(h), _ptr(uh), _ptr(vh), _ptr(uhtr), _ptr(vhtr), _ptr(eta_av), _ptr(taux),
            _ptr(tauy), C.c_double(dt), C.c_int(int(calc_dtbt)), hooks))

    # -- MOM_tracer_advect / tridiagonal solvers ----------------------------------------------
    def tracer_advect_init(self, dt_dyn, scheme=0, useHuynhStencilBug=False):
        """tracer_advect_init (MOM_tracer_advect.F90:1155); scheme: 0 PLM (default), 1 PPM:H3, 2 PPM."""
        check(self.lib, self.lib.mom6x_tracer_advect_init(self.ctx, C.c_double(dt_dyn), C.c_int(scheme), C.c_int(int(useHuynhStencilBug))))

    def advect_tracer(self, h_end, uhtr, vhtr, dt, tracers, schemes=None, x_first_in=-1, max_iter_in=0, uhr_out=None, vhr_out=None):
        """advect_tracer (MOM_tracer_advect.F90:53); `tracers` is the registry as a list of 3-D fields."""
        n = len(tracers)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tracers])
        sch = (C.c_int * n)(*(schemes if schemes is not None else [-1] * n))
        iters = C.c_int(0)
        check(self.lib, self.lib.mom6x_advect_tracer(self.ctx, _ptr(h_end), _ptr(uhtr), _ptr(vhtr), C.c_double(dt), ptrs, sch, n,
                                                     C.c_int(x_first_in), C.c_int(max_iter_in), _ptr(uhr_out), _ptr(vhr_out), C.byref(iters)))
        return iters.value

    def triDiagTS(self, hold, ea, eb, T, S=None, rng=None):
        """triDiagTS (MOM_diabatic_aux.F90:394)."""
        is_, ie, js, je = rng if rng else (0, self.dims.ni - 1, 0, self.dims.nj - 1)
        check(self.lib, self.lib.mom6x_triDiagTS(self.ctx, is_, ie, js, je, _ptr(hold), _ptr(ea), _ptr(eb), _ptr(T), _ptr(S)))

    def triDiagTS_Eulerian(self, hold, ent, T, S=None, rng=None):
        """triDiagTS_Eulerian (MOM_diabatic_aux.F90:444)."""
        is_, ie, js, je = rng if rng else (0, self.dims.ni - 1, 0, self.dims.nj - 1)
        check(self.lib, self.lib.mom6x_triDiagTS_Eulerian(self.ctx, is_, ie, js, je, _ptr(hold), _ptr(ent), _ptr(T), _ptr(S)))

    def tracer_vertdiff(self, h_old, ea, eb, dt, tr, sfc_flux=None, btm_flux=None, convert_flux=True):
        """tracer_vertdiff (MOM_tracer_diabatic.F90:25), no-sinking branch."""
        check(self.lib, self.lib.mom6x_tracer_vertdiff(self.ctx, _ptr(h_old), _ptr(ea), _ptr(eb), C.c_double(dt), _ptr(tr),
                                                       _ptr(sfc_flux), _ptr(btm_flux), C.c_int(int(convert_flux))))

    def tracer_vertdiff_sink(self, h_old, ea, eb, dt, tr, sink_rate, sfc_flux=None, btm_flux=None, btm_reservoir=None, convert_flux=True,
                             eulerian=False):
        """tracer_vertdiff / tracer_vertdiff_Eulerian (ea = ent, eb ignored) with sink_rate, optionally btm_reservoir
        (MOM_tracer_diabatic.F90:123-179 / :315-380)."""
        if eulerian:
            check(self.lib, self.lib.mom6x_tracer_vertdiff_Eulerian_sink(self.ctx, _ptr(h_old), _ptr(ea), C.c_double(dt), _ptr(tr), _ptr(sfc_flux),
                                                                         _ptr(btm_flux), _ptr(btm_reservoir), C.c_double(sink_rate),
                                                                         C.c_int(int(convert_flux))))
        else:
            check(self.lib, self.lib.mom6x_tracer_vertdiff_sink(self.ctx, _ptr(h_old), _ptr(ea), _ptr(eb), C.c_double(dt), _ptr(tr), _ptr(sfc_flux),
                                                                _ptr(btm_flux), _ptr(btm_reservoir), C.c_double(sink_rate),
                                                                C.c_int(int(convert_flux))))

    def tracer_vertdiff_Eulerian(self, h_old, ent, dt, tr, sfc_flux=None, btm_flux=None, convert_flux=True):
        """tracer_vertdiff_Eulerian (MOM_tracer_diabatic.F90:224), no-sinking branch."""
        check(self.lib, self.lib.mom6x_tracer_vertdiff_Eulerian(self.ctx, _ptr(h_old), _ptr(ent), C.c_double(dt), _ptr(tr),
                                                                _ptr(sfc_flux), _ptr(btm_flux), C.c_int(int(convert_flux))))


def _view(ptr, shape, device):
    """torch tensor aliasing device memory owned by the C library."""
    n = int(np.prod(shape))

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device).view(shape)


def prof_enable(dyc, on=True):
    check(dyc.lib, dyc.lib.mom6x_prof_enable(dyc.ctx, C.c_int(int(on))))


def prof_reset(dyc):
    check(dyc.lib, dyc.lib.mom6x_prof_reset(dyc.ctx))


def prof_report(dyc):
    """dict kernel-name -> (launch count, total ms) measured with HIP events on the compute stream."""
    n = dyc.lib.mom6x_prof_report(dyc.ctx, None, C.c_int(0))
    buf = C.create_string_buffer(max(n, 1) + 16)
    dyc.lib.mom6x_prof_report(dyc.ctx, buf, C.c_int(len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split("\t")
        out[name] = (int(cnt), float(ms))
    return out


def _es25_16(x):
    """Fortran ES25.16 followed by the 1X of chk_sum_msg3's format."""
    m, e = ("%.16E" % x).split("E")
    return ("%sE%+03d" % (m, int(e))).rjust(25) + " "
