"""ctypes mirror of include/mom6x.h and loader of the HIP C-ABI library.

The product path has NO CPU fallback: `load_library()` raises if
mom6_amd/lib/libmom6x.so is missing, and every compute entry point needs a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmom6x.so")

c_double_p = C.POINTER(C.c_double)


class Dims(C.Structure):
    """mom6x_dims (include/mom6x.h); MOM_hor_index.F90:14-44 extents."""
    _fields_ = [
        ("ni", C.c_int), ("nj", C.c_int), ("nk", C.c_int), ("halo", C.c_int),
        ("ioff", C.c_int), ("joff", C.c_int), ("pitch", C.c_int), ("slab", C.c_int),
        ("i_glob0", C.c_int), ("j_glob0", C.c_int), ("ni_glob", C.c_int), ("nj_glob", C.c_int),
        ("reentrant_x", C.c_int), ("reentrant_y", C.c_int),
    ]

    @property
    def nrows(self):
        return self.nj + 2 * self.halo + 1

    def shape2(self):
        return (self.nrows, self.pitch)

    def shape3(self, nk=None):
        return (self.nk if nk is None else nk, self.nrows, self.pitch)

    def sl(self, i0, i1, j0, j1):
        """numpy slices [j, i] for local inclusive ranges i0..i1, j0..j1."""
        return (slice(j0 + self.joff, j1 + self.joff + 1), slice(i0 + self.ioff, i1 + self.ioff + 1))


def dims_init(ni, nj, nk, halo=4, ni_glob=None, nj_glob=None, i_glob0=0, j_glob0=0,
              reentrant_x=False, reentrant_y=False):
    """Python twin of mom6x_dims_init (tested equal to the C one)."""
    if halo + 1 > 16:
        raise ValueError("halo too wide")
    d = Dims()
    d.ni, d.nj, d.nk, d.halo = ni, nj, nk, halo
    d.ioff = 16
    d.joff = halo + 1
    d.pitch = ((d.ioff + ni + halo + 15) // 16) * 16
    d.slab = d.pitch * (nj + 2 * halo + 1)
    d.ni_glob = ni if ni_glob is None else ni_glob
    d.nj_glob = nj if nj_glob is None else nj_glob
    d.i_glob0, d.j_glob0 = i_glob0, j_glob0
    d.reentrant_x, d.reentrant_y = int(reentrant_x), int(reentrant_y)
    return d


class VGrid(C.Structure):
    """mom6x_vgrid; MOM_verticalGrid.F90:25-90."""
    _fields_ = [
        ("g_Earth", C.c_double), ("Rho0", C.c_double), ("Angstrom_H", C.c_double),
        ("H_subroundoff", C.c_double), ("dZ_subroundoff", C.c_double),
        ("H_to_Z", C.c_double), ("Z_to_H", C.c_double), ("H_to_RZ", C.c_double),
        ("RZ_to_H", C.c_double), ("Boussinesq", C.c_int),
    ]


def vgrid_default(g_Earth=9.80, Rho0=1035.0, Angstrom=1e-10):
    """Defaults of verticalGridInit (MOM_verticalGrid.F90:100-230) with all US scalings = 1."""
    gv = VGrid()
    gv.g_Earth, gv.Rho0, gv.Angstrom_H = g_Earth, Rho0, Angstrom
    # verticalGrid.F90:214-218: H_subroundoff = 1e-20 * max(Angstrom_H, m_to_H*1e-17)
    gv.H_subroundoff = 1e-20 * max(Angstrom, 1e-17)
    gv.dZ_subroundoff = 1e-20 * max(Angstrom, 1e-17)
    gv.H_to_Z = gv.Z_to_H = 1.0
    gv.H_to_RZ = Rho0
    gv.RZ_to_H = 1.0 / Rho0
    gv.Boussinesq = 1
    return gv


class ContinuityParams(C.Structure):
    """mom6x_continuity_params; continuity_PPM_CS (MOM_continuity_PPM.F90:35-68)."""
    _fields_ = [
        ("upwind_1st", C.c_int), ("monotonic", C.c_int), ("simple_2nd", C.c_int),
        ("tol_eta", C.c_double), ("tol_vel", C.c_double), ("CFL_limit_adjust", C.c_double),
        ("aggress_adjust", C.c_int), ("vol_CFL", C.c_int), ("better_iter", C.c_int),
        ("use_visc_rem_max", C.c_int), ("marginal_faces", C.c_int), ("sum_order", C.c_int),
    ]


def continuity_params_default(nk, Angstrom=1e-10):
    """Defaults of continuity_PPM_init (MOM_continuity_PPM.F90:2674-2754)."""
    p = ContinuityParams()
    p.upwind_1st = p.monotonic = p.simple_2nd = 0
    p.tol_eta = 0.5 * nk * Angstrom
    p.tol_vel = 3.0e8
    p.CFL_limit_adjust = 0.5
    p.aggress_adjust = p.vol_CFL = 0
    p.better_iter = p.use_visc_rem_max = p.marginal_faces = 1
    p.sum_order = default_sum_order(nk)
    return p


SUM_REFERENCE, SUM_TREE16, SUM_TREE16_FMA = 0, 1, 2


def default_sum_order(nk):
    """Order of the column sums of the mass-flux kernels (mom6x_continuity_params.sum_order): the 16-lane tree of the
    wave-owned kernel with fused multiply-adds at fixed sites (MOM6X_SUM_TREE16_FMA, the default since round 6: 4 % faster, the
    same distance from the reference's arithmetic as the un-fused tree) unless MOM6X_SUMS=tree asks for the un-fused tree,
    MOM6X_SUMS=exact for the reference's sequential order (bit-identical to the Fortran loop nest, slower), or the column is
    deeper than the wave-owned kernel carries."""
    want = os.environ.get("MOM6X_SUMS", "").lower()
    if want in ("exact", "reference", "0") or nk > 128:
        return SUM_REFERENCE
    if want in ("tree", "tree16", "1"):
        return SUM_TREE16
    return SUM_TREE16_FMA


class BTCont(C.Structure):
    """mom6x_BT_cont; BT_cont_type (MOM_variables.F90:315-350)."""
    _names = ["FA_u_EE", "FA_u_E0", "FA_u_W0", "FA_u_WW", "uBT_WW", "uBT_EE",
              "FA_v_NN", "FA_v_N0", "FA_v_S0", "FA_v_SS", "vBT_SS", "vBT_NN", "h_u", "h_v"]
    _fields_ = [(n, C.c_void_p) for n in _names]


BT_THICK_FROM_BT_CONT, BT_THICK_HYBRID, BT_THICK_HARMONIC, BT_THICK_ARITHMETIC = 0, 1, 2, 3   # MOM6X_BT_THICK_*


class BarotropicParams(C.Structure):
    """mom6x_barotropic_params; barotropic_CS (MOM_barotropic.F90:108-330)."""
    _fields_ = [
        ("bebt", C.c_double), ("dtbt", C.c_double), ("dt_bt_filter", C.c_double),
        ("BT_project_velocity", C.c_int), ("Sadourny", C.c_int), ("strong_drag", C.c_int),
        ("wt_uv_bug", C.c_int), ("use_old_coriolis_bracket_bug", C.c_int),
        ("visc_rem_u_uh0", C.c_int), ("clip_velocity", C.c_int),
        ("CFL_trunc", C.c_double), ("vel_underflow", C.c_double), ("G_extra", C.c_double),
        ("BT_Coriolis_scale", C.c_double), ("maxCFL_BT_cont", C.c_double),
        ("bound_BT_corr", C.c_int), ("BT_cont_bounds", C.c_int),
        ("dtbt_fraction", C.c_double), ("Z_ref", C.c_double),
        ("use_wide_halos", C.c_int), ("BTHALO", C.c_int), ("min_stencil", C.c_int),
        ("nonlinear_continuity", C.c_int), ("nonlin_cont_update_period", C.c_int),
        ("bt_thick_scheme", C.c_int), ("maxvel", C.c_double),
    ]


def barotropic_params_default(dtbt):
    """Defaults read in barotropic_init (MOM_barotropic.F90:5403-5713)."""
    p = BarotropicParams()
    p.bebt, p.dtbt, p.dt_bt_filter = 0.1, dtbt, -0.25
    p.use_wide_halos, p.BTHALO, p.min_stencil = 1, 0, 0
    p.nonlinear_continuity, p.nonlin_cont_update_period = 0, 1
    p.bt_thick_scheme, p.maxvel = BT_THICK_FROM_BT_CONT, 3.0e8
    p.BT_project_velocity = 0
    p.Sadourny = 1
    p.strong_drag = 0
    p.wt_uv_bug = 1
    p.use_old_coriolis_bracket_bug = 0
    p.visc_rem_u_uh0 = 0
    p.clip_velocity = 0
    p.CFL_trunc, p.vel_underflow, p.G_extra = 0.5, 0.0, 0.0
    p.BT_Coriolis_scale, p.maxCFL_BT_cont = 1.0, 0.25
    p.bound_BT_corr, p.BT_cont_bounds = 0, 1
    p.dtbt_fraction, p.Z_ref = 0.98, 0.0
    return p


SADOURNY75_ENERGY, ARAKAWA_HSU90, ROBUST_ENSTRO, SADOURNY75_ENSTRO, ARAKAWA_LAMB81, AL_BLEND = 1, 2, 3, 4, 5, 6
KE_ARAKAWA, KE_SIMPLE_GUDONOV, KE_GUDONOV = 10, 11, 12
PV_ADV_CENTERED, PV_ADV_UPWIND1 = 21, 22


class CoriolisParams(C.Structure):
    """mom6x_coriolis_params; CoriolisAdv_CS (MOM_CoriolisAdv.F90:29-100)."""
    _fields_ = [("Coriolis_Scheme", C.c_int), ("KE_Scheme", C.c_int), ("bound_Coriolis", C.c_int),
                ("no_slip", C.c_int), ("Coriolis_En_Dis", C.c_int), ("PV_Adv_Scheme", C.c_int),
                ("F_eff_max_blend", C.c_double), ("wt_lin_blend", C.c_double)]


def coriolis_params_default():
    """Defaults of CoriolisAdv_init (MOM_CoriolisAdv.F90:1054-1320)."""
    p = CoriolisParams()
    p.Coriolis_Scheme, p.KE_Scheme = SADOURNY75_ENERGY, KE_ARAKAWA
    p.bound_Coriolis = p.no_slip = p.Coriolis_En_Dis = 0
    p.PV_Adv_Scheme, p.F_eff_max_blend, p.wt_lin_blend = PV_ADV_CENTERED, 4.0, 0.125   # :1126-1139, :1178-1192
    return p


class PGFParams(C.Structure):
    """mom6x_pgf_params; PressureForce_FV_CS (MOM_PressureForce_FV.F90:40-110)."""
    _fields_ = [("rho_ref", C.c_double), ("rho_ref_bug", C.c_int), ("Z_ref", C.c_double)]


class VertviscParams(C.Structure):
    """mom6x_vertvisc_params; vertvisc_CS (MOM_vert_friction.F90:39-180)."""
    _fields_ = [("Kv", C.c_double), ("Kvml_invZ2", C.c_double), ("Hmix", C.c_double), ("Hbbl", C.c_double),
                ("harm_BL_val", C.c_double), ("Kv_extra_bbl", C.c_double), ("harmonic_visc", C.c_int),
                ("bottomdraglaw", C.c_int), ("answer_date", C.c_int)]


def vertvisc_params_default(Kv=1.0e-4, Hmix=20.0, Hbbl=10.0):
    """vertvisc_init :3135 defaults; KV, HMIX_FIXED and HBBL have none in MOM6 (fail_if_missing)."""
    p = VertviscParams()
    p.Kv = Kv; p.Kvml_invZ2 = 0.0; p.Hmix = Hmix; p.Hbbl = Hbbl; p.harm_BL_val = 0.0; p.Kv_extra_bbl = 0.0
    p.harmonic_visc = 0; p.bottomdraglaw = 1; p.answer_date = 99991231
    return p


class HorViscParams(C.Structure):
    """mom6x_hor_visc_params; hor_visc_CS (MOM_hor_visc.F90:36-259)."""
    _fields_ = [("Laplacian", C.c_int), ("biharmonic", C.c_int), ("Kh", C.c_double), ("Kh_bg_min", C.c_double),
                ("Kh_vel_scale", C.c_double), ("Smagorinsky_Kh", C.c_int), ("Smag_Lap_const", C.c_double), ("bound_Kh", C.c_int),
                ("better_bound_Kh", C.c_int), ("add_LES_viscosity", C.c_int), ("Ah", C.c_double), ("Ah_vel_scale", C.c_double),
                ("Ah_time_scale", C.c_double), ("Smagorinsky_Ah", C.c_int), ("Smag_bi_const", C.c_double), ("bound_Ah", C.c_int),
                ("better_bound_Ah", C.c_int), ("bound_Coriolis", C.c_int), ("bound_Cor_vel", C.c_double), ("use_land_mask", C.c_int),
                ("bound_coef", C.c_double), ("no_slip", C.c_int), ("backscatter_underbound", C.c_int), ("dt", C.c_double),
                ("Leith_Kh", C.c_int), ("Leith_Lap_const", C.c_double), ("Leith_Ah", C.c_int), ("Leith_bi_const", C.c_double),
                ("modified_Leith", C.c_int), ("use_beta_in_Leith", C.c_int)]


def hor_visc_params_default(dt, Laplacian=False, biharmonic=True):
    """hor_visc_init :2403-2720 defaults (LAPLACIAN F, BIHARMONIC T, bounds on, everything else off / zero)."""
    p = HorViscParams()
    p.Laplacian = int(Laplacian); p.biharmonic = int(biharmonic)
    p.Kh = 0.0; p.Kh_bg_min = 0.0; p.Kh_vel_scale = 0.0; p.Smagorinsky_Kh = 0; p.Smag_Lap_const = 0.0
    p.bound_Kh = 1; p.better_bound_Kh = 1; p.add_LES_viscosity = 0
    p.Ah = 0.0; p.Ah_vel_scale = 0.0; p.Ah_time_scale = 0.0; p.Smagorinsky_Ah = 0; p.Smag_bi_const = 0.0
    p.bound_Ah = 1; p.better_bound_Ah = 1; p.bound_Coriolis = 0; p.bound_Cor_vel = 3.0e8
    p.use_land_mask = 1; p.bound_coef = 0.8; p.no_slip = 0; p.backscatter_underbound = 1; p.dt = dt
    return p


REMAP_PCM, REMAP_PLM, REMAP_PPM_H4, REMAP_PPM_IH4 = 0, 2, 4, 5   # enum mom6x_remap_scheme


class RemappingParams(C.Structure):
    """mom6x_remapping_params; remapping_CS (MOM_remapping.F90:47-84)."""
    _fields_ = [("scheme", C.c_int), ("boundary_extrapolation", C.c_int), ("force_bounds_in_subcell", C.c_int),
                ("force_bounds_in_target", C.c_int), ("om4_remap_via_sub_cells", C.c_int), ("answer_date", C.c_int),
                ("h_neglect", C.c_double), ("h_neglect_edge", C.c_double)]


def remapping_params_default(scheme=REMAP_PLM, h_neglect=1.0e-30, **kw):
    """initialize_remapping :1654 with the type's defaults (:47-84): boundary extrapolation on, target values bounded,
    sub-cell values not, the non-OM4 sub-cell integrator; h_neglect as in remapping_unit_tests (1e-30 H)."""
    p = RemappingParams()
    p.scheme = scheme; p.boundary_extrapolation = 1; p.force_bounds_in_subcell = 0; p.force_bounds_in_target = 1
    p.om4_remap_via_sub_cells = 0; p.answer_date = 99991231; p.h_neglect = h_neglect; p.h_neglect_edge = h_neglect
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class ChksumResult(C.Structure):
    """mom6x_chksum_result: the numbers of the two message lines of chksum_{h,u,v,B}_{2d,3d} (MOM_checksums.F90)."""
    _fields_ = [("mean", C.c_double), ("amin", C.c_double), ("amax", C.c_double), ("bc0", C.c_int), ("bc", C.c_int * 4),
                ("nbc", C.c_int), ("bc_kind", C.c_int)]


CHK_NONE, CHK_CORNERS, CHK_NSEW, CHK_W, CHK_S = range(5)    # enum mom6x_chksum_kind


class SumOutputParams(C.Structure):
    """mom6x_sum_output_params: the members of Sum_output_CS (MOM_sum_output.F90:60-140) the sums of write_energy read."""
    _fields_ = [("do_APE_calc", C.c_int), ("use_temperature", C.c_int), ("dt_in_T", C.c_double), ("D_list_min_inc", C.c_double),
                ("Z_ref", C.c_double), ("C_p", C.c_double)]


def sum_output_params_default(dt, **kw):
    """CALCULATE_APE = True, ENABLE_THERMODYNAMICS off unless asked, DEPTH_LIST_MIN_INC = 1e-10 m, C_P = 3991.86795711963."""
    p = SumOutputParams()
    p.do_APE_calc = 1; p.use_temperature = 0; p.dt_in_T = dt; p.D_list_min_inc = 1.0e-10; p.Z_ref = 0.0; p.C_p = 3991.86795711963
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class EnergySums(C.Structure):
    """mom6x_energy_sums."""
    _fields_ = [("mass_tot", C.c_double), ("KE_tot", C.c_double), ("PE_tot", C.c_double), ("max_CFL", C.c_double * 2),
                ("mass_EFP", C.c_int64 * 6), ("salt_EFP", C.c_int64 * 6), ("heat_EFP", C.c_int64 * 6)]


class RegridZstarParams(C.Structure):
    """mom6x_regrid_zstar_params; the members of regridding_CS (MOM_regridding.F90:40-140) the z* branch reads."""
    _fields_ = [("min_thickness", C.c_double), ("old_grid_weight", C.c_double), ("depth_of_time_filter_shallow", C.c_double),
                ("depth_of_time_filter_deep", C.c_double), ("Z_ref", C.c_double)]


def regrid_zstar_params_default(**kw):
    """MIN_THICKNESS = 0.001 m, no time filtering (REGRID_TIME_SCALE = 0), Z_ref = 0."""
    p = RegridZstarParams()
    p.min_thickness = 1.0e-3; p.old_grid_weight = 0.0; p.depth_of_time_filter_shallow = 0.0; p.depth_of_time_filter_deep = 0.0
    p.Z_ref = 0.0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


INTERP_P1M_H2, INTERP_PLM, INTERP_PPM_H4 = 0, 3, 5   # enum mom6x_interp_scheme
INTERP_P1M_H4 = 1                                      # (restated in the oracle only)


class RegridRhoParams(C.Structure):
    """mom6x_regrid_rho_params: what REGRIDDING_RHO / REGRIDDING_HYCOM1 read of regridding_CS on top of the z* members."""
    _fields_ = [("f", RegridZstarParams), ("interp_scheme", C.c_int), ("boundary_extrapolation", C.c_int), ("ref_pressure", C.c_double),
                ("compressibility_fraction", C.c_double), ("integrate_downward_for_e", C.c_int)]


def regrid_rho_params_default(**kw):
    """INTERPOLATION_SCHEME = P1M_H2, BOUNDARY_EXTRAPOLATION = F, P_REF = 2e7 Pa, no compressibility, heights integrated
    downward from the surface (MOM_regridding.F90:90, :106, :118, :210-240)."""
    p = RegridRhoParams()
    p.f = regrid_zstar_params_default()
    p.interp_scheme = INTERP_P1M_H2; p.boundary_extrapolation = 0; p.ref_pressure = 2.0e7; p.compressibility_fraction = 0.0
    p.integrate_downward_for_e = 1
    for k, v in kw.items():
        if hasattr(p.f, k):
            setattr(p.f, k, v)
        else:
            setattr(p, k, v)
    return p


LINEAR, WRIGHT, WRIGHT_FULL, WRIGHT_REDUCED, UNESCO, ROQUET_RHO, JACKETT06, ROQUET_SPV = 1, 2, 3, 4, 5, 6, 7, 8   # enum mom6x_eos_form


class EOSParams(C.Structure):
    """mom6x_eos_params: tv%eqn_of_state + the EOS-only switches of PressureForce_FV_CS."""
    _fields_ = [("form", C.c_int), ("Rho_T0_S0", C.c_double), ("dRho_dT", C.c_double), ("dRho_dS", C.c_double),
                ("dRho_dp", C.c_double), ("MassWghtInterp", C.c_int), ("use_SSH_in_Z0p", C.c_int),
                ("Recon_Scheme", C.c_int), ("boundary_extrap", C.c_int), ("MassWghtInterpVanOnly", C.c_int), ("h_nonvanished", C.c_double),
                ("EOS_quadrature", C.c_int)]


def eos_params_default(form=WRIGHT):
    """EQN_OF_STATE (default WRIGHT), RHO_T0_S0 = 1000, DRHO_DT = -0.2, DRHO_DS = 0.8 (MOM_EOS.F90:1562-1600)."""
    p = EOSParams()
    p.form = form
    p.Rho_T0_S0 = 1000.0; p.dRho_dT = -0.2; p.dRho_dS = 0.8; p.dRho_dp = 0.0
    p.MassWghtInterp = 0; p.use_SSH_in_Z0p = 0
    p.Recon_Scheme = 0; p.boundary_extrap = 1; p.MassWghtInterpVanOnly = 0; p.h_nonvanished = 1.0e-6; p.EOS_quadrature = 0   # (Recon_Scheme: 1 with USE_REGRIDDING)
    return p


def pgf_params_default(Rho0=1035.0):
    p = PGFParams()
    p.rho_ref, p.rho_ref_bug, p.Z_ref = Rho0, 1, 0.0
    return p


def layer_densities(nk, Rho0=1035.0, g_Earth=9.80, drho=2.0):
    """GV%Rlay and GV%g_prime for a simple linear layer-density profile (COORD_CONFIG="linear",
    MOM_coord_initialization.F90:126-170): Rlay(k) = Rlay(1) + (k-1)*drho/(nk-1) style spacing and
    g_prime(k) = g*(Rlay(k)-Rlay(k-1))/Rho0, g_prime(1) = g (free surface)."""
    import numpy as np
    Rlay = Rho0 - 0.5 * drho + drho * (np.arange(nk) / max(nk - 1, 1))
    g_prime = np.empty(nk)
    g_prime[0] = g_Earth
    g_prime[1:] = (g_Earth / Rho0) * (Rlay[1:] - Rlay[:-1])
    return np.ascontiguousarray(Rlay), np.ascontiguousarray(g_prime)


class RK2Params(C.Structure):
    """mom6x_rk2_params; MOM_dyn_split_RK2_CS (MOM_dynamics_split_RK2.F90:85-273)."""
    _fields_ = [("be", C.c_double), ("begw", C.c_double), ("split_bottom_stress", C.c_int),
                ("BT_use_layer_fluxes", C.c_int), ("store_CAu", C.c_int), ("visc_rem_dt_bug", C.c_int), ("remap_aux", C.c_int),
                ("no_BT_cont", C.c_int)]


def rk2_params_default():
    """Defaults read in initialize_dyn_split_RK2 (MOM_dynamics_split_RK2.F90:1427-1495)."""
    p = RK2Params()
    p.be, p.begw = 0.6, 0.0
    p.split_bottom_stress = 0
    p.BT_use_layer_fluxes = p.store_CAu = p.visc_rem_dt_bug = 1
    p.remap_aux = 0
    p.no_BT_cont = 0                                                   # USE_BT_CONT_TYPE = True
    return p


VERTVISC_COEF_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double)
HOR_VISC_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p)


class RK2Hooks(C.Structure):
    """mom6x_rk2_hooks."""
    _fields_ = [("user", C.c_void_p), ("vertvisc_coef", VERTVISC_COEF_HOOK), ("horizontal_viscosity", HOR_VISC_HOOK)]


RK2_FIELDS = ["CAu", "CAv", "CAu_pred", "CAv_pred", "PFu", "PFv", "diffu", "diffv", "visc_rem_u", "visc_rem_v",
              "u_accel_bt", "v_accel_bt", "u_av", "v_av", "h_av", "pbce", "eta", "eta_PF", "uhbt", "vhbt",
              "taux_bot", "tauy_bot", "BT_h_u", "BT_h_v"]
# mom6x_dyn_split_RK2_restart_fills: the restart variables the caller has uploaded (include/mom6x.h MOM6X_RK2_HAVE_*)
RK2_HAVE_ETA, RK2_HAVE_DIFFU, RK2_HAVE_U2, RK2_HAVE_CAU, RK2_HAVE_UH, RK2_HAVE_H2 = 1, 2, 4, 8, 16, 32
RK2_FIELDS_2D = {"eta", "eta_PF", "uhbt", "vhbt", "taux_bot", "tauy_bot"}


# Metric plane indices: enum mom6x_metric
METRICS = [
    "mask2dT", "mask2dCu", "mask2dCv", "mask2dBu",
    "dxT", "dyT", "IdxT", "IdyT",
    "dxCu", "dyCu", "IdxCu", "IdyCu",
    "dxCv", "dyCv", "IdxCv", "IdyCv",
    "dxBu", "dyBu", "IdxBu", "IdyBu",
    "areaT", "IareaT", "areaBu", "IareaBu",
    "areaCu", "areaCv", "IareaCu", "IareaCv",
    "dy_Cu", "dx_Cv", "bathyT", "CoriolisBu", "Coriolis2Bu",
]
G = {n: i for i, n in enumerate(METRICS)}
G_COUNT = len(METRICS)

MOM6X_OK = 0

_lib = None


ABI_VERSION = 6   # include/mom6x.h MOM6X_ABI_VERSION


def load_library(path=None):
    """Load the HIP C-ABI shared library.  Raises (never falls back) if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"mom6_amd: HIP extension {p} is missing; run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    if lib.mom6x_abi_version() != ABI_VERSION:
        raise RuntimeError(f"mom6_amd: {p} has ABI version {lib.mom6x_abi_version()}, this host mirror was written for "
                           f"{ABI_VERSION} (include/mom6x.h MOM6X_ABI_VERSION); rebuild the library")
    lib.mom6x_last_error.restype = C.c_char_p
    lib.mom6x_ctx_stream.restype = C.c_void_p
    lib.mom6x_ctx_dims.restype = C.POINTER(Dims)
    lib.mom6x_ctx_metrics_dev.restype = C.c_void_p
    for name in ("mom6x_barotropic_field", "mom6x_rk2_field"):
        if hasattr(lib, name):
            getattr(lib, name).restype = C.c_void_p
    if path is None:
        _lib = lib
    return lib


class Mom6xError(RuntimeError):
    pass


def check(lib, rc):
    if rc != MOM6X_OK:
        msg = lib.mom6x_last_error()
        raise Mom6xError(f"mom6x error {rc}: {msg.decode() if msg else ''}")
