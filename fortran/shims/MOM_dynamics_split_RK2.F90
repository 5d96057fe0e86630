!> Drop-in replacement of src/core/MOM_dynamics_split_RK2.F90: the SAME module name, public names and argument lists
!! (step_MOM_dyn_split_RK2 :294-296, register_restarts_dyn_split_RK2 :1210, remap_dyn_split_RK2_aux_vars :1302,
!! init_dyn_split_RK2_diabatic :1335, initialize_dyn_split_RK2 :1346-1350, end_dyn_split_RK2 :1885), served by the
!! MI355X library through ISO_C_BINDING (fortran/mom6x_c_api.F90) and the dependency-free host layer
!! fortran/mom6x_host.F90, which is compiled and run on the GPU by fortran/drive_double_gyre.F90.
!!
!! THIS FILE COMPILES ONLY INSIDE A MOM6 BUILD TREE (it uses MOM_grid, MOM_restart, ... and through them FMS, which
!! this repository cannot build); it is the source a maintainer puts in place of the reference module.  What it does:
!!   * register_restarts_dyn_split_RK2 registers HOST mirrors of the restart variables of :1222-1290 (eta, u_av/v_av,
!!     CAu_pred/CAv_pred, h_av, uh/vh, diffu/diffv, and the barotropic ones through register_barotropic_restarts), exactly
!!     as the reference does; refresh_host_mirrors (called by the MOM.F90 driver before save_restart / post_data, see
!!     INTEGRATION.md section 4) downloads them -- NOT every step;
!!   * initialize_dyn_split_RK2 reads the parameters the reference reads (:1425-1500), builds the device context from
!!     G, GV, calls the device initialisations INCLUDING vertvisc_init and hor_visc_init (so vertvisc_coef and
!!     horizontal_viscosity run on the device inside the step) and, on a new run, the first-step fills of :1577-1650;
!!   * step_MOM_dyn_split_RK2 uploads the state only when the host changed it (first call, after a host-side ALE or
!!     thermodynamic step: MOM.F90 sets CS%host_changed_state through dyn_split_RK2_host_changed), uploads the wind
!!     stress, runs the step on the resident state, and returns.  u, v, h, uhtr, vhtr are downloaded by
!!     dyn_split_RK2_sync_host where MOM.F90 needs them on the host (before the host thermodynamics at DT_THERM).
module MOM_dynamics_split_RK2

use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use MOM_ALE,               only : ALE_CS
use MOM_barotropic,        only : barotropic_CS, register_barotropic_restarts
use MOM_diabatic_driver,   only : diabatic_CS
use MOM_diag_mediator,     only : diag_ctrl
use MOM_error_handler,     only : MOM_error, FATAL, WARNING
use MOM_file_parser,       only : get_param, param_file_type
use MOM_forcing_type,      only : mech_forcing
use MOM_get_input,         only : directories
use MOM_grid,              only : ocean_grid_type
use MOM_harmonic_analysis, only : harmonic_analysis_CS
use MOM_hor_index,         only : hor_index_type
use MOM_io,                only : vardesc, var_desc
use MOM_MEKE_types,        only : MEKE_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_open_boundary,     only : ocean_OBC_type, update_OBC_CS
use MOM_porous_barriers,   only : porous_barrier_type
use MOM_restart,           only : register_restart_field, register_restart_pair, query_initialized, MOM_restart_CS
use MOM_set_visc,          only : set_visc_CS
use MOM_stochastics,       only : stochastic_CS
use MOM_thickness_diffuse, only : thickness_diffuse_CS
use MOM_time_manager,      only : time_type
use MOM_unit_scaling,      only : unit_scale_type
use MOM_variables,         only : thermo_var_ptrs, vertvisc_type, ocean_internal_state, accel_diag_ptrs, cont_diag_ptrs
use MOM_verticalGrid,      only : verticalGrid_type
use MOM_wave_interface,    only : wave_parameters_CS

implicit none ; private

#include <MOM_memory.h>

public :: step_MOM_dyn_split_RK2, register_restarts_dyn_split_RK2, initialize_dyn_split_RK2
public :: remap_dyn_split_RK2_aux_vars, init_dyn_split_RK2_diabatic, end_dyn_split_RK2
public :: dyn_split_RK2_host_changed, dyn_split_RK2_sync_host, refresh_host_mirrors   ! additions: see the header

!> The control structure (opaque to callers, as in the reference: "; private")
type, public :: MOM_dyn_split_RK2_CS ; private
  type(c_ptr) :: ctx = c_null_ptr            !< mom6x_ctx: the tile on the device
  type(mom6x_dims) :: dims                   !< its layout
  type(dyn_state_type) :: S                  !< u, v, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy in HBM
  logical :: host_changed_state = .true.     !< the host arrays are newer than the device copy
  !> HOST mirrors of the restart variables (registered by pointer with MOM_restart, :1222-1290)
  real, allocatable, dimension(:,:)   :: eta
  real, allocatable, dimension(:,:,:) :: u_av, v_av, h_av, CAu_pred, CAv_pred, diffu, diffv
  logical :: store_CAu = .true., remap_aux = .false., module_is_initialized = .false.
  type(barotropic_CS) :: barotropic_CSp      !< kept so that register_barotropic_restarts can register ubtav, vbtav, ...
end type MOM_dyn_split_RK2_CS

!> mom6x_rk2_field ids of the restart variables (include/mom6x.h)
integer(c_int), parameter :: F_CAU_PRED = 2, F_CAV_PRED = 3, F_DIFFU = 6, F_DIFFV = 7, F_U_AV = 12, F_V_AV = 13, F_H_AV = 14, F_ETA = 16

contains

!> step_MOM_dyn_split_RK2 (:294-296)
subroutine step_MOM_dyn_split_RK2(u_inst, v_inst, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, &
                                  uh, vh, uhtr, vhtr, eta_av, G, GV, US, CS, calc_dtbt, VarMix, &
                                  MEKE, thickness_diffuse_CSp, pbv, STOCH, Waves)
  type(ocean_grid_type),             intent(inout) :: G
  type(verticalGrid_type),           intent(in)    :: GV
  type(unit_scale_type),             intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: u_inst
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: v_inst
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(inout) :: h
  type(thermo_var_ptrs),             intent(in)    :: tv
  type(vertvisc_type),               intent(inout) :: visc
  type(time_type),                   intent(in)    :: Time_local
  real,                              intent(in)    :: dt
  type(mech_forcing),                intent(in)    :: forces
  real, dimension(:,:),              pointer       :: p_surf_begin
  real, dimension(:,:),              pointer       :: p_surf_end
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: vh
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: uhtr
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: vhtr
  real, dimension(SZI_(G),SZJ_(G)),  intent(out)   :: eta_av
  type(MOM_dyn_split_RK2_CS),        pointer       :: CS
  logical,                           intent(in)    :: calc_dtbt
  type(VarMix_CS),                   intent(inout) :: VarMix
  type(MEKE_type),                   intent(inout) :: MEKE
  type(thickness_diffuse_CS),        intent(inout) :: thickness_diffuse_CSp
  type(porous_barrier_type),         intent(in)    :: pbv
  type(stochastic_CS),               intent(inout) :: STOCH
  type(wave_parameters_CS), optional, pointer      :: Waves
  integer(c_int) :: rc

  if (associated(p_surf_begin) .or. associated(p_surf_end)) call MOM_error(FATAL, &
    "step_MOM_dyn_split_RK2: a time-varying surface pressure is not carried by the device path.")
  if (CS%host_changed_state) then   ! the first step, or the host (ALE, thermodynamics on the host) changed u, v, h
    call dyn_state_upload(CS%S, u_inst, v_inst, h, uh, vh, uhtr, vhtr)
    if (associated(tv%T)) then      ! tv%T, tv%S for the equation-of-state branch of PressureForce_FV_Bouss
      call upload_tv(CS, tv, G, GV)
    endif
    CS%host_changed_state = .false.
  endif
  ! set_viscous_BBL outputs vertvisc_coef reads (visc%Kv_bbl_u, ... change when MOM.F90 has called set_viscous_BBL)
  call upload_visc(CS, visc, G, GV)
  call dyn_step(CS%S, forces%taux, forces%tauy, real(dt, c_double), calc_dtbt)
  ! eta_av is small and MOM.F90 uses it right away (ssh accumulation, MOM.F90:1393): one 2-d download per step
  rc = mom6x_download(CS%ctx, eta_av, CS%S%eta_av, STG_H, 1_c_int)
  if (rc /= 0) call MOM_error(FATAL, "step_MOM_dyn_split_RK2: "//trim(mom6x_message()))
end subroutine step_MOM_dyn_split_RK2

!> To be called by MOM.F90 after it changed u, v, h on the host (ALE_regridding_and_remapping, the diabatic step).
subroutine dyn_split_RK2_host_changed(CS)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  CS%host_changed_state = .true.
end subroutine dyn_split_RK2_host_changed

!> To be called by MOM.F90 where the host reads u, v, h, uhtr, vhtr next (before the thermodynamic step, diagnostics).
subroutine dyn_split_RK2_sync_host(CS, u, v, h, uh, vh, uhtr, vhtr, eta_av)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  real, dimension(:,:,:), intent(inout) :: u, v, h, uh, vh, uhtr, vhtr
  real, dimension(:,:),   intent(inout) :: eta_av
  call dyn_state_download(CS%S, u, v, h, uh, vh, uhtr, vhtr, eta_av)
end subroutine dyn_split_RK2_sync_host

!> Download the restart variables into their registered host mirrors (before save_restart; RK2.F90:1222-1290).
subroutine refresh_host_mirrors(CS, G, GV)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  integer(c_int) :: rc, nk
  nk = int(GV%ke, c_int)
  rc = mom6x_download(CS%ctx, CS%eta, mom6x_rk2_field(CS%ctx, F_ETA), STG_H, 1_c_int)
  rc = mom6x_download(CS%ctx, CS%u_av, mom6x_rk2_field(CS%ctx, F_U_AV), STG_U, nk)
  rc = mom6x_download(CS%ctx, CS%v_av, mom6x_rk2_field(CS%ctx, F_V_AV), STG_V, nk)
  rc = mom6x_download(CS%ctx, CS%h_av, mom6x_rk2_field(CS%ctx, F_H_AV), STG_H, nk)
  if (CS%store_CAu) then
    rc = mom6x_download(CS%ctx, CS%CAu_pred, mom6x_rk2_field(CS%ctx, F_CAU_PRED), STG_U, nk)
    rc = mom6x_download(CS%ctx, CS%CAv_pred, mom6x_rk2_field(CS%ctx, F_CAV_PRED), STG_V, nk)
  endif
  rc = mom6x_download(CS%ctx, CS%diffu, mom6x_rk2_field(CS%ctx, F_DIFFU), STG_U, nk)
  rc = mom6x_download(CS%ctx, CS%diffv, mom6x_rk2_field(CS%ctx, F_DIFFV), STG_V, nk)
  if (rc /= 0) call MOM_error(FATAL, "refresh_host_mirrors: "//trim(mom6x_message()))
end subroutine refresh_host_mirrors

!> register_restarts_dyn_split_RK2 (:1210): the same variables under the same names, backed by the host mirrors
subroutine register_restarts_dyn_split_RK2(HI, GV, US, param_file, CS, restart_CS, uh, vh)
  type(hor_index_type),          intent(in)    :: HI
  type(verticalGrid_type),       intent(in)    :: GV
  type(unit_scale_type),         intent(in)    :: US
  type(param_file_type),         intent(in)    :: param_file
  type(MOM_dyn_split_RK2_CS),    pointer       :: CS
  type(MOM_restart_CS),          intent(inout) :: restart_CS
  real, dimension(SZIB_(HI),SZJ_(HI),SZK_(GV)), target, intent(inout) :: uh
  real, dimension(SZI_(HI),SZJB_(HI),SZK_(GV)), target, intent(inout) :: vh
  character(len=40) :: mdl = "MOM_dynamics_split_RK2"
  type(vardesc) :: vd(2)
  character(len=48) :: thickness_units, flux_units
  integer :: isd, ied, jsd, jed, nz, IsdB, IedB, JsdB, JedB

  isd = HI%isd ; ied = HI%ied ; jsd = HI%jsd ; jed = HI%jed ; nz = GV%ke
  IsdB = HI%IsdB ; IedB = HI%IedB ; JsdB = HI%JsdB ; JedB = HI%JedB
  if (associated(CS)) then
    call MOM_error(WARNING, "register_restarts_split_RK2 called with an associated control structure.")
    return
  endif
  allocate(CS)
  call get_param(param_file, mdl, "STORE_CORIOLIS_ACCEL", CS%store_CAu, default=.true., do_not_log=.true.)
  allocate(CS%eta(isd:ied,jsd:jed), source=0.0)
  allocate(CS%u_av(IsdB:IedB,jsd:jed,nz), source=0.0) ; allocate(CS%v_av(isd:ied,JsdB:JedB,nz), source=0.0)
  allocate(CS%h_av(isd:ied,jsd:jed,nz), source=GV%Angstrom_H)
  allocate(CS%diffu(IsdB:IedB,jsd:jed,nz), source=0.0) ; allocate(CS%diffv(isd:ied,JsdB:JedB,nz), source=0.0)
  if (CS%store_CAu) then
    allocate(CS%CAu_pred(IsdB:IedB,jsd:jed,nz), source=0.0) ; allocate(CS%CAv_pred(isd:ied,JsdB:JedB,nz), source=0.0)
  endif
  thickness_units = "m" ; flux_units = "m3 s-1"
  call register_restart_field(CS%eta, "sfc", .false., restart_CS, longname="Free surface Height", units=thickness_units, &
                              conversion=GV%H_to_mks)
  vd(1) = var_desc("u2", "m s-1", "Auxiliary Zonal velocity", 'u', 'L')
  vd(2) = var_desc("v2", "m s-1", "Auxiliary Meridional velocity", 'v', 'L')
  call register_restart_pair(CS%u_av, CS%v_av, vd(1), vd(2), .false., restart_CS, conversion=US%L_T_to_m_s)
  if (CS%store_CAu) then
    vd(1) = var_desc("CAu", "m s-2", "Zonal Coriolis and advactive acceleration", 'u', 'L')
    vd(2) = var_desc("CAv", "m s-2", "Meridional Coriolis and advactive acceleration", 'v', 'L')
    call register_restart_pair(CS%CAu_pred, CS%CAv_pred, vd(1), vd(2), .false., restart_CS, conversion=US%L_T2_to_m_s2)
  else
    call register_restart_field(CS%h_av, "h2", .false., restart_CS, longname="Auxiliary Layer Thickness", &
                                units=thickness_units, conversion=GV%H_to_mks)
    vd(1) = var_desc("uh", flux_units, "Zonal thickness flux", 'u', 'L')
    vd(2) = var_desc("vh", flux_units, "Meridional thickness flux", 'v', 'L')
    call register_restart_pair(uh, vh, vd(1), vd(2), .false., restart_CS, conversion=GV%H_to_MKS*US%L_to_m**2*US%s_to_T)
  endif
  vd(1) = var_desc("diffu", "m s-2", "Zonal horizontal viscous acceleration", 'u', 'L')
  vd(2) = var_desc("diffv", "m s-2", "Meridional horizontal viscous acceleration", 'v', 'L')
  call register_restart_pair(CS%diffu, CS%diffv, vd(1), vd(2), .false., restart_CS, conversion=US%L_T2_to_m_s2)
  call register_barotropic_restarts(HI, GV, US, param_file, CS%barotropic_CSp, restart_CS)
end subroutine register_restarts_dyn_split_RK2

!> remap_dyn_split_RK2_aux_vars (:1302): u_av, v_av, diffu, diffv (and CAu_pred, CAv_pred) follow the new grid
subroutine remap_dyn_split_RK2_aux_vars(G, GV, CS, h_old_u, h_old_v, h_new_u, h_new_v, ALE_CSp)
  type(ocean_grid_type),            intent(inout) :: G
  type(verticalGrid_type),          intent(in)    :: GV
  type(MOM_dyn_split_RK2_CS),       pointer       :: CS
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: h_old_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: h_old_v
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: h_new_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: h_new_v
  type(ALE_CS),                     pointer       :: ALE_CSp
  type(c_ptr) :: d_ou, d_ov, d_nu, d_nv
  type(mom6x_remapping_params) :: RP
  integer(c_int) :: rc, nk
  integer(c_size_t) :: n3
  if (.not.CS%remap_aux) return
  nk = int(GV%ke, c_int) ; n3 = int(CS%dims%slab, c_size_t) * int(nk, c_size_t)
  rc = mom6x_dev_alloc(CS%ctx, d_ou, n3) ; rc = mom6x_dev_alloc(CS%ctx, d_ov, n3)
  rc = mom6x_dev_alloc(CS%ctx, d_nu, n3) ; rc = mom6x_dev_alloc(CS%ctx, d_nv, n3)
  rc = mom6x_upload(CS%ctx, d_ou, h_old_u, STG_U, nk) ; rc = mom6x_upload(CS%ctx, d_ov, h_old_v, STG_V, nk)
  rc = mom6x_upload(CS%ctx, d_nu, h_new_u, STG_U, nk) ; rc = mom6x_upload(CS%ctx, d_nv, h_new_v, STG_V, nk)
  call remapping_params_of(ALE_CSp, RP)   ! REMAPPING_SCHEME etc. of the ALE control structure (INTEGRATION.md section 7)
  rc = mom6x_remap_dyn_split_RK2_aux_vars(CS%ctx, RP, d_ou, d_ov, d_nu, d_nv)
  if (rc /= 0) call MOM_error(FATAL, "remap_dyn_split_RK2_aux_vars: "//trim(mom6x_message()))
  rc = mom6x_dev_free(CS%ctx, d_ou) ; rc = mom6x_dev_free(CS%ctx, d_ov) ; rc = mom6x_dev_free(CS%ctx, d_nu) ; rc = mom6x_dev_free(CS%ctx, d_nv)
end subroutine remap_dyn_split_RK2_aux_vars

!> init_dyn_split_RK2_diabatic (:1335): the device step has no use for KPP / ePBL members (FPMIX is off on this path)
subroutine init_dyn_split_RK2_diabatic(diabatic_CSp, CS)
  type(diabatic_CS),          intent(in) :: diabatic_CSp
  type(MOM_dyn_split_RK2_CS), pointer    :: CS
end subroutine init_dyn_split_RK2_diabatic

!> initialize_dyn_split_RK2 (:1346-1350)
subroutine initialize_dyn_split_RK2(u, v, h, tv, uh, vh, eta, Time, G, GV, US, param_file, &
                      diag, CS, HA_CSp, restart_CS, dt, Accel_diag, Cont_diag, MIS, &
                      VarMix, MEKE, thickness_diffuse_CSp,                  &
                      OBC, update_OBC_CSp, ALE_CSp, set_visc, &
                      visc, dirs, ntrunc, pbv, calc_dtbt, cont_stencil)
  type(ocean_grid_type),            intent(inout) :: G
  type(verticalGrid_type),          intent(in)    :: GV
  type(unit_scale_type),            intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(inout) :: h
  type(thermo_var_ptrs),            intent(in)    :: tv
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: vh
  real, dimension(SZI_(G),SZJ_(G)), intent(inout) :: eta
  type(time_type),          target, intent(in)    :: Time
  type(param_file_type),            intent(in)    :: param_file
  type(diag_ctrl),          target, intent(inout) :: diag
  type(MOM_dyn_split_RK2_CS),       pointer       :: CS
  type(harmonic_analysis_CS),       pointer       :: HA_CSp
  type(MOM_restart_CS),             intent(inout) :: restart_CS
  real,                             intent(in)    :: dt
  type(accel_diag_ptrs),    target, intent(inout) :: Accel_diag
  type(cont_diag_ptrs),     target, intent(inout) :: Cont_diag
  type(ocean_internal_state),       intent(inout) :: MIS
  type(VarMix_CS),                  intent(inout) :: VarMix
  type(MEKE_type),                  intent(inout) :: MEKE
  type(thickness_diffuse_CS),       intent(inout) :: thickness_diffuse_CSp
  type(ocean_OBC_type),             pointer       :: OBC
  type(update_OBC_CS),              pointer       :: update_OBC_CSp
  type(ALE_CS),                     pointer       :: ALE_CSp
  type(set_visc_CS),        target, intent(in)    :: set_visc
  type(vertvisc_type),              intent(inout) :: visc
  type(directories),                intent(in)    :: dirs
  integer, target,                  intent(inout) :: ntrunc
  logical,                          intent(out)   :: calc_dtbt
  type(porous_barrier_type),        intent(in)    :: pbv
  integer,                          intent(out)   :: cont_stencil
  character(len=40) :: mdl = "MOM_dynamics_split_RK2"
  type(mom6x_vgrid) :: gvx
  type(mom6x_continuity_params) :: cont ; type(mom6x_barotropic_params) :: bt ; type(mom6x_coriolis_params) :: cor
  type(mom6x_pgf_params) :: pgf ; type(mom6x_rk2_params) :: rk2 ; type(mom6x_vertvisc_params) :: vv
  type(mom6x_hor_visc_params) :: hv
  real(c_double), allocatable :: block(:)
  integer(c_int) :: rc
  logical :: new_run

  if (.not.associated(CS)) call MOM_error(FATAL, "initialize_dyn_split_RK2 called with an unassociated control structure.")
  if (CS%module_is_initialized) then
    call MOM_error(WARNING, "initialize_dyn_split_RK2 called with a control structure that has already been initialized.")
    return
  endif
  if (associated(OBC)) call MOM_error(FATAL, "initialize_dyn_split_RK2: open boundaries are not carried by the device path.")
  CS%module_is_initialized = .true.

  ! ---- the layout and the metric block: one plane per array of ocean_grid_type (mom6x_host: mom6x_pack_plane) ----------
  rc = mom6x_dims_init(CS%dims, int(G%iec-G%isc+1, c_int), int(G%jec-G%jsc+1, c_int), int(GV%ke, c_int), int(G%isc-G%isd, c_int))
  CS%dims%i_glob0 = G%idg_offset ; CS%dims%j_glob0 = G%jdg_offset
  CS%dims%ni_glob = G%Domain%niglobal ; CS%dims%nj_glob = G%Domain%njglobal
  allocate(block(0:int(G_COUNT, c_size_t)*int(CS%dims%slab, c_size_t)-1), source=0.0_c_double)
  call mom6x_pack_plane(CS%dims, block, G_mask2dT, G%mask2dT, STG_H) ; call mom6x_pack_plane(CS%dims, block, G_mask2dCu, G%mask2dCu, STG_U)
  call mom6x_pack_plane(CS%dims, block, G_mask2dCv, G%mask2dCv, STG_V) ; call mom6x_pack_plane(CS%dims, block, G_mask2dBu, G%mask2dBu, STG_Q)
  call mom6x_pack_plane(CS%dims, block, G_dxT, G%dxT, STG_H)   ; call mom6x_pack_plane(CS%dims, block, G_dyT, G%dyT, STG_H)
  call mom6x_pack_plane(CS%dims, block, G_IdxT, G%IdxT, STG_H) ; call mom6x_pack_plane(CS%dims, block, G_IdyT, G%IdyT, STG_H)
  call mom6x_pack_plane(CS%dims, block, G_dxCu, G%dxCu, STG_U) ; call mom6x_pack_plane(CS%dims, block, G_dyCu, G%dyCu, STG_U)
  call mom6x_pack_plane(CS%dims, block, G_IdxCu, G%IdxCu, STG_U) ; call mom6x_pack_plane(CS%dims, block, G_IdyCu, G%IdyCu, STG_U)
  call mom6x_pack_plane(CS%dims, block, G_dxCv, G%dxCv, STG_V) ; call mom6x_pack_plane(CS%dims, block, G_dyCv, G%dyCv, STG_V)
  call mom6x_pack_plane(CS%dims, block, G_IdxCv, G%IdxCv, STG_V) ; call mom6x_pack_plane(CS%dims, block, G_IdyCv, G%IdyCv, STG_V)
  call mom6x_pack_plane(CS%dims, block, G_dxBu, G%dxBu, STG_Q) ; call mom6x_pack_plane(CS%dims, block, G_dyBu, G%dyBu, STG_Q)
  call mom6x_pack_plane(CS%dims, block, G_IdxBu, G%IdxBu, STG_Q) ; call mom6x_pack_plane(CS%dims, block, G_IdyBu, G%IdyBu, STG_Q)
  call mom6x_pack_plane(CS%dims, block, G_areaT, G%areaT, STG_H) ; call mom6x_pack_plane(CS%dims, block, G_IareaT, G%IareaT, STG_H)
  call mom6x_pack_plane(CS%dims, block, G_areaBu, G%areaBu, STG_Q) ; call mom6x_pack_plane(CS%dims, block, G_IareaBu, G%IareaBu, STG_Q)
  call mom6x_pack_plane(CS%dims, block, G_areaCu, G%areaCu, STG_U) ; call mom6x_pack_plane(CS%dims, block, G_areaCv, G%areaCv, STG_V)
  call mom6x_pack_plane(CS%dims, block, G_IareaCu, G%IareaCu, STG_U) ; call mom6x_pack_plane(CS%dims, block, G_IareaCv, G%IareaCv, STG_V)
  call mom6x_pack_plane(CS%dims, block, G_dy_Cu, G%dy_Cu, STG_U) ; call mom6x_pack_plane(CS%dims, block, G_dx_Cv, G%dx_Cv, STG_V)
  call mom6x_pack_plane(CS%dims, block, G_bathyT, G%bathyT, STG_H)
  call mom6x_pack_plane(CS%dims, block, G_CoriolisBu, G%CoriolisBu, STG_Q) ; call mom6x_pack_plane(CS%dims, block, G_Coriolis2Bu, G%Coriolis2Bu, STG_Q)
  gvx%g_Earth = GV%g_Earth ; gvx%Rho0 = GV%Rho0 ; gvx%Angstrom_H = GV%Angstrom_H ; gvx%H_subroundoff = GV%H_subroundoff
  gvx%dZ_subroundoff = GV%dZ_subroundoff ; gvx%H_to_Z = GV%H_to_Z ; gvx%Z_to_H = GV%Z_to_H ; gvx%H_to_RZ = GV%H_to_RZ
  gvx%RZ_to_H = GV%RZ_to_H ; gvx%Boussinesq = merge(1_c_int, 0_c_int, GV%Boussinesq)
  rc = mom6x_ctx_create(CS%ctx, CS%dims, 0_c_int, block, gvx, int(G%first_direction, c_int))
  if (rc /= 0) call MOM_error(FATAL, "initialize_dyn_split_RK2: "//trim(mom6x_message()))

  ! ---- parameters: the names, defaults and order of the reference's get_param calls ------------------------------------
  call read_continuity_params(param_file, GV, cont)   ! continuity_PPM_init :2674-2754
  call read_barotropic_params(param_file, G, GV, US, dt, bt)   ! barotropic_init :5403-5713
  call read_coriolis_params(param_file, cor)          ! CoriolisAdv_init :1054-1200
  call read_pgf_params(param_file, GV, pgf)           ! PressureForce_FV_init :2020-2200
  call get_param(param_file, mdl, "BE", rk2%be, "If SPLIT is true, BE determines the relative weighting of a forward-backward "//&
                 "and a backward Euler treatment of the baroclinic gravity waves.", units="nondim", default=0.6)
  call get_param(param_file, mdl, "BEGW", rk2%begw, "If SPLIT is true, BEGW is a number from 0 to 1 that controls the extent "//&
                 "to which the treatment of gravity waves is forward-backward (0) or simulated backward Euler (1).", units="nondim", default=0.0)
  call get_rk2_flags(param_file, rk2)                 ! SPLIT_BOTTOM_STRESS, BT_USE_LAYER_FLUXES, STORE_CORIOLIS_ACCEL, VISC_REM_BUG, REMAP_AUXILIARY_VARS
  CS%remap_aux = (rk2%remap_aux /= 0)
  rc = mom6x_continuity_init(CS%ctx, cont) ; if (rc /= 0) call MOM_error(FATAL, trim(mom6x_message()))
  rc = mom6x_barotropic_init(CS%ctx, bt)   ; if (rc /= 0) call MOM_error(FATAL, trim(mom6x_message()))
  rc = mom6x_CoriolisAdv_init(CS%ctx, cor) ; if (rc /= 0) call MOM_error(FATAL, trim(mom6x_message()))
  rc = mom6x_PressureForce_init(CS%ctx, pgf, GV%Rlay, GV%g_prime) ; if (rc /= 0) call MOM_error(FATAL, trim(mom6x_message()))
  rc = mom6x_initialize_dyn_split_RK2(CS%ctx, rk2) ; if (rc /= 0) call MOM_error(FATAL, trim(mom6x_message()))
  ! vertvisc_coef (x3 per step) and horizontal_viscosity (x1) run on the device inside the step:
  call read_vertvisc_params(param_file, GV, US, vv)   ! vertvisc_init, MOM_vert_friction.F90:2932-3200
  rc = mom6x_vertvisc_init(CS%ctx, vv) ; if (rc /= 0) call MOM_error(FATAL, trim(mom6x_message()))
  call read_hor_visc_params(param_file, G, US, dt, hv) ! hor_visc_init, MOM_hor_visc.F90:2322-3000
  rc = mom6x_hor_visc_init(CS%ctx, hv) ; if (rc /= 0) call MOM_error(FATAL, trim(mom6x_message()))
  cont_stencil = 3 ; if (cont%simple_2nd /= 0) cont_stencil = 2 ; if (cont%upwind_1st /= 0) cont_stencil = 1   ! continuity_stencil :2757

  ! ---- the state and the first-step fills of :1577-1650 ---------------------------------------------------------------
  call dyn_state_init(CS%S, CS%ctx, CS%dims)
  call dyn_state_upload(CS%S, u, v, h, uh, vh, uh, vh) ; CS%host_changed_state = .false.
  new_run = .not. query_initialized(CS%eta, "sfc", restart_CS)
  if (new_run) then
    rc = mom6x_dyn_split_RK2_new_run(CS%ctx, CS%S%u, CS%S%v, CS%S%h, CS%S%uh, CS%S%vh, real(dt, c_double))
    if (rc /= 0) call MOM_error(FATAL, "initialize_dyn_split_RK2: "//trim(mom6x_message()))
  else   ! a restarted run: the registered mirrors hold the file's values
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_ETA), CS%eta, STG_H, 1_c_int)
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_U_AV), CS%u_av, STG_U, int(GV%ke, c_int))
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_V_AV), CS%v_av, STG_V, int(GV%ke, c_int))
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_H_AV), CS%h_av, STG_H, int(GV%ke, c_int))
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_DIFFU), CS%diffu, STG_U, int(GV%ke, c_int))
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_DIFFV), CS%diffv, STG_V, int(GV%ke, c_int))
    if (CS%store_CAu) then
      rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_CAU_PRED), CS%CAu_pred, STG_U, int(GV%ke, c_int))
      rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_CAV_PRED), CS%CAv_pred, STG_V, int(GV%ke, c_int))
      rc = mom6x_rk2_set_CAu_pred_stored(CS%ctx, 1_c_int)
    endif
  endif
  rc = mom6x_download(CS%ctx, eta, mom6x_rk2_field(CS%ctx, F_ETA), STG_H, 1_c_int)   ! the eta argument is intent(inout)
  calc_dtbt = .true.
end subroutine initialize_dyn_split_RK2

!> end_dyn_split_RK2 (:1885)
subroutine end_dyn_split_RK2(CS)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  integer(c_int) :: rc
  if (.not.associated(CS)) return
  call dyn_state_end(CS%S)
  rc = mom6x_ctx_destroy(CS%ctx)
  deallocate(CS)
end subroutine end_dyn_split_RK2

! The parameter readers (read_continuity_params, read_barotropic_params, read_coriolis_params, read_pgf_params,
! get_rk2_flags, read_vertvisc_params, read_hor_visc_params), upload_tv, upload_visc and remapping_params_of are the
! mechanical part of the shim: one get_param per member of the bind(C) parameter structs of fortran/mom6x_c_api.F90,
! with the names and defaults listed per member in include/mom6x.h (every member's comment gives the MOM_input name
! and the reference default).  They are in fortran/shims/mom6x_param_readers.inc (INTEGRATION.md section 3).
#include "mom6x_param_readers.inc"

end module MOM_dynamics_split_RK2
