!> Drop-in for the entry points of src/parameterizations/vertical/MOM_vert_friction.F90 on the dynamical core's path:
!! vertvisc :557, vertvisc_remnant :1229, vertvisc_coef :1357, vertvisc_init :3135, vertvisc_end :3676,
!! updateCFLtruncationValue :3636 and the type vertvisc_CS -- same names and argument lists.  vertvisc_coef leaves its
!! coefficient set (a_u, a_v, h_u, h_v of the reference's CS) inside the device context, where vertvisc and
!! vertvisc_remnant find it, exactly as the reference's three routines communicate through CS.
!! vertvisc_limit_vel (CFL truncation with its point-acceleration files) and vertFPmix stay host Fortran.
module MOM_vert_friction
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_diag_mediator,         only : diag_ctrl
use MOM_error_handler,         only : MOM_error, FATAL
use MOM_file_parser,           only : get_param, log_version, param_file_type
use MOM_forcing_type,          only : mech_forcing
use MOM_get_input,             only : directories
use MOM_grid,                  only : ocean_grid_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_open_boundary,         only : ocean_OBC_type
use MOM_time_manager,          only : time_type
use MOM_unit_scaling,          only : unit_scale_type
use MOM_variables,             only : thermo_var_ptrs, vertvisc_type, ocean_internal_state, accel_diag_ptrs, cont_diag_ptrs
use MOM_verticalGrid,          only : verticalGrid_type
use MOM_wave_interface,        only : wave_parameters_CS
implicit none ; private
#include <MOM_memory.h>
public :: vertvisc, vertvisc_remnant, vertvisc_coef, vertvisc_init, vertvisc_end, updateCFLtruncationValue
public :: vertvisc_upload_visc, vertvisc_read_params

type, public :: vertvisc_CS ; private
  logical :: initialized = .false.
  type(c_ptr) :: ctx = c_null_ptr
  type(mom6x_vertvisc_params) :: p
  real :: Hmix_stress = 0.0          !< HMIX_STRESS with DIRECT_STRESS (0: the stress goes into the top layer)
end type vertvisc_CS

contains

!> The vertvisc_type members vertvisc_coef and the solves read (set_viscous_BBL outputs; :1519-1525, :2439, Ray_u :640)
subroutine vertvisc_upload_visc(ctx, visc, GV, slot0)
  type(c_ptr), intent(in) :: ctx ; type(vertvisc_type), intent(in) :: visc ; type(verticalGrid_type), intent(in) :: GV
  integer, intent(in) :: slot0      !< the first of seven scratch slots to use
  type(c_ptr) :: p(7)
  integer(c_int) :: rc
  p(:) = c_null_ptr
  if (allocated(visc%Kv_bbl_u)) p(1) = shim_up2(slot0, visc%Kv_bbl_u, STG_U)
  if (allocated(visc%Kv_bbl_v)) p(2) = shim_up2(slot0+1, visc%Kv_bbl_v, STG_V)
  if (allocated(visc%bbl_thick_u)) p(3) = shim_up2(slot0+2, visc%bbl_thick_u, STG_U)
  if (allocated(visc%bbl_thick_v)) p(4) = shim_up2(slot0+3, visc%bbl_thick_v, STG_V)
  if (allocated(visc%Kv_shear)) p(5) = shim_up3(slot0+4, visc%Kv_shear, STG_H, GV%ke+1)
  if (allocated(visc%Ray_u)) p(6) = shim_up3(slot0+5, visc%Ray_u, STG_U, GV%ke)
  if (allocated(visc%Ray_v)) p(7) = shim_up3(slot0+6, visc%Ray_v, STG_V, GV%ke)
  rc = mom6x_vertvisc_set_visc(ctx, p(1), p(2), p(3), p(4), p(5), p(6), p(7)) ; call shim_check(rc, "vertvisc_coef (visc)")
end subroutine vertvisc_upload_visc

!> vertvisc_coef (:1357).  dz is thickness_to_dz(h) in Boussinesq mode (= GV%H_to_Z * h), which the device forms itself.
subroutine vertvisc_coef(u, v, h, dz, forces, visc, tv, dt, G, GV, US, CS, OBC, VarMix)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: dz
  type(mech_forcing),      intent(in)    :: forces
  type(vertvisc_type),     intent(in)    :: visc
  type(thermo_var_ptrs),   intent(in)    :: tv
  real,                    intent(in)    :: dt
  type(vertvisc_CS),       intent(inout) :: CS
  type(ocean_OBC_type),    pointer       :: OBC
  type(VarMix_CS),         intent(in)    :: VarMix
  integer(c_int) :: rc
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_vert_friction(coef): Module must be initialized before it is used.")
  if (associated(OBC)) call MOM_error(FATAL, "vertvisc_coef: open boundaries are not carried by the MI355X path.")
  call vertvisc_upload_visc(CS%ctx, visc, GV, 10)
  rc = mom6x_vertvisc_coef(CS%ctx, shim_up3(1, u, STG_U, GV%ke), shim_up3(2, v, STG_V, GV%ke), shim_up3(3, h, STG_H, GV%ke), &
                           real(dt, c_double))
  call shim_check(rc, "vertvisc_coef")
end subroutine vertvisc_coef

!> vertvisc (:557)
subroutine vertvisc(u, v, h, forces, visc, dt, OBC, ADp, CDp, G, GV, US, CS, taux_bot, tauy_bot, fpmix, Waves)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: h
  type(mech_forcing),    intent(in)      :: forces
  type(vertvisc_type),   intent(inout)   :: visc
  real,                  intent(in)      :: dt
  type(ocean_OBC_type),  pointer         :: OBC
  type(accel_diag_ptrs), intent(inout)   :: ADp
  type(cont_diag_ptrs),  intent(inout)   :: CDp
  type(vertvisc_CS),     pointer         :: CS
  real, dimension(SZIB_(G),SZJ_(G)), optional, intent(out) :: taux_bot
  real, dimension(SZI_(G),SZJB_(G)), optional, intent(out) :: tauy_bot
  logical,               optional, intent(in) :: fpmix
  type(wave_parameters_CS), optional, pointer :: Waves
  type(c_ptr) :: d_u, d_v, d_h, p_txb, p_tyb
  integer(c_int) :: rc
  if (.not.associated(CS)) call MOM_error(FATAL, "MOM_vert_friction(visc): Module must be initialized before it is used.")
  if (associated(OBC)) call MOM_error(FATAL, "vertvisc: open boundaries are not carried by the MI355X path.")
  if (present(fpmix)) then ; if (fpmix) call MOM_error(FATAL, "vertvisc: FPMIX is not carried by the MI355X path.") ; endif
  if (present(Waves)) then ; if (associated(Waves)) call MOM_error(FATAL, "vertvisc: Stokes drift is not carried by the MI355X path.") ; endif
  d_u = shim_up3(1, u, STG_U, GV%ke) ; d_v = shim_up3(2, v, STG_V, GV%ke)
  if (CS%Hmix_stress > 0.0) then      ! DIRECT_STRESS spreads the wind stress over HMIX_STRESS of h (:707-735)
    d_h = shim_up3(3, h, STG_H, GV%ke)
    rc = mom6x_vertvisc_set_direct_stress(CS%ctx, real(CS%Hmix_stress, c_double), d_h) ; call shim_check(rc, "vertvisc (DIRECT_STRESS)")
  endif
  p_txb = c_null_ptr ; p_tyb = c_null_ptr
  if (present(taux_bot)) p_txb = shim_buf(6, 1)
  if (present(tauy_bot)) p_tyb = shim_buf(7, 1)
  rc = mom6x_vertvisc(CS%ctx, d_u, d_v, shim_up2(4, forces%taux, STG_U), shim_up2(5, forces%tauy, STG_V), real(dt, c_double), p_txb, p_tyb)
  call shim_check(rc, "vertvisc")
  call shim_down3(u, d_u, STG_U, GV%ke) ; call shim_down3(v, d_v, STG_V, GV%ke)
  if (present(taux_bot)) call shim_down2(taux_bot, p_txb, STG_U)
  if (present(tauy_bot)) call shim_down2(tauy_bot, p_tyb, STG_V)
end subroutine vertvisc

!> vertvisc_remnant (:1229)
subroutine vertvisc_remnant(visc, visc_rem_u, visc_rem_v, dt, G, GV, US, CS)
  type(ocean_grid_type), intent(in)   :: G
  type(verticalGrid_type), intent(in) :: GV
  type(vertvisc_type),   intent(in)   :: visc
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: visc_rem_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: visc_rem_v
  real,                  intent(in)    :: dt
  type(unit_scale_type), intent(in)    :: US
  type(vertvisc_CS),     pointer       :: CS
  type(c_ptr) :: d_u, d_v
  integer(c_int) :: rc
  if (.not.associated(CS)) call MOM_error(FATAL, "MOM_vert_friction(remnant): Module must be initialized before it is used.")
  d_u = shim_up3(1, visc_rem_u, STG_U, GV%ke) ; d_v = shim_up3(2, visc_rem_v, STG_V, GV%ke)
  rc = mom6x_vertvisc_remnant(CS%ctx, d_u, d_v, real(dt, c_double)) ; call shim_check(rc, "vertvisc_remnant")
  call shim_down3(visc_rem_u, d_u, STG_U, GV%ke) ; call shim_down3(visc_rem_v, d_v, STG_V, GV%ke)
end subroutine vertvisc_remnant

!> The parameters of vertvisc_init :3160-3420 the device path reads, under their MOM_input names
subroutine vertvisc_read_params(param_file, GV, US, p, Hmix_stress)
  type(param_file_type), intent(in) :: param_file ; type(verticalGrid_type), intent(in) :: GV ; type(unit_scale_type), intent(in) :: US
  type(mom6x_vertvisc_params), intent(out) :: p ; real, intent(out) :: Hmix_stress
  character(len=40) :: mdl = "MOM_vert_friction"
  integer :: default_answer_date, answer_date
  logical :: flag, direct_stress
  call get_param(param_file, mdl, "DEFAULT_ANSWER_DATE", default_answer_date, default=99991231, do_not_log=.true.)
  call get_param(param_file, mdl, "VERT_FRICTION_ANSWER_DATE", answer_date, "The vintage of the order of arithmetic and expressions "//&
                 "in the viscous calculations.", default=default_answer_date)
  p%answer_date = answer_date
  call get_param(param_file, mdl, "BOTTOMDRAGLAW", flag, "If true, the bottom stress is calculated with a drag law of the form "//&
                 "c_drag*|u|*u.", default=.true.)
  p%bottomdraglaw = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "DIRECT_STRESS", direct_stress, "If true, the wind stress is distributed over the topmost "//&
                 "HMIX_STRESS of fluid (like in HYCOM), and KVML may be set to a very small value.", default=.false.)
  call get_param(param_file, mdl, "HARMONIC_VISC", flag, "If true, use the harmonic mean thicknesses for calculating the "//&
                 "vertical viscosity.", default=.false.)
  p%harmonic_visc = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "HARMONIC_BL_SCALE", p%harm_BL_val, "A scale to determine when water is in the boundary "//&
                 "layers based solely on harmonic mean thicknesses.", units="nondim", default=0.0)
  call get_param(param_file, mdl, "HMIX_FIXED", p%Hmix, "The prescribed depth over which the near-surface viscosity and "//&
                 "diffusivity are elevated when the bulk mixed layer is not used.", units="m", scale=US%m_to_Z, default=0.0)
  Hmix_stress = 0.0
  if (direct_stress) call get_param(param_file, mdl, "HMIX_STRESS", Hmix_stress, "The depth over which the wind stress is applied "//&
                 "if DIRECT_STRESS is true.", units="m", default=real(p%Hmix*US%Z_to_m), scale=GV%m_to_H)
  call get_param(param_file, mdl, "KV", p%Kv, "The background kinematic viscosity in the interior.", units="m2 s-1", &
                 fail_if_missing=.true., scale=GV%m2_s_to_HZ_T)
  call get_param(param_file, mdl, "KV_ML_INVZ2", p%Kvml_invZ2, "An extra kinematic viscosity in a mixed layer of thickness "//&
                 "HMIX_FIXED, with the actual viscosity scaling as 1/(z*HMIX_FIXED)^2.", units="m2 s-1", default=0.0, scale=GV%m2_s_to_HZ_T)
  call get_param(param_file, mdl, "HBBL", p%Hbbl, "The thickness of a bottom boundary layer with a viscosity increased by "//&
                 "KV_EXTRA_BBL if BOTTOMDRAGLAW is not defined, or the thickness over which near-bottom velocities are averaged "//&
                 "for the drag law if BOTTOMDRAGLAW is defined.", units="m", fail_if_missing=.true., scale=US%m_to_Z)
  call get_param(param_file, mdl, "KV_EXTRA_BBL", p%Kv_extra_bbl, "An extra kinematic viscosity in the benthic boundary layer.", &
                 units="m2 s-1", default=0.0, scale=GV%m2_s_to_HZ_T)
  call must_be("DYNAMIC_VISCOUS_ML", .false.) ; call must_be("FIXED_DEPTH_LOTW_ML", .false.)
  call must_be("USE_GL90_IN_SSW", .false.) ; call must_be("BULKMIXEDLAYER", .false.) ; call must_be("DEBUG_TRUNCATIONS", .false.)
contains
  subroutine must_be(name, default)
    character(len=*), intent(in) :: name ; logical, intent(in) :: default
    logical :: val
    call get_param(param_file, mdl, name, val, default=default, do_not_log=.true.)
    if (val .neqv. default) call MOM_error(FATAL, "vertvisc_init: "//trim(name)//" is not carried by the MI355X path.")
  end subroutine must_be
end subroutine vertvisc_read_params

!> vertvisc_init (:3135)
subroutine vertvisc_init(MIS, Time, G, GV, US, param_file, diag, ADp, dirs, ntrunc, CS, fpmix)
  type(ocean_internal_state), target, intent(in) :: MIS
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(accel_diag_ptrs),   intent(inout) :: ADp
  type(directories),       intent(in)    :: dirs
  integer, target,         intent(inout) :: ntrunc
  type(vertvisc_CS),       pointer       :: CS
  logical, optional,       intent(in)    :: fpmix
  integer(c_int) :: rc
  if (associated(CS)) then
    call MOM_error(FATAL, "vertvisc_init called with an associated control structure.")   ! (a WARNING + return in the reference)
    return
  endif
  allocate(CS)
  CS%initialized = .true.
  if (present(fpmix)) then ; if (fpmix) call MOM_error(FATAL, "vertvisc_init: FPMIX is not carried by the MI355X path.") ; endif
  call log_version(param_file, "MOM_vert_friction", "mom6x", "")
  call vertvisc_read_params(param_file, GV, US, CS%p, CS%Hmix_stress)
  call shim_set_domain_flags(param_file)
  CS%ctx = shim_ctx(G, GV)
  rc = mom6x_vertvisc_init(CS%ctx, CS%p) ; call shim_check(rc, "vertvisc_init")
end subroutine vertvisc_init

!> updateCFLtruncationValue (:3636): the CFL-truncation ramp belongs to vertvisc_limit_vel, which stays on the host
subroutine updateCFLtruncationValue(Time, CS, US, activate)
  type(time_type), target, intent(in)    :: Time
  type(vertvisc_CS),       pointer       :: CS
  type(unit_scale_type),   intent(in)    :: US
  logical, optional,       intent(in)    :: activate
end subroutine updateCFLtruncationValue

!> vertvisc_end (:3676)
subroutine vertvisc_end(CS)
  type(vertvisc_CS), intent(inout) :: CS
  CS%initialized = .false. ; CS%ctx = c_null_ptr
end subroutine vertvisc_end

end module MOM_vert_friction
