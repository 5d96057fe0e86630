!> Drop-in for src/core/MOM_barotropic.F90: btstep :455-459, set_dtbt :3509, btcalc :4360, bt_mass_source :5243,
!! barotropic_init :5301, barotropic_get_tav :6193, barotropic_end :6216, register_barotropic_restarts :6253 and the type
!! barotropic_CS -- same module name, procedure names and argument lists, served by the mom6x_bt* entry points.
!!
!! State.  The device context owns what the reference keeps in barotropic_CS (ubtav, vbtav, eta_cor, frhatu/v, IDatu/v,
!! dtbt).  The RESTART variables of :6253-6296 (ubtav, vbtav, DTBT) are HOST mirrors in this CS, registered with
!! MOM_restart by pointer exactly as the reference registers its own arrays, and
!!   * barotropic_refresh_restart_mirrors downloads them (MOM_dynamics_split_RK2 calls it before the host can write a restart),
!!   * barotropic_init, on a restarted run (query_initialized), uploads ubtav / vbtav and keeps the file's DTBT (:5965-5968).
!! Inside step_MOM_dyn_split_RK2 none of the procedures below is called -- the device step runs btcalc, bt_mass_source and
!! btstep on resident arrays; they serve the other callers (MOM_dynamics_split_RK2b, diagnostics, MOM_hor_visc's
!! barotropic_get_tav) with HOST arrays: upload, compute, download.
module MOM_barotropic
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_cpu_clock,         only : cpu_clock_id, cpu_clock_begin, cpu_clock_end, CLOCK_MODULE, CLOCK_ROUTINE
use MOM_diag_mediator,     only : diag_ctrl
use MOM_error_handler,     only : MOM_error, FATAL, WARNING
use MOM_file_parser,       only : get_param, log_version, param_file_type
use MOM_forcing_type,      only : mech_forcing
use MOM_grid,              only : ocean_grid_type
use MOM_harmonic_analysis, only : harmonic_analysis_CS
use MOM_hor_index,         only : hor_index_type
use MOM_io,                only : vardesc, var_desc
use MOM_open_boundary,     only : ocean_OBC_type
use MOM_restart,           only : register_restart_field, register_restart_pair, query_initialized, MOM_restart_CS
use MOM_self_attr_load,    only : SAL_CS
use MOM_time_manager,      only : time_type
use MOM_unit_scaling,      only : unit_scale_type
use MOM_variables,         only : BT_cont_type, accel_diag_ptrs
use MOM_verticalGrid,      only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>

public :: btcalc, bt_mass_source, btstep, barotropic_init, barotropic_end
public :: register_barotropic_restarts, set_dtbt, barotropic_get_tav
public :: barotropic_refresh_restart_mirrors   ! addition (see the header)
public :: barotropic_uses_BT_cont_type         ! addition: USE_BT_CONT_TYPE as barotropic_init read it (the reference tells its caller by allocating BT_cont)

type, public :: barotropic_CS ; private
  logical :: module_is_initialized = .false.
  type(c_ptr) :: ctx = c_null_ptr
  type(mom6x_barotropic_params) :: p
  !> Host mirrors of the restart variables (:6279-6296)
  real, allocatable, dimension(:,:) :: ubtav, vbtav
  real :: dtbt = 0.0
  logical :: restarted = .false.     !< the mirrors came from a restart file
  logical :: use_BT_cont_type = .true.   !< USE_BT_CONT_TYPE (:5407)
end type barotropic_CS

integer :: id_clock_sync = -1, id_clock_calc = -1

contains

!> btstep (:455-459).  forces%taux / tauy, the optional-by-association pointers (taux_bot, tauy_bot, uh0 ... v_vh0,
!! eta_PF_start) and etaav keep the reference's presence semantics; BT_cont must be the one the device filled (the host
!! type cannot be handed over), so a call from the host requires it unassociated.
subroutine btstep(U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, forces, pbce, &
                  eta_PF_in, U_Cor, V_Cor, accel_layer_u, accel_layer_v, &
                  eta_out, uhbtav, vhbtav, G, GV, US, CS, &
                  visc_rem_u, visc_rem_v, SpV_avg, ADp, OBC, BT_cont, eta_PF_start, &
                  taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav)
  type(ocean_grid_type),                      intent(inout) :: G
  type(verticalGrid_type),                    intent(in)  :: GV
  type(unit_scale_type),                      intent(in)  :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)  :: U_in
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)  :: V_in
  real, dimension(SZI_(G),SZJ_(G)),           intent(in)  :: eta_in
  real,                                       intent(in)  :: dt
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)  :: bc_accel_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)  :: bc_accel_v
  type(mech_forcing),                         intent(in)  :: forces
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)  :: pbce
  real, dimension(SZI_(G),SZJ_(G)),           intent(in)  :: eta_PF_in
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)  :: U_Cor
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)  :: V_Cor
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(out) :: accel_layer_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(out) :: accel_layer_v
  real, dimension(SZI_(G),SZJ_(G)),           intent(out) :: eta_out
  real, dimension(SZIB_(G),SZJ_(G)),          intent(out) :: uhbtav
  real, dimension(SZI_(G),SZJB_(G)),          intent(out) :: vhbtav
  type(barotropic_CS),                        intent(inout) :: CS
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)  :: visc_rem_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)  :: visc_rem_v
  real, dimension(SZI_(G),SZJ_(G)),           intent(in)  :: SpV_avg
  type(accel_diag_ptrs),                      pointer     :: ADp
  type(ocean_OBC_type),                       pointer     :: OBC
  type(BT_cont_type),                         pointer     :: BT_cont
  real, dimension(:,:),                       pointer     :: eta_PF_start
  real, dimension(:,:),                       pointer     :: taux_bot
  real, dimension(:,:),                       pointer     :: tauy_bot
  real, dimension(:,:,:),                     pointer     :: uh0
  real, dimension(:,:,:),                     pointer     :: u_uh0
  real, dimension(:,:,:),                     pointer     :: vh0
  real, dimension(:,:,:),                     pointer     :: v_vh0
  real, dimension(SZI_(G),SZJ_(G)), optional, intent(out) :: etaav
  type(c_ptr) :: d(20), p_txb, p_tyb, p_uh0, p_vh0, p_uuh0, p_vvh0, p_etaav
  integer(c_int) :: rc
  integer :: nk

  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "btstep: Module MOM_barotropic must be initialized before it is used.")
  if (associated(OBC)) call MOM_error(FATAL, "btstep: open boundaries are not carried by the MI355X path.")
  if (associated(BT_cont)) call MOM_error(FATAL, "btstep (MI355X): a host BT_cont_type cannot be handed to the device; "//&
      "BT_cont is produced and consumed inside step_MOM_dyn_split_RK2.")
  if (associated(eta_PF_start)) call MOM_error(FATAL, "btstep (MI355X): eta_PF_start (DYNAMIC_SURFACE_PRESSURE) is not carried.")
  call cpu_clock_begin(id_clock_calc)
  nk = GV%ke
  d(1) = shim_up3(1, U_in, STG_U, nk) ; d(2) = shim_up3(2, V_in, STG_V, nk) ; d(3) = shim_up2(3, eta_in, STG_H)
  d(4) = shim_up3(4, bc_accel_u, STG_U, nk) ; d(5) = shim_up3(5, bc_accel_v, STG_V, nk)
  d(6) = shim_up2(6, forces%taux, STG_U) ; d(7) = shim_up2(7, forces%tauy, STG_V)
  d(8) = shim_up3(8, pbce, STG_H, nk) ; d(9) = shim_up2(9, eta_PF_in, STG_H)
  d(10) = shim_up3(10, U_Cor, STG_U, nk) ; d(11) = shim_up3(11, V_Cor, STG_V, nk)
  d(12) = shim_out3(12, accel_layer_u, nk) ; d(13) = shim_out3(13, accel_layer_v, nk)
  d(14) = shim_out2(14, eta_out) ; d(15) = shim_out2(15, uhbtav) ; d(16) = shim_out2(16, vhbtav)
  d(17) = shim_up3(17, visc_rem_u, STG_U, nk) ; d(18) = shim_up3(18, visc_rem_v, STG_V, nk)
  p_txb = c_null_ptr ; p_tyb = c_null_ptr ; p_uh0 = c_null_ptr ; p_vh0 = c_null_ptr ; p_uuh0 = c_null_ptr ; p_vvh0 = c_null_ptr
  p_etaav = c_null_ptr
  if (associated(taux_bot) .and. associated(tauy_bot)) then
    p_txb = shim_up2(19, taux_bot, STG_U) ; p_tyb = shim_up2(20, tauy_bot, STG_V)
  endif
  if (associated(uh0)) then   ! (:770-772: all four or none)
    p_uh0 = shim_up3(21, uh0, STG_U, nk) ; p_vh0 = shim_up3(22, vh0, STG_V, nk)
    p_uuh0 = shim_up3(23, u_uh0, STG_U, nk) ; p_vvh0 = shim_up3(24, v_vh0, STG_V, nk)
  endif
  if (present(etaav)) p_etaav = shim_out2(25, etaav)
  rc = mom6x_btstep(CS%ctx, d(1), d(2), d(3), real(dt, c_double), d(4), d(5), d(6), d(7), d(8), d(9), d(10), d(11), &
                    d(12), d(13), d(14), d(15), d(16), d(17), d(18), c_null_ptr, p_txb, p_tyb, p_uh0, p_vh0, p_uuh0, p_vvh0, p_etaav)
  call shim_check(rc, "btstep")
  call shim_down3(accel_layer_u, d(12), STG_U, nk) ; call shim_down3(accel_layer_v, d(13), STG_V, nk)
  call shim_down2(eta_out, d(14), STG_H) ; call shim_down2(uhbtav, d(15), STG_U) ; call shim_down2(vhbtav, d(16), STG_V)
  if (present(etaav)) call shim_down2(etaav, p_etaav, STG_H)
  call btstep_report_warnings(CS)
  call cpu_clock_end(id_clock_calc)
end subroutine btstep

!> The reference's WARNING "btstep: eta has dropped below bathyT" (:2738-2745), counted on the device sub-step by sub-step
subroutine btstep_report_warnings(CS)
  type(barotropic_CS), intent(in) :: CS
  integer(c_long_long) :: cnt ; real(c_double) :: info(4) ; integer(c_int) :: rc
  character(len=200) :: mesg
  rc = mom6x_btstep_warnings(CS%ctx, 1_c_int, cnt, info)
  if (rc == 0 .and. cnt > 0) then
    write(mesg, '("btstep: eta has dropped below bathyT: ",ES12.4," vs. ",ES12.4," at tile point ",2I6," (",I0," times)")') &
        info(1), info(2), int(info(3)), int(info(4)), cnt
    call MOM_error(WARNING, trim(mesg), all_print=.true.)
  endif
end subroutine btstep_report_warnings

!> set_dtbt (:3509).  BT_cont (SET_DTBT_USE_BT_CONT) is not carried.  With pbce the face areas follow :3576-3582: from eta when
!! NONLINEAR_BT_CONTINUITY is set and eta is present, otherwise find_face_areas(add_max=SSH_add).
subroutine set_dtbt(G, GV, US, CS, pbce, gtot_est, BT_cont, eta, SSH_add)
  type(ocean_grid_type),        intent(inout) :: G
  type(verticalGrid_type),      intent(in)    :: GV
  type(unit_scale_type),        intent(in)    :: US
  type(barotropic_CS),          intent(inout) :: CS
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), optional, intent(in) :: pbce
  real,               optional, intent(in)    :: gtot_est
  type(BT_cont_type), optional, pointer       :: BT_cont
  real, dimension(SZI_(G),SZJ_(G)), optional, intent(in) :: eta
  real,               optional, intent(in)    :: SSH_add
  real(c_double) :: dtbt_out, ssh
  type(c_ptr) :: p_eta
  integer(c_int) :: rc
  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "set_dtbt: Module MOM_barotropic must be initialized before it is used.")
  if (present(BT_cont)) then ; if (associated(BT_cont)) call MOM_error(FATAL, &
      "set_dtbt (MI355X): SET_DTBT_USE_BT_CONT is not carried.") ; endif
  ssh = 0.0 ; if (present(SSH_add)) ssh = SSH_add
  if (present(pbce)) then
    p_eta = c_null_ptr
    if (present(eta) .and. CS%p%nonlinear_continuity /= 0) p_eta = shim_up2(2, eta, STG_H)   ! :3578
    rc = mom6x_set_dtbt_pbce_eta(CS%ctx, shim_up3(1, pbce, STG_H, GV%ke), p_eta, ssh, dtbt_out)
  elseif (present(gtot_est)) then
    rc = mom6x_set_dtbt(CS%ctx, c_null_ptr, real(gtot_est, c_double), ssh, dtbt_out)
  else
    call MOM_error(FATAL, "set_dtbt: Either pbce or gtot_est must be present.") ; rc = 0
  endif
  call shim_check(rc, "set_dtbt")
  CS%dtbt = dtbt_out
end subroutine set_dtbt

!> btcalc (:4360): the fractional thicknesses frhatu, frhatv of the barotropic solver, from h (or from h_u, h_v)
subroutine btcalc(h, G, GV, CS, h_u, h_v, may_use_default, OBC)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h
  type(barotropic_CS),     intent(inout) :: CS
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(in) :: h_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(in) :: h_v
  logical,                 optional, intent(in) :: may_use_default
  type(ocean_OBC_type),    optional, pointer    :: OBC
  type(c_ptr) :: p_hu, p_hv
  integer(c_int) :: rc
  logical :: use_default
  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "btcalc: Module MOM_barotropic must be initialized before it is used.")
  if (present(OBC)) then ; if (associated(OBC)) call MOM_error(FATAL, "btcalc: open boundaries are not carried by the MI355X path.") ; endif
  if (present(h_u) .neqv. present(h_v)) call MOM_error(FATAL, "btcalc: Either both h_u and h_v or neither one must be present.")
  p_hu = c_null_ptr ; p_hv = c_null_ptr
  if (present(h_u)) then ; p_hu = shim_up3(2, h_u, STG_U, GV%ke) ; p_hv = shim_up3(3, h_v, STG_V, GV%ke) ; endif
  use_default = .false. ; if (present(may_use_default)) use_default = may_use_default
  if (use_default) then
    rc = mom6x_btcalc(CS%ctx, shim_up3(1, h, STG_H, GV%ke), p_hu, p_hv)
  else   ! :4426-4429: FROM_BT_CONT without h_u, h_v is "Inconsistent settings of optional arguments and hvel_scheme."
    rc = mom6x_btcalc_strict(CS%ctx, shim_up3(1, h, STG_H, GV%ke), p_hu, p_hv)
  endif
  call shim_check(rc, "btcalc")
end subroutine btcalc

!> bt_mass_source (:5243)
subroutine bt_mass_source(h, eta, set_cor, G, GV, CS)
  type(ocean_grid_type),              intent(in) :: G
  type(verticalGrid_type),            intent(in) :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in) :: h
  real, dimension(SZI_(G),SZJ_(G)),   intent(in) :: eta
  logical,                            intent(in) :: set_cor
  type(barotropic_CS),                intent(inout) :: CS
  integer(c_int) :: rc
  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "bt_mass_source: Module MOM_barotropic must be initialized before it is used.")
  rc = mom6x_bt_mass_source(CS%ctx, shim_up3(1, h, STG_H, GV%ke), shim_up2(2, eta, STG_H), merge(1_c_int, 0_c_int, set_cor))
  call shim_check(rc, "bt_mass_source")
end subroutine bt_mass_source

!> barotropic_init (:5301): the parameters of :5403-5713 (SURVEY.md 8(b.1) lists the defaults), the device initialisation,
!! and the restart branch of :5962-5970 / :6124-6135.
subroutine barotropic_init(u, v, h, Time, G, GV, US, param_file, diag, CS, &
                           restart_CS, calc_dtbt, BT_cont, OBC, SAL_CSp, HA_CSp)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h
  type(time_type), target, intent(in)    :: Time
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(barotropic_CS),     intent(inout) :: CS
  type(MOM_restart_CS),    intent(in)    :: restart_CS
  logical,                 intent(out)   :: calc_dtbt
  type(BT_cont_type),      pointer       :: BT_cont
  type(ocean_OBC_type),    pointer       :: OBC
  type(SAL_CS), target,    optional      :: SAL_CSp
  type(harmonic_analysis_CS), target, optional :: HA_CSp
  character(len=40) :: mdl = "MOM_barotropic"
  real :: dtbt_input, dtbt_restart
  character(len=40) :: hvel_str
  integer :: bt_halo_sz, min_stencil
  real(c_double), target :: dtbt_c
  integer(c_int) :: rc

  if (CS%module_is_initialized) then
    call MOM_error(WARNING, "barotropic_init called with a control structure that has already been initialized.")
    return
  endif
  CS%module_is_initialized = .true.
  if (associated(OBC)) call MOM_error(FATAL, "barotropic_init: open boundaries are not carried by the MI355X path.")
  call log_version(param_file, mdl, "mom6x", "")
  call get_param(param_file, mdl, "BEBT", CS%p%bebt, "BEBT determines whether the barotropic time stepping uses the forward-backward "//&
                 "time-stepping scheme or a backward Euler scheme.", units="nondim", default=0.1)
  call get_param(param_file, mdl, "DTBT", dtbt_input, "The barotropic time step, in s; if negative, the fraction of the stability "//&
                 "limit to use.", units="s or nondim", default=-0.98)
  CS%p%dtbt_fraction = 0.98 ; CS%p%dtbt = 0.0
  if (dtbt_input < 0.0) then ; CS%p%dtbt_fraction = -dtbt_input ; else ; CS%p%dtbt = dtbt_input * US%s_to_T ; endif
  call get_param(param_file, mdl, "DT_BT_FILTER", CS%p%dt_bt_filter, "A time-scale over which the barotropic mode solutions are "//&
                 "filtered, in seconds if positive, or as a fraction of DT if negative.", units="sec or nondim", default=-0.25)
  call flag_param("BT_PROJECT_VELOCITY", CS%p%BT_project_velocity, .false.)
  call flag_param("SADOURNY", CS%p%Sadourny, .true.)
  call flag_param("BT_STRONG_DRAG", CS%p%strong_drag, .false.)
  call flag_param("VISC_REM_BT_WEIGHT_BUG", CS%p%wt_uv_bug, .true.)
  call flag_param("BT_USE_OLD_CORIOLIS_BRACKET_BUG", CS%p%use_old_coriolis_bracket_bug, .false.)
  call flag_param("BT_USE_VISC_REM_U_UH0", CS%p%visc_rem_u_uh0, .false.)
  call flag_param("CLIP_BT_VELOCITY", CS%p%clip_velocity, .false.)
  call get_param(param_file, mdl, "CFL_TRUNCATE", CS%p%CFL_trunc, "The CFL above which CLIP_BT_VELOCITY truncates.", &
                 units="nondim", default=0.5)
  call get_param(param_file, mdl, "VEL_UNDERFLOW", CS%p%vel_underflow, "A negligibly small velocity magnitude below which "//&
                 "velocity components are set to 0.", units="m s-1", default=0.0, scale=US%m_s_to_L_T)
  call get_param(param_file, mdl, "G_BT_EXTRA", CS%p%G_extra, "A nondimensional factor by which gtot is enhanced.", &
                 units="nondim", default=0.0)
  call get_param(param_file, mdl, "BT_CORIOLIS_SCALE", CS%p%BT_Coriolis_scale, "A factor by which the barotropic Coriolis "//&
                 "anomaly terms are scaled.", units="nondim", default=1.0)
  call get_param(param_file, mdl, "MAXCFL_BT_CONT", CS%p%maxCFL_BT_cont, "The maximum permitted CFL number associated with the "//&
                 "barotropic accelerations from the summed velocities times the time-derivatives of thicknesses.", &
                 units="nondim", default=0.25)
  call flag_param("BOUND_BT_CORRECTION", CS%p%bound_BT_corr, .false.)
  call flag_param("BT_CONT_CORR_BOUNDS", CS%p%BT_cont_bounds, .true.)
  CS%p%Z_ref = G%Z_ref
  call flag_param("BT_USE_WIDE_HALOS", CS%p%use_wide_halos, .true.)
  call get_param(param_file, mdl, "BTHALO", bt_halo_sz, "The minimum halo size for the barotropic solver.", default=0, layoutParam=.true.)
  CS%p%BTHALO = int(bt_halo_sz, c_int)
  call get_param(param_file, mdl, "BT_WIDE_HALO_MIN_STENCIL", min_stencil, "The minimum stencil width to use with the wide halo "//&
                 "iterations.", default=0, layoutParam=.true.)
  CS%p%min_stencil = int(min_stencil, c_int)
  ! read by btstep when it is called without a BT_cont_type (:5466-5474)
  call flag_param("NONLINEAR_BT_CONTINUITY", CS%p%nonlinear_continuity, .false.)
  call get_param(param_file, mdl, "NONLIN_BT_CONT_UPDATE_PERIOD", min_stencil, "If NONLINEAR_BT_CONTINUITY is true, the number of "//&
                 "barotropic time steps between updates to the face areas, or 0 to update only before the barotropic stepping.", &
                 default=1)
  CS%p%nonlin_cont_update_period = int(min_stencil, c_int)
  ! (the barotropic domain's halo on the device is the tile context's: shim_ctx makes it G's, so BTHALO > NIHALO is refused by
  !  mom6x_barotropic_init with the instruction to widen the context; the answers do not depend on it)
  call get_param(param_file, mdl, "USE_BT_CONT_TYPE", CS%use_BT_cont_type, "If true, use a structure with elements that describe "//&
                 "effective face areas from the summed continuity solver as a function the barotropic flow in coupling between "//&
                 "the barotropic and baroclinic flow.  This is only used if SPLIT is true.", default=.true.)
  ! BT_THICK_SCHEME :5566-5591.  Behind a BT_cont_type the MI355X step takes the thicknesses from BT_cont%h_u, h_v (FROM_BT_CONT,
  ! the default); without one FROM_BT_CONT is the reference's own FATAL and the three interpolations are carried.
  call get_param(param_file, mdl, "BT_THICK_SCHEME", hvel_str, "A string describing the scheme that is used to set the open face "//&
                 "areas used for barotropic transport and the relative weights of the accelerations: ARITHMETIC, HARMONIC, "//&
                 "HYBRID or FROM_BT_CONT.", default="FROM_BT_CONT")
  select case (trim(hvel_str))
    case ("HYBRID") ; CS%p%bt_thick_scheme = 1_c_int
    case ("HARMONIC") ; CS%p%bt_thick_scheme = 2_c_int
    case ("ARITHMETIC") ; CS%p%bt_thick_scheme = 3_c_int
    case ("FROM_BT_CONT") ; CS%p%bt_thick_scheme = 0_c_int
    case default
      call MOM_error(FATAL, "barotropic_init: Unrecognized setting #define BT_THICK_SCHEME "//trim(hvel_str)//" found in input file.")
  end select
  if ((CS%p%bt_thick_scheme == 0) .and. .not.CS%use_BT_cont_type) call MOM_error(FATAL, &
      "barotropic_init: BT_THICK_SCHEME FROM_BT_CONT can only be used if USE_BT_CONT_TYPE is defined.")
  if ((CS%p%bt_thick_scheme /= 0) .and. CS%use_BT_cont_type) call MOM_error(FATAL, &
      "barotropic_init: with USE_BT_CONT_TYPE the MI355X path carries BT_THICK_SCHEME = FROM_BT_CONT only.")
  call get_param(param_file, mdl, "MAXVEL", CS%p%maxvel, "The maximum velocity allowed before the velocity components are "//&
                 "truncated.", units="m s-1", default=3.0e8, scale=US%m_s_to_L_T)
  call must_be("INTEGRAL_BT_CONTINUITY", .false.)
  call must_be("ADJUST_BT_CONT", .false.) ; call must_be("GRADUAL_BT_ICS", .false.)
  call must_be("BT_NONLIN_STRESS", .false.) ; call must_be("DYNAMIC_SURFACE_PRESSURE", .false.)
  call must_be("BT_LINEAR_WAVE_DRAG", .false.) ; call must_be("LINEARIZED_BT_CORIOLIS", .true.)
  call must_be("TIDES", .false.) ; call must_be("CALCULATE_SAL", .false.) ; call must_be("USE_FILTER", .false.)

  call shim_set_domain_flags(param_file)
  CS%ctx = shim_ctx(G, GV)
  rc = mom6x_barotropic_init(CS%ctx, CS%p) ; call shim_check(rc, "barotropic_init")
  if (.not.allocated(CS%ubtav)) call MOM_error(FATAL, "barotropic_init: register_barotropic_restarts must be called first.")

  ! :5945-5970.  The device initialisation has made the initial estimate of set_dtbt (gtot_est, SSH_EXTRA) and taken a
  ! positive DTBT as given; a restart file's DTBT replaces the estimate when DTBT is a fraction; the first step need not
  ! recompute it only when both a file value and a fixed DTBT exist (the reference's own rule, :5970).
  CS%restarted = query_initialized(CS%ubtav, "ubtav", restart_CS) .and. query_initialized(CS%vbtav, "vbtav", restart_CS)
  dtbt_restart = -1.0
  if (query_initialized(CS%dtbt, "DTBT", restart_CS)) dtbt_restart = CS%dtbt
  if (dtbt_input <= 0.0 .and. dtbt_restart > 0.0) then
    dtbt_c = dtbt_restart
    rc = mom6x_barotropic_dtbt(CS%ctx, c_null_ptr, c_loc(dtbt_c)) ; call shim_check(rc, "barotropic_init (DTBT)")
  endif
  calc_dtbt = .true. ; if ((dtbt_restart > 0.0) .and. (dtbt_input > 0.0)) calc_dtbt = .false.
  ! :6124-6135: ubtav, vbtav from the file, else they are formed from u, v by the new-run initialisation of the dynamics
  if (CS%restarted) then
    rc = mom6x_upload(CS%ctx, mom6x_barotropic_field(CS%ctx, 0_c_int), CS%ubtav, STG_U, 1_c_int) ; call shim_check(rc, "barotropic_init (ubtav)")
    rc = mom6x_upload(CS%ctx, mom6x_barotropic_field(CS%ctx, 1_c_int), CS%vbtav, STG_V, 1_c_int) ; call shim_check(rc, "barotropic_init (vbtav)")
  endif
  id_clock_calc = cpu_clock_id('(Ocean BT calcs only)', grain=CLOCK_ROUTINE)
  id_clock_sync = cpu_clock_id('(Ocean BT global synch)', grain=CLOCK_ROUTINE)

contains
  subroutine flag_param(name, flag, default)
    character(len=*), intent(in) :: name ; integer(c_int), intent(out) :: flag ; logical, intent(in) :: default
    logical :: val
    call get_param(param_file, mdl, name, val, default=default)
    flag = merge(1_c_int, 0_c_int, val)
  end subroutine flag_param
  subroutine must_be(name, default)
    character(len=*), intent(in) :: name ; logical, intent(in) :: default
    logical :: val
    call get_param(param_file, mdl, name, val, default=default, do_not_log=.true.)
    if (val .neqv. default) call MOM_error(FATAL, "barotropic_init: "//trim(name)//" is not carried by the MI355X path.")
  end subroutine must_be
end subroutine barotropic_init

!> USE_BT_CONT_TYPE as barotropic_init read it: step_MOM_dyn_split_RK2 runs without a BT_cont_type when it is false
logical function barotropic_uses_BT_cont_type(CS)
  type(barotropic_CS), intent(in) :: CS
  barotropic_uses_BT_cont_type = CS%use_BT_cont_type
end function barotropic_uses_BT_cont_type

!> barotropic_get_tav (:6193): the time-mean barotropic velocities (MOM_hor_visc uses them with its bounds on the viscosity)
subroutine barotropic_get_tav(CS, ubtav, vbtav, G, US)
  type(barotropic_CS),               intent(in)    :: CS
  type(ocean_grid_type),             intent(in)    :: G
  real, dimension(SZIB_(G),SZJ_(G)), intent(inout) :: ubtav
  real, dimension(SZI_(G),SZJB_(G)), intent(inout) :: vbtav
  type(unit_scale_type),             intent(in)    :: US
  call shim_down2(ubtav, mom6x_barotropic_field(CS%ctx, 0_c_int), STG_U)
  call shim_down2(vbtav, mom6x_barotropic_field(CS%ctx, 1_c_int), STG_V)
end subroutine barotropic_get_tav

!> Device -> the registered host mirrors (ubtav, vbtav, DTBT): to be current whenever the host may write a restart file.
subroutine barotropic_refresh_restart_mirrors(CS)
  type(barotropic_CS), intent(inout) :: CS
  real(c_double), target :: dtbt_c
  integer(c_int) :: rc
  if (.not.CS%module_is_initialized) return
  call shim_down2(CS%ubtav, mom6x_barotropic_field(CS%ctx, 0_c_int), STG_U)
  call shim_down2(CS%vbtav, mom6x_barotropic_field(CS%ctx, 1_c_int), STG_V)
  rc = mom6x_barotropic_dtbt(CS%ctx, c_loc(dtbt_c), c_null_ptr) ; call shim_check(rc, "barotropic_refresh_restart_mirrors")
  CS%dtbt = dtbt_c
end subroutine barotropic_refresh_restart_mirrors

!> barotropic_end (:6216)
subroutine barotropic_end(CS)
  type(barotropic_CS), intent(inout) :: CS
  if (allocated(CS%ubtav)) deallocate(CS%ubtav)
  if (allocated(CS%vbtav)) deallocate(CS%vbtav)
  CS%module_is_initialized = .false. ; CS%ctx = c_null_ptr
end subroutine barotropic_end

!> register_barotropic_restarts (:6253): ubtav, vbtav and DTBT under the reference's names, units and descriptions, backed
!! by the host mirrors.  (ubt_IC, vbt_IC belong to GRADUAL_BT_ICS, which barotropic_init refuses.)
subroutine register_barotropic_restarts(HI, GV, US, param_file, CS, restart_CS)
  type(hor_index_type),    intent(in) :: HI
  type(verticalGrid_type), intent(in) :: GV
  type(unit_scale_type),   intent(in) :: US
  type(param_file_type),   intent(in) :: param_file
  type(barotropic_CS),     intent(inout) :: CS
  type(MOM_restart_CS),    intent(inout) :: restart_CS
  type(vardesc) :: vd(2)
  allocate(CS%ubtav(HI%IsdB:HI%IedB,HI%jsd:HI%jed), source=0.0)
  allocate(CS%vbtav(HI%isd:HI%ied,HI%JsdB:HI%JedB), source=0.0)
  vd(1) = var_desc("ubtav", "m s-1", "Time mean barotropic zonal velocity", hor_grid='u', z_grid='1')
  vd(2) = var_desc("vbtav", "m s-1", "Time mean barotropic meridional velocity", hor_grid='v', z_grid='1')
  call register_restart_pair(CS%ubtav, CS%vbtav, vd(1), vd(2), .false., restart_CS, conversion=US%L_T_to_m_s)
  call register_restart_field(CS%dtbt, "DTBT", .false., restart_CS, longname="Barotropic timestep", units="seconds", &
                              conversion=US%T_to_s)
end subroutine register_barotropic_restarts

end module MOM_barotropic
