!> Drop-in for src/core/MOM_CoriolisAdv.F90: CorAdCalc :125, CoriolisAdv_init :1054, CoriolisAdv_end :1322 and the type
!! CoriolisAdv_CS -- same names and argument lists, served by mom6x_CoriolisAdv_init / mom6x_CorAdCalc.  Host arrays in,
!! host arrays out (the device step itself calls the device routine on resident arrays).
module MOM_CoriolisAdv
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_diag_mediator,   only : diag_ctrl
use MOM_error_handler,   only : MOM_error, FATAL, WARNING
use MOM_file_parser,     only : get_param, log_version, param_file_type
use MOM_grid,            only : ocean_grid_type
use MOM_open_boundary,   only : ocean_OBC_type
use MOM_porous_barriers, only : porous_barrier_type
use MOM_time_manager,    only : time_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_variables,       only : accel_diag_ptrs
use MOM_verticalGrid,    only : verticalGrid_type
use MOM_wave_interface,  only : wave_parameters_CS
implicit none ; private
#include <MOM_memory.h>
public :: CorAdCalc, CoriolisAdv_init, CoriolisAdv_end

type, public :: CoriolisAdv_CS ; private
  logical :: initialized = .false.
  type(c_ptr) :: ctx = c_null_ptr
  type(mom6x_coriolis_params) :: p
end type CoriolisAdv_CS

contains

!> CorAdCalc (:125)
subroutine CorAdCalc(u, v, h, uh, vh, CAu, CAv, OBC, AD, G, GV, US, CS, pbv, Waves)
  type(ocean_grid_type),                      intent(in)    :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: vh
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(out)   :: CAu
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(out)   :: CAv
  type(ocean_OBC_type),                       pointer       :: OBC
  type(accel_diag_ptrs),                      intent(inout) :: AD
  type(unit_scale_type),                      intent(in)    :: US
  type(CoriolisAdv_CS),                       intent(in)    :: CS
  type(porous_barrier_type),                  intent(in)    :: pbv
  type(wave_parameters_CS),         optional, pointer       :: Waves
  type(c_ptr) :: d_CAu, d_CAv
  integer(c_int) :: rc
  integer :: nk
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_CoriolisAdv: Module must be initialized before it is used.")
  if (associated(OBC)) call MOM_error(FATAL, "CorAdCalc: open boundaries are not carried by the MI355X path.")
  if (present(Waves)) then ; if (associated(Waves)) call MOM_error(FATAL, "CorAdCalc: Stokes drift is not carried by the MI355X path.") ; endif
  nk = GV%ke
  d_CAu = shim_out3(6, CAu, nk) ; d_CAv = shim_out3(7, CAv, nk)   ! (the resident copies of CAu, CAv if the host has handed them over)
  rc = mom6x_CorAdCalc(CS%ctx, shim_up3(1, u, STG_U, nk), shim_up3(2, v, STG_V, nk), shim_up3(3, h, STG_H, nk), &
                       shim_up3(4, uh, STG_U, nk), shim_up3(5, vh, STG_V, nk), d_CAu, d_CAv)
  call shim_check(rc, "CorAdCalc")
  call shim_down3(CAu, d_CAu, STG_U, nk) ; call shim_down3(CAv, d_CAv, STG_V, nk)
end subroutine CorAdCalc

!> CoriolisAdv_init (:1054): CORIOLIS_SCHEME, KE_SCHEME, BOUND_CORIOLIS, NOSLIP, CORIOLIS_EN_DIS (:1090-1200)
subroutine CoriolisAdv_init(Time, G, GV, US, param_file, diag, AD, CS)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(accel_diag_ptrs),   target, intent(inout) :: AD
  type(CoriolisAdv_CS),    intent(inout) :: CS
  character(len=40) :: mdl = "MOM_CoriolisAdv"
  character(len=40) :: tmpstr
  logical :: flag
  integer(c_int) :: rc
  CS%initialized = .true.
  call log_version(param_file, mdl, "mom6x", "")
  call get_param(param_file, mdl, "NOSLIP", flag, "If true, no slip boundary conditions are used; otherwise free slip "//&
                 "boundary conditions are assumed.", default=.false.)
  CS%p%no_slip = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "CORIOLIS_EN_DIS", flag, "If true, two estimates of the thickness fluxes are used to "//&
                 "estimate the Coriolis term, and the one that dissipates energy relative to the other one is used.", default=.false.)
  CS%p%Coriolis_En_Dis = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "CORIOLIS_SCHEME", tmpstr, "CORIOLIS_SCHEME selects the discretization for the Coriolis terms.", &
                 default="SADOURNY75_ENERGY")
  select case (trim(tmpstr))      ! the integer codes of the reference module (:103-116)
    case ("SADOURNY75_ENERGY") ; CS%p%Coriolis_Scheme = 1
    case ("ARAKAWA_HSU90")     ; CS%p%Coriolis_Scheme = 2
    case ("SADOURNY75_ENSTRO") ; CS%p%Coriolis_Scheme = 4
    case ("ARAKAWA_LAMB81")    ; CS%p%Coriolis_Scheme = 5
    case ("ARAKAWA_LAMB_BLEND") ; CS%p%Coriolis_Scheme = 6
    case ("ROBUST_ENSTRO")     ; CS%p%Coriolis_Scheme = 3
    case default ; call MOM_error(FATAL, "CoriolisAdv_init: Unrecognized setting #define CORIOLIS_SCHEME "//trim(tmpstr)//" found in input file.")
  end select
  CS%p%F_eff_max_blend = 4.0 ; CS%p%wt_lin_blend = 0.125
  if (CS%p%Coriolis_Scheme == 6) then   ! :1125-1142
    call get_param(param_file, mdl, "CORIOLIS_BLEND_WT_LIN", CS%p%wt_lin_blend, "A weighting value for the ratio of inverse "//&
                 "thicknesses, beyond which the blending between Sadourny Energy and Arakawa & Hsu goes linearly to 0 when "//&
                 "CORIOLIS_SCHEME is ARAWAKA_LAMB_BLEND. This must be between 1 and 1e-16.", units="nondim", default=0.125)
    call get_param(param_file, mdl, "CORIOLIS_BLEND_F_EFF_MAX", CS%p%F_eff_max_blend, "The factor by which the maximum "//&
                 "effective Coriolis acceleration from any point can be increased when blending different discretizations "//&
                 "with the ARAKAWA_LAMB_BLEND Coriolis scheme.  This must be greater than 2.0 (the max value for Sadourny "//&
                 "energy).", units="nondim", default=4.0)
    CS%p%wt_lin_blend = min(1.0, max(CS%p%wt_lin_blend, 1e-16))
    if (CS%p%F_eff_max_blend < 2.0) call MOM_error(WARNING, "CoriolisAdv_init: CORIOLIS_BLEND_F_EFF_MAX should be at least 2.")
  endif
  call get_param(param_file, mdl, "BOUND_CORIOLIS", flag, "If true, the Coriolis terms at u-points are bounded by the four "//&
                 "estimates of (f+rv)v from the four neighboring v-points, and similarly at v-points.", default=.false.)
  CS%p%bound_Coriolis = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "KE_SCHEME", tmpstr, "KE_SCHEME selects the discretization for acceleration due to the "//&
                 "kinetic energy gradient.", default="KE_ARAKAWA")
  select case (trim(tmpstr))
    case ("KE_ARAKAWA") ; CS%p%KE_Scheme = 10
    case ("KE_SIMPLE_GUDONOV") ; CS%p%KE_Scheme = 11
    case ("KE_GUDONOV") ; CS%p%KE_Scheme = 12
    case default ; call MOM_error(FATAL, "CoriolisAdv_init: #define KE_SCHEME "//trim(tmpstr)//" in input file is invalid.")
  end select
  call get_param(param_file, mdl, "PV_ADV_SCHEME", tmpstr, "PV_ADV_SCHEME selects the discretization for PV advection.", &
                 default="PV_ADV_CENTERED")
  select case (trim(tmpstr))      ! :1186-1192
    case ("PV_ADV_CENTERED") ; CS%p%PV_Adv_Scheme = 21
    case ("PV_ADV_UPWIND1") ; CS%p%PV_Adv_Scheme = 22
    case default ; call MOM_error(FATAL, "CoriolisAdv_init: #DEFINE PV_ADV_SCHEME in input file is invalid.")
  end select
  call shim_set_domain_flags(param_file)
  CS%ctx = shim_ctx(G, GV)
  rc = mom6x_CoriolisAdv_init(CS%ctx, CS%p) ; call shim_check(rc, "CoriolisAdv_init")
end subroutine CoriolisAdv_init

!> CoriolisAdv_end (:1322)
subroutine CoriolisAdv_end(CS)
  type(CoriolisAdv_CS), intent(inout) :: CS
  CS%initialized = .false. ; CS%ctx = c_null_ptr
end subroutine CoriolisAdv_end

end module MOM_CoriolisAdv
