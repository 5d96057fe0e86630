!> Drop-in for the entry points of src/parameterizations/lateral/MOM_hor_visc.F90: horizontal_viscosity :266, hor_visc_init :2322,
!! hor_visc_end :3515, hor_visc_vel_stencil :3305 and the type hor_visc_CS -- same names and argument lists, served by
!! mom6x_hor_visc_init / mom6x_horizontal_viscosity (SURVEY 8f-2).  The split RK2 step on the device calls the device routine
!! itself on resident arrays; this module is for a host that calls horizontal_viscosity on its own arrays (an unsplit or RK3
!! dynamical core, or the diagnostics of MOM.F90) and for initialize_dyn_split_RK2, which calls hor_visc_init as the reference does.
!! Carried: LAPLACIAN and / or BIHARMONIC with KH / KH_VEL_SCALE / KH_BG_MIN / AH / AH_VEL_SCALE / AH_TIME_SCALE, SMAGORINSKY_KH / _AH,
!! LEITH_KH / _AH (USE_BETA_IN_LEITH, MODIFIED_LEITH), BOUND_KH / BOUND_AH in both forms, BOUND_CORIOLIS_BIHARM, ADD_LES_VISCOSITY,
!! USE_LAND_MASK_FOR_HVISC, NOSLIP, BACKSCATTER_UNDERBOUND.  Refused with the reference's parameter name: USE_LEITHY,
!! USE_QG_LEITH_VISC, USE_MEKE, USE_GME, ANISOTROPIC_VISCOSITY, USE_ZB2020, USE_KH_BG_2D, USE_CONT_THICKNESS, KH_SIN_LAT; open
!! boundaries; the FrictWork diagnostics.
module MOM_hor_visc
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_barotropic,            only : barotropic_CS
use MOM_diag_mediator,         only : diag_ctrl
use MOM_error_handler,         only : MOM_error, FATAL, WARNING
use MOM_file_parser,           only : get_param, log_version, param_file_type
use MOM_grid,                  only : ocean_grid_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_MEKE_types,            only : MEKE_type
use MOM_open_boundary,         only : ocean_OBC_type
use MOM_stochastics,           only : stochastic_CS
use MOM_thickness_diffuse,     only : thickness_diffuse_CS
use MOM_time_manager,          only : time_type
use MOM_unit_scaling,          only : unit_scale_type
use MOM_variables,             only : accel_diag_ptrs, thermo_var_ptrs
use MOM_verticalGrid,          only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>
public :: horizontal_viscosity, hor_visc_init, hor_visc_end, hor_visc_vel_stencil

type, public :: hor_visc_CS ; private
  logical :: initialized = .false.
  type(c_ptr) :: ctx = c_null_ptr
  type(mom6x_hor_visc_params) :: p
end type hor_visc_CS

contains

!> horizontal_viscosity (:266).  h is intent(inout) in the reference for its halo updates under USE_CONT_THICKNESS only; uh, vh are
!! read by the GME / ZB2020 branches alone.  Needs u, v on (is-2:ie+2, js-2:je+2) and h on (is-1:ie+1, js-1:je+1) as the reference.
subroutine horizontal_viscosity(u, v, h, uh, vh, diffu, diffv, MEKE, VarMix, G, GV, US, &
                                CS, tv, dt, OBC, BT, TD, ADp, hu_cont, hv_cont, STOCH)
  type(ocean_grid_type),                      intent(in)    :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(inout) :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: vh
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(out)   :: diffu
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(out)   :: diffv
  type(MEKE_type),                            intent(inout) :: MEKE
  type(VarMix_CS),                            intent(inout) :: VarMix
  type(unit_scale_type),                      intent(in)    :: US
  type(hor_visc_CS),                          intent(inout) :: CS
  type(thermo_var_ptrs),                      intent(in)    :: tv
  real,                                       intent(in)    :: dt
  type(ocean_OBC_type),             optional, pointer       :: OBC
  type(barotropic_CS),              optional, intent(in)    :: BT
  type(thickness_diffuse_CS),       optional, intent(in)    :: TD
  type(accel_diag_ptrs),            optional, intent(in)    :: ADp
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(inout) :: hu_cont
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(inout) :: hv_cont
  type(stochastic_CS),              optional, intent(inout) :: STOCH
  type(c_ptr) :: d_diffu, d_diffv
  integer(c_int) :: rc
  integer :: nk
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_hor_visc: Module must be initialized before it is used.")
  if (present(OBC)) then ; if (associated(OBC)) &
    call MOM_error(FATAL, "horizontal_viscosity: open boundaries are not carried by the MI355X path.") ; endif
  if (present(hu_cont) .or. present(hv_cont)) &
    call MOM_error(FATAL, "horizontal_viscosity: USE_CONT_THICKNESS is not carried by the MI355X path.")
  if (.not.(CS%p%Laplacian /= 0 .or. CS%p%biharmonic /= 0)) return      ! :1947 (the accelerations stay what they were)
  nk = GV%ke
  d_diffu = shim_out3(4, diffu, nk) ; d_diffv = shim_out3(5, diffv, nk)
  rc = mom6x_horizontal_viscosity(CS%ctx, shim_up3(1, u, STG_U, nk), shim_up3(2, v, STG_V, nk), shim_up3(3, h, STG_H, nk), d_diffu, d_diffv)
  call shim_check(rc, "horizontal_viscosity")
  call shim_down3(diffu, d_diffu, STG_U, nk) ; call shim_down3(diffv, d_diffv, STG_V, nk)
end subroutine horizontal_viscosity

!> hor_visc_init (:2322): every member of mom6x_hor_visc_params with the reference's parameter names, defaults and order of
!! dependence (:2403-2720); the static metric planes (:2740-3290) are formed on the device by mom6x_hor_visc_init.
subroutine hor_visc_init(Time, G, GV, US, param_file, diag, CS, ADp)
  type(time_type),         intent(in)    :: Time
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(hor_visc_CS),       intent(inout) :: CS
  type(accel_diag_ptrs), intent(in), optional :: ADp
  character(len=40) :: mdl = "MOM_hor_visc"
  real :: dt
  integer(c_int) :: rc
  CS%initialized = .true.
  call log_version(param_file, mdl, "mom6x", "")
  dt = 0.0
  call read_hor_visc_params(param_file, G, US, dt, CS%p)      ! (p%dt = 0 so far)
  if (CS%p%Laplacian /= 0 .or. CS%p%biharmonic /= 0) then     ! :2713-2718
    call get_param(param_file, mdl, "DT", dt, "The (baroclinic) dynamics time step.", units="s", scale=US%s_to_T, fail_if_missing=.true.)
    CS%p%dt = dt
  endif
  call shim_set_domain_flags(param_file)
  CS%ctx = shim_ctx(G, GV)
  rc = mom6x_hor_visc_init(CS%ctx, CS%p) ; call shim_check(rc, "hor_visc_init")
end subroutine hor_visc_init

!> hor_visc_vel_stencil (:3305)
function hor_visc_vel_stencil(CS) result(stencil)
  type(hor_visc_CS), intent(in) :: CS
  integer :: stencil
  stencil = 2
  if ((CS%p%Leith_Kh /= 0) .or. (CS%p%Leith_Ah /= 0)) stencil = 3
end function hor_visc_vel_stencil

!> hor_visc_end (:3515)
subroutine hor_visc_end(CS)
  type(hor_visc_CS), intent(inout) :: CS
  CS%initialized = .false. ; CS%ctx = c_null_ptr
end subroutine hor_visc_end

subroutine gp_flag(pf, mdl, name, flag, default, desc)   ! a logical parameter as the C int of the struct
  type(param_file_type), intent(in) :: pf ; character(len=*), intent(in) :: mdl, name
  integer(c_int), intent(out) :: flag ; logical, intent(in) :: default ; character(len=*), optional, intent(in) :: desc
  logical :: val
  if (present(desc)) then ; call get_param(pf, mdl, name, val, desc, default=default)
  else ; call get_param(pf, mdl, name, val, default=default) ; endif
  flag = merge(1_c_int, 0_c_int, val)
end subroutine gp_flag

subroutine reject(pf, mdl, name, default)          ! a switch that must keep its default on the device path
  type(param_file_type), intent(in) :: pf ; character(len=*), intent(in) :: mdl, name ; logical, intent(in) :: default
  logical :: val
  call get_param(pf, mdl, name, val, default=default, do_not_log=.true.)
  if (val .neqv. default) call MOM_error(FATAL, trim(mdl)//": "//trim(name)//" is not carried by the MI355X path.")
end subroutine reject

!> hor_visc_init, MOM_hor_visc.F90:2403-2720: every member of mom6x_hor_visc_params (the struct is zeroed first)
subroutine read_hor_visc_params(pf, G, US, dt, p)
  type(param_file_type), intent(in) :: pf ; type(ocean_grid_type), intent(in) :: G ; type(unit_scale_type), intent(in) :: US
  real, intent(in) :: dt ; type(mom6x_hor_visc_params), intent(out) :: p
  character(len=40) :: mdl = "MOM_hor_visc"
  logical :: bound_Cor_def
  real :: maxvel
  call gp_flag(pf, mdl, "LAPLACIAN", p%Laplacian, .false.) ; call gp_flag(pf, mdl, "BIHARMONIC", p%biharmonic, .true.)
  call get_param(pf, mdl, "KH", p%Kh, units="m2 s-1", default=0.0, scale=US%m_to_L**2*US%T_to_s)
  call get_param(pf, mdl, "KH_BG_MIN", p%Kh_bg_min, units="m2 s-1", default=0.0, scale=US%m_to_L**2*US%T_to_s)
  call get_param(pf, mdl, "KH_VEL_SCALE", p%Kh_vel_scale, units="m s-1", default=0.0, scale=US%m_s_to_L_T)
  call gp_flag(pf, mdl, "SMAGORINSKY_KH", p%Smagorinsky_Kh, .false.)
  call get_param(pf, mdl, "SMAG_LAP_CONST", p%Smag_Lap_const, units="nondim", default=0.0)
  call gp_flag(pf, mdl, "LEITH_KH", p%Leith_Kh, .false.)
  call get_param(pf, mdl, "LEITH_LAP_CONST", p%Leith_Lap_const, units="nondim", default=0.0)
  call gp_flag(pf, mdl, "BOUND_KH", p%bound_Kh, .true.)
  call gp_flag(pf, mdl, "BETTER_BOUND_KH", p%better_bound_Kh, (p%bound_Kh /= 0))
  call gp_flag(pf, mdl, "ADD_LES_VISCOSITY", p%add_LES_viscosity, .false.)
  call get_param(pf, mdl, "AH", p%Ah, units="m4 s-1", default=0.0, scale=US%m_to_L**4*US%T_to_s)
  call get_param(pf, mdl, "AH_VEL_SCALE", p%Ah_vel_scale, units="m s-1", default=0.0, scale=US%m_s_to_L_T)
  call get_param(pf, mdl, "AH_TIME_SCALE", p%Ah_time_scale, units="s", default=0.0, scale=US%s_to_T)
  call gp_flag(pf, mdl, "SMAGORINSKY_AH", p%Smagorinsky_Ah, .false.)
  call get_param(pf, mdl, "SMAG_BI_CONST", p%Smag_bi_const, units="nondim", default=0.0)
  call gp_flag(pf, mdl, "LEITH_AH", p%Leith_Ah, .false.)
  call get_param(pf, mdl, "LEITH_BI_CONST", p%Leith_bi_const, units="nondim", default=0.0)
  call gp_flag(pf, mdl, "USE_BETA_IN_LEITH", p%use_beta_in_Leith, .false.)
  call gp_flag(pf, mdl, "MODIFIED_LEITH", p%modified_Leith, .false.)
  call gp_flag(pf, mdl, "BOUND_AH", p%bound_Ah, .true.)
  call gp_flag(pf, mdl, "BETTER_BOUND_AH", p%better_bound_Ah, (p%bound_Ah /= 0))
  call get_param(pf, mdl, "BOUND_CORIOLIS", bound_Cor_def, default=.false.)
  call gp_flag(pf, mdl, "BOUND_CORIOLIS_BIHARM", p%bound_Coriolis, bound_Cor_def)
  call get_param(pf, mdl, "MAXVEL", maxvel, units="m s-1", default=3.0e8)
  call get_param(pf, mdl, "BOUND_CORIOLIS_VEL", p%bound_Cor_vel, units="m s-1", default=maxvel, scale=US%m_s_to_L_T)
  call gp_flag(pf, mdl, "USE_LAND_MASK_FOR_HVISC", p%use_land_mask, .true.)
  call get_param(pf, mdl, "HORVISC_BOUND_COEF", p%bound_coef, units="nondim", default=0.8)
  call gp_flag(pf, mdl, "NOSLIP", p%no_slip, .false.)
  call gp_flag(pf, mdl, "BACKSCATTER_UNDERBOUND", p%backscatter_underbound, .true.)
  p%dt = dt
  call reject(pf, mdl, "USE_LEITHY", .false.) ; call reject(pf, mdl, "USE_QG_LEITH_VISC", .false.) ; call reject(pf, mdl, "USE_MEKE", .false.)
  call reject(pf, mdl, "USE_GME", .false.) ; call reject(pf, mdl, "ANISOTROPIC_VISCOSITY", .false.) ; call reject(pf, mdl, "USE_ZB2020", .false.)
  call reject(pf, mdl, "USE_KH_BG_2D", .false.) ; call reject(pf, mdl, "USE_CONT_THICKNESS", .false.) ; call reject(pf, mdl, "KH_SIN_LAT", .false.)
end subroutine read_hor_visc_params

end module MOM_hor_visc
