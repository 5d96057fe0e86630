!> Drop-in for the entry points of src/core/MOM_continuity_PPM.F90 that the split dynamical core and MOM_continuity.F90's
!! first four pass-throughs bind (MOM_continuity.F90:6-9): continuity_PPM :86-87, continuity_PPM_init :2674,
!! continuity_PPM_stencil :2757 and the type continuity_PPM_CS -- same module name, procedure names and argument lists,
!! optional-argument semantics included (presence changes behaviour: :590-592, :637, :737, :756), served by
!! mom6x_continuity_init / mom6x_continuity_PPM.
!!
!! Inside the device step (MOM_dynamics_split_RK2) this boundary is never crossed: the step calls the device routine
!! directly on resident arrays.  It exists for the callers outside it (MOM_dynamics_unsplit*, initialisation code), which
!! hand over HOST arrays: each call uploads its inputs and downloads its outputs (PCIe-bound; DESIGN.md section 4).
!! The other public names of the reference module (continuity_fluxes, continuity_adjust_vel, zonal_mass_flux, ... used by
!! thickness diffusion, OBC and the offline code) stay host Fortran: INTEGRATION.md section 3 says how a build keeps them.
module MOM_continuity_PPM
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_cpu_clock,       only : cpu_clock_id, cpu_clock_begin, cpu_clock_end, CLOCK_ROUTINE
use MOM_diag_mediator,   only : diag_ctrl
use MOM_error_handler,   only : MOM_error, FATAL
use MOM_file_parser,     only : get_param, log_version, param_file_type
use MOM_grid,            only : ocean_grid_type
use MOM_open_boundary,   only : ocean_OBC_type
use MOM_porous_barriers, only : porous_barrier_type
use MOM_time_manager,    only : time_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_variables,       only : BT_cont_type
use MOM_verticalGrid,    only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>
public :: continuity_PPM, continuity_PPM_init, continuity_PPM_stencil, continuity_PPM_CS

!> The control structure: the parameters continuity_PPM_init read (kept for continuity_PPM_stencil) and the tile's context
type :: continuity_PPM_CS ; private
  logical :: initialized = .false.
  type(c_ptr) :: ctx = c_null_ptr
  type(mom6x_continuity_params) :: p
end type continuity_PPM_CS

integer :: id_clock_update = -1

contains

!> continuity_PPM (:86-87)
subroutine continuity_PPM(u, v, hin, h, uh, vh, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, &
                          visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont, du_cor, dv_cor)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: hin
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(inout) :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(out)   :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(out)   :: vh
  real,                    intent(in)    :: dt
  type(unit_scale_type),   intent(in)    :: US
  type(continuity_PPM_CS), intent(in)    :: CS
  type(ocean_OBC_type),    pointer       :: OBC
  type(porous_barrier_type), intent(in)  :: pbv
  real, dimension(SZIB_(G),SZJ_(G)), optional, intent(in)    :: uhbt
  real, dimension(SZI_(G),SZJB_(G)), optional, intent(in)    :: vhbt
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(in)  :: visc_rem_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(in)  :: visc_rem_v
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(out) :: u_cor
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(out) :: v_cor
  type(BT_cont_type),      optional, pointer     :: BT_cont
  real, dimension(SZIB_(G),SZJ_(G)), optional, intent(out)   :: du_cor
  real, dimension(SZI_(G),SZJB_(G)), optional, intent(out)   :: dv_cor
  integer(c_int) :: rc
  integer :: nk
  type(c_ptr) :: d_u, d_v, d_hin, d_h, d_uh, d_vh, p_uhbt, p_vhbt, p_vru, p_vrv, p_ucor, p_vcor, p_ducor, p_dvcor

  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_continuity_PPM: Module must be initialized before it is used.")
  if (associated(OBC)) call MOM_error(FATAL, "continuity_PPM: open boundaries are not carried by the MI355X path.")
  if (present(visc_rem_u) .neqv. present(visc_rem_v)) call MOM_error(FATAL, "MOM_continuity_PPM: Either both "//&
      "visc_rem_u and visc_rem_v or neither one must be present in call to continuity_PPM.")
  if (present(BT_cont)) then ; if (associated(BT_cont)) call MOM_error(FATAL, &
      "continuity_PPM (MI355X): BT_cont is filled on the device inside step_MOM_dyn_split_RK2 only.") ; endif
  call cpu_clock_begin(id_clock_update)
  nk = GV%ke
  d_u = shim_up3(1, u, STG_U, nk) ; d_v = shim_up3(2, v, STG_V, nk) ; d_hin = shim_up3(3, hin, STG_H, nk)
  d_h = shim_up3(4, h, STG_H, nk)     ! intent(inout): cells the routine does not write keep the caller's values
  d_uh = shim_buf(5, nk) ; d_vh = shim_buf(6, nk)
  p_uhbt = c_null_ptr ; p_vhbt = c_null_ptr ; p_vru = c_null_ptr ; p_vrv = c_null_ptr
  p_ucor = c_null_ptr ; p_vcor = c_null_ptr ; p_ducor = c_null_ptr ; p_dvcor = c_null_ptr
  if (present(uhbt)) p_uhbt = shim_up2(7, uhbt, STG_U)
  if (present(vhbt)) p_vhbt = shim_up2(8, vhbt, STG_V)
  if (present(visc_rem_u)) then
    p_vru = shim_up3(9, visc_rem_u, STG_U, nk) ; p_vrv = shim_up3(10, visc_rem_v, STG_V, nk)
  endif
  if (present(u_cor)) p_ucor = shim_buf(11, nk)
  if (present(v_cor)) p_vcor = shim_buf(12, nk)
  if (present(du_cor)) p_ducor = shim_buf(13, 1)
  if (present(dv_cor)) p_dvcor = shim_buf(14, 1)
  rc = mom6x_continuity_PPM(CS%ctx, d_u, d_v, d_hin, d_h, d_uh, d_vh, real(dt, c_double), p_uhbt, p_vhbt, &
                            p_vru, p_vrv, p_ucor, p_vcor, c_null_ptr, p_ducor, p_dvcor)
  call shim_check(rc, "continuity_PPM")
  call shim_down3(h, d_h, STG_H, nk) ; call shim_down3(uh, d_uh, STG_U, nk) ; call shim_down3(vh, d_vh, STG_V, nk)
  if (present(u_cor)) call shim_down3(u_cor, p_ucor, STG_U, nk)
  if (present(v_cor)) call shim_down3(v_cor, p_vcor, STG_V, nk)
  if (present(du_cor)) call shim_down2(du_cor, p_ducor, STG_U)
  if (present(dv_cor)) call shim_down2(dv_cor, p_dvcor, STG_V)
  call cpu_clock_end(id_clock_update)
end subroutine continuity_PPM

!> continuity_PPM_init (:2674): the parameters of :2690-2750 under their MOM_input names, logged as the reference logs them
subroutine continuity_PPM_init(Time, G, GV, US, param_file, diag, CS)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(continuity_PPM_CS), intent(inout) :: CS
  character(len=40) :: mdl = "MOM_continuity_PPM"
  character(len=24) :: sums
  logical :: flag, aggress
  integer(c_int) :: rc

  CS%initialized = .true.
  call log_version(param_file, mdl, "mom6x", "")
  call get_param(param_file, mdl, "MONOTONIC_CONTINUITY", flag, &
                 "If true, CONTINUITY_PPM uses the Colella and Woodward monotonic limiter.  The default (false) is to use "//&
                 "a simple positive definite limiter.", default=.false.)
  CS%p%monotonic = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "SIMPLE_2ND_PPM_CONTINUITY", flag, &
                 "If true, CONTINUITY_PPM uses a simple 2nd order (arithmetic mean) interpolation of the edge values.", &
                 default=.false.)
  CS%p%simple_2nd = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "UPWIND_1ST_CONTINUITY", flag, &
                 "If true, CONTINUITY_PPM becomes a 1st-order upwind continuity solver.", default=.false.)
  CS%p%upwind_1st = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "ETA_TOLERANCE", CS%p%tol_eta, &
                 "The tolerance for the differences between the barotropic and baroclinic estimates of the sea surface height "//&
                 "due to the fluxes through each face.", units="m", default=0.5*GV%ke*GV%Angstrom_m, scale=GV%m_to_H)
  call get_param(param_file, mdl, "VELOCITY_TOLERANCE", CS%p%tol_vel, &
                 "The tolerance for barotropic velocity discrepancies between the barotropic solution and the sum of the "//&
                 "layer thicknesses.", units="m s-1", default=3.0e8, scale=US%m_s_to_L_T)
  call get_param(param_file, mdl, "CONT_PPM_AGGRESS_ADJUST", flag, &
                 "If true, allow the adjusted velocities to have a relative CFL change up to 0.5.", default=.false.)
  CS%p%aggress_adjust = merge(1_c_int, 0_c_int, flag)
  aggress = flag   ! (MOM_continuity_PPM.F90:2728: CS%vol_CFL = CS%aggress_adjust, read only when that is false)
  call get_param(param_file, mdl, "CONT_PPM_VOLUME_BASED_CFL", flag, &
                 "If true, use the ratio of the open face lengths to the tracer cell areas when estimating CFL numbers.  "//&
                 "The default is set by CONT_PPM_AGGRESS_ADJUST.", default=aggress, do_not_read=aggress)
  CS%p%vol_CFL = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "CONTINUITY_CFL_LIMIT", CS%p%CFL_limit_adjust, &
                 "The maximum CFL of the adjusted velocities.", units="nondim", default=0.5)
  call get_param(param_file, mdl, "CONT_PPM_BETTER_ITER", flag, &
                 "If true, stop corrective iterations using a velocity based criterion and only stop if the iteration is "//&
                 "better than all predecessors.", default=.true.)
  CS%p%better_iter = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "CONT_PPM_USE_VISC_REM_MAX", flag, &
                 "If true, use more appropriate limiting bounds for corrections in strongly viscous columns.", default=.true.)
  CS%p%use_visc_rem_max = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl, "CONT_PPM_MARGINAL_FACE_AREAS", flag, &
                 "If true, use the marginal face areas from the continuity solver for use as the weights in the barotropic "//&
                 "solver.  Otherwise use the transport averaged areas.", default=.true.)
  CS%p%marginal_faces = merge(1_c_int, 0_c_int, flag)
  ! The one parameter the reference does not have: the order of the column sums of the mass-flux kernels (include/mom6x.h).
  call get_param(param_file, mdl, "MOM6X_CONTINUITY_SUMS", sums, &
                 "The order of the column sums of the MI355X mass-flux kernels: TREE16 (a 16-lane tree, the fast kernel; "//&
                 "answers within 1e-11 of range of the reference after 10 steps) or REFERENCE (sequential in k, bit-identical "//&
                 "to the Fortran loop nest), or TREE16_FMA (TREE16 with fused multiply-adds at fixed sites of the flux and "//&
                 "edge-value formulas; 4 % faster, the same distance from the reference's arithmetic).", default="TREE16_FMA")
  select case (trim(sums))
    case ("TREE16") ; CS%p%sum_order = 1_c_int
    case ("REFERENCE") ; CS%p%sum_order = 0_c_int
    case ("TREE16_FMA") ; CS%p%sum_order = 2_c_int
    case default ; call MOM_error(FATAL, "continuity_PPM_init: MOM6X_CONTINUITY_SUMS must be TREE16, REFERENCE or TREE16_FMA.")
  end select
  call shim_set_domain_flags(param_file)
  CS%ctx = shim_ctx(G, GV)
  rc = mom6x_continuity_init(CS%ctx, CS%p) ; call shim_check(rc, "continuity_PPM_init")
  id_clock_update = cpu_clock_id('(Ocean continuity update)', grain=CLOCK_ROUTINE)
end subroutine continuity_PPM_init

!> continuity_PPM_stencil (:2757)
function continuity_PPM_stencil(CS) result(stencil)
  type(continuity_PPM_CS), intent(in) :: CS
  integer :: stencil
  stencil = 3 ; if (CS%p%simple_2nd /= 0) stencil = 2 ; if (CS%p%upwind_1st /= 0) stencil = 1
end function continuity_PPM_stencil

end module MOM_continuity_PPM
