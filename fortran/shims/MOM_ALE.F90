!> Drop-in for the entry points of src/ALE/MOM_ALE.F90 that MOM.F90's ALE_regridding_and_remapping (:1751) calls every
!! thermodynamic step: ALE_init :168, ALE_end :445, pre_ALE_adjustments :489, ALE_regrid :518, ALE_remap_tracers :760, ALE_remap_set_h_vel :882,
!! ALE_remap_velocities :1089, ALE_update_regrid_weights :1719, ALE_remap_init_conds :1711, ALE_set_extrap_boundaries :347 and the
!! type ALE_CS -- same names and argument lists, served by mom6x_ALE_regrid_zstar / _rho, mom6x_ALE_convective_adjustment,
!! mom6x_ALE_remap_tracers, mom6x_ALE_remap_set_h_vel and mom6x_ALE_remap_velocities(_conserve_ke) (SURVEY 8f-3).
!! Carried: REGRIDDING_COORDINATE_MODE = ZSTAR ("Z*") or RHO with ALE_COORDINATE_CONFIG = UNIFORM[:N[,dz]] (N = NK), MIN_THICKNESS,
!! REGRID_TIME_SCALE with the two filter depths, INTERPOLATION_SCHEME = P1M_H2 / PLM / PPM_H4, BOUNDARY_EXTRAPOLATION, P_REF;
!! REMAPPING_SCHEME / VELOCITY_REMAPPING_SCHEME = PCM, PLM, PPM_H4, PPM_IH4 with REMAP_BOUND_INTERMEDIATE_VALUES,
!! REMAP_BOUNDARY_EXTRAP / INIT_BOUNDARY_EXTRAP, REMAPPING_USE_OM4_SUBCELLS, REMAP_VEL_CONSERVE_KE.  Refused with the reference's
!! parameter name: the other coordinate modes and configurations (the device carries HYCOM1 as mom6x_ALE_regrid_hycom1, whose target
!! densities come from files this module does not read), REMAP_UV_USING_OLD_ALG, PARTIAL_CELL_VELOCITY_REMAP, the near-bottom
!! velocity masks, answer dates before 2019, ice shelves, PCM_cell masks, open boundaries.  The diagnostics (ALE_register_diags,
!! pre_ALE_diagnostics) stay host Fortran.
module MOM_ALE
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_error_handler,   only : MOM_error, FATAL, WARNING
use MOM_file_parser,     only : get_param, log_version, param_file_type
use MOM_grid,            only : ocean_grid_type
use MOM_open_boundary,   only : ocean_OBC_type
use mom6x_eos_reader,    only : shim_read_eos
use MOM_tracer_registry, only : tracer_registry_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_variables,       only : thermo_var_ptrs
use MOM_verticalGrid,    only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>
public :: ALE_init, ALE_end, ALE_regrid, ALE_remap_tracers, ALE_remap_set_h_vel, ALE_remap_velocities, pre_ALE_adjustments
public :: ALE_update_regrid_weights, ALE_remap_init_conds, ALE_set_extrap_boundaries, ALE_getCoordinate
public :: ALE_vel_remap_params

integer, parameter :: REGRIDDING_ZSTAR = 2, REGRIDDING_RHO = 3      !< regrid_consts.F90:14-15

type, public :: ALE_CS ; private
  type(c_ptr) :: ctx = c_null_ptr                  !< the tile's device context (found at the first call that has G)
  integer :: regridding_scheme = REGRIDDING_ZSTAR  !< REGRIDDING_COORDINATE_MODE
  integer :: nk = 0
  type(mom6x_regrid_rho_params) :: rg              !< regridding_CS as the device takes it (rg%f: the z* members)
  real(c_double), allocatable :: coordinateResolution(:)   !< nk values [Z] (RHO: [R])
  real(c_double), allocatable :: target_density(:)         !< nk+1 interface densities [R] (RHO)
  type(mom6x_remapping_params) :: remap, vel_remap !< CS%remapCS, CS%vel_remapCS
  logical :: remap_boundary_extrap = .false.       !< REMAP_BOUNDARY_EXTRAP, applied by ALE_set_extrap_boundaries
  logical :: do_conv_adj = .false.                 !< regridding_preadjust_reqs: RHO wants a statically stable column first
  logical :: conserve_ke = .false.                 !< REMAP_VEL_CONSERVE_KE
  logical :: remap_after_initialization = .true.   !< REMAP_AFTER_INITIALIZATION
  real :: regrid_time_scale = 0.0                  !< REGRID_TIME_SCALE [T]
  type(mom6x_eos_params) :: eos ; logical :: have_eos = .false.
end type ALE_CS

contains

!> ALE_init (:168) + ALE_initRegridding :1667 -> initialize_regridding (MOM_regridding.F90:186), the parameters of the carried modes
subroutine ALE_init(param_file, GV, US, max_depth, CS)
  type(param_file_type),   intent(in) :: param_file
  type(verticalGrid_type), intent(in) :: GV
  type(unit_scale_type),   intent(in) :: US
  real,                    intent(in) :: max_depth
  type(ALE_CS),            pointer    :: CS
  character(len=40) :: mdl = "MOM_ALE"
  character(len=80) :: string, vel_string, coord_mode, config
  logical :: flag, init_boundary_extrap, force_bounds_in_subcell, om4_remap_via_sub_cells
  integer :: default_answer_date, answer_date, regrid_answer_date, k, ke, ic
  real :: filter_shallow_depth, filter_deep_depth, tmpReal, dz_uniform, BBL_h_vel_mask, rho_light, rho_heavy, tot
  if (associated(CS)) then
    call MOM_error(WARNING, "ALE_init called with an associated control structure.")
    return
  endif
  allocate(CS)
  call log_version(param_file, mdl, "mom6x", "")
  call get_param(param_file, mdl, "REMAP_UV_USING_OLD_ALG", flag, "If true, uses the old remapping-via-a-delta-z method for "//&
                 "remapping u and v. If false, uses the new method that remaps between grids described by an old and new "//&
                 "thickness.", default=.false.)
  if (flag) call MOM_error(FATAL, trim(mdl)//": REMAP_UV_USING_OLD_ALG is not carried by the MI355X path.")

  ! ---- ALE_initRegridding :1667, initialize_regridding MOM_regridding.F90:186-830
  call get_param(param_file, mdl, "REGRIDDING_COORDINATE_MODE", coord_mode, "Coordinate mode for vertical regridding.", &
                 default="LAYER", fail_if_missing=.true.)
  select case (trim(coord_mode))
    case ("ZSTAR", "Z*") ; CS%regridding_scheme = REGRIDDING_ZSTAR
    case ("RHO") ; CS%regridding_scheme = REGRIDDING_RHO
    case default ; call MOM_error(FATAL, trim(mdl)//": REGRIDDING_COORDINATE_MODE = "//trim(coord_mode)//" is not carried by the "//&
                                  "MI355X path's Fortran boundary (ZSTAR and RHO are; HYCOM1 through mom6x_ALE_regrid_hycom1).")
  end select
  CS%rg%interp_scheme = 0 ; CS%rg%boundary_extrapolation = 0 ; CS%rg%ref_pressure = 2.0e7 ; CS%rg%compressibility_fraction = 0.0
  if (CS%regridding_scheme == REGRIDDING_RHO) then      ! the state-dependent coordinates' own parameters :283-331
    call get_param(param_file, mdl, "INTERPOLATION_SCHEME", string, "This sets the interpolation scheme to use to determine the "//&
                   "new grid.", default="P1M_H2")
    select case (trim(string))      ! regrid_interp.F90:38-49
      case ("P1M_H2") ; CS%rg%interp_scheme = 0
      case ("PLM") ; CS%rg%interp_scheme = 3
      case ("PPM_H4") ; CS%rg%interp_scheme = 5
      case default ; call MOM_error(FATAL, trim(mdl)//": INTERPOLATION_SCHEME = "//trim(string)//" is not carried by the MI355X path "//&
                                    "(P1M_H2, PLM and PPM_H4 are).")
    end select
    call get_param(param_file, mdl, "DEFAULT_ANSWER_DATE", default_answer_date, default=99991231, do_not_log=.true.)
    call get_param(param_file, mdl, "REGRIDDING_ANSWER_DATE", regrid_answer_date, "The vintage of the expressions and order of "//&
                   "arithmetic to use for regridding.", default=20181231)
    if (regrid_answer_date < 20190101) call MOM_error(FATAL, trim(mdl)//": REGRIDDING_ANSWER_DATE < 20190101 is not carried by the "//&
                                                      "MI355X path (the reference's default is 20181231: set it).")
    call get_param(param_file, mdl, "BOUNDARY_EXTRAPOLATION", flag, "When defined, a proper high-order reconstruction scheme is used "//&
                   "within boundary cells rather than PCM.", default=.false.)
    CS%rg%boundary_extrapolation = merge(1_c_int, 0_c_int, flag)
  endif
  call get_param(param_file, mdl, "ALE_COORDINATE_CONFIG", config, "Determines how to specify the coordinate resolution.", &
                 default="UNIFORM")
  ke = GV%ke ; dz_uniform = max_depth
  if (trim(config) /= "UNIFORM") then      ! "UNIFORM:N" or "UNIFORM:N,dz" :371-374
    if (index(trim(config), "UNIFORM:") /= 1 .or. len_trim(config) <= 8) &
      call MOM_error(FATAL, trim(mdl)//": ALE_COORDINATE_CONFIG = "//trim(config)//" is not carried by the MI355X path's Fortran "//&
                     "boundary (UNIFORM[:N[,dz]] is; hand the resolution of any other to mom6x_ALE_regrid_zstar).")
    ic = index(config, ",")
    if (ic > 0) then ; read(config(9:ic-1), *) ke ; read(config(ic+1:len_trim(config)), *) dz_uniform
    else ; read(config(9:len_trim(config)), *) ke ; endif
  endif
  if (ke /= GV%ke) call MOM_error(FATAL, trim(mdl)//": the number of levels of ALE_COORDINATE_CONFIG must be NK on the MI355X path.")
  CS%nk = ke
  allocate(CS%coordinateResolution(ke)) ; allocate(CS%target_density(ke+1)) ; CS%target_density(:) = 0.0
  if (CS%regridding_scheme == REGRIDDING_RHO) then       ! uniformResolution :1972, setCoordinateResolution scale = US%kg_m3_to_R
    rho_light = US%R_to_kg_m3*(GV%Rlay(1) + 0.5*(GV%Rlay(1)-GV%Rlay(min(2,ke))))
    rho_heavy = US%R_to_kg_m3*(GV%Rlay(ke) + 0.5*(GV%Rlay(ke)-GV%Rlay(max(ke-1,1))))
    CS%coordinateResolution(:) = ((rho_heavy - rho_light) / real(ke)) * US%kg_m3_to_R
    if (ke == 1) then       ! set_target_densities_from_GV :2069
      CS%target_density(1) = 0.0 ; CS%target_density(2) = 2.0*GV%Rlay(1)
    else
      CS%target_density(1) = (GV%Rlay(1) + 0.5*(GV%Rlay(1)-GV%Rlay(2)))
      CS%target_density(ke+1) = (GV%Rlay(ke) + 0.5*(GV%Rlay(ke)-GV%Rlay(ke-1)))
      do k=2,ke ; CS%target_density(k) = CS%target_density(k-1) + CS%coordinateResolution(k) ; enddo
    endif
  else
    CS%coordinateResolution(:) = dz_uniform / real(ke)
    tot = sum(CS%coordinateResolution(:))                ! the target grid made consistent with MAXIMUM_DEPTH :563-582
    if (tot < max_depth) then
      CS%coordinateResolution(ke) = CS%coordinateResolution(ke) + (max_depth - tot)
    elseif (tot > max_depth) then
      if (CS%coordinateResolution(ke) + (max_depth - tot) > 0.) then
        CS%coordinateResolution(ke) = CS%coordinateResolution(ke) + (max_depth - tot)
      else
        call MOM_error(FATAL, trim(mdl)//", initialize_regridding: MAXIMUM_DEPTH was too shallow to adjust bottom layer of DZ!"//trim(config))
      endif
    endif
    CS%coordinateResolution(:) = CS%coordinateResolution(:) * US%m_to_Z
  endif
  if (CS%regridding_scheme == REGRIDDING_RHO) then
    call get_param(param_file, mdl, "P_REF", CS%rg%ref_pressure, "The pressure that is used for calculating the coordinate density.", &
                   units="Pa", default=2.0e7, scale=US%Pa_to_RL2_T2)
    call get_param(param_file, mdl, "REGRID_COMPRESSIBILITY_FRACTION", CS%rg%compressibility_fraction, units="nondim", default=0.)
    call shim_read_eos(param_file, GV, US, CS%eos, CS%have_eos)
    if (.not.CS%have_eos) call MOM_error(FATAL, trim(mdl)//": REGRIDDING_COORDINATE_MODE = RHO needs ENABLE_THERMODYNAMICS.")
  endif
  call get_param(param_file, mdl, "MIN_THICKNESS", CS%rg%f%min_thickness, "When regridding, this is the minimum layer thickness allowed.", &
                 units="m", scale=GV%m_to_H, default=1.e-3)
  CS%do_conv_adj = (CS%regridding_scheme == REGRIDDING_RHO)      ! regridding_preadjust_reqs :966-984

  ! ---- the remapping ALE orchestrates :209-281
  call get_param(param_file, mdl, "REMAPPING_SCHEME", string, "This sets the reconstruction scheme used for vertical remapping for all "//&
                 "variables.", default="PLM")
  call get_param(param_file, mdl, "VELOCITY_REMAPPING_SCHEME", vel_string, "This sets the reconstruction scheme used for vertical "//&
                 "remapping of velocities. By default it is the same as REMAPPING_SCHEME.", default=trim(string))
  call get_param(param_file, mdl, "FATAL_CHECK_RECONSTRUCTIONS", flag, default=.false.)
  if (flag) call MOM_error(FATAL, trim(mdl)//": FATAL_CHECK_RECONSTRUCTIONS is not carried by the MI355X path.")
  call get_param(param_file, mdl, "FATAL_CHECK_REMAPPING", flag, default=.false.)
  if (flag) call MOM_error(FATAL, trim(mdl)//": FATAL_CHECK_REMAPPING is not carried by the MI355X path.")
  call get_param(param_file, mdl, "REMAP_BOUND_INTERMEDIATE_VALUES", force_bounds_in_subcell, "If true, the values on the "//&
                 "intermediate grid used for remapping are forced to be bounded, which might not be the case due to round off.", &
                 default=.false.)
  call get_param(param_file, mdl, "REMAP_BOUNDARY_EXTRAP", CS%remap_boundary_extrap, "If true, values at the interfaces of boundary "//&
                 "cells are extrapolated instead of piecewise constant", default=.false.)
  call get_param(param_file, mdl, "INIT_BOUNDARY_EXTRAP", init_boundary_extrap, "If true, values at the interfaces of boundary cells "//&
                 "are extrapolated instead of piecewise constant during initialization.", default=CS%remap_boundary_extrap)
  call get_param(param_file, mdl, "DEFAULT_ANSWER_DATE", default_answer_date, default=99991231)
  call get_param(param_file, mdl, "REMAPPING_USE_OM4_SUBCELLS", om4_remap_via_sub_cells, "This selects the remapping algorithm used in "//&
                 "OM4 that does not use the full reconstruction for the top- and lower-most sub-layers.", default=.true.)
  call get_param(param_file, mdl, "REMAPPING_ANSWER_DATE", answer_date, "The vintage of the expressions and order of arithmetic to use "//&
                 "for remapping.", default=default_answer_date)
  if (answer_date < 20190101) call MOM_error(FATAL, trim(mdl)//": REMAPPING_ANSWER_DATE < 20190101 is not carried by the MI355X path.")
  call set_remap(CS%remap, string, "REMAPPING_SCHEME")
  call set_remap(CS%vel_remap, vel_string, "VELOCITY_REMAPPING_SCHEME")
  call get_param(param_file, mdl, "PARTIAL_CELL_VELOCITY_REMAP", flag, default=.false.)
  if (flag) call MOM_error(FATAL, trim(mdl)//": PARTIAL_CELL_VELOCITY_REMAP is not carried by the MI355X path.")
  call get_param(param_file, mdl, "REMAP_AFTER_INITIALIZATION", CS%remap_after_initialization, "If true, applies regridding and "//&
                 "remapping immediately after initialization so that the state is ALE consistent.", default=.true.)
  call get_param(param_file, mdl, "REGRID_TIME_SCALE", CS%regrid_time_scale, "The time-scale used in blending between the current "//&
                 "(old) grid and the target (new) grid.", units="s", default=0., scale=US%s_to_T)
  call get_param(param_file, mdl, "REGRID_FILTER_SHALLOW_DEPTH", filter_shallow_depth, "The depth above which no time-filtering is "//&
                 "applied.", units="m", default=0., scale=GV%m_to_H)
  call get_param(param_file, mdl, "REGRID_FILTER_DEEP_DEPTH", filter_deep_depth, "The depth below which full time-filtering is applied "//&
                 "with time-scale REGRID_TIME_SCALE.", units="m", default=0., scale=GV%m_to_H)
  CS%rg%f%depth_of_time_filter_shallow = filter_shallow_depth ; CS%rg%f%depth_of_time_filter_deep = filter_deep_depth
  CS%rg%f%old_grid_weight = 0.0 ; CS%rg%f%Z_ref = 0.0
  call get_param(param_file, mdl, "REGRID_USE_OLD_DIRECTION", flag, default=.true., do_not_log=.true.)
  CS%rg%integrate_downward_for_e = merge(0_c_int, 1_c_int, flag)
  call get_param(param_file, mdl, "REMAP_VEL_MASK_BBL_THICK", BBL_h_vel_mask, "A thickness of a bottom boundary layer below which "//&
                 "velocities in thin layers are zeroed out after remapping, or a negative value to avoid such filtering altogether.", &
                 default=-0.001, units="m", scale=GV%m_to_H)
  if (BBL_h_vel_mask > 0.0) call MOM_error(FATAL, trim(mdl)//": REMAP_VEL_MASK_BBL_THICK > 0 is not carried by the MI355X path.")
  call get_param(param_file, mdl, "REMAP_VEL_CONSERVE_KE", CS%conserve_ke, "If true, a correction is applied to the baroclinic component "//&
                 "of velocity after remapping so that total KE is conserved.", default=.false.)

contains
  subroutine set_remap(p, scheme, pname)      ! initialize_remapping (MOM_remapping.F90:1654) with ALE_init's arguments
    type(mom6x_remapping_params), intent(out) :: p ; character(len=*), intent(in) :: scheme, pname
    select case (trim(scheme))      ! the codes of MOM_remapping.F90:86-96
      case ("PCM") ; p%scheme = 0
      case ("PLM") ; p%scheme = 2
      case ("PPM_H4") ; p%scheme = 4
      case ("PPM_IH4") ; p%scheme = 5
      case default ; call MOM_error(FATAL, trim(mdl)//": "//trim(pname)//" = "//trim(scheme)//" is not carried by the MI355X path "//&
                                    "(PCM, PLM, PPM_H4, PPM_IH4 are).")
    end select
    p%boundary_extrapolation = merge(1_c_int, 0_c_int, init_boundary_extrap)
    p%force_bounds_in_subcell = merge(1_c_int, 0_c_int, force_bounds_in_subcell)
    p%force_bounds_in_target = 1      ! the default of initialize_remapping
    p%om4_remap_via_sub_cells = merge(1_c_int, 0_c_int, om4_remap_via_sub_cells)
    p%answer_date = answer_date
    p%h_neglect = GV%H_subroundoff ; p%h_neglect_edge = GV%H_subroundoff      ! answer dates >= 20190101 (:258-259)
  end subroutine set_remap
end subroutine ALE_init

!> ALE_set_extrap_boundaries (:347): after initialisation the run-time REMAP_BOUNDARY_EXTRAP replaces INIT_BOUNDARY_EXTRAP
subroutine ALE_set_extrap_boundaries(param_file, CS)
  type(param_file_type), intent(in) :: param_file
  type(ALE_CS),          pointer    :: CS
  logical :: remap_boundary_extrap
  call get_param(param_file, "MOM_ALE", "REMAP_BOUNDARY_EXTRAP", remap_boundary_extrap, "If true, values at the interfaces of boundary "//&
                 "cells are extrapolated instead of piecewise constant", default=.false.)
  CS%remap%boundary_extrapolation = merge(1_c_int, 0_c_int, remap_boundary_extrap)
  CS%vel_remap%boundary_extrapolation = CS%remap%boundary_extrapolation
end subroutine ALE_set_extrap_boundaries

!> ALE_end (:445)
subroutine ALE_end(CS)
  type(ALE_CS), pointer :: CS
  if (.not.associated(CS)) return
  if (allocated(CS%coordinateResolution)) deallocate(CS%coordinateResolution)
  if (allocated(CS%target_density)) deallocate(CS%target_density)
  deallocate(CS)
end subroutine ALE_end

!> ALE_remap_init_conds (:1711)
logical function ALE_remap_init_conds(CS)
  type(ALE_CS), pointer :: CS
  ALE_remap_init_conds = .false.
  if (associated(CS)) ALE_remap_init_conds = CS%remap_after_initialization
end function ALE_remap_init_conds

!> ALE_update_regrid_weights (:1719)
subroutine ALE_update_regrid_weights(dt, CS)
  real,         intent(in) :: dt
  type(ALE_CS), pointer    :: CS
  real :: w
  if (associated(CS)) then
    w = 0.0
    if (CS%regrid_time_scale > 0.0) w = CS%regrid_time_scale / (CS%regrid_time_scale + dt)
    CS%rg%f%old_grid_weight = w
  endif
end subroutine ALE_update_regrid_weights

!> ALE_getCoordinate (:1688): the target interface positions [Z] (RHO: the interface densities [R])
function ALE_getCoordinate(CS)
  type(ALE_CS), pointer :: CS
  real, dimension(CS%nk+1) :: ALE_getCoordinate
  integer :: k
  if (CS%regridding_scheme == REGRIDDING_RHO) then
    ALE_getCoordinate(:) = CS%target_density(:)
  else
    ALE_getCoordinate(1) = 0.0
    do k=1,CS%nk ; ALE_getCoordinate(k+1) = ALE_getCoordinate(k) - CS%coordinateResolution(k) ; enddo   ! getStaticThickness / :2263
  endif
end function ALE_getCoordinate

!> CS%vel_remapCS as the device takes it: what remap_dyn_split_RK2_aux_vars hands to mom6x_remap_dyn_split_RK2_aux_vars
function ALE_vel_remap_params(CS) result(p)
  type(ALE_CS), pointer :: CS
  type(mom6x_remapping_params) :: p
  if (.not.associated(CS)) call MOM_error(FATAL, "ALE_vel_remap_params: the ALE control structure is not associated.")
  p = CS%vel_remap
end function ALE_vel_remap_params

!> pre_ALE_adjustments (:489): the column-wise convective adjustment the RHO coordinate asks for (regridding_preadjust_reqs :966,
!! convective_adjustment MOM_regridding.F90:1905), in place on h, tv%T, tv%S; nothing to do for z*.
subroutine pre_ALE_adjustments(G, GV, US, h, tv, Reg, CS, u, v)
  type(ocean_grid_type),                      intent(in)    :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  type(unit_scale_type),                      intent(in)    :: US
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(inout) :: h
  type(thermo_var_ptrs),                      intent(inout) :: tv
  type(tracer_registry_type),                 pointer       :: Reg
  type(ALE_CS),                               pointer       :: CS
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(inout) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(inout) :: v
  type(c_ptr) :: d_h, d_T, d_S
  integer(c_int) :: rc
  integer :: nk
  if (.not.associated(CS)) call MOM_error(FATAL, "pre_ALE_adjustments: the ALE control structure is not associated.")
  if (.not.CS%do_conv_adj) return
  if (.not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, "pre_ALE_adjustments: convective adjustment needs tv%T and tv%S.")
  nk = GV%ke
  if (.not.c_associated(CS%ctx)) CS%ctx = shim_ctx(G, GV)
  d_h = shim_up3(1, h, STG_H, nk) ; d_T = shim_up3(4, tv%T, STG_H, nk) ; d_S = shim_up3(5, tv%S, STG_H, nk)
  rc = mom6x_ALE_convective_adjustment(CS%ctx, CS%eos, d_h, d_T, d_S) ; call shim_check(rc, "pre_ALE_adjustments")
  call shim_down3(h, d_h, STG_H, nk) ; call shim_down3(tv%T, d_T, STG_H, nk) ; call shim_down3(tv%S, d_S, STG_H, nk)
end subroutine pre_ALE_adjustments

!> ALE_regrid (:518) -> regridding_main (MOM_regridding.F90:862).  (REGRIDDING_RHO wants a statically stable column: MOM.F90 calls
!! pre_ALE_adjustments before it, as it does with the reference.)
subroutine ALE_regrid(G, GV, US, h, h_new, dzRegrid, tv, CS, frac_shelf_h, PCM_cell)
  type(ocean_grid_type),                       intent(in)    :: G
  type(verticalGrid_type),                     intent(in)    :: GV
  type(unit_scale_type),                       intent(in)    :: US
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),   intent(in)    :: h
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),   intent(out)   :: h_new
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)+1), intent(out)   :: dzRegrid
  type(thermo_var_ptrs),                       intent(inout) :: tv
  type(ALE_CS),                                pointer       :: CS
  real, dimension(SZI_(G),SZJ_(G)),               optional, intent(in)  :: frac_shelf_h
  logical, dimension(SZI_(G),SZJ_(G),SZK_(GV)),   optional, intent(out) :: PCM_cell
  type(c_ptr) :: d_h, d_hn, d_dz
  integer(c_int) :: rc
  integer :: nk
  if (.not.associated(CS)) call MOM_error(FATAL, "ALE_regrid: the ALE control structure is not associated.")
  if (present(frac_shelf_h)) call MOM_error(FATAL, "ALE_regrid: ice shelves are not carried by the MI355X path.")
  nk = GV%ke
  if (.not.c_associated(CS%ctx)) CS%ctx = shim_ctx(G, GV)
  CS%rg%f%Z_ref = G%Z_ref
  d_h = shim_up3(1, h, STG_H, nk) ; d_hn = shim_out3(2, h_new, nk) ; d_dz = shim_out3(3, dzRegrid, nk+1)
  if (CS%regridding_scheme == REGRIDDING_ZSTAR) then
    rc = mom6x_ALE_regrid_zstar(CS%ctx, CS%rg%f, CS%coordinateResolution, d_h, d_hn, d_dz)
  else
    if (.not.(associated(tv%T) .and. associated(tv%S))) &
      call MOM_error(FATAL, "ALE_regrid: REGRIDDING_COORDINATE_MODE = RHO needs tv%T and tv%S.")
    rc = mom6x_ALE_regrid_rho(CS%ctx, CS%rg, CS%eos, CS%target_density, d_h, shim_up3(4, tv%T, STG_H, nk), shim_up3(5, tv%S, STG_H, nk), &
                              d_hn, d_dz)
  endif
  call shim_check(rc, "ALE_regrid")
  call shim_down3(h_new, d_hn, STG_H, nk) ; call shim_down3(dzRegrid, d_dz, STG_H, nk+1)
  if (present(PCM_cell)) PCM_cell(:,:,:) = .false.      ! (only the HYBGEN coordinate sets any)
end subroutine ALE_regrid

!> ALE_remap_tracers (:760): every registered tracer from the old onto the new grid, in place.  The tendency diagnostics (dt) are
!! host Fortran.
subroutine ALE_remap_tracers(CS, G, GV, h_old, h_new, Reg, debug, dt, PCM_cell)
  type(ALE_CS),                              intent(in)    :: CS
  type(ocean_grid_type),                     intent(in)    :: G
  type(verticalGrid_type),                   intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in)    :: h_old
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in)    :: h_new
  type(tracer_registry_type),                pointer       :: Reg
  logical,                         optional, intent(in)    :: debug
  real,                            optional, intent(in)    :: dt
  logical, dimension(SZI_(G),SZJ_(G),SZK_(GV)), optional, intent(in) :: PCM_cell
  integer, parameter :: NTR_MAX = 32
  type(c_ptr), target :: d_tr(NTR_MAX)
  type(c_ptr) :: d_ho, d_hn, ctx
  integer(c_int) :: rc
  integer :: m, ntr, nk
  ntr = 0 ; if (associated(Reg)) ntr = Reg%ntr
  if (ntr < 1) return
  if (ntr > NTR_MAX) call MOM_error(FATAL, "ALE_remap_tracers: more tracers than the shim's scratch slots.")
  if (present(PCM_cell)) then ; if (any(PCM_cell)) &
    call MOM_error(FATAL, "ALE_remap_tracers: PCM_cell masks are not carried by the MI355X path.") ; endif
  nk = GV%ke
  ctx = CS%ctx ; if (.not.c_associated(ctx)) ctx = shim_ctx(G, GV)
  d_ho = shim_up3(1, h_old, STG_H, nk) ; d_hn = shim_up3(2, h_new, STG_H, nk)
  do m=1,ntr ; d_tr(m) = shim_up3(8+m, Reg%Tr(m)%t, STG_H, nk) ; enddo
  rc = mom6x_ALE_remap_tracers(ctx, CS%remap, d_ho, d_hn, c_loc(d_tr), int(ntr, c_int)) ; call shim_check(rc, "ALE_remap_tracers")
  do m=1,ntr ; call shim_down3(Reg%Tr(m)%t, d_tr(m), STG_H, nk) ; enddo
end subroutine ALE_remap_tracers

!> ALE_remap_set_h_vel (:882)
subroutine ALE_remap_set_h_vel(CS, G, GV, h_new, h_u, h_v, OBC, debug)
  type(ALE_CS),                               intent(in)    :: CS
  type(ocean_grid_type),                      intent(in)    :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: h_new
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: h_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: h_v
  type(ocean_OBC_type),                       pointer       :: OBC
  logical,                          optional, intent(in)    :: debug
  type(c_ptr) :: d_hu, d_hv, ctx
  integer(c_int) :: rc
  integer :: nk
  if (associated(OBC)) call MOM_error(FATAL, "ALE_remap_set_h_vel: open boundaries are not carried by the MI355X path.")
  nk = GV%ke
  ctx = CS%ctx ; if (.not.c_associated(ctx)) ctx = shim_ctx(G, GV)
  d_hu = shim_up3(2, h_u, STG_U, nk) ; d_hv = shim_up3(3, h_v, STG_V, nk)      ! (inout: closed faces keep their values)
  rc = mom6x_ALE_remap_set_h_vel(ctx, shim_up3(1, h_new, STG_H, nk), d_hu, d_hv) ; call shim_check(rc, "ALE_remap_set_h_vel")
  call shim_down3(h_u, d_hu, STG_U, nk) ; call shim_down3(h_v, d_hv, STG_V, nk)
end subroutine ALE_remap_set_h_vel

!> ALE_remap_velocities (:1089); with REMAP_VEL_CONSERVE_KE and allow_preserve_variance the KE-conserving correction :1166-1195
subroutine ALE_remap_velocities(CS, G, GV, h_old_u, h_old_v, h_new_u, h_new_v, u, v, debug, dt, allow_preserve_variance)
  type(ALE_CS),                               intent(in)    :: CS
  type(ocean_grid_type),                      intent(in)    :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: h_old_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: h_old_v
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: h_new_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: h_new_v
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: v
  logical,                          optional, intent(in)    :: debug
  real,                             optional, intent(in)    :: dt
  logical,                          optional, intent(in)    :: allow_preserve_variance
  type(c_ptr) :: d_u, d_v, d_hou, d_hov, d_hnu, d_hnv, ctx
  integer(c_int) :: rc
  integer :: nk
  logical :: preserve
  nk = GV%ke
  ctx = CS%ctx ; if (.not.c_associated(ctx)) ctx = shim_ctx(G, GV)
  preserve = .false. ; if (present(allow_preserve_variance)) preserve = allow_preserve_variance .and. CS%conserve_ke   ! :1133-1136
  d_hou = shim_up3(1, h_old_u, STG_U, nk) ; d_hov = shim_up3(2, h_old_v, STG_V, nk)
  d_hnu = shim_up3(3, h_new_u, STG_U, nk) ; d_hnv = shim_up3(4, h_new_v, STG_V, nk)
  d_u = shim_up3(5, u, STG_U, nk) ; d_v = shim_up3(6, v, STG_V, nk)
  if (preserve) then
    rc = mom6x_ALE_remap_velocities_conserve_ke(ctx, CS%vel_remap, d_hou, d_hov, d_hnu, d_hnv, d_u, d_v)
  else
    rc = mom6x_ALE_remap_velocities(ctx, CS%vel_remap, d_hou, d_hov, d_hnu, d_hnv, d_u, d_v)
  endif
  call shim_check(rc, "ALE_remap_velocities")
  call shim_down3(u, d_u, STG_U, nk) ; call shim_down3(v, d_v, STG_V, nk)
end subroutine ALE_remap_velocities

end module MOM_ALE
