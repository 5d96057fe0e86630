!> Drop-in for src/core/MOM_PressureForce.F90: PressureForce :41, PressureForce_init :85 and the type PressureForce_CS --
!! same names and argument lists.  The reference module is a dispatcher (analytic finite volume, Montgomery potential,
!! ...); the MI355X path carries PressureForce_FV_Bouss (MOM_PressureForce_FV.F90:947) with Set_pbce_Bouss
!! (MOM_PressureForce_Mont.F90:649), so ANALYTIC_FV_PGF must keep its default and the model must be Boussinesq.
!! The equation of state of the use_EOS branch is read from the parameter file the way EOS_init reads it
!! (MOM_EOS.F90:1562-1640): EOS_type is opaque to other modules, so its coefficients cannot be taken from tv%eqn_of_state.
module MOM_PressureForce
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use mom6x_eos_reader,    only : shim_read_eos
use MOM_ALE,             only : ALE_CS
use MOM_diag_mediator,   only : diag_ctrl
use MOM_error_handler,   only : MOM_error, FATAL
use MOM_file_parser,     only : get_param, log_version, param_file_type
use MOM_grid,            only : ocean_grid_type
use MOM_self_attr_load,  only : SAL_CS
use MOM_tidal_forcing,   only : tidal_forcing_CS
use MOM_time_manager,    only : time_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_variables,       only : accel_diag_ptrs, thermo_var_ptrs
use MOM_verticalGrid,    only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>
public :: PressureForce, PressureForce_init, PressureForce_read_eos

type, public :: PressureForce_CS ; private
  logical :: initialized = .false.
  type(c_ptr) :: ctx = c_null_ptr
  type(mom6x_pgf_params) :: p
  type(mom6x_eos_params) :: eos
  logical :: have_eos = .false.
end type PressureForce_CS

contains

!> PressureForce (:41)
subroutine PressureForce(h, tv, PFu, PFv, G, GV, US, CS, ALE_CSp, ADp, p_atm, pbce, eta)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  type(unit_scale_type),   intent(in)  :: US
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)  :: h
  type(thermo_var_ptrs),   intent(in)  :: tv
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(out) :: PFu
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(out) :: PFv
  type(PressureForce_CS),  intent(inout) :: CS
  type(ALE_CS),            pointer     :: ALE_CSp
  type(accel_diag_ptrs),   pointer     :: ADp
  real, dimension(:,:),    pointer     :: p_atm
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), optional, intent(out) :: pbce
  real, dimension(SZI_(G),SZJ_(G)),          optional, intent(out) :: eta
  type(c_ptr) :: d_PFu, d_PFv, p_pbce, p_eta
  type(mom6x_eos_params), target :: eos
  integer(c_int) :: rc
  integer :: nk
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_PressureForce: Module must be initialized before it is used.")
  if (associated(p_atm)) call MOM_error(FATAL, "PressureForce: an atmospheric pressure field is not carried by the MI355X path.")
  nk = GV%ke
  if (associated(tv%T) .and. associated(tv%S)) then   ! the use_EOS branch (FV.F90:1206)
    if (.not.CS%have_eos) call MOM_error(FATAL, "PressureForce: tv%T is associated but no equation of state was read.")
    eos = CS%eos
    ! (slots 38 / 39 hold tv%T, tv%S and nothing else -- the RK2 shim's upload_tv uses the same two -- so the pointers the context
    !  keeps stay valid whichever shim ran last; the low slots are reused as output buffers by the other modules)
    rc = mom6x_PressureForce_set_tv(CS%ctx, shim_up3(38, tv%T, STG_H, nk), shim_up3(39, tv%S, STG_H, nk), c_loc(eos))
  else
    rc = mom6x_PressureForce_set_tv(CS%ctx, c_null_ptr, c_null_ptr, c_null_ptr)
  endif
  call shim_check(rc, "PressureForce (tv)")
  d_PFu = shim_buf(2, nk) ; d_PFv = shim_buf(3, nk)
  p_pbce = c_null_ptr ; p_eta = c_null_ptr
  if (present(pbce)) p_pbce = shim_buf(4, nk)
  if (present(eta)) p_eta = shim_buf(5, 1)
  rc = mom6x_PressureForce(CS%ctx, shim_up3(1, h, STG_H, nk), d_PFu, d_PFv, p_pbce, p_eta)
  call shim_check(rc, "PressureForce")
  call shim_down3(PFu, d_PFu, STG_U, nk) ; call shim_down3(PFv, d_PFv, STG_V, nk)
  if (present(pbce)) call shim_down3(pbce, p_pbce, STG_H, nk)
  if (present(eta)) call shim_down2(eta, p_eta, STG_H)
end subroutine PressureForce

!> PressureForce_init (:85) -> PressureForce_FV_init (MOM_PressureForce_FV.F90:2020)
subroutine PressureForce_init(Time, G, GV, US, param_file, diag, CS, ADp, SAL_CSp, tides_CSp)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(PressureForce_CS),  intent(inout) :: CS
  type(accel_diag_ptrs),   pointer       :: ADp
  type(SAL_CS),           intent(in), optional :: SAL_CSp
  type(tidal_forcing_CS), intent(in), optional :: tides_CSp
  character(len=40) :: mdl = "MOM_PressureForce", mdl_fv = "MOM_PressureForce_FV"
  logical :: flag
  integer(c_int) :: rc
  CS%initialized = .true.
  call log_version(param_file, mdl, "mom6x", "")
  call get_param(param_file, mdl, "ANALYTIC_FV_PGF", flag, "If true the pressure gradient forces are calculated with a finite "//&
                 "volume form that analytically integrates the equations of state in pressure.", default=.true.)
  if (.not.flag) call MOM_error(FATAL, "PressureForce_init: the Montgomery-potential form (ANALYTIC_FV_PGF = False) is not "//&
                 "carried by the MI355X path.")
  call get_param(param_file, mdl_fv, "RHO_PGF_REF", CS%p%rho_ref, "The reference density that is subtracted off when calculating "//&
                 "pressure gradient forces.", units="kg m-3", default=GV%Rho0*US%R_to_kg_m3, scale=US%kg_m3_to_R)
  call get_param(param_file, mdl_fv, "RHO_PGF_REF_BUG", flag, "If true, recover a bug that RHO_0 (the mean seawater density in "//&
                 "Boussinesq mode) and RHO_PGF_REF are both used in the Boussinesq pressure force.", default=.true.)
  CS%p%rho_ref_bug = merge(1_c_int, 0_c_int, flag)
  CS%p%Z_ref = G%Z_ref
  call must_be("USE_STANLEY_PGF", .false.) ; call must_be("CORRECTION_INTXPA", .false.) ; call must_be("RESET_INTXPA_INTEGRAL", .false.)
  call must_be("TIDES", .false.) ; call must_be("CALCULATE_SAL", .false.)
  call PressureForce_read_eos(param_file, GV, US, CS%eos, CS%have_eos)
  call shim_set_domain_flags(param_file)
  CS%ctx = shim_ctx(G, GV)
  rc = mom6x_PressureForce_init(CS%ctx, CS%p, GV%Rlay, GV%g_prime) ; call shim_check(rc, "PressureForce_init")
contains
  subroutine must_be(name, default)
    character(len=*), intent(in) :: name ; logical, intent(in) :: default
    logical :: val
    call get_param(param_file, mdl_fv, name, val, default=default, do_not_log=.true.)
    if (val .neqv. default) call MOM_error(FATAL, "PressureForce_init: "//trim(name)//" is not carried by the MI355X path.")
  end subroutine must_be
end subroutine PressureForce_init

!> The equation of state and the EOS-only switches of PressureForce_FV_CS as mom6x_eos_params: EQN_OF_STATE and its
!! coefficients as EOS_init reads them (MOM_EOS.F90:1562-1640), MASS_WEIGHT_IN_PRESSURE_GRADIENT(_TOP),
!! MASS_WEIGHT_IN_PGF_VANISHED_ONLY, SSH_IN_EOS_PRESSURE_FOR_PGF, RECONSTRUCT_FOR_PRESSURE, PRESSURE_RECONSTRUCTION_SCHEME,
!! BOUNDARY_EXTRAPOLATION_PRESSURE (MOM_PressureForce_FV.F90:2111-2190).  have_eos is false when ENABLE_THERMODYNAMICS is.
subroutine PressureForce_read_eos(param_file, GV, US, eos, have_eos)
  type(param_file_type),   intent(in)  :: param_file
  type(verticalGrid_type), intent(in)  :: GV
  type(unit_scale_type),   intent(in)  :: US
  type(mom6x_eos_params),  intent(out) :: eos
  logical,                 intent(out) :: have_eos
  call shim_read_eos(param_file, GV, US, eos, have_eos)      ! (mom6x_eos_reader: MOM_ALE reads the same)
end subroutine PressureForce_read_eos

end module MOM_PressureForce
