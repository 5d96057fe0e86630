!> mom6x_eos_reader -- the equation of state as mom6x_eos_params, read from the parameter file the way EOS_init reads it
!! (MOM_EOS.F90:1562-1640): EOS_type is opaque to other modules, so its coefficients cannot be taken from tv%eqn_of_state.  Its own
!! module because both MOM_PressureForce (which uses MOM_ALE for ALE_CS, as the reference does) and MOM_ALE (the RHO coordinate's
!! densities) need it.
module mom6x_eos_reader
use, intrinsic :: iso_c_binding
use mom6x_c_api
use MOM_error_handler,   only : MOM_error, FATAL
use MOM_file_parser,     only : get_param, param_file_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_verticalGrid,    only : verticalGrid_type
implicit none ; private
public :: shim_read_eos

contains

!> The equation of state and the EOS-only switches of PressureForce_FV_CS as mom6x_eos_params: EQN_OF_STATE and its
!! coefficients as EOS_init reads them (MOM_EOS.F90:1562-1640), MASS_WEIGHT_IN_PRESSURE_GRADIENT(_TOP),
!! MASS_WEIGHT_IN_PGF_VANISHED_ONLY, SSH_IN_EOS_PRESSURE_FOR_PGF, RECONSTRUCT_FOR_PRESSURE, PRESSURE_RECONSTRUCTION_SCHEME,
!! BOUNDARY_EXTRAPOLATION_PRESSURE (MOM_PressureForce_FV.F90:2111-2190).  have_eos is false when ENABLE_THERMODYNAMICS is.
subroutine shim_read_eos(param_file, GV, US, eos, have_eos)
  type(param_file_type),   intent(in)  :: param_file
  type(verticalGrid_type), intent(in)  :: GV
  type(unit_scale_type),   intent(in)  :: US
  type(mom6x_eos_params),  intent(out) :: eos
  logical,                 intent(out) :: have_eos
  character(len=40) :: mdl = "MOM_EOS", mdl_fv = "MOM_PressureForce_FV"
  character(len=40) :: tmpstr
  logical :: flag, use_ALE, reconstruct
  real :: rho_ref, Tref, Sref, pref, h_nv
  eos%form = 0 ; eos%Rho_T0_S0 = 0.0 ; eos%dRho_dT = 0.0 ; eos%dRho_dS = 0.0 ; eos%dRho_dp = 0.0
  eos%MassWghtInterp = 0 ; eos%use_SSH_in_Z0p = 0 ; eos%Recon_Scheme = 0 ; eos%boundary_extrap = 1
  eos%MassWghtInterpVanOnly = 0 ; eos%h_nonvanished = 0.0 ; eos%EOS_quadrature = 0
  call get_param(param_file, "MOM", "ENABLE_THERMODYNAMICS", have_eos, default=.true., do_not_log=.true.)
  if (.not.have_eos) return
  call get_param(param_file, mdl, "EQN_OF_STATE", tmpstr, default="WRIGHT", do_not_log=.true.)
  select case (trim(tmpstr))
    case ("LINEAR") ; eos%form = 1
    case ("WRIGHT") ; eos%form = 2
    case ("WRIGHT_FULL") ; eos%form = 3
    case ("WRIGHT_REDUCED") ; eos%form = 4
    case ("UNESCO", "JACKETT_MCD") ; eos%form = 5       ! (MOM_EOS.F90:1572-1573: JACKETT_MCD is the UNESCO refit)
    case ("JACKETT_06") ; eos%form = 7
    case ("ROQUET_SPV") ; eos%form = 8
    case ("ROQUET_RHO", "NEMO") ; eos%form = 6
    case default ; call MOM_error(FATAL, "PressureForce_init: EQN_OF_STATE "//trim(tmpstr)//" is not carried by the MI355X path "//&
                                  "(LINEAR, WRIGHT, WRIGHT_FULL, WRIGHT_REDUCED, UNESCO, ROQUET_RHO, JACKETT_06 and ROQUET_SPV are; TEOS10 is not).")
  end select
  call get_param(param_file, mdl, "EOS_QUADRATURE", flag, default=.false., do_not_log=.true.)   ! MOM_EOS.F90:1654
  eos%EOS_quadrature = merge(1_c_int, 0_c_int, flag)
  if (eos%form == 1) then   ! RHO_T0_S0 from the reference state when it is not given (MOM_EOS.F90:1598-1636)
    call get_param(param_file, mdl, "RHO_REF_LINEAR_EOS", rho_ref, units="kg m-3", default=1000.0, do_not_log=.true.)
    call get_param(param_file, mdl, "T_REF_LINEAR_EOS", Tref, units="degC", default=0.0, do_not_log=.true.)
    call get_param(param_file, mdl, "S_REF_LINEAR_EOS", Sref, units="psu", default=0.0, do_not_log=.true.)
    call get_param(param_file, mdl, "P_REF_LINEAR_EOS", pref, units="Pa", default=0.0, do_not_log=.true.)
    call get_param(param_file, mdl, "DRHO_DT", eos%dRho_dT, units="kg m-3 K-1", default=-0.2, do_not_log=.true.)
    call get_param(param_file, mdl, "DRHO_DS", eos%dRho_dS, units="kg m-3 ppt-1", default=0.8, do_not_log=.true.)
    call get_param(param_file, mdl, "DRHO_DP", eos%dRho_dp, units="s2 m-2", default=0.0, do_not_log=.true.)
    call get_param(param_file, mdl, "RHO_T0_S0", eos%Rho_T0_S0, units="kg m-3", &
                   default=rho_ref - (eos%dRho_dT*Tref + eos%dRho_dS*Sref + eos%dRho_dp*pref), do_not_log=.true.)
  endif
  call get_param(param_file, mdl_fv, "SSH_IN_EOS_PRESSURE_FOR_PGF", flag, default=.false., do_not_log=.true.)
  eos%use_SSH_in_Z0p = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl_fv, "MASS_WEIGHT_IN_PRESSURE_GRADIENT", flag, default=.false., do_not_log=.true.)
  if (flag) eos%MassWghtInterp = ibset(eos%MassWghtInterp, 0)
  call get_param(param_file, mdl_fv, "MASS_WEIGHT_IN_PRESSURE_GRADIENT_TOP", flag, default=.false., do_not_log=.true.)
  if (flag) eos%MassWghtInterp = ibset(eos%MassWghtInterp, 1)
  call get_param(param_file, mdl_fv, "MASS_WEIGHT_IN_PGF_VANISHED_ONLY", flag, default=.false., do_not_log=.true.)
  eos%MassWghtInterpVanOnly = merge(1_c_int, 0_c_int, flag)
  call get_param(param_file, mdl_fv, "USE_REGRIDDING", use_ALE, default=.false., do_not_log=.true.)
  call get_param(param_file, mdl_fv, "RECONSTRUCT_FOR_PRESSURE", reconstruct, default=use_ALE, do_not_log=.true.)
  if (reconstruct) then
    call get_param(param_file, mdl_fv, "PRESSURE_RECONSTRUCTION_SCHEME", eos%Recon_Scheme, default=1, do_not_log=.true.)
    call get_param(param_file, mdl_fv, "BOUNDARY_EXTRAPOLATION_PRESSURE", flag, default=.true., do_not_log=.true.)
    eos%boundary_extrap = merge(1_c_int, 0_c_int, flag)
  endif
  call get_param(param_file, mdl_fv, "RESET_INTXPA_H_NONVANISHED", h_nv, units="m", default=1.0e-6, scale=GV%m_to_H, do_not_log=.true.)
  eos%h_nonvanished = h_nv
end subroutine shim_read_eos

end module mom6x_eos_reader
