!> Drop-in replacement of src/core/MOM_dynamics_split_RK2.F90: the SAME module name, public names and argument lists
!! (step_MOM_dyn_split_RK2 :294-296, register_restarts_dyn_split_RK2 :1210, remap_dyn_split_RK2_aux_vars :1302,
!! init_dyn_split_RK2_diabatic :1335, initialize_dyn_split_RK2 :1346-1350, end_dyn_split_RK2 :1885), served by the
!! MI355X library through ISO_C_BINDING (fortran/mom6x_c_api.F90), the dependency-free host layer fortran/mom6x_host.F90
!! and the sibling shims that carry the reference's sub-module names (MOM_continuity_PPM, MOM_barotropic, MOM_CoriolisAdv,
!! MOM_PressureForce, MOM_vert_friction): initialize_dyn_split_RK2 calls THEIR *_init procedures with the reference's
!! argument lists, as the reference does (:1552-1600), so every parameter is read and logged by the module that owns it.
!!
!! This file uses MOM_grid, MOM_restart, ... and therefore compiles inside a MOM6 build tree -- or against the
!! interface-only stand-ins of tests/fortran_stubs/, which is how this repository compiles it (tests/test_fortran_shims_cpu.py)
!! and RUNS it on the GPU (tests/fortran_stubs/drive_shims.F90: new run, two steps, save_restart, end, restore_state,
!! the third step; bit-identical to the uninterrupted run and to the committed fixture).
!!
!! Two modes (MOM_input parameter MOM6X_RESIDENT_STATE, an addition):
!!   * False (default) -- behind an UNCHANGED MOM.F90.  Every call of step_MOM_dyn_split_RK2 uploads u, v, h, uhtr, vhtr and
!!     the set_viscous_BBL / T, S inputs, runs the step on the device and downloads u, v, h, uh, vh, uhtr, vhtr, eta_av and
!!     the restart variables' host mirrors.  Correct whatever the host does between steps; PCIe-bound (DESIGN.md section 4).
!!   * True -- the state stays in HBM between steps.  MOM.F90 then has to say when the host changed it
!!     (dyn_split_RK2_host_changed: after a host-side ALE or thermodynamic step) and when it needs it
!!     (dyn_split_RK2_sync_host: before the host thermodynamics / diagnostics; refresh_host_mirrors: before save_restart).
!!     Three one-line additions to MOM.F90, INTEGRATION.md section 4.
module MOM_dynamics_split_RK2

use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_ALE,               only : ALE_CS, ALE_vel_remap_params
use MOM_barotropic,        only : barotropic_init, barotropic_CS, register_barotropic_restarts, barotropic_end
use MOM_barotropic,        only : barotropic_refresh_restart_mirrors, barotropic_uses_BT_cont_type
use MOM_continuity_PPM,    only : continuity_init=>continuity_PPM_init, continuity_stencil=>continuity_PPM_stencil
use MOM_continuity_PPM,    only : continuity_CS=>continuity_PPM_CS
use MOM_CoriolisAdv,       only : CoriolisAdv_init, CoriolisAdv_end, CoriolisAdv_CS
use MOM_cpu_clock,         only : cpu_clock_id, cpu_clock_begin, cpu_clock_end, CLOCK_MODULE_DRIVER, CLOCK_MODULE, CLOCK_ROUTINE
use MOM_diabatic_driver,   only : diabatic_CS
use MOM_diag_mediator,     only : diag_ctrl
use MOM_error_handler,     only : MOM_error, FATAL, WARNING, callTree_enter, callTree_leave
use MOM_file_parser,       only : get_param, log_version, param_file_type
use MOM_forcing_type,      only : mech_forcing
use MOM_get_input,         only : directories
use MOM_grid,              only : ocean_grid_type
use MOM_harmonic_analysis, only : harmonic_analysis_CS
use MOM_hor_visc,          only : hor_visc_CS, hor_visc_init, hor_visc_end, hor_visc_vel_stencil
use MOM_hor_index,         only : hor_index_type
use MOM_io,                only : vardesc, var_desc
use MOM_MEKE_types,        only : MEKE_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_open_boundary,     only : ocean_OBC_type, update_OBC_CS
use MOM_porous_barriers,   only : porous_barrier_type
use MOM_PressureForce,     only : PressureForce_init, PressureForce_CS, PressureForce_read_eos
use MOM_restart,           only : register_restart_field, register_restart_pair, query_initialized, MOM_restart_CS
use MOM_set_visc,          only : set_visc_CS
use MOM_stochastics,       only : stochastic_CS
use MOM_thickness_diffuse, only : thickness_diffuse_CS
use MOM_time_manager,      only : time_type
use MOM_unit_scaling,      only : unit_scale_type
use MOM_variables,         only : thermo_var_ptrs, vertvisc_type, ocean_internal_state, accel_diag_ptrs, cont_diag_ptrs, BT_cont_type
use MOM_vert_friction,     only : vertvisc_init, vertvisc_end, vertvisc_CS, vertvisc_upload_visc
use MOM_verticalGrid,      only : verticalGrid_type
use MOM_wave_interface,    only : wave_parameters_CS

implicit none ; private

#include <MOM_memory.h>

public :: step_MOM_dyn_split_RK2, register_restarts_dyn_split_RK2, initialize_dyn_split_RK2
public :: remap_dyn_split_RK2_aux_vars, init_dyn_split_RK2_diabatic, end_dyn_split_RK2
public :: dyn_split_RK2_host_changed, dyn_split_RK2_sync_host, refresh_host_mirrors   ! MOM6X_RESIDENT_STATE = True only

!> The control structure (opaque to callers, as in the reference: "; private")
type, public :: MOM_dyn_split_RK2_CS ; private
  type(c_ptr) :: ctx = c_null_ptr            !< mom6x_ctx: the tile on the device (mom6x_shim_ctx owns it)
  type(mom6x_dims) :: dims                   !< its layout
  type(dyn_state_type) :: S                  !< u, v, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy in HBM
  logical :: resident = .false.              !< MOM6X_RESIDENT_STATE
  logical :: host_changed_state = .true.     !< (resident mode) the host arrays are newer than the device copy
  !> HOST mirrors of the restart variables (registered by pointer with MOM_restart, :1222-1290)
  real, allocatable, dimension(:,:)   :: eta
  real, allocatable, dimension(:,:,:) :: u_av, v_av, h_av, CAu_pred, CAv_pred, diffu, diffv
  !> HOST mirrors of the arrays the reference hands to MOM_diagnostics and MOM.F90 by pointer: Accel_diag%PFu, PFv, CAu, CAv,
  !! u_accel_bt, v_accel_bt (+ diffu, diffv above) and MIS%pbce (:1512-1534); refreshed with the restart mirrors
  real, allocatable, dimension(:,:,:) :: PFu, PFv, CAu, CAv, u_accel_bt, v_accel_bt, pbce
  logical :: store_CAu = .true., remap_aux = .false., module_is_initialized = .false.
  logical :: diag_mirrors = .true.           !< MOM6X_ACCEL_DIAG_MIRRORS: keep the arrays behind Accel_diag / MIS current
  type(mom6x_eos_params) :: eos              !< tv%eqn_of_state + the EOS switches of the pressure force
  logical :: have_eos = .false.
  type(accel_diag_ptrs), pointer :: ADp => NULL()
  type(BT_cont_type), pointer :: BT_cont => NULL()    !< stays unassociated: the device owns the BT_cont arrays
  type(ocean_OBC_type), pointer :: OBC => NULL()
  !> The control structures of the sub-modules, as in the reference (:233-262)
  type(continuity_CS)    :: continuity_CSp
  type(CoriolisAdv_CS)   :: CoriolisAdv
  type(PressureForce_CS) :: PressureForce_CSp
  type(hor_visc_CS)      :: hor_visc
  type(vertvisc_CS), pointer :: vertvisc_CSp => NULL()
  type(barotropic_CS)    :: barotropic_CSp
end type MOM_dyn_split_RK2_CS

!> mom6x_rk2_field ids of the restart variables (include/mom6x.h)
integer(c_int), parameter :: F_CAU_PRED = 2, F_CAV_PRED = 3, F_DIFFU = 6, F_DIFFV = 7, F_U_AV = 12, F_V_AV = 13, F_H_AV = 14, F_ETA = 16
integer(c_int), parameter :: F_CAU = 0, F_CAV = 1, F_PFU = 4, F_PFV = 5, F_U_ACCEL_BT = 10, F_V_ACCEL_BT = 11, F_PBCE = 15

integer :: id_clock_step = -1, id_clock_xfer = -1

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

  if (.not.associated(CS)) call MOM_error(FATAL, "step_MOM_dyn_split_RK2: Module must be initialized before it is used.")
  if (associated(p_surf_begin) .or. associated(p_surf_end)) call MOM_error(FATAL, &
    "step_MOM_dyn_split_RK2: a time-varying surface pressure is not carried by the MI355X path.")
  if (present(Waves)) then ; if (associated(Waves)) call MOM_error(FATAL, &
    "step_MOM_dyn_split_RK2: wave coupling is not carried by the MI355X path.") ; endif
  call callTree_enter("step_MOM_dyn_split_RK2(), MOM_dynamics_split_RK2.F90 [MI355X]")
  call cpu_clock_begin(id_clock_xfer)
  if (.not.CS%resident .or. CS%host_changed_state) then
    ! uh, vh are pure outputs of the step (:646-1079 overwrite them before reading): no upload
    rc = mom6x_upload(CS%ctx, CS%S%u, u_inst, STG_U, CS%S%nk) ; call shim_check(rc, "step_MOM_dyn_split_RK2 (u)")
    rc = mom6x_upload(CS%ctx, CS%S%v, v_inst, STG_V, CS%S%nk) ; call shim_check(rc, "step_MOM_dyn_split_RK2 (v)")
    rc = mom6x_upload(CS%ctx, CS%S%h, h, STG_H, CS%S%nk) ; call shim_check(rc, "step_MOM_dyn_split_RK2 (h)")
    rc = mom6x_upload(CS%ctx, CS%S%uhtr, uhtr, STG_U, CS%S%nk) ; call shim_check(rc, "step_MOM_dyn_split_RK2 (uhtr)")
    rc = mom6x_upload(CS%ctx, CS%S%vhtr, vhtr, STG_V, CS%S%nk) ; call shim_check(rc, "step_MOM_dyn_split_RK2 (vhtr)")
    CS%S%on_device = .true.
    if (associated(tv%T)) call upload_tv(CS, tv, GV)   ! tv%T, tv%S for the equation-of-state branch of PressureForce_FV_Bouss
    CS%host_changed_state = .false.
  endif
  ! what set_viscous_BBL left in visc for vertvisc_coef (MOM.F90 calls it before every dynamics step)
  call vertvisc_upload_visc(CS%ctx, visc, GV, 30)
  call cpu_clock_end(id_clock_xfer)

  call cpu_clock_begin(id_clock_step)
  call dyn_step(CS%S, forces%taux, forces%tauy, real(dt, c_double), calc_dtbt)
  call cpu_clock_end(id_clock_step)

  call cpu_clock_begin(id_clock_xfer)
  if (CS%resident) then
    ! eta_av is small and MOM.F90 uses it right away (the SSH accumulation of MOM.F90:1393): one 2-d download per step
    rc = mom6x_download(CS%ctx, eta_av, CS%S%eta_av, STG_H, 1_c_int) ; call shim_check(rc, "step_MOM_dyn_split_RK2 (eta_av)")
  else
    call dyn_state_download(CS%S, u_inst, v_inst, h, uh, vh, uhtr, vhtr, eta_av)
    call refresh_host_mirrors(CS, G, GV)
  endif
  call cpu_clock_end(id_clock_xfer)
  call callTree_leave("step_MOM_dyn_split_RK2()")
end subroutine step_MOM_dyn_split_RK2

!> (resident mode) to be called by MOM.F90 after it changed u, v, h on the host (ALE_regridding_and_remapping, the diabatic step)
subroutine dyn_split_RK2_host_changed(CS)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  CS%host_changed_state = .true.
end subroutine dyn_split_RK2_host_changed

!> (resident mode) to be called by MOM.F90 where the host reads u, v, h, uhtr, vhtr next (the thermodynamic step, diagnostics)
subroutine dyn_split_RK2_sync_host(CS, u, v, h, uh, vh, uhtr, vhtr, eta_av)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  real, dimension(:,:,:), intent(inout) :: u, v, h, uh, vh, uhtr, vhtr
  real, dimension(:,:),   intent(inout) :: eta_av
  call dyn_state_download(CS%S, u, v, h, uh, vh, uhtr, vhtr, eta_av)
end subroutine dyn_split_RK2_sync_host

!> Download the restart variables into their registered host mirrors (RK2.F90:1222-1290, MOM_barotropic.F90:6279-6296).
!! Non-resident mode: every step.  Resident mode: MOM.F90 calls it before save_restart.
subroutine refresh_host_mirrors(CS, G, GV)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  integer :: nk
  type(c_ptr) :: pa, pb
  nk = GV%ke
  call shim_down2(CS%eta, mom6x_rk2_field(CS%ctx, F_ETA), STG_H)
  call shim_down3(CS%u_av, mom6x_rk2_field(CS%ctx, F_U_AV), STG_U, nk)
  call shim_down3(CS%v_av, mom6x_rk2_field(CS%ctx, F_V_AV), STG_V, nk)
  call shim_down3(CS%h_av, mom6x_rk2_field(CS%ctx, F_H_AV), STG_H, nk)
  if (CS%store_CAu) then
    call shim_down3(CS%CAu_pred, mom6x_rk2_field(CS%ctx, F_CAU_PRED), STG_U, nk)
    call shim_down3(CS%CAv_pred, mom6x_rk2_field(CS%ctx, F_CAV_PRED), STG_V, nk)
  endif
  call shim_down3(CS%diffu, mom6x_rk2_field(CS%ctx, F_DIFFU), STG_U, nk)
  call shim_down3(CS%diffv, mom6x_rk2_field(CS%ctx, F_DIFFV), STG_V, nk)
  call barotropic_refresh_restart_mirrors(CS%barotropic_CSp)   ! ubtav, vbtav, DTBT
  if (CS%diag_mirrors) then   ! what Accel_diag and MIS point to (mom6x_rk2_field forms the deferred u_accel_bt, v_accel_bt on request)
    call shim_down3(CS%PFu, mom6x_rk2_field(CS%ctx, F_PFU), STG_U, nk) ; call shim_down3(CS%PFv, mom6x_rk2_field(CS%ctx, F_PFV), STG_V, nk)
    call shim_down3(CS%CAu, mom6x_rk2_field(CS%ctx, F_CAU), STG_U, nk) ; call shim_down3(CS%CAv, mom6x_rk2_field(CS%ctx, F_CAV), STG_V, nk)
    ! (the layer accelerations are formed on request from what btstep left behind; a btstep called from outside in between
    !  has replaced that, and the field comes back null: the mirrors then keep the last step's values)
    pa = mom6x_rk2_field(CS%ctx, F_U_ACCEL_BT) ; pb = mom6x_rk2_field(CS%ctx, F_V_ACCEL_BT)
    if (c_associated(pa) .and. c_associated(pb)) then
      call shim_down3(CS%u_accel_bt, pa, STG_U, nk) ; call shim_down3(CS%v_accel_bt, pb, STG_V, nk)
    else
      call MOM_error(WARNING, "MOM_dynamics_split_RK2 (mom6x): u_accel_bt / v_accel_bt of the last step are gone "//&
                     "(btstep has run since); Accel_diag%u_accel_bt keeps its previous values.")
    endif
    call shim_down3(CS%pbce, mom6x_rk2_field(CS%ctx, F_PBCE), STG_H, nk)
  endif
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
  integer(c_int) :: rc
  integer :: nk
  if (.not.CS%remap_aux) return
  nk = GV%ke
  ! (ALE_CSp%vel_remapCS, as the reference's ALE_remap_velocities(ALE_CSp, ...) calls use it :1318-1328)
  rc = mom6x_remap_dyn_split_RK2_aux_vars(CS%ctx, ALE_vel_remap_params(ALE_CSp), shim_up3(1, h_old_u, STG_U, nk), shim_up3(2, h_old_v, STG_V, nk), &
                                          shim_up3(3, h_new_u, STG_U, nk), shim_up3(4, h_new_v, STG_V, nk))
  call shim_check(rc, "remap_dyn_split_RK2_aux_vars")
  if (.not.CS%resident) call refresh_host_mirrors(CS, G, GV)
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
  type(mom6x_rk2_params) :: rk2
  real, allocatable :: zero_u(:,:,:), zero_v(:,:,:)
  integer(c_int) :: rc
  integer :: nk
  integer(c_int) :: have

  if (.not.associated(CS)) call MOM_error(FATAL, "initialize_dyn_split_RK2 called with an unassociated control structure.")
  if (CS%module_is_initialized) then
    call MOM_error(WARNING, "initialize_dyn_split_RK2 called with a control structure that has already been initialized.")
    return
  endif
  if (associated(OBC)) call MOM_error(FATAL, "initialize_dyn_split_RK2: open boundaries are not carried by the MI355X path.")
  CS%module_is_initialized = .true.
  nk = GV%ke

  call log_version(param_file, mdl, "mom6x", "")
  call get_param(param_file, mdl, "MOM6X_RESIDENT_STATE", CS%resident, "If true, the prognostic state stays in HBM between calls "//&
                 "of step_MOM_dyn_split_RK2 and MOM.F90 has to call dyn_split_RK2_host_changed / dyn_split_RK2_sync_host / "//&
                 "refresh_host_mirrors; if false the state crosses PCIe twice per step and MOM.F90 is unchanged.", default=.false.)
  call get_param(param_file, mdl, "BE", rk2%be, "If SPLIT is true, BE determines the relative weighting of a forward-backward "//&
                 "and a backward Euler treatment of the baroclinic gravity waves.", units="nondim", default=0.6)
  call get_param(param_file, mdl, "BEGW", rk2%begw, "If SPLIT is true, BEGW is a number from 0 to 1 that controls the extent "//&
                 "to which the treatment of gravity waves is forward-backward (0) or simulated backward Euler (1).", units="nondim", default=0.0)
  call get_param(param_file, mdl, "MOM6X_ACCEL_DIAG_MIRRORS", CS%diag_mirrors, "If true, the host arrays behind Accel_diag%PFu, "//&
                 "PFv, CAu, CAv, u_accel_bt, v_accel_bt and MIS%pbce are refreshed from the device whenever the restart mirrors are "//&
                 "(every step without MOM6X_RESIDENT_STATE; on refresh_host_mirrors with it).  If false they stay associated and zero.", &
                 default=.true.)
  call get_rk2_flags(param_file, rk2)     ! SPLIT_BOTTOM_STRESS, BT_USE_LAYER_FLUXES, STORE_CORIOLIS_ACCEL, VISC_REM_BUG, REMAP_AUXILIARY_VARS
  CS%remap_aux = (rk2%remap_aux /= 0)
  call PressureForce_read_eos(param_file, GV, US, CS%eos, CS%have_eos)

  ! ---- the sub-modules, in the reference's order and with its argument lists (:1552-1600).  The first of them creates
  !      the device context of this PE's tile (mom6x_shim_ctx: layout, metric block, LAYOUT > 1: RCCL communicator).
  call continuity_init(Time, G, GV, US, param_file, diag, CS%continuity_CSp)
  cont_stencil = continuity_stencil(CS%continuity_CSp)
  call CoriolisAdv_init(Time, G, GV, US, param_file, diag, Accel_diag, CS%CoriolisAdv)
  call PressureForce_init(Time, G, GV, US, param_file, diag, CS%PressureForce_CSp, CS%ADp)
  CS%ctx = shim_ctx(G, GV) ; CS%dims = shim_dims()
  call hor_visc_init(Time, G, GV, US, param_file, diag, CS%hor_visc, ADp=Accel_diag)      ! :1564
  call vertvisc_init(MIS, Time, G, GV, US, param_file, diag, Accel_diag, dirs, ntrunc, CS%vertvisc_CSp)
  call barotropic_init(u, v, h, Time, G, GV, US, param_file, diag, CS%barotropic_CSp, restart_CS, calc_dtbt, CS%BT_cont, &
                       CS%OBC)
  rk2%no_BT_cont = merge(0_c_int, 1_c_int, barotropic_uses_BT_cont_type(CS%barotropic_CSp))   ! (:467-469: associated(CS%BT_cont))
  rc = mom6x_initialize_dyn_split_RK2(CS%ctx, rk2) ; call shim_check(rc, "initialize_dyn_split_RK2")

  ! ---- the arrays other modules reach by pointer (:1512-1534): host mirrors of the device's, refreshed by refresh_host_mirrors
  allocate(CS%PFu(G%IsdB:G%IedB,G%jsd:G%jed,nk), source=0.0) ; allocate(CS%PFv(G%isd:G%ied,G%JsdB:G%JedB,nk), source=0.0)
  allocate(CS%CAu(G%IsdB:G%IedB,G%jsd:G%jed,nk), source=0.0) ; allocate(CS%CAv(G%isd:G%ied,G%JsdB:G%JedB,nk), source=0.0)
  allocate(CS%u_accel_bt(G%IsdB:G%IedB,G%jsd:G%jed,nk), source=0.0) ; allocate(CS%v_accel_bt(G%isd:G%ied,G%JsdB:G%JedB,nk), source=0.0)
  allocate(CS%pbce(G%isd:G%ied,G%jsd:G%jed,nk), source=0.0)
  MIS%diffu => CS%diffu ; MIS%diffv => CS%diffv ; MIS%PFu => CS%PFu ; MIS%PFv => CS%PFv ; MIS%CAu => CS%CAu ; MIS%CAv => CS%CAv
  MIS%pbce => CS%pbce ; MIS%u_accel_bt => CS%u_accel_bt ; MIS%v_accel_bt => CS%v_accel_bt ; MIS%u_av => CS%u_av ; MIS%v_av => CS%v_av
  CS%ADp => Accel_diag
  Accel_diag%diffu => CS%diffu ; Accel_diag%diffv => CS%diffv ; Accel_diag%PFu => CS%PFu ; Accel_diag%PFv => CS%PFv
  Accel_diag%CAu => CS%CAu ; Accel_diag%CAv => CS%CAv
  Accel_diag%u_accel_bt => CS%u_accel_bt ; Accel_diag%v_accel_bt => CS%v_accel_bt

  ! ---- the state, and the first-step fills of :1577-1650 or the restart file's values ------------------------------------
  call dyn_state_init(CS%S, CS%ctx, CS%dims)
  allocate(zero_u(G%IsdB:G%IedB,G%jsd:G%jed,nk), source=0.0) ; allocate(zero_v(G%isd:G%ied,G%JsdB:G%JedB,nk), source=0.0)
  call dyn_state_upload(CS%S, u, v, h, uh, vh, zero_u, zero_v) ; CS%host_changed_state = .false.
  deallocate(zero_u, zero_v)
  if (associated(tv%T)) call upload_tv(CS, tv, GV)
  call vertvisc_upload_visc(CS%ctx, visc, GV, 30)
  ! :1577-1668, variable by variable as the reference decides with query_initialized: what the restart file held is uploaded from
  ! its registered mirror and named in `have`; the device forms every other one the way the reference does (a new run: have = 0)
  have = 0_c_int
  if (query_initialized(CS%eta, "sfc", restart_CS)) then
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_ETA), CS%eta, STG_H, 1_c_int) ; call shim_check(rc, "restart: sfc")
    have = ior(have, MOM6X_RK2_HAVE_ETA)
  endif
  if (query_initialized(CS%diffu, "diffu", restart_CS) .and. query_initialized(CS%diffv, "diffv", restart_CS)) then
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_DIFFU), CS%diffu, STG_U, int(nk, c_int)) ; call shim_check(rc, "restart: diffu")
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_DIFFV), CS%diffv, STG_V, int(nk, c_int)) ; call shim_check(rc, "restart: diffv")
    have = ior(have, MOM6X_RK2_HAVE_DIFFU)
  endif
  if (query_initialized(CS%u_av, "u2", restart_CS) .and. query_initialized(CS%v_av, "v2", restart_CS)) then
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_U_AV), CS%u_av, STG_U, int(nk, c_int)) ; call shim_check(rc, "restart: u2")
    rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_V_AV), CS%v_av, STG_V, int(nk, c_int)) ; call shim_check(rc, "restart: v2")
    have = ior(have, MOM6X_RK2_HAVE_U2)
  endif
  if (CS%store_CAu) then
    if (query_initialized(CS%CAu_pred, "CAu", restart_CS) .and. query_initialized(CS%CAv_pred, "CAv", restart_CS)) then
      rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_CAU_PRED), CS%CAu_pred, STG_U, int(nk, c_int)) ; call shim_check(rc, "restart: CAu")
      rc = mom6x_upload(CS%ctx, mom6x_rk2_field(CS%ctx, F_CAV_PRED), CS%CAv_pred, STG_V, int(nk, c_int)) ; call shim_check(rc, "restart: CAv")
      have = ior(have, MOM6X_RK2_HAVE_CAU)
    endif
    ! (:1621-1628 would look for "uh", "vh", "h2" of an older file with only_read_from_restarts; without them the device takes the
    !  reference's other branch, :1629-1636: h_av from one continuity call, then CorAdCalc)
  endif
  rc = mom6x_dyn_split_RK2_restart_fills(CS%ctx, CS%S%u, CS%S%v, CS%S%h, CS%S%uh, CS%S%vh, real(dt, c_double), have)
  call shim_check(rc, "initialize_dyn_split_RK2 (restart variables)")
  call shim_down2(eta, mom6x_rk2_field(CS%ctx, F_ETA), STG_H)   ! the eta argument is intent(inout): :1578-1590
  call refresh_host_mirrors(CS, G, GV)
  id_clock_step = cpu_clock_id('(Ocean dynamics on the device)', grain=CLOCK_MODULE_DRIVER)
  id_clock_xfer = cpu_clock_id('(Ocean dynamics host-device transfers)', grain=CLOCK_ROUTINE)
end subroutine initialize_dyn_split_RK2

!> end_dyn_split_RK2 (:1885)
subroutine end_dyn_split_RK2(CS)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  if (.not.associated(CS)) return
  call barotropic_end(CS%barotropic_CSp)
  if (associated(CS%vertvisc_CSp)) then ; call vertvisc_end(CS%vertvisc_CSp) ; deallocate(CS%vertvisc_CSp) ; endif
  call hor_visc_end(CS%hor_visc)
  call CoriolisAdv_end(CS%CoriolisAdv)
  call dyn_state_end(CS%S)
  call shim_ctx_end()
  deallocate(CS)
end subroutine end_dyn_split_RK2

!> tv%T, tv%S -> the device, with the equation of state read at initialisation (the use_EOS branch of PressureForce_FV_Bouss)
subroutine upload_tv(CS, tv, GV)
  type(MOM_dyn_split_RK2_CS), pointer :: CS ; type(thermo_var_ptrs), intent(in) :: tv ; type(verticalGrid_type), intent(in) :: GV
  type(mom6x_eos_params), target :: eos
  integer(c_int) :: rc
  if (.not.CS%have_eos) call MOM_error(FATAL, "step_MOM_dyn_split_RK2: tv%T is associated but ENABLE_THERMODYNAMICS is false.")
  eos = CS%eos
  rc = mom6x_PressureForce_set_tv(CS%ctx, shim_up3(38, tv%T, STG_H, GV%ke), shim_up3(39, tv%S, STG_H, GV%ke), c_loc(eos))
  call shim_check(rc, "step_MOM_dyn_split_RK2 (tv)")
end subroutine upload_tv

! get_rk2_flags: one get_param per member of the bind(C) parameter struct this module itself owns (the other structs are read by
! the sub-modules' own *_init: hor_visc_init, ALE_init, ...), fortran/shims/mom6x_param_readers.inc
#include "mom6x_param_readers.inc"

end module MOM_dynamics_split_RK2
