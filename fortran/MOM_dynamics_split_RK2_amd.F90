!> Drop-in shim: the public interface of MOM_dynamics_split_RK2 (src/core/MOM_dynamics_split_RK2.F90)
!! served by the MI355X-native library through ISO_C_BINDING.
!!
!! THIS FILE COMPILES ONLY INSIDE A MOM6 BUILD TREE (it uses MOM_grid, MOM_variables, ... and through
!! them FMS); it is the source a maintainer drops in place of the module used at MOM.F90:1312.  The
!! procedure names, argument lists and the opaque control structure are those of the reference
!! (:294-296 step_MOM_dyn_split_RK2, :1346 initialize_dyn_split_RK2, :1885 end_dyn_split_RK2).  State
!! stays resident in HBM between steps; host arrays are refreshed where MOM6 reads them next
!! (MOM.F90:1362, :1393, :1469-1471; save_restart through the pointers registered at RK2.F90:1210).
module MOM_dynamics_split_RK2_amd

use, intrinsic :: iso_c_binding
use mom6x_c_api
use MOM_error_handler,  only : MOM_error, FATAL
use MOM_grid,           only : ocean_grid_type
use MOM_verticalGrid,   only : verticalGrid_type
use MOM_unit_scaling,   only : unit_scale_type
use MOM_variables,      only : thermo_var_ptrs, vertvisc_type
use MOM_forcing_type,   only : mech_forcing
use MOM_time_manager,   only : time_type

implicit none ; private

#include <MOM_memory.h>

public :: step_MOM_dyn_split_RK2, initialize_dyn_split_RK2_amd, end_dyn_split_RK2

!> The control structure: the device context plus the device copies of the prognostic state
type, public :: MOM_dyn_split_RK2_CS ; private
  type(c_ptr) :: ctx = c_null_ptr        !< mom6x_ctx
  type(c_ptr) :: d_u, d_v, d_h, d_uh, d_vh, d_uhtr, d_vhtr, d_eta_av, d_taux, d_tauy
  logical :: state_on_device = .false.   !< true once u,v,h,... have been uploaded
end type MOM_dyn_split_RK2_CS

contains

!> step_MOM_dyn_split_RK2 with the reference's argument list (RK2.F90:294-296).
subroutine step_MOM_dyn_split_RK2(u_inst, v_inst, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, &
                                  uh, vh, uhtr, vhtr, eta_av, G, GV, US, CS, calc_dtbt, VarMix, MEKE, &
                                  thickness_diffuse_CSp, pbv, STOCH, Waves)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: u_inst
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: v_inst
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(inout) :: h
  type(thermo_var_ptrs),   intent(in)    :: tv
  type(vertvisc_type),     intent(inout) :: visc
  type(time_type),         intent(in)    :: Time_local
  real,                    intent(in)    :: dt
  type(mech_forcing),      intent(in)    :: forces
  real, dimension(:,:),    pointer       :: p_surf_begin, p_surf_end
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: vh
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: uhtr
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: vhtr
  real, dimension(SZI_(G),SZJ_(G)),        intent(out)   :: eta_av
  type(unit_scale_type),   intent(in)    :: US
  type(MOM_dyn_split_RK2_CS), pointer    :: CS
  logical,                 intent(in)    :: calc_dtbt
  class(*), optional :: VarMix, MEKE, thickness_diffuse_CSp, pbv, STOCH, Waves  ! handed to the host hooks
  integer(c_int) :: rc
  integer :: nk
  nk = GV%ke
  if (associated(p_surf_begin) .or. associated(p_surf_end)) call MOM_error(FATAL, &
    "step_MOM_dyn_split_RK2(amd): dynamic surface pressure is not supported on the device path")
  if (.not.CS%state_on_device) then   ! first step (or after a host-side change such as ALE remapping)
    rc = mom6x_upload(CS%ctx, CS%d_u, u_inst, 1, nk) ; rc = mom6x_upload(CS%ctx, CS%d_v, v_inst, 2, nk)
    rc = mom6x_upload(CS%ctx, CS%d_h, h, 0, nk)
    rc = mom6x_upload(CS%ctx, CS%d_uh, uh, 1, nk) ; rc = mom6x_upload(CS%ctx, CS%d_vh, vh, 2, nk)
    rc = mom6x_upload(CS%ctx, CS%d_uhtr, uhtr, 1, nk) ; rc = mom6x_upload(CS%ctx, CS%d_vhtr, vhtr, 2, nk)
    CS%state_on_device = .true.
  endif
  rc = mom6x_upload(CS%ctx, CS%d_taux, forces%taux, 1, 1) ; rc = mom6x_upload(CS%ctx, CS%d_tauy, forces%tauy, 2, 1)
  ! hooks = c_null_ptr: vertvisc_coef outputs / diffu,diffv frozen over the step.  A full integration passes a
  ! mom6x_rk2_hooks whose bind(C) callbacks download up,vp,h, call vertvisc_coef / horizontal_viscosity and
  ! upload the results (INTEGRATION.md section 5).
  rc = mom6x_step_dyn_split_RK2(CS%ctx, CS%d_u, CS%d_v, CS%d_h, CS%d_uh, CS%d_vh, CS%d_uhtr, CS%d_vhtr, CS%d_eta_av, &
                                CS%d_taux, CS%d_tauy, real(dt, c_double), merge(1_c_int, 0_c_int, calc_dtbt), c_null_ptr)
  if (rc /= 0) call MOM_error(FATAL, "step_MOM_dyn_split_RK2(amd): "//c_message())
  ! MOM.F90 uses u, v, h, uhtr, vhtr and eta_av right after the call:
  rc = mom6x_download(CS%ctx, u_inst, CS%d_u, 1, nk) ; rc = mom6x_download(CS%ctx, v_inst, CS%d_v, 2, nk)
  rc = mom6x_download(CS%ctx, h, CS%d_h, 0, nk)
  rc = mom6x_download(CS%ctx, uhtr, CS%d_uhtr, 1, nk) ; rc = mom6x_download(CS%ctx, vhtr, CS%d_vhtr, 2, nk)
  rc = mom6x_download(CS%ctx, eta_av, CS%d_eta_av, 0, 1)
end subroutine step_MOM_dyn_split_RK2

!> The part of initialize_dyn_split_RK2 (:1346) that creates the device context from G, GV and the parameters.
subroutine initialize_dyn_split_RK2_amd(G, GV, metrics_block, dims, gvx, cont, bt, cor, pgf, rk2, device, CS)
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  real(c_double),          intent(in) :: metrics_block(*) !< the MOM6X_G_* planes packed from G%...
  type(mom6x_dims), intent(in) :: dims ; type(mom6x_vgrid), intent(in) :: gvx
  type(mom6x_continuity_params), intent(in) :: cont ; type(mom6x_barotropic_params), intent(in) :: bt
  type(mom6x_coriolis_params), intent(in) :: cor ; type(mom6x_pgf_params), intent(in) :: pgf
  type(mom6x_rk2_params), intent(in) :: rk2 ; integer, intent(in) :: device
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  integer(c_int) :: rc
  integer(c_size_t) :: n3, n2
  allocate(CS)
  rc = mom6x_ctx_create(CS%ctx, dims, int(device, c_int), metrics_block, gvx, int(G%first_direction, c_int))
  if (rc /= 0) call MOM_error(FATAL, "initialize_dyn_split_RK2(amd): "//c_message())
  rc = mom6x_continuity_init(CS%ctx, cont) ; rc = mom6x_barotropic_init(CS%ctx, bt)
  rc = mom6x_CoriolisAdv_init(CS%ctx, cor) ; rc = mom6x_PressureForce_init(CS%ctx, pgf, GV%Rlay, GV%g_prime)
  rc = mom6x_initialize_dyn_split_RK2(CS%ctx, rk2)
  if (rc /= 0) call MOM_error(FATAL, "initialize_dyn_split_RK2(amd): "//c_message())
  n2 = int(dims%slab, c_size_t) ; n3 = n2 * int(dims%nk, c_size_t)
  rc = mom6x_dev_alloc(CS%ctx, CS%d_u, n3) ; rc = mom6x_dev_alloc(CS%ctx, CS%d_v, n3) ; rc = mom6x_dev_alloc(CS%ctx, CS%d_h, n3)
  rc = mom6x_dev_alloc(CS%ctx, CS%d_uh, n3) ; rc = mom6x_dev_alloc(CS%ctx, CS%d_vh, n3)
  rc = mom6x_dev_alloc(CS%ctx, CS%d_uhtr, n3) ; rc = mom6x_dev_alloc(CS%ctx, CS%d_vhtr, n3)
  rc = mom6x_dev_alloc(CS%ctx, CS%d_eta_av, n2) ; rc = mom6x_dev_alloc(CS%ctx, CS%d_taux, n2) ; rc = mom6x_dev_alloc(CS%ctx, CS%d_tauy, n2)
end subroutine initialize_dyn_split_RK2_amd

subroutine end_dyn_split_RK2(CS)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  integer(c_int) :: rc
  if (associated(CS)) then ; rc = mom6x_ctx_destroy(CS%ctx) ; deallocate(CS) ; endif
end subroutine end_dyn_split_RK2

function c_message() result(msg)
  character(len=480) :: msg
  character(kind=c_char), pointer :: p(:)
  integer :: n
  msg = "" ; call c_f_pointer(mom6x_last_error(), p, [480])
  do n=1,480 ; if (p(n) == c_null_char) exit ; msg(n:n) = p(n) ; enddo
end function c_message

end module MOM_dynamics_split_RK2_amd
