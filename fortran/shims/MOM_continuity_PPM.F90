!> Drop-in replacement of the routine other modules call directly: continuity_PPM (src/core/MOM_continuity_PPM.F90:86-87),
!! same module name, argument list and optional-argument semantics (presence changes behaviour: :590-592, :637, :737,
!! :756), served by mom6x_continuity_PPM.  Inside the device step (MOM_dynamics_split_RK2) this boundary is never
!! crossed; it exists for the callers outside it (MOM_dynamics_unsplit*, thickness diffusion's continuity_adjust_vel ...),
!! which hand over HOST arrays: each call uploads its inputs and downloads its outputs (PCIe-bound; DESIGN.md section 4).
!! Compiles only inside a MOM6 tree.
module MOM_continuity_PPM
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use MOM_error_handler,   only : MOM_error, FATAL
use MOM_grid,            only : ocean_grid_type
use MOM_open_boundary,   only : ocean_OBC_type
use MOM_porous_barriers, only : porous_barrier_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_variables,       only : BT_cont_type
use MOM_verticalGrid,    only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>
public :: continuity_PPM, continuity_PPM_CS

!> The control structure: the device context the dynamics shim created (shared), and scratch device arrays
type :: continuity_PPM_CS ; private
  type(c_ptr) :: ctx = c_null_ptr
  type(c_ptr) :: d(13) = c_null_ptr   !< u, v, hin, h, uh, vh, uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, du_cor|dv_cor pair
  logical :: initialized = .false.
end type continuity_PPM_CS

contains

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
  integer(c_int) :: rc, nk
  type(c_ptr) :: p_uhbt, p_vhbt, p_vru, p_vrv, p_ucor, p_vcor, p_ducor, p_dvcor

  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_continuity_PPM: Module must be initialized before it is used.")
  if (associated(OBC)) call MOM_error(FATAL, "continuity_PPM: open boundaries are not carried by the device path.")
  if (present(visc_rem_u) .neqv. present(visc_rem_v)) call MOM_error(FATAL, "MOM_continuity_PPM: Either both "//&
      "visc_rem_u and visc_rem_v or neither one must be present in call to continuity_PPM.")
  if (present(BT_cont)) then ; if (associated(BT_cont)) call MOM_error(FATAL, &
      "continuity_PPM shim: BT_cont is filled on the device inside the split step only.") ; endif
  nk = int(GV%ke, c_int)
  rc = mom6x_upload(CS%ctx, CS%d(1), u, STG_U, nk) ; rc = mom6x_upload(CS%ctx, CS%d(2), v, STG_V, nk)
  rc = mom6x_upload(CS%ctx, CS%d(3), hin, STG_H, nk)
  p_uhbt = c_null_ptr ; p_vhbt = c_null_ptr ; p_vru = c_null_ptr ; p_vrv = c_null_ptr
  p_ucor = c_null_ptr ; p_vcor = c_null_ptr ; p_ducor = c_null_ptr ; p_dvcor = c_null_ptr
  if (present(uhbt)) then ; rc = mom6x_upload(CS%ctx, CS%d(7), uhbt, STG_U, 1_c_int) ; p_uhbt = CS%d(7) ; endif
  if (present(vhbt)) then ; rc = mom6x_upload(CS%ctx, CS%d(8), vhbt, STG_V, 1_c_int) ; p_vhbt = CS%d(8) ; endif
  if (present(visc_rem_u)) then
    rc = mom6x_upload(CS%ctx, CS%d(9), visc_rem_u, STG_U, nk) ; rc = mom6x_upload(CS%ctx, CS%d(10), visc_rem_v, STG_V, nk)
    p_vru = CS%d(9) ; p_vrv = CS%d(10)
  endif
  if (present(u_cor)) p_ucor = CS%d(11) ; if (present(v_cor)) p_vcor = CS%d(12)
  if (present(du_cor)) p_ducor = CS%d(13) ; if (present(dv_cor)) p_dvcor = CS%d(13)   ! (never both in the reference's calls)
  rc = mom6x_continuity_PPM(CS%ctx, CS%d(1), CS%d(2), CS%d(3), CS%d(4), CS%d(5), CS%d(6), real(dt, c_double), p_uhbt, p_vhbt, &
                            p_vru, p_vrv, p_ucor, p_vcor, c_null_ptr, p_ducor, p_dvcor)
  if (rc /= 0) call MOM_error(FATAL, "continuity_PPM: "//trim(mom6x_message()))
  rc = mom6x_download(CS%ctx, h, CS%d(4), STG_H, nk)
  rc = mom6x_download(CS%ctx, uh, CS%d(5), STG_U, nk) ; rc = mom6x_download(CS%ctx, vh, CS%d(6), STG_V, nk)
  if (present(u_cor)) rc = mom6x_download(CS%ctx, u_cor, CS%d(11), STG_U, nk)
  if (present(v_cor)) rc = mom6x_download(CS%ctx, v_cor, CS%d(12), STG_V, nk)
  if (present(du_cor)) rc = mom6x_download(CS%ctx, du_cor, CS%d(13), STG_U, 1_c_int)
  if (present(dv_cor)) rc = mom6x_download(CS%ctx, dv_cor, CS%d(13), STG_V, 1_c_int)
end subroutine continuity_PPM

end module MOM_continuity_PPM
