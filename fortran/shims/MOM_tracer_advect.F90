!> Drop-in replacement of advect_tracer (src/tracer/MOM_tracer_advect.F90:53-54): same module name and argument list,
!! served by mom6x_advect_tracer.  Called once per thermodynamic step by MOM.F90 (step_MOM_tracer_dyn), with the
!! transports uhtr, vhtr the dynamics accumulated ON THE DEVICE: h_end, uhtr, vhtr are taken from the resident state
!! when the dynamics shim says its copy is current (no upload), the tracers of the registry are uploaded, advected and
!! downloaded (they belong to the host-side tracer packages).  The offline-transport arguments (vol_prev,
!! update_vol_prev) and open boundaries are rejected, as the device routine does.  Compiles only inside a MOM6 tree.
module MOM_tracer_advect
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use MOM_error_handler,   only : MOM_error, FATAL
use MOM_grid,            only : ocean_grid_type
use MOM_open_boundary,   only : ocean_OBC_type
use MOM_tracer_registry, only : tracer_registry_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_verticalGrid,    only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>
public :: advect_tracer, tracer_advect_CS

type :: tracer_advect_CS ; private
  type(c_ptr) :: ctx = c_null_ptr
  type(c_ptr) :: d_h = c_null_ptr, d_uhtr = c_null_ptr, d_vhtr = c_null_ptr   !< the dynamics' resident arrays, or scratch
  type(c_ptr), allocatable :: d_tr(:)      !< one device array per registered tracer
  integer(c_int), allocatable :: schemes(:) !< TRACER_ADVECTION_SCHEME per tracer (0 PLM, 1 PPM:H3, 2 PPM; Reg%Tr(m)%advect_scheme or CS default)
  logical :: state_is_resident = .false.   !< h, uhtr, vhtr above ARE the dynamics shim's device arrays
end type tracer_advect_CS

contains

subroutine advect_tracer(h_end, uhtr, vhtr, OBC, dt, G, GV, US, CS, Reg, x_first_in, &
                         vol_prev, max_iter_in, update_vol_prev, uhr_out, vhr_out)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h_end
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: uhtr
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: vhtr
  type(ocean_OBC_type),    pointer       :: OBC
  real,                    intent(in)    :: dt
  type(unit_scale_type),   intent(in)    :: US
  type(tracer_advect_CS),  pointer       :: CS
  type(tracer_registry_type), pointer    :: Reg
  logical,       optional, intent(in)    :: x_first_in
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), optional, intent(inout) :: vol_prev
  integer,       optional, intent(in)    :: max_iter_in
  logical,       optional, intent(in)    :: update_vol_prev
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(out) :: uhr_out
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(out) :: vhr_out
  integer(c_int) :: rc, nk, x_first, max_iter, iters
  integer :: m
  type(c_ptr) :: p_uhr, p_vhr

  if (.not.associated(CS)) call MOM_error(FATAL, "MOM_tracer_advect: tracer_advect_init must be called before advect_tracer.")
  if (.not.associated(Reg)) call MOM_error(FATAL, "MOM_tracer_advect: register_tracer must be called before advect_tracer.")
  if (Reg%ntr == 0) return
  if (associated(OBC)) call MOM_error(FATAL, "advect_tracer: open boundaries are not carried by the device path.")
  if (present(vol_prev) .or. present(update_vol_prev)) call MOM_error(FATAL, &
      "advect_tracer: the offline-transport arguments are not carried by the device path.")
  nk = int(GV%ke, c_int)
  if (.not.CS%state_is_resident) then
    rc = mom6x_upload(CS%ctx, CS%d_h, h_end, STG_H, nk)
    rc = mom6x_upload(CS%ctx, CS%d_uhtr, uhtr, STG_U, nk) ; rc = mom6x_upload(CS%ctx, CS%d_vhtr, vhtr, STG_V, nk)
  endif
  do m = 1, Reg%ntr ; rc = mom6x_upload(CS%ctx, CS%d_tr(m), Reg%Tr(m)%t, STG_H, nk) ; enddo
  x_first = -1 ; if (present(x_first_in)) x_first = merge(1_c_int, 0_c_int, x_first_in)     ! -1: absent (G%first_direction decides)
  max_iter = -1 ; if (present(max_iter_in)) max_iter = int(max_iter_in, c_int)
  p_uhr = c_null_ptr ; p_vhr = c_null_ptr   ! (uhr_out / vhr_out: scratch device arrays when present; see INTEGRATION.md)
  rc = mom6x_advect_tracer(CS%ctx, CS%d_h, CS%d_uhtr, CS%d_vhtr, real(dt, c_double), CS%d_tr, CS%schemes, int(Reg%ntr, c_int), &
                           x_first, max_iter, p_uhr, p_vhr, iters)
  if (rc /= 0) call MOM_error(FATAL, "advect_tracer: "//trim(mom6x_message()))
  do m = 1, Reg%ntr ; rc = mom6x_download(CS%ctx, Reg%Tr(m)%t, CS%d_tr(m), STG_H, nk) ; enddo
end subroutine advect_tracer

end module MOM_tracer_advect
