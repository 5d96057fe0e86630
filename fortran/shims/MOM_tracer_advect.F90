!> Drop-in for src/tracer/MOM_tracer_advect.F90: advect_tracer :53-54, tracer_advect_init :1155, tracer_advect_end :1205 and
!! the type tracer_advect_CS -- same module name, procedure names and argument lists, served by mom6x_tracer_advect_init /
!! mom6x_advect_tracer.  Called once per thermodynamic step by MOM.F90 (step_MOM_tracer_dyn) with host arrays: h_end, the
!! accumulated transports and the tracers of the registry are uploaded, advected and the tracers (and uhr_out / vhr_out when
!! present: the transports the iteration could not use, :336-348) downloaded.  The offline-transport arguments (vol_prev,
!! update_vol_prev) and open boundaries are refused, as the device routine refuses them.
module MOM_tracer_advect
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_cpu_clock,       only : cpu_clock_id, cpu_clock_begin, cpu_clock_end, CLOCK_MODULE
use MOM_diag_mediator,   only : diag_ctrl
use MOM_error_handler,   only : MOM_error, FATAL
use MOM_file_parser,     only : get_param, log_version, param_file_type
use MOM_grid,            only : ocean_grid_type
use MOM_open_boundary,   only : ocean_OBC_type
use MOM_time_manager,    only : time_type
use MOM_tracer_registry, only : tracer_registry_type, tracer_type
use MOM_tracer_advect_schemes, only : ADVECT_PLM, ADVECT_PPMH3, ADVECT_PPM, set_tracer_advect_scheme, TracerAdvectionSchemeDoc
use MOM_unit_scaling,    only : unit_scale_type
use MOM_verticalGrid,    only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>
public :: advect_tracer, tracer_advect_init, tracer_advect_end

type, public :: tracer_advect_CS ; private
  real    :: dt                        !< the (baroclinic) dynamics time step, DT (:1177)
  integer :: default_advect_scheme     !< TRACER_ADVECTION_SCHEME as ADVECT_PLM / ADVECT_PPMH3 / ADVECT_PPM
  logical :: useHuynhStencilBug = .false.
  logical :: device_ready = .false.    !< mom6x_tracer_advect_init has run (it needs G, GV, which advect_tracer brings)
  integer :: last_iterations = 0       !< passes of the last call (the reference reports them through its DEBUG messages)
end type tracer_advect_CS

integer :: id_clock_advect = -1

contains

!> advect_tracer (:53-54)
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
  type(c_ptr) :: ctx, d_h, d_uhtr, d_vhtr, p_uhr, p_vhr
  type(c_ptr), allocatable, target :: d_tr(:)
  integer(c_int), allocatable, target :: schemes(:)
  integer(c_int), target :: iters
  integer(c_int) :: rc, x_first, max_iter
  integer :: m, nk

  if (.not.associated(CS)) call MOM_error(FATAL, "MOM_tracer_advect: tracer_advect_init must be called before advect_tracer.")
  if (.not.associated(Reg)) call MOM_error(FATAL, "MOM_tracer_advect: register_tracer must be called before advect_tracer.")
  if (Reg%ntr == 0) return
  if (associated(OBC)) call MOM_error(FATAL, "advect_tracer: open boundaries are not carried by the MI355X path.")
  if (present(vol_prev) .or. present(update_vol_prev)) call MOM_error(FATAL, &
      "advect_tracer: the offline-transport arguments (vol_prev, update_vol_prev) are not carried by the MI355X path.")
  call cpu_clock_begin(id_clock_advect)
  ctx = shim_ctx(G, GV) ; nk = GV%ke
  if (.not.CS%device_ready) then
    rc = mom6x_tracer_advect_init(ctx, real(CS%dt, c_double), int(CS%default_advect_scheme, c_int), &
                                  merge(1_c_int, 0_c_int, CS%useHuynhStencilBug))
    call shim_check(rc, "tracer_advect_init") ; CS%device_ready = .true.
  endif
  d_h = shim_up3(1, h_end, STG_H, nk) ; d_uhtr = shim_up3(2, uhtr, STG_U, nk) ; d_vhtr = shim_up3(3, vhtr, STG_V, nk)
  allocate(d_tr(Reg%ntr), schemes(Reg%ntr))
  do m = 1, Reg%ntr
    d_tr(m) = shim_up3(5 + m, Reg%Tr(m)%t, STG_H, nk)
    schemes(m) = int(CS%default_advect_scheme, c_int)                 ! Reg%Tr(m)%advect_scheme < 0: the module default (:300-305)
    if (Reg%Tr(m)%advect_scheme >= 0) schemes(m) = int(Reg%Tr(m)%advect_scheme, c_int)
  enddo
  x_first = -1 ; if (present(x_first_in)) x_first = merge(1_c_int, 0_c_int, x_first_in)   ! absent: G%first_direction decides (:131)
  max_iter = -1 ; if (present(max_iter_in)) max_iter = int(max_iter_in, c_int)
  p_uhr = c_null_ptr ; p_vhr = c_null_ptr
  if (present(uhr_out)) p_uhr = shim_buf(4, nk)
  if (present(vhr_out)) p_vhr = shim_buf(5, nk)
  rc = mom6x_advect_tracer(ctx, d_h, d_uhtr, d_vhtr, real(dt, c_double), c_loc(d_tr), c_loc(schemes), int(Reg%ntr, c_int), &
                           x_first, max_iter, p_uhr, p_vhr, c_loc(iters))
  call shim_check(rc, "advect_tracer")
  CS%last_iterations = iters
  do m = 1, Reg%ntr ; call shim_down3(Reg%Tr(m)%t, d_tr(m), STG_H, nk) ; enddo
  if (present(uhr_out)) call shim_down3(uhr_out, p_uhr, STG_U, nk)
  if (present(vhr_out)) call shim_down3(vhr_out, p_vhr, STG_V, nk)
  call cpu_clock_end(id_clock_advect)
end subroutine advect_tracer

!> tracer_advect_init (:1155): DT, TRACER_ADVECTION_SCHEME, USE_HUYNH_STENCIL_BUG (:1176-1196)
subroutine tracer_advect_init(Time, G, US, param_file, diag, CS)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(tracer_advect_CS),  pointer       :: CS
  character(len=40)  :: mdl = "MOM_tracer_advect"
  character(len=256) :: mesg
  if (associated(CS)) then
    call MOM_error(FATAL, "tracer_advect_init called with associated control structure.")   ! (a WARNING + return in the reference)
    return
  endif
  allocate(CS)
  call log_version(param_file, mdl, "mom6x", "")
  call get_param(param_file, mdl, "DT", CS%dt, fail_if_missing=.true., desc="The (baroclinic) dynamics time step.", units="s", &
                 scale=US%s_to_T)
  call get_param(param_file, mdl, "TRACER_ADVECTION_SCHEME", mesg, desc="The horizontal transport scheme for tracers:\n"//&
                 trim(TracerAdvectionSchemeDoc), default='PLM')
  call set_tracer_advect_scheme(CS%default_advect_scheme, mesg)
  if (CS%default_advect_scheme == ADVECT_PPMH3) then
    call get_param(param_file, mdl, "USE_HUYNH_STENCIL_BUG", CS%useHuynhStencilBug, desc="If true, use a stencil width of 2 in "//&
                   "PPM:H3 tracer advection. This is incorrect and will produce regressions in certain configurations, but may "//&
                   "be required to reproduce results in legacy simulations.", default=.false.)
  endif
  call shim_set_domain_flags(param_file)
  id_clock_advect = cpu_clock_id('(Ocean advect tracer)', grain=CLOCK_MODULE)
end subroutine tracer_advect_init

!> tracer_advect_end (:1205)
subroutine tracer_advect_end(CS)
  type(tracer_advect_CS), pointer :: CS
  if (associated(CS)) deallocate(CS)
end subroutine tracer_advect_end

end module MOM_tracer_advect
