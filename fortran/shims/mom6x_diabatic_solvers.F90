!> The vertical tridiagonal solves of the thermodynamic step on the device, under the reference's procedure names and
!! argument lists: triDiagTS (src/parameterizations/vertical/MOM_diabatic_aux.F90:394), triDiagTS_Eulerian (:444),
!! tracer_vertdiff (src/tracer/MOM_tracer_diabatic.F90:25), tracer_vertdiff_Eulerian (:224), and diabatic's own early
!! return for a single layer (MOM_diabatic_driver.F90:277, `if (GV%ke == 1) return`).
!!
!! MOM_diabatic_aux and MOM_tracer_diabatic hold a great deal of host physics besides these four solvers
!! (applyBoundaryFluxesInOut, make_frazil, set_pen_shortwave, applyTracerBoundaryFluxesInOut ...), so the two modules are
!! not replaced: a build points the `use ..., only : triDiagTS, triDiagTS_Eulerian` of MOM_diabatic_driver.F90:14-15 and
!! the `use MOM_tracer_diabatic, only : tracer_vertdiff, tracer_vertdiff_Eulerian` of :71 and of the tracer packages
!! (DOME_tracer.F90:21 ...) at this module instead -- a one-line change per `use`, INTEGRATION.md section 3.
!! The sinking / bottom-reservoir forms of tracer_vertdiff (sink_rate, btm_reservoir) are carried since round 4.
module mom6x_diabatic_solvers
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_error_handler, only : MOM_error, FATAL
use MOM_grid,          only : ocean_grid_type
use MOM_verticalGrid,  only : verticalGrid_type
implicit none ; private
#include <MOM_memory.h>
public :: triDiagTS, triDiagTS_Eulerian, tracer_vertdiff, tracer_vertdiff_Eulerian, diabatic_is_trivial

contains

!> triDiagTS (MOM_diabatic_aux.F90:394); is, ie, js, je are MOM6's local indices (G%isc ... of the caller)
subroutine triDiagTS(G, GV, is, ie, js, je, hold, ea, eb, T, S)
  type(ocean_grid_type),                     intent(in)    :: G
  type(verticalGrid_type),                   intent(in)    :: GV
  integer,                                   intent(in)    :: is, ie, js, je
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in)    :: hold, ea, eb
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(inout) :: T, S
  type(c_ptr) :: ctx, d_T, d_S
  integer(c_int) :: rc
  ctx = shim_ctx(G, GV)
  d_T = shim_up3(4, T, STG_H, GV%ke) ; d_S = shim_up3(5, S, STG_H, GV%ke)
  rc = mom6x_triDiagTS(ctx, int(is - G%isc, c_int), int(ie - G%isc, c_int), int(js - G%jsc, c_int), int(je - G%jsc, c_int), &
                       shim_up3(1, hold, STG_H, GV%ke), shim_up3(2, ea, STG_H, GV%ke), shim_up3(3, eb, STG_H, GV%ke), d_T, d_S)
  call shim_check(rc, "triDiagTS")
  call shim_down3(T, d_T, STG_H, GV%ke) ; call shim_down3(S, d_S, STG_H, GV%ke)
end subroutine triDiagTS

!> triDiagTS_Eulerian (MOM_diabatic_aux.F90:444)
subroutine triDiagTS_Eulerian(G, GV, is, ie, js, je, hold, ent, T, S)
  type(ocean_grid_type),                     intent(in)    :: G
  type(verticalGrid_type),                   intent(in)    :: GV
  integer,                                   intent(in)    :: is, ie, js, je
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in)    :: hold
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)+1), intent(in)  :: ent
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(inout) :: T, S
  type(c_ptr) :: ctx, d_T, d_S
  integer(c_int) :: rc
  ctx = shim_ctx(G, GV)
  d_T = shim_up3(4, T, STG_H, GV%ke) ; d_S = shim_up3(5, S, STG_H, GV%ke)
  rc = mom6x_triDiagTS_Eulerian(ctx, int(is - G%isc, c_int), int(ie - G%isc, c_int), int(js - G%jsc, c_int), int(je - G%jsc, c_int), &
                                shim_up3(1, hold, STG_H, GV%ke), shim_up3(2, ent, STG_H, GV%ke + 1), d_T, d_S)
  call shim_check(rc, "triDiagTS_Eulerian")
  call shim_down3(T, d_T, STG_H, GV%ke) ; call shim_down3(S, d_S, STG_H, GV%ke)
end subroutine triDiagTS_Eulerian

!> tracer_vertdiff (MOM_tracer_diabatic.F90:25)
subroutine tracer_vertdiff(h_old, ea, eb, dt, tr, G, GV, sfc_flux, btm_flux, btm_reservoir, sink_rate, convert_flux_in)
  type(ocean_grid_type),                     intent(in)    :: G
  type(verticalGrid_type),                   intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in)    :: h_old, ea, eb
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(inout) :: tr
  real,                                      intent(in)    :: dt
  real, dimension(SZI_(G),SZJ_(G)), optional,intent(in)    :: sfc_flux, btm_flux
  real, dimension(SZI_(G),SZJ_(G)), optional,intent(inout) :: btm_reservoir
  real,                             optional,intent(in)    :: sink_rate
  logical,                          optional,intent(in)    :: convert_flux_in
  type(c_ptr) :: ctx, d_tr, p_sfc, p_btm, p_res
  integer(c_int) :: rc, convert
  ctx = shim_ctx(G, GV)
  p_sfc = c_null_ptr ; p_btm = c_null_ptr ; p_res = c_null_ptr
  if (present(sfc_flux)) p_sfc = shim_up2(5, sfc_flux, STG_H)
  if (present(btm_flux)) p_btm = shim_up2(6, btm_flux, STG_H)
  convert = 1 ; if (present(convert_flux_in)) convert = merge(1_c_int, 0_c_int, convert_flux_in)   ! default .true. (:64)
  d_tr = shim_up3(4, tr, STG_H, GV%ke)
  if (present(sink_rate)) then       ! :123-179; btm_reservoir is only read (and updated) on this branch
    if (present(btm_reservoir)) p_res = shim_up2(7, btm_reservoir, STG_H)
    rc = mom6x_tracer_vertdiff_sink(ctx, shim_up3(1, h_old, STG_H, GV%ke), shim_up3(2, ea, STG_H, GV%ke), shim_up3(3, eb, STG_H, GV%ke), &
                                    real(dt, c_double), d_tr, p_sfc, p_btm, p_res, real(sink_rate, c_double), convert)
    call shim_check(rc, "tracer_vertdiff")
    if (present(btm_reservoir)) call shim_down2(btm_reservoir, p_res, STG_H)
  else
    rc = mom6x_tracer_vertdiff(ctx, shim_up3(1, h_old, STG_H, GV%ke), shim_up3(2, ea, STG_H, GV%ke), shim_up3(3, eb, STG_H, GV%ke), &
                               real(dt, c_double), d_tr, p_sfc, p_btm, convert)
    call shim_check(rc, "tracer_vertdiff")
  endif
  call shim_down3(tr, d_tr, STG_H, GV%ke)
end subroutine tracer_vertdiff

!> tracer_vertdiff_Eulerian (MOM_tracer_diabatic.F90:224)
subroutine tracer_vertdiff_Eulerian(h_old, ent, dt, tr, G, GV, sfc_flux, btm_flux, btm_reservoir, sink_rate, convert_flux_in)
  type(ocean_grid_type),                     intent(in)    :: G
  type(verticalGrid_type),                   intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in)    :: h_old
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)+1), intent(in)  :: ent
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(inout) :: tr
  real,                                      intent(in)    :: dt
  real, dimension(SZI_(G),SZJ_(G)), optional,intent(in)    :: sfc_flux, btm_flux
  real, dimension(SZI_(G),SZJ_(G)), optional,intent(inout) :: btm_reservoir
  real,                             optional,intent(in)    :: sink_rate
  logical,                          optional,intent(in)    :: convert_flux_in
  type(c_ptr) :: ctx, d_tr, p_sfc, p_btm, p_res
  integer(c_int) :: rc, convert
  ctx = shim_ctx(G, GV)
  p_sfc = c_null_ptr ; p_btm = c_null_ptr ; p_res = c_null_ptr
  if (present(sfc_flux)) p_sfc = shim_up2(5, sfc_flux, STG_H)
  if (present(btm_flux)) p_btm = shim_up2(6, btm_flux, STG_H)
  convert = 1 ; if (present(convert_flux_in)) convert = merge(1_c_int, 0_c_int, convert_flux_in)
  d_tr = shim_up3(4, tr, STG_H, GV%ke)
  if (present(sink_rate)) then       ! :315-380
    if (present(btm_reservoir)) p_res = shim_up2(7, btm_reservoir, STG_H)
    rc = mom6x_tracer_vertdiff_Eulerian_sink(ctx, shim_up3(1, h_old, STG_H, GV%ke), shim_up3(2, ent, STG_H, GV%ke + 1), &
                                             real(dt, c_double), d_tr, p_sfc, p_btm, p_res, real(sink_rate, c_double), convert)
    call shim_check(rc, "tracer_vertdiff_Eulerian")
    if (present(btm_reservoir)) call shim_down2(btm_reservoir, p_res, STG_H)
  else
    rc = mom6x_tracer_vertdiff_Eulerian(ctx, shim_up3(1, h_old, STG_H, GV%ke), shim_up3(2, ent, STG_H, GV%ke + 1), &
                                        real(dt, c_double), d_tr, p_sfc, p_btm, convert)
    call shim_check(rc, "tracer_vertdiff_Eulerian")
  endif
  call shim_down3(tr, d_tr, STG_H, GV%ke)
end subroutine tracer_vertdiff_Eulerian

!> diabatic's early return (MOM_diabatic_driver.F90:277 ff.: nothing to do with one layer)
logical function diabatic_is_trivial(G, GV)
  type(ocean_grid_type), intent(in) :: G ; type(verticalGrid_type), intent(in) :: GV
  diabatic_is_trivial = (mom6x_diabatic_is_trivial(shim_ctx(G, GV)) /= 0)
end function diabatic_is_trivial

end module mom6x_diabatic_solvers
