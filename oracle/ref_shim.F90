!> ORACLE SUPPORT (test infrastructure only): bind(C) doors into the REAL reference modules that compile from their own
!! source files without FMS / netCDF (src/ALE/PLM_functions.F90, src/ALE/PCM_functions.F90: no `use` statements at all).
!! oracle/Makefile compiles those two files where they lie under /root/reference together with this file into
!! oracle/_ref/libref_ale.so; tests/test_remap_cpu.py then holds orc_remap.c's PLM / PCM reconstructions to the
!! reference's own code, bit for bit, on random columns.  Nothing of the reference is copied here.
module ref_shim
  use, intrinsic :: iso_c_binding
  use PLM_functions, only : PLM_reconstruction, PLM_boundary_extrapolation, PLM_slope_wa, PLM_monotonized_slope, &
                            PLM_extrapolate_slope
  use PCM_functions, only : PCM_reconstruction
  implicit none
contains
  subroutine ref_PLM_reconstruction(n, h, u, edges, coefs, h_neglect, extrapolate) bind(C, name="ref_PLM_reconstruction")
    integer(c_int), value :: n, extrapolate
    real(c_double), intent(in) :: h(n), u(n)
    real(c_double), intent(inout) :: edges(n,2), coefs(n,2)
    real(c_double), value :: h_neglect
    call PLM_reconstruction(int(n), h, u, edges, coefs, h_neglect)
    if (extrapolate /= 0) call PLM_boundary_extrapolation(int(n), h, u, edges, coefs, h_neglect)
  end subroutine
  subroutine ref_PCM_reconstruction(n, u, edges, coefs) bind(C, name="ref_PCM_reconstruction")
    integer(c_int), value :: n
    real(c_double), intent(in) :: u(n)
    real(c_double), intent(inout) :: edges(n,2), coefs(n,1)
    call PCM_reconstruction(int(n), u, edges, coefs)
  end subroutine
  real(c_double) function ref_PLM_slope_wa(h_l, h_c, h_r, h_neglect, u_l, u_c, u_r) bind(C, name="ref_PLM_slope_wa")
    real(c_double), value :: h_l, h_c, h_r, h_neglect, u_l, u_c, u_r
    ref_PLM_slope_wa = PLM_slope_wa(h_l, h_c, h_r, h_neglect, u_l, u_c, u_r)
  end function
  real(c_double) function ref_PLM_monotonized_slope(u_l, u_c, u_r, s_l, s_c, s_r) bind(C, name="ref_PLM_monotonized_slope")
    real(c_double), value :: u_l, u_c, u_r, s_l, s_c, s_r
    ref_PLM_monotonized_slope = PLM_monotonized_slope(u_l, u_c, u_r, s_l, s_c, s_r)
  end function
  real(c_double) function ref_PLM_extrapolate_slope(h_l, h_c, h_neglect, u_l, u_c) bind(C, name="ref_PLM_extrapolate_slope")
    real(c_double), value :: h_l, h_c, h_neglect, u_l, u_c
    ref_PLM_extrapolate_slope = PLM_extrapolate_slope(h_l, h_c, h_neglect, u_l, u_c)
  end function
end module ref_shim
